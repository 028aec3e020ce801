"""Architecture registry and checkpoint I/O with the reference's signatures
(src/lib/models/model.py:24-95): ``create_model(arch, heads, head_conv)``,
``load_model(model, path, ...)``, ``save_model(path, epoch, model, optimizer=None)``.

``load_model`` keeps the reference's tolerant behaviour (model.py:31-67): ``module.`` prefixes of
DataParallel checkpoints are stripped, tensors whose shape does not match are skipped with a
message, parameters the checkpoint lacks keep the model's value, and the final load is
non-strict -- so every zoo checkpoint of a supported architecture loads unchanged.
"""
import importlib

import torch

# arch prefix -> (module under centernet_amd.networks, factory name)
_ARCHS = {
    'res': ('resnet', 'get_pose_net'),                      # msra_resnet.py
    'resdcn': ('resnet', 'get_pose_net_dcn'),               # resnet_dcn.py
    'dla': ('pose_dla_dcn', 'get_pose_net'),                # pose_dla_dcn.py
    'hourglass': ('large_hourglass', 'get_large_hourglass_net'),  # large_hourglass.py
}

_HINT = ('If you see this, your model does not fully load the pre-trained weight. Please make '
         'sure you have correctly specified --arch xxx or set the correct --num_classes for '
         'your own dataset.')


def _split_arch(arch):
    """'resdcn_18' -> ('resdcn', 18); 'hourglass' -> ('hourglass', 0)."""
    name, sep, depth = arch.partition('_')
    return name, int(depth) if sep else 0


def create_model(arch, heads, head_conv):
    name, depth = _split_arch(arch)
    if name not in _ARCHS:
        raise KeyError("arch '%s' is outside the MI355X hot path (supported: %s)"
                       % (name, sorted(_ARCHS)))
    module, factory = _ARCHS[name]
    build = getattr(importlib.import_module('.networks.' + module, __package__), factory)
    return build(num_layers=depth, heads=heads, head_conv=head_conv)


def _without_dataparallel_prefix(state_dict):
    def clean(key):
        return key[7:] if key.startswith('module') and not key.startswith('module_list') else key
    return {clean(k): v for k, v in state_dict.items()}


def _reconcile(loaded, own):
    """Drop-in the checkpoint tensors that fit; report and keep the model's own value otherwise."""
    merged = dict(own)
    for key, tensor in loaded.items():
        if key not in own:
            print('Drop parameter {}.'.format(key) + _HINT)
        elif tensor.shape != own[key].shape:
            print('Skip loading parameter {}, required shape{}, loaded shape{}. {}'.format(
                key, own[key].shape, tensor.shape, _HINT))
        else:
            merged[key] = tensor
    for key in own:
        if key not in loaded:
            print('No param {}.'.format(key) + _HINT)
    return merged


def _read_checkpoint(model_path):
    """Zoo checkpoints are plain {'epoch', 'state_dict', 'optimizer'} dicts of tensors, so the
    safe tensors-only unpickler is enough.  Files that need full unpickling (arbitrary code
    execution from a downloaded file) load only with CENTERNET_UNSAFE_LOAD=1, with a warning."""
    import os
    try:
        return torch.load(model_path, map_location='cpu', weights_only=True)
    except Exception as err:
        if os.environ.get('CENTERNET_UNSAFE_LOAD') != '1':
            raise RuntimeError(
                "%s does not load with weights_only=True (%s). If you trust the file, set "
                "CENTERNET_UNSAFE_LOAD=1 to unpickle it fully." % (model_path, err)) from err
        print('WARNING: unpickling {} without weights_only (CENTERNET_UNSAFE_LOAD=1)'.format(model_path))
        return torch.load(model_path, map_location='cpu', weights_only=False)


def _decayed_lr(lr, lr_step, epoch):
    """Base rate divided by ten for every schedule step already behind ``epoch``."""
    return lr * (0.1 ** sum(1 for step in (lr_step or ()) if epoch >= step))


def load_model(model, model_path, optimizer=None, resume=False, lr=None, lr_step=None):
    """models/model.py:31-84: tolerant state-dict load; with ``optimizer`` the return value is
    ``(model, optimizer, start_epoch)`` and, when resuming, the optimizer state and the
    step-decayed learning rate are restored."""
    checkpoint = _read_checkpoint(model_path)
    print('loaded {}, epoch {}'.format(model_path, checkpoint.get('epoch', '?')))
    loaded = _without_dataparallel_prefix(checkpoint['state_dict'])
    model.load_state_dict(_reconcile(loaded, model.state_dict()), strict=False)
    if optimizer is None:
        return model
    start_epoch = 0
    if resume:
        if 'optimizer' in checkpoint:
            optimizer.load_state_dict(checkpoint['optimizer'])
            start_epoch = checkpoint['epoch']
            rate = _decayed_lr(lr, lr_step, start_epoch)
            for group in optimizer.param_groups:
                group['lr'] = rate
            print('Resumed optimizer with start lr', rate)
        else:
            print('No optimizer parameters in checkpoint.')
    return model, optimizer, start_epoch


def save_model(path, epoch, model, optimizer=None):
    net = model.module if isinstance(model, torch.nn.DataParallel) else model
    payload = {'epoch': epoch, 'state_dict': net.state_dict()}
    if optimizer is not None:
        payload['optimizer'] = optimizer.state_dict()
    torch.save(payload, path)
