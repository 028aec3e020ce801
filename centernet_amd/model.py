"""Architecture registry and checkpoint I/O (mirror of src/lib/models/model.py).

``create_model(arch, heads, head_conv)`` and ``load_model(model, path, ...)`` keep the
reference's signatures and its tolerant-load behaviour (model.py:31-67): DataParallel
``module.`` prefixes are stripped, parameters with a mismatching shape are skipped with
a message, missing ones keep the model's value, ``strict=False``.
"""
import torch

from .networks.resnet import get_pose_net, get_pose_net_dcn


def _lazy_dla(num_layers, heads, head_conv):
    from .networks.pose_dla_dcn import get_pose_net as f
    return f(num_layers=num_layers, heads=heads, head_conv=head_conv)


def _lazy_hourglass(num_layers, heads, head_conv):
    from .networks.large_hourglass import get_large_hourglass_net as f
    return f(num_layers=num_layers, heads=heads, head_conv=head_conv)


_model_factory = {
    'res': get_pose_net,          # msra_resnet.py
    'resdcn': get_pose_net_dcn,   # resnet_dcn.py
    'dla': _lazy_dla,             # pose_dla_dcn.py
    'hourglass': _lazy_hourglass,  # large_hourglass.py
}


def create_model(arch, heads, head_conv):
    # model.py:24-29
    num_layers = int(arch[arch.find('_') + 1:]) if '_' in arch else 0
    arch = arch[:arch.find('_')] if '_' in arch else arch
    if arch not in _model_factory:
        raise KeyError("arch '%s' is outside the MI355X hot path (supported: %s)"
                       % (arch, sorted(_model_factory)))
    return _model_factory[arch](num_layers=num_layers, heads=heads, head_conv=head_conv)


def load_model(model, model_path, optimizer=None, resume=False, lr=None, lr_step=None):
    # model.py:31-84 (the optimizer/resume branch is training-only and not supported)
    if optimizer is not None:
        raise NotImplementedError("training state (optimizer/resume) is out of scope")
    checkpoint = torch.load(model_path, map_location=lambda storage, loc: storage,
                            weights_only=False)
    print('loaded {}, epoch {}'.format(model_path, checkpoint.get('epoch', '?')))
    state_dict_ = checkpoint['state_dict']
    state_dict = {}
    for k in state_dict_:
        if k.startswith('module') and not k.startswith('module_list'):
            state_dict[k[7:]] = state_dict_[k]
        else:
            state_dict[k] = state_dict_[k]
    model_state_dict = model.state_dict()
    msg = ('If you see this, your model does not fully load the pre-trained weight. Please make '
           'sure you have correctly specified --arch xxx or set the correct --num_classes for '
           'your own dataset.')
    for k in state_dict:
        if k in model_state_dict:
            if state_dict[k].shape != model_state_dict[k].shape:
                print('Skip loading parameter {}, required shape{}, loaded shape{}. {}'.format(
                    k, model_state_dict[k].shape, state_dict[k].shape, msg))
                state_dict[k] = model_state_dict[k]
        else:
            print('Drop parameter {}.'.format(k) + msg)
    for k in model_state_dict:
        if k not in state_dict:
            print('No param {}.'.format(k) + msg)
            state_dict[k] = model_state_dict[k]
    model.load_state_dict(state_dict, strict=False)
    return model


def save_model(path, epoch, model, optimizer=None):
    # model.py:86-95
    state_dict = model.module.state_dict() if isinstance(model, torch.nn.DataParallel) \
        else model.state_dict()
    data = {'epoch': epoch, 'state_dict': state_dict}
    if optimizer is not None:
        data['optimizer'] = optimizer.state_dict()
    torch.save(data, path)
