"""Configuration flags (mirror of src/lib/opts.py; same flag names, defaults and derived
fields so ``opts().init([...])`` is call-compatible).

Training / dataset / debug-visualisation flags are accepted so that existing command
lines parse, but only the inference-relevant ones are consumed by this package:
task, --arch, --head_conv, --down_ratio, --input_res/_h/_w, --load_model, --gpus, --K,
--flip_test, --test_scales, --nms, --keep_res/--fix_res, --cat_spec_wh, --not_reg_offset,
--not_hm_hp, --not_reg_hp_offset, --debug, --vis_thresh; for the ddd task --not_reg_bbox and
--peak_thresh, for exdet --scores_thresh, --center_thresh, --aggr_weight, --agnostic_ex.
"""
import argparse
import os

# (flag, kwargs) -- table form; names and defaults as in opts.py:13-225
_FLAGS = [
    ("task", dict(default="ctdet", nargs="?", help="ctdet | multi_pose | ddd | exdet")),
    ("--dataset", dict(default="coco")), ("--exp_id", dict(default="default")),
    ("--test", dict(action="store_true")), ("--debug", dict(type=int, default=0)),
    ("--demo", dict(default="")), ("--load_model", dict(default="")),
    ("--resume", dict(action="store_true")), ("--gpus", dict(default="0", help="-1 for CPU (unsupported here: raises)")),
    ("--num_workers", dict(type=int, default=4)), ("--not_cuda_benchmark", dict(action="store_true")),
    ("--seed", dict(type=int, default=317)), ("--print_iter", dict(type=int, default=0)),
    ("--hide_data_time", dict(action="store_true")), ("--save_all", dict(action="store_true")),
    ("--metric", dict(default="loss")), ("--vis_thresh", dict(type=float, default=0.3)),
    ("--debugger_theme", dict(default="white", choices=["white", "black"])),
    ("--arch", dict(default="dla_34", help="res_18 | res_101 | resdcn_18 | resdcn_101 | dla_34 | hourglass")),
    ("--head_conv", dict(type=int, default=-1)), ("--down_ratio", dict(type=int, default=4)),
    ("--input_res", dict(type=int, default=-1)), ("--input_h", dict(type=int, default=-1)),
    ("--input_w", dict(type=int, default=-1)),
    ("--lr", dict(type=float, default=1.25e-4)), ("--lr_step", dict(type=str, default="90,120")),
    ("--num_epochs", dict(type=int, default=140)), ("--batch_size", dict(type=int, default=32)),
    ("--master_batch_size", dict(type=int, default=-1)), ("--num_iters", dict(type=int, default=-1)),
    ("--val_intervals", dict(type=int, default=5)), ("--trainval", dict(action="store_true")),
    ("--flip_test", dict(action="store_true")), ("--test_scales", dict(type=str, default="1")),
    ("--nms", dict(action="store_true")), ("--K", dict(type=int, default=100)),
    ("--not_prefetch_test", dict(action="store_true")), ("--fix_res", dict(action="store_true")),
    ("--keep_res", dict(action="store_true")), ("--not_rand_crop", dict(action="store_true")),
    # new (not in the reference): keep pre_process on the host instead of the device kernels
    ("--host_pre_process", dict(action="store_true")),
    # new: compute on the plain fp32 matrix instruction instead of f32s (three fp16 MFMAs per
    # product on range-controlled fp16 pairs) -- the A/B reference mode
    ("--fp32_mfma", dict(action="store_true")),
    ("--shift", dict(type=float, default=0.1)), ("--scale", dict(type=float, default=0.4)),
    ("--rotate", dict(type=float, default=0)), ("--flip", dict(type=float, default=0.5)),
    ("--no_color_aug", dict(action="store_true")), ("--aug_rot", dict(type=float, default=0)),
    ("--aug_ddd", dict(type=float, default=0.5)), ("--rect_mask", dict(action="store_true")),
    ("--kitti_split", dict(default="3dop")), ("--mse_loss", dict(action="store_true")),
    ("--reg_loss", dict(default="l1")), ("--hm_weight", dict(type=float, default=1)),
    ("--off_weight", dict(type=float, default=1)), ("--wh_weight", dict(type=float, default=0.1)),
    ("--hp_weight", dict(type=float, default=1)), ("--hm_hp_weight", dict(type=float, default=1)),
    ("--dep_weight", dict(type=float, default=1)), ("--dim_weight", dict(type=float, default=1)),
    ("--rot_weight", dict(type=float, default=1)), ("--peak_thresh", dict(type=float, default=0.2)),
    ("--norm_wh", dict(action="store_true")), ("--dense_wh", dict(action="store_true")),
    ("--cat_spec_wh", dict(action="store_true")), ("--not_reg_offset", dict(action="store_true")),
    ("--agnostic_ex", dict(action="store_true")), ("--scores_thresh", dict(type=float, default=0.1)),
    ("--center_thresh", dict(type=float, default=0.1)), ("--aggr_weight", dict(type=float, default=0.0)),
    ("--dense_hp", dict(action="store_true")), ("--not_hm_hp", dict(action="store_true")),
    ("--not_reg_hp_offset", dict(action="store_true")), ("--not_reg_bbox", dict(action="store_true")),
    ("--eval_oracle_hm", dict(action="store_true")), ("--eval_oracle_wh", dict(action="store_true")),
    ("--eval_oracle_offset", dict(action="store_true")), ("--eval_oracle_kps", dict(action="store_true")),
    ("--eval_oracle_hmhp", dict(action="store_true")), ("--eval_oracle_hp_offset", dict(action="store_true")),
    ("--eval_oracle_dep", dict(action="store_true")),
]

_DATASET_DEFAULTS = {  # opts.py:337-353
    "ctdet": dict(default_resolution=[512, 512], num_classes=80, mean=[0.408, 0.447, 0.470],
                  std=[0.289, 0.274, 0.278], dataset="coco"),
    "multi_pose": dict(default_resolution=[512, 512], num_classes=1, mean=[0.408, 0.447, 0.470],
                       std=[0.289, 0.274, 0.278], dataset="coco_hp", num_joints=17,
                       flip_idx=[[1, 2], [3, 4], [5, 6], [7, 8], [9, 10], [11, 12], [13, 14],
                                 [15, 16]]),
    "exdet": dict(default_resolution=[512, 512], num_classes=80, mean=[0.408, 0.447, 0.470],
                  std=[0.289, 0.274, 0.278], dataset="coco"),
    "ddd": dict(default_resolution=[384, 1280], num_classes=3, mean=[0.485, 0.456, 0.406],
                std=[0.229, 0.224, 0.225], dataset="kitti"),
}


class _Struct:
    def __init__(self, entries):
        for k, v in entries.items():
            setattr(self, k, v)


class opts(object):
    def __init__(self):
        self.parser = argparse.ArgumentParser()
        for flag, kw in _FLAGS:
            self.parser.add_argument(flag, **kw)

    def parse(self, args=''):
        # opts.py:227-282
        opt = self.parser.parse_args() if args == '' else self.parser.parse_args(args)
        # exdet scores K^4 groupings per image; the kernel takes K <= 64 and ExtremeNet's own setting is 40.  A
        # command line that does not pass --K (the reference default of 100) runs at 40 instead of failing.
        import sys as _sys
        argv = _sys.argv[1:] if args == '' else list(args)
        if opt.task == 'exdet' and not any(a == '--K' or str(a).startswith('--K=') for a in argv):
            opt.K = 40
        opt.gpus_str = opt.gpus
        opt.gpus = [int(g) for g in opt.gpus.split(',')]
        opt.gpus = [i for i in range(len(opt.gpus))] if opt.gpus[0] >= 0 else [-1]
        opt.lr_step = [int(i) for i in opt.lr_step.split(',')]
        opt.test_scales = [float(i) for i in opt.test_scales.split(',')]
        opt.fix_res = not opt.keep_res
        opt.reg_offset = not opt.not_reg_offset
        opt.reg_bbox = not opt.not_reg_bbox
        opt.hm_hp = not opt.not_hm_hp
        opt.reg_hp_offset = (not opt.not_reg_hp_offset) and opt.hm_hp
        if opt.head_conv == -1:
            opt.head_conv = 256 if 'dla' in opt.arch else 64
        opt.pad = 127 if 'hourglass' in opt.arch else 31
        opt.num_stacks = 2 if opt.arch == 'hourglass' else 1
        # Derived fields reference-style scripts read off the namespace (opts.py:251-281): the
        # debug overrides, the per-GPU batch split (first GPU's share, the rest spread evenly,
        # remainder to the lowest ranks), and the experiment directories under the package root.
        if opt.trainval:
            opt.val_intervals = 100000000
        if opt.debug > 0:
            opt.num_workers, opt.batch_size, opt.master_batch_size = 0, 1, -1
            opt.gpus = opt.gpus[:1]
        n = len(opt.gpus)
        if opt.master_batch_size == -1:
            opt.master_batch_size = opt.batch_size // n
        rest = opt.batch_size - opt.master_batch_size
        opt.chunk_sizes = [opt.master_batch_size] + \
            [rest // (n - 1) + (1 if i < rest % (n - 1) else 0) for i in range(n - 1)]
        opt.root_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
        opt.data_dir = os.path.join(opt.root_dir, 'data')
        opt.exp_dir = os.path.join(opt.root_dir, 'exp', opt.task)
        opt.save_dir = os.path.join(opt.exp_dir, opt.exp_id)
        opt.debug_dir = os.path.join(opt.save_dir, 'debug')
        if opt.resume and opt.load_model == '':
            base = opt.save_dir[:-4] if opt.save_dir.endswith('TEST') else opt.save_dir
            opt.load_model = os.path.join(base, 'model_last.pth')
        return opt

    def update_dataset_info_and_set_heads(self, opt, dataset):
        # opts.py:284-334
        input_h, input_w = dataset.default_resolution
        opt.mean, opt.std = dataset.mean, dataset.std
        opt.num_classes = dataset.num_classes
        input_h = opt.input_res if opt.input_res > 0 else input_h
        input_w = opt.input_res if opt.input_res > 0 else input_w
        opt.input_h = opt.input_h if opt.input_h > 0 else input_h
        opt.input_w = opt.input_w if opt.input_w > 0 else input_w
        opt.output_h = opt.input_h // opt.down_ratio
        opt.output_w = opt.input_w // opt.down_ratio
        opt.input_res = max(opt.input_h, opt.input_w)
        opt.output_res = max(opt.output_h, opt.output_w)
        if opt.task == 'ctdet':
            opt.heads = {'hm': opt.num_classes,
                         'wh': 2 if not opt.cat_spec_wh else 2 * opt.num_classes}
            if opt.reg_offset:
                opt.heads.update({'reg': 2})
        elif opt.task == 'multi_pose':
            opt.flip_idx = dataset.flip_idx
            opt.heads = {'hm': opt.num_classes, 'wh': 2, 'hps': 34}
            if opt.reg_offset:
                opt.heads.update({'reg': 2})
            if opt.hm_hp:
                opt.heads.update({'hm_hp': 17})
            if opt.reg_hp_offset:
                opt.heads.update({'hp_offset': 2})
        elif opt.task == 'ddd':
            opt.heads = {'hm': opt.num_classes, 'dep': 1, 'rot': 8, 'dim': 3}
            if opt.reg_bbox:
                opt.heads.update({'wh': 2})
            if opt.reg_offset:
                opt.heads.update({'reg': 2})
        elif opt.task == 'exdet':
            edge_maps = 1 if opt.agnostic_ex else opt.num_classes
            opt.heads = {'hm_t': edge_maps, 'hm_l': edge_maps, 'hm_b': edge_maps, 'hm_r': edge_maps,
                         'hm_c': opt.num_classes}
            if opt.reg_offset:
                opt.heads.update({'reg_t': 2, 'reg_l': 2, 'reg_b': 2, 'reg_r': 2})
        else:
            raise NotImplementedError("task '%s' is not defined (ctdet, multi_pose, ddd, exdet)" % opt.task)
        return opt

    def init(self, args=''):
        # opts.py:336-362
        opt = self.parse(args)
        if opt.task not in _DATASET_DEFAULTS:
            raise NotImplementedError("task '%s' is not defined (ctdet, multi_pose, ddd, exdet)" % opt.task)
        dataset = _Struct(_DATASET_DEFAULTS[opt.task])
        opt.dataset = dataset.dataset
        opt = self.update_dataset_info_and_set_heads(opt, dataset)
        return opt
