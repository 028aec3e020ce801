"""oracle/ -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  Nothing under ``centernet_amd/`` imports it, and the
product path raises if its HIP library is missing instead of falling back here.

Contents
--------
``dcn_v2_oracle.c``  scalar C restatement of DCNv2 forward (reference
                     ``DCNv2/src/cuda/dcn_v2_im2col_cuda.cu:18-47,118-180``,
                     ``DCNv2/src/dcn_v2_cuda.c:40-97``).  Parity pinned only by the
                     reference's zero-offset identity KAT + analytic cases
                     (the reference DCNv2 cannot be built here).
``decode_oracle.c``  scalar C restatement of ``_nms/_topk/_topk_channel/
                     ctdet_decode/multi_pose_decode`` (reference
                     ``models/decode.py:9-15,92-119,464-571``).  Parity pinned by
                     golden vectors produced by the reference's own Python on CPU
                     (``tests/golden/gen_golden.py``).
``cref.py``          ctypes binding of the two C files (numpy in / numpy out).
``net_oracle.py``    torch-CPU functional restatement of the reference network
                     graphs (``msra_resnet.py``, ``resnet_dcn.py``, ...) driven by a
                     state-dict, with DCN layers evaluated by ``dcn_v2_oracle.c``.
``post_oracle.py``   numpy restatement of ``ctdet_post_process`` / ``transform_preds``.
"""
