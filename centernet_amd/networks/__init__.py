"""Parameter containers + plan descriptions for the CenterNet backbones.

State-dict names and shapes equal the reference's (src/lib/models/networks/*.py) so
model-zoo checkpoints load through ``centernet_amd.model.load_model``.
"""
