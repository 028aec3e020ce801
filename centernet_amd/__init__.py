"""centernet_amd -- MI355X-native (gfx950) CenterNet inference hot path.

Host side mirrors the reference's interface for this path
(``opts().init`` -> ``detector_factory[opt.task](opt).run(img)``); the compute is
hand-written HIP behind the C ABI of ``include/centernet_amd.h``.  No CPU fallback.
"""
__version__ = "0.1.0"
