"""Task name -> detector class, the lookup the reference's users go through
(``detector_factory[opt.task](opt)``, src/lib/detectors/detector_factory.py:10-15).
Only the tasks on the MI355X hot path are registered; the decoders of 'ddd' and 'exdet' exist
(centernet_amd.decode) but their detector classes are not built."""
from . import ctdet, multi_pose

detector_factory = dict(ctdet=ctdet.CtdetDetector, multi_pose=multi_pose.MultiPoseDetector)
