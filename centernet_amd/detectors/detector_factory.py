"""Task name -> detector class, the lookup the reference's users go through
(``detector_factory[opt.task](opt)``, src/lib/detectors/detector_factory.py:10-15)."""
from . import ctdet, ddd, exdet, multi_pose

detector_factory = dict(ctdet=ctdet.CtdetDetector, multi_pose=multi_pose.MultiPoseDetector,
                        ddd=ddd.DddDetector, exdet=exdet.ExdetDetector)
