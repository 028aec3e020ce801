"""detector_factory (mirror of src/lib/detectors/detector_factory.py:10-15).
'exdet' and 'ddd' are outside the MI355X hot path."""
from .ctdet import CtdetDetector
from .multi_pose import MultiPoseDetector

detector_factory = {
    'ctdet': CtdetDetector,
    'multi_pose': MultiPoseDetector,
}
