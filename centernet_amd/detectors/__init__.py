from .detector_factory import detector_factory  # noqa: F401
