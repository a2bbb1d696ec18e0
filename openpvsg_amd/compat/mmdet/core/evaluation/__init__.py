from . import panoptic_utils  # noqa: F401
