from . import base_panoptic_fusion_head  # noqa: F401
