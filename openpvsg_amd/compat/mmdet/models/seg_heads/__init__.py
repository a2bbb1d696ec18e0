from . import panoptic_fusion_heads  # noqa: F401
