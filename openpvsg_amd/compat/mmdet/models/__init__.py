import openpvsg_amd.backbone  # noqa: F401  (register the backend's modules before anything else)
import openpvsg_amd.blocks  # noqa: F401
import openpvsg_amd.detectors  # noqa: F401
import openpvsg_amd.fusion  # noqa: F401
import openpvsg_amd.heads  # noqa: F401
from . import builder, dense_heads, detectors, seg_heads, utils  # noqa: F401
from .builder import (BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, build_backbone,  # noqa: F401
                      build_detector, build_head, build_loss, build_neck)
