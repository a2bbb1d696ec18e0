from openpvsg_amd import registry as R
from openpvsg_amd.registry import (build_backbone, build_detector, build_head,  # noqa: F401
                                   build_loss, build_neck)
from openpvsg_amd.compat._policy import protect

import openpvsg_amd.backbone, openpvsg_amd.blocks, openpvsg_amd.detectors  # noqa: F401,E401
import openpvsg_amd.fusion, openpvsg_amd.heads  # noqa: F401,E401

BACKBONES = protect(R.BACKBONES)
NECKS = protect(R.NECKS)
HEADS = protect(R.HEADS)
DETECTORS = protect(R.DETECTORS)
LOSSES = protect(R.LOSSES)
