"""`import models` (tools/test.py:26): importing registers the backend's detectors / heads under the names
the reference's configs use.  The training-only head variants are not part of the backend."""
import openpvsg_amd.backbone  # noqa: F401
import openpvsg_amd.blocks  # noqa: F401
import openpvsg_amd.detectors  # noqa: F401
import openpvsg_amd.fusion  # noqa: F401
import openpvsg_amd.heads  # noqa: F401
from openpvsg_amd.blocks import SinePositionalEncoding3D  # noqa: F401
from openpvsg_amd.detectors import (Mask2FormerCustom, Mask2FormerVideoCustom,  # noqa: F401
                                    Mask2FormerVideoCustomMinVIS)
from openpvsg_amd.fusion import MaskFormerFusionHeadCustom  # noqa: F401
from openpvsg_amd.heads import Mask2FormerHeadCustom, Mask2FormerVideoHead  # noqa: F401
from . import mask2former_vps, relation_head, unitrack  # noqa: F401
from .unitrack.test_mots_from_mask2former import eval_seq  # noqa: F401
