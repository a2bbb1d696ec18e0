from openpvsg_amd.blocks import MultiScaleDeformableAttention  # noqa: F401
from openpvsg_amd.compat._policy import _training_only

point_sample = _training_only('point_sample')
