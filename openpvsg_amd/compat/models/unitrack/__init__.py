"""`models.unitrack.*` (tools/prepare_query_tube_ips.py:30, models/__init__.py:14-20): the IPS tube association,
answered by openpvsg_amd.unitrack.  Tracking evaluation (`eval/`), pose/box propagation and the other
appearance encoders of UniTrack are outside the backend."""
from . import basetrack, core, data, mask, model, multitracker, utils  # noqa: F401
from .test_mots_from_mask2former import eval_seq  # noqa: F401
