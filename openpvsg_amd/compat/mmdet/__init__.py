"""`mmdet` names used by the reference's inference path, answered by openpvsg_amd."""
__version__ = '2.25.0'
from . import apis, core, datasets, models, utils  # noqa: F401,E402
