from openpvsg_amd.config import Config, ConfigDict, DictAction  # noqa: F401
from openpvsg_amd.registry import Registry, build_from_cfg  # noqa: F401
