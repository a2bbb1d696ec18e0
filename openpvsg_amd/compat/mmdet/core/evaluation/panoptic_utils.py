from openpvsg_amd.fusion import INSTANCE_OFFSET  # noqa: F401
