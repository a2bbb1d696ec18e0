from openpvsg_amd.fusion import mask2bbox  # noqa: F401
