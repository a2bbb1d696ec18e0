from openpvsg_amd.unitrack import mask2box  # noqa: F401
