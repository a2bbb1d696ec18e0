from openpvsg_amd.unitrack import LoadOutputsFromMask2Former  # noqa: F401
