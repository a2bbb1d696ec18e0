from . import box, io, mask  # noqa: F401
