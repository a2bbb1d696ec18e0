from . import matching  # noqa: F401
