from . import kalman_filter  # noqa: F401
