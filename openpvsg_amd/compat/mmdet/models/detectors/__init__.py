from . import single_stage  # noqa: F401
