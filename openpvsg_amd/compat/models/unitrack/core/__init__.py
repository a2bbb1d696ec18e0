from . import association, motion  # noqa: F401
