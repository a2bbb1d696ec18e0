from . import registry, transformer  # noqa: F401
