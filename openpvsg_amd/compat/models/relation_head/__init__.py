from . import base, convolution, test_utils, train_utils, transformer  # noqa: F401
