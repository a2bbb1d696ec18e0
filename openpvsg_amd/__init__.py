"""openpvsg_amd -- MI355X (gfx950) backend for the OpenPVSG inference hot path.

Layout:
  csrc/        hand-written HIP kernels + the C ABI declared in include/openpvsg_hip.h
  _lib.py      ctypes loader for lib/libopenpvsg_hip.so (fails loudly when it is missing)
  ops.py       tensor-level wrappers (device pointers + current HIP stream -> C ABI)
  (modules)    host-side mirror of the mmcv/mmdet registry surface the reference configs name
"""
__version__ = '0.1.0'
