import numpy as np

from openpvsg_amd.detectors import bbox2result  # noqa: F401
from openpvsg_amd.fusion import INSTANCE_OFFSET  # noqa: F401
from openpvsg_amd.compat._policy import _training_only
from . import evaluation, mask, visualization  # noqa: F401

build_assigner = lambda *a, **k: None    # noqa: E731  (train_cfg is None at inference)
build_sampler = lambda *a, **k: None     # noqa: E731
reduce_mean = _training_only('reduce_mean')
BitmapMasks = _training_only('BitmapMasks')


def multi_apply(func, *args, **kwargs):
    from functools import partial
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


def encode_mask_results(mask_results):
    """[3P] mmdet.core.encode_mask_results: per-class lists of binary masks -> lists of COCO RLE dicts.  The backend's own
    codec (openpvsg_amd/tubes.py; device-side run boundaries for masks that are still on the GPU) -- pycocotools is not a
    dependency."""
    from openpvsg_amd.detectors import encode_mask_results as _enc
    return _enc(mask_results)
