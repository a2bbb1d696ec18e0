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
    """[3P] RLE-encode instance masks (needs pycocotools, which the backend does not depend on)."""
    import pycocotools.mask as mask_util
    cls_segms = mask_results[0] if isinstance(mask_results, tuple) else mask_results
    return [[mask_util.encode(np.array(m[:, :, np.newaxis], order='F', dtype='uint8'))[0] for m in c] for c in cls_segms]
