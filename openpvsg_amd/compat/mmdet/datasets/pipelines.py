"""Pipeline names the reference's data code imports; the data path itself (cv2 I/O) is outside the
MI355X hot path -- only the light-weight pieces are implemented."""
import numpy as np
import torch

from .builder import PIPELINES
from openpvsg_amd.compat._policy import _training_only


def to_tensor(data):
    if isinstance(data, torch.Tensor):
        return data
    if isinstance(data, np.ndarray):
        return torch.from_numpy(data)
    if isinstance(data, (list, tuple)):
        return torch.tensor(data)
    if isinstance(data, int):
        return torch.LongTensor([data])
    if isinstance(data, float):
        return torch.FloatTensor([data])
    raise TypeError('type %s cannot be converted to tensor.' % type(data))


class Compose:
    def __init__(self, transforms):
        from openpvsg_amd.registry import build_from_cfg
        self.transforms = [build_from_cfg(t, PIPELINES) if isinstance(t, dict) else t for t in transforms]

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data


class _Placeholder:
    def __init__(self, *a, **k):
        raise NotImplementedError('%s: image I/O pipelines are outside the backend (feed tensors)' % type(self).__name__)


class Resize(_Placeholder): pass            # noqa: E701
class RandomFlip(_Placeholder): pass        # noqa: E701
class Pad(_Placeholder): pass               # noqa: E701
class Normalize(_Placeholder): pass         # noqa: E701
class LoadAnnotations(_Placeholder): pass   # noqa: E701
class LoadImageFromFile(_Placeholder): pass # noqa: E701
