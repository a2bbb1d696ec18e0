"""`mmcv` names used by the reference's inference path, answered by openpvsg_amd."""
import json
import os
import pickle

from openpvsg_amd.config import Config, ConfigDict, DictAction  # noqa: F401
from . import cnn, ops, parallel, runner, utils  # noqa: F401
from .parallel import DataContainer  # noqa: F401

__version__ = '1.4.0'


def mkdir_or_exist(dir_name, mode=0o777):
    if dir_name:
        os.makedirs(os.path.expanduser(dir_name), mode=mode, exist_ok=True)


def dump(obj, file=None, file_format=None, **kwargs):
    fmt = file_format or (str(file).rsplit('.', 1)[-1] if file else 'pkl')
    if fmt in ('pkl', 'pickle'):
        if file is None:
            return pickle.dumps(obj, **kwargs)
        with open(file, 'wb') as f:
            pickle.dump(obj, f, **kwargs)
    elif fmt == 'json':
        if file is None:
            return json.dumps(obj, **kwargs)
        with open(file, 'w') as f:
            json.dump(obj, f, **kwargs)
    else:
        raise TypeError('unsupported format: %s' % fmt)


def load(file, file_format=None, **kwargs):
    fmt = file_format or str(file).rsplit('.', 1)[-1]
    if fmt in ('pkl', 'pickle'):
        with open(file, 'rb') as f:
            return pickle.load(f, **kwargs)
    if fmt == 'json':
        with open(file) as f:
            return json.load(f, **kwargs)
    raise TypeError('unsupported format: %s' % fmt)


def imread(path, flag='color', channel_order='bgr', backend=None):
    """Image decode through PIL (cv2 is not a dependency of the backend)."""
    import numpy as np
    from PIL import Image
    img = np.asarray(Image.open(path).convert('RGB' if flag == 'color' else 'L'))
    return img[..., ::-1].copy() if (flag == 'color' and channel_order == 'bgr') else img
