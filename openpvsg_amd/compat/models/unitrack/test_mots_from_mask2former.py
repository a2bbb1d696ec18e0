from openpvsg_amd.unitrack import eval_seq as _eval_seq
from .mask import MaskAssociationTracker


def eval_seq(data_cfg, tracker_cfg, outputs, classes, save_root, return_results=False, frames=None, app_model=None):
    """models/unitrack/test_mots_from_mask2former.py:29-34 signature (+ `frames`: the video's images, since the
    backend has no png reader; see openpvsg_amd.unitrack.LoadOutputsFromMask2Former)."""
    return _eval_seq(data_cfg, tracker_cfg, outputs, classes, save_root, return_results, frames=frames,
                     app_model=app_model, tracker_cls=MaskAssociationTracker)
