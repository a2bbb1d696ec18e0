"""`models.mask2former_vps.utils` names (SimpleTracker is the class query_feats.pickle refers to)."""
from openpvsg_amd.tubes import SimpleTracker as _ST
from openpvsg_amd.tubes import concat_seq as _concat_seq


class SimpleTracker(_ST):
    pass


def concat_seq(outputs, save_root):
    return _concat_seq(outputs, save_root, tracker_cls=SimpleTracker)[0]
