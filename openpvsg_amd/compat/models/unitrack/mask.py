from openpvsg_amd.unitrack import MaskAssociationTracker as _Tracker
from .data.query_feat_tracklet import QueryFeatTube


class MaskAssociationTracker(_Tracker):
    tube_cls = QueryFeatTube
