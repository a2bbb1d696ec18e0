from . import query_feat_tracklet, single_video  # noqa: F401
