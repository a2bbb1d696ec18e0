from openpvsg_amd.unitrack import QueryFeatTube as _QFT


class QueryFeatTube(_QFT):
    """defined here so that query_feats.pickle names models.unitrack.data.query_feat_tracklet.QueryFeatTube"""
