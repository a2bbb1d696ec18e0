"""`utils.relation_matching` names used on the inference side (tools/rel_test_full.py, datasets/datasets/
pvsg_relation.py:9,67).  The ground-truth tube matching that builds relation TRAINING sets
(match_tubes, match_and_process_gt_tubes, translate_gt_relations, ...) is outside the backend."""
import os
import pickle

import numpy as np

from openpvsg_amd.tubes import process_feats_and_relations, process_pairs, read_mots_results  # noqa: F401
from openpvsg_amd.tubes import process_feats as _process_feats


def load_pickle(filepath):
    with open(filepath, 'rb') as f:
        return pickle.load(f)


def save_pickle(filepath, data):
    with open(filepath, 'wb') as f:
        pickle.dump(data, f)


def get_pred_mask_tubes_one_video(vid, work_dir):
    return read_mots_results(os.path.join(work_dir, vid, 'quantitive', 'masks.txt'))


def process_feats(pred_feat_tubes, d=256):
    """accepts the reference's {tube id: [per-frame dict or None]} or the list of tracker objects"""
    if isinstance(pred_feat_tubes, dict):
        return process_feats_and_relations([], pred_feat_tubes, d)['feats']
    return _process_feats(pred_feat_tubes, d)


def calculate_iou(gt_mask, pred_mask):
    union = np.logical_or(gt_mask, pred_mask).sum()
    return 0 if union == 0 else np.logical_and(gt_mask, pred_mask).sum() / union
