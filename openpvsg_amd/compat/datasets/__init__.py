"""`from datasets import PVSGRelationDataset` (tools/rel_test.py:4): the relation-set reader whose item
layout is part of the rel_test boundary.  Image / video datasets (cv2 I/O) are outside
the backend."""
import json
import os
import pickle

import numpy as np


class PVSGRelationDataset:
    """datasets/datasets/pvsg_relation.py:15-79 (return_mask=False path): one item per video =
    {'feats': float64 [N,T,256], 'relations': [...indices remapped to 0..N-1...], 'pairs', 'vid'}."""

    def __init__(self, anno_file, split='train', work_dir='./work_dirs/train_save_qf_1106', return_mask=False):
        if return_mask:
            raise NotImplementedError('mask tubes (MOTS files) are read by tools/rel_test_full.py only')
        with open(anno_file, 'r') as f:
            anno = json.load(f)
        self.video_ids = [v for src in ('vidor', 'epic_kitchen', 'ego4d') for v in anno['split'][src][split]]
        self.work_dir, self.split = work_dir, split
        self.classes = anno['objects']['thing'] + anno['objects']['stuff']
        self.relations = anno['relations']
        self.videos = {v['video_id']: v for v in anno['data']}

    def __len__(self):
        return len(self.video_ids)

    def __getitem__(self, index):
        vid = self.video_ids[index]
        with open(os.path.join(self.work_dir, vid, 'relations.pickle'), 'rb') as f:
            item = pickle.load(f)
        item['vid'] = vid
        order = list(item['feats'])
        remap = {key: i for i, key in enumerate(order)}
        item['feats'] = np.array([item['feats'][k] for k in order])
        for rel in item['relations']:
            rel['subject_index'] = remap[rel['subject_index']]
            rel['object_index'] = remap[rel['object_index']]
        item['pairs'] = [[r['subject_index'], r['object_index']] for r in item['relations']]
        return item
