import csv
import os

import numpy as np


def save_metrics_to_csv(final_metrics, pair_recall_list, K_values, csv_file_path, model_name):
    """utils/show_log.py:5-31: one row per evaluation, header on first write."""
    new = not os.path.isfile(csv_file_path)
    os.makedirs(os.path.dirname(csv_file_path) or '.', exist_ok=True)
    header = ['Model', 'Pair Recall'] + ['R/mR@%d' % K for K in K_values] + ['wR/wmR@%d' % K for K in K_values]
    row = [model_name, '%.2f' % (100 * np.array(pair_recall_list).mean())]
    row += ['%.2f/%.2f' % (100 * final_metrics[K]['recall'], 100 * final_metrics[K]['mean_recall']) for K in K_values]
    row += ['%.2f/%.2f' % (100 * final_metrics[K]['weak_recall'], 100 * final_metrics[K]['weak_mean_recall'])
            for K in K_values]
    with open(csv_file_path, mode='a', newline='') as f:
        w = csv.writer(f)
        if new:
            w.writerow(header)
        w.writerow(row)
