from openpvsg_amd.relation import (calculate_final_metrics, calculate_iou,  # noqa: F401
                                   calculate_pair_recall_at_k)
