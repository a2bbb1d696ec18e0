from openpvsg_amd.relation import (generate_pairwise_results, generate_results,  # noqa: F401
                                   pick_top_pairs_eval)
