from openpvsg_amd.unitrack import (category_gate, fuse_motion, iou_distance, linear_assignment,  # noqa: F401
                                    reconsdot_distance)
