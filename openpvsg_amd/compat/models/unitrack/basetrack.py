from openpvsg_amd.unitrack import (BaseTrack, STrack, TrackState, joint_stracks,  # noqa: F401
                                    remove_duplicate_stracks, sub_stracks)
