"""Library-GEMM selection for the clip-size shapes of the path (hipBLASLt / rocBLAS solution per shape).

PyTorch's TunableOp timed every rocBLAS / hipBLASLt solution for the fp32 GEMMs `bench.py` issues at
32 x 720p frames (and, round 2, at 16 / 8 / 4 frames: what a rank holds when the clip is sharded over 2 / 4 / 8 GPUs) (encoder projections 618 240 x 256 -> 544 / 256 / 1024, FFN 1024 -> 256, decoder K/V projections
at 29 440 / 117 760 / 471 040 keys, the backbone's stride-1 1x1 convolutions as batched W @ x[b], ...) and the winners are stored in `gemm_gfx950.csv` (`scripts/tune_gemms.sh`
regenerates it, ~140 s).  `enable()` only LOADS that table: no tuning happens at run time, shapes that are not in it
and library versions that do not match the file's validators fall back to PyTorch's default choice.
Same arithmetic (fp32 in, fp32 accumulate); only the tiling of the library kernel changes.  +4 % end to end (187 -> 194 frames/s)."""
import os

import torch

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gemm_gfx950.csv')


def enable(table=TABLE):
    """Returns True when the table was handed to TunableOp."""
    t = getattr(torch.cuda, 'tunable', None)
    if t is None or not os.path.exists(table) or not torch.cuda.is_available():
        return False
    t.enable(True)
    t.tuning_enable(False)
    if hasattr(t, 'write_file_on_exit'):
        t.write_file_on_exit(False)          # load only: N ranks must not rewrite the tracked table at exit
    t.set_filename(table, insert_device_ordinal=False)
    return True
