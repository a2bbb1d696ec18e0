"""The C-ABI library loads and exports every symbol include/openpvsg_hip.h declares (CPU only: no
compute calls), and the ctypes signature table covers the header."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'openpvsg_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(pvsg_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_something():
    fns = header_functions()
    assert 'pvsg_ms_deform_attn_forward' in fns and len(fns) >= 10


def test_library_exports_every_declared_symbol(hip_lib):
    for fn in header_functions():
        assert hasattr(hip_lib, fn), 'missing export: ' + fn
    assert hip_lib.pvsg_abi_version() == 11
    assert b'gfx950' in hip_lib.pvsg_version()


def test_signature_table_matches_header():
    from openpvsg_amd import _lib
    declared = set(header_functions()) - {'pvsg_last_error', 'pvsg_version', 'pvsg_abi_version'}
    assert declared == set(_lib.SIGNATURES)


def test_invalid_args_report_errors_without_gpu(hip_lib):
    # argument validation happens before any launch: exercisable on CPU
    rc = hip_lib.pvsg_ms_deform_attn_forward(None, None, None, None, None, None, 1, 1, 8, 32, 1, 3, 4, 64, None)
    assert rc == 1 and b'null pointer' in hip_lib.pvsg_last_error()
    one = ctypes.c_void_p(16)
    rc = hip_lib.pvsg_mask_logits_forward(one, one, one, 1, 1, 200, 256, 64, None)
    assert rc == 2 and b'Q<=112' in hip_lib.pvsg_last_error()
    rc = hip_lib.pvsg_pair_score_forward(one, one, one, one, one, one, one, None, one, 0, 4, 256, 1024, None)
    assert rc == 1
    # IPS association entries: sizes are checked before anything is launched
    assert hip_lib.pvsg_reconsdot_workspace_bytes(0, 10, 3, 10) == 0 and hip_lib.pvsg_reconsdot_workspace_bytes(2, 33, 3, 64) > 0
    rc = hip_lib.pvsg_reconsdot_cost(one, one, one, 2, 2000, 3, 64, ctypes.c_float(100.0), None, one, one, None)
    assert rc == 1 and b'1024 cells' in hip_lib.pvsg_last_error()
    rc = hip_lib.pvsg_reconsdot_cost(one, None, one, 2, 20, 3, 64, ctypes.c_float(100.0), None, one, one, None)
    assert rc == 1 and b'null pointer' in hip_lib.pvsg_last_error()
    counts = (ctypes.c_longlong * 5)(3, 2, 40, 1, 7)                  # host codec: runs of one mask
    seg = (ctypes.c_longlong * 1)(5)
    out, lens = (ctypes.c_ubyte * 80)(), (ctypes.c_longlong * 1)()
    n = hip_lib.pvsg_rle_counts_to_chars(counts, seg, 1, out, lens)
    from openpvsg_amd import tubes
    assert bytes(out[:n]).decode() == tubes.rle_counts_to_string([3, 2, 40, 1, 7]) and lens[0] == n
    assert hip_lib.pvsg_rle_counts_to_chars(None, None, 1, None, None) == -1


def test_cpu_tensors_are_rejected(hip_lib):
    import torch
    from openpvsg_amd import ops
    with pytest.raises(RuntimeError, match='no CPU path'):
        ops.mask_logits(torch.zeros(1, 100, 256), torch.zeros(1, 256, 8, 8))
    with pytest.raises(RuntimeError, match='no CPU path'):
        ops.pair_score(torch.zeros(3, 2, 256), torch.zeros(3, 2, 256), torch.zeros(1024, 512),
                       torch.zeros(1024), torch.zeros(1, 1024), torch.zeros(1))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from openpvsg_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.BackendMissingError):
        _lib.load()


def test_gemm_table_is_loadable_and_load_only():
    """openpvsg_amd/tuning: the committed hipBLASLt/rocBLAS selection table carries TunableOp validators and is
    handed over with tuning switched off (nothing is timed at run time)."""
    import torch
    from openpvsg_amd import tuning
    rows = [r.strip().split(',') for r in open(tuning.TABLE) if r.strip()]
    assert {r[1] for r in rows if r[0] == 'Validator'} >= {'PT_VERSION', 'GCN_ARCH_NAME', 'ROCBLAS_VERSION', 'HIPBLASLT_VERSION'}
    ops = [r for r in rows if r[0] != 'Validator']
    assert len(ops) >= 10 and all(r[0].startswith('Gemm') and '_float_' in r[0] for r in ops)
    assert any(r[0].startswith('GemmStridedBatched') for r in ops)    # the backbone's 1x1 convolutions as W @ x[b]
    assert any('618240' in r[1] for r in ops)                      # the 32 x 720p encoder shapes
    if not torch.cuda.is_available():
        assert tuning.enable() is False              # nothing to configure without a device
    elif getattr(torch.cuda, 'tunable', None) is not None:
        was = (torch.cuda.tunable.is_enabled(), torch.cuda.tunable.tuning_is_enabled(), torch.cuda.tunable.get_filename())
        try:
            assert tuning.enable() and torch.cuda.tunable.is_enabled() and not torch.cuda.tunable.tuning_is_enabled()
            assert torch.cuda.tunable.get_filename() == tuning.TABLE
        finally:
            torch.cuda.tunable.enable(was[0])
            torch.cuda.tunable.tuning_enable(was[1])
            torch.cuda.tunable.set_filename(was[2], insert_device_ordinal=False)
