"""Disassemble every gfx950 code object of the built library and look for the write-after-read pattern that corrupted
bottleneck_tail64_kernel in round 5: a 16-bit-input MFMA (v_mfma_f32_16x16x32_{f16,bf16},
v_mfma_f32_32x32x16_{f16,bf16}) whose A or B source registers are overwritten by a PACKED-f32 VALU instruction (v_pk_mul_f32,
v_pk_add_f32, v_pk_fma_f32, v_pk_mov_b32: 64-bit destinations, issued over two passes) within `window` issue slots behind it,
with no other MFMA in between.  The matrix core reads A / B over several passes and hipcc 7.2's hazard recogniser guards SrcC
only; ordinary 32-bit VALU writes behind an MFMA (v_cvt_pk_bf16_f32, v_cndmask, v_and ... -- the bf16x3 GEMMs and the masked
attention are full of them, `packed_only=False` lists them) have been bit-exact over every round's repeat tests, the packed
form is the one that was caught corrupting lanes 48-63.  Test infrastructure (tests/test_mfma_hazard.py); the lab form is scripts/lab/mfma_war_scan.py."""
import os
import re
import shutil
import subprocess
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
_REG = re.compile(r'v\[(\d+):(\d+)\]|v(\d+)')
_LABEL = re.compile(r'^[0-9a-f]+ <(.+)>:$')
_MFMA16 = ('v_mfma_f32_16x16x32_f16', 'v_mfma_f32_16x16x32_bf16', 'v_mfma_f32_32x32x16_f16', 'v_mfma_f32_32x32x16_bf16')


def _regs(tok):
    m = _REG.fullmatch(tok.strip())
    if not m:
        return set()
    if m.group(1):
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return {int(m.group(3))}


def disassemble(lib_path):
    """-> list of (kernel, instruction text) over all gfx950 code objects embedded in the shared library"""
    tmp = tempfile.mkdtemp(prefix='pvsg_scan_')
    try:
        so = os.path.join(tmp, os.path.basename(lib_path))
        shutil.copy(lib_path, so)                      # --offloading drops the bundles next to its input
        subprocess.run([OBJDUMP, '--offloading', so], cwd=tmp, check=True, capture_output=True)
        ins = []
        for f in sorted(os.listdir(tmp)):
            if 'gfx950' not in f:
                continue
            out = subprocess.run([OBJDUMP, '-d', os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            kernel = None
            for ln in out.split('\n'):
                t = ln.strip()
                m = _LABEL.match(t)
                if m:
                    kernel = m.group(1)
                    continue
                if not t or kernel is None:
                    continue
                t = t.split('//')[0].strip()
                if t:
                    ins.append((kernel, t))
        return ins
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


_PACKED = ('v_pk_mul_f32', 'v_pk_add_f32', 'v_pk_fma_f32', 'v_pk_mov_b32')


def scan(ins, window=6, packed_only=True):
    """-> {kernel: [(mfma, overwriting instruction, distance)]}"""
    hits = {}
    for i, (k, t) in enumerate(ins):
        if not t.startswith(_MFMA16):
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(',')]
        src = _regs(ops[1]) | _regs(ops[2])
        n = 0
        for k2, u in ins[i + 1:i + 1 + 4 * window]:
            if k2 != k or u.startswith(('v_mfma', 's_endpgm', 's_cbranch', 's_branch', 's_setpc')):
                break
            if u.startswith('s_nop'):
                n += int(u.split()[1], 0) + 1
            elif u.startswith('v_'):
                dst = u.split(None, 1)[1].split(',')[0] if ' ' in u else ''
                if _regs(dst) & src and (not packed_only or u.startswith(_PACKED)):
                    hits.setdefault(k, []).append((t, u, n))
                n += 1
            else:
                n += 1
            if n >= window:
                break
    return hits
