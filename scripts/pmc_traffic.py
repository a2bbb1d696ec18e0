"""Regenerates profiles/pmc_traffic.json: HBM-side bytes per launch of every hand-written kernel of ONE bench step.

Runs on the MI355X box.  Two separate rocprofv3 passes of `bench.py` (FETCH_SIZE and WRITE_SIZE do not fit one pass,
MI355X_MICROARCH.md "rocprofv3 PMC slots"), `--kernel-trace` only, then averages the counters per kernel function and
applies the gfx950 correction the guide prescribes (FETCH_SIZE tallies 128-byte requests at 64 B for 16 B/lane
streaming reads: x2; WRITE_SIZE as reported).  bench.py reads the file for `roofline.traffic`, keyed by C-ABI entry.

  python scripts/pmc_traffic.py [--frames 32] [--out profiles/pmc_traffic.json] [--keep-csv profiles/rNN_pmc]
"""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# kernel-function substring -> C-ABI entry (first match wins)
ENTRY = [('msda_fused', 'pvsg_msda_fused_forward'), ('msda_fwd', 'pvsg_ms_deform_attn_forward'),
         ('mask_gemm_kernel<0', 'pvsg_mask_logits_forward'), ('mask_gemm_kernel<1', 'pvsg_attn_mask_bits_forward'),
         ('xattn_partial', 'pvsg_masked_xattn_partial'), ('xattn_combine', 'pvsg_xattn_combine'),
         ('conv1x1_affine', 'pvsg_conv1x1_affine'), ('winograd_f2x3', 'pvsg_conv3x3_winograd'), ('conv3x3s2_kernel', 'pvsg_conv3x3s2_affine'), ('stem7x7_f16x2_kernel', 'pvsg_stem7x7_f16x2_bn_relu_pool'), ('stem7x7_kernel', 'pvsg_stem7x7_bn_relu_pool'), ('conv1x1_bf16x3_kernel', 'pvsg_conv1x1_bf16x3'), ('conv1x1_bf16x3_k32_kernel', 'pvsg_conv1x1_bf16x3'),
         ('gemm_bf16x3_kernel', 'pvsg_gemm_bf16x3'), ('gemm_bf16x3_k32_kernel', 'pvsg_gemm_bf16x3'), ('affine_act_nchw', 'pvsg_affine_act_nchw'),
         ('add_layernorm', 'pvsg_add_layernorm'), ('center_downsample', 'pvsg_center_downsample'),
         ('decoder_rows_post', 'pvsg_decoder_rows_post'), ('decoder_rows_pre', 'pvsg_decoder_rows_pre'),
         ('pan_owner', 'pvsg_panoptic_fuse_sel'), ('pan_decide', 'pvsg_panoptic_fuse_sel'), ('pan_paint', 'pvsg_panoptic_fuse_sel'),
         ('gn_partial', 'pvsg_group_norm_affine'), ('gn_finish', 'pvsg_group_norm_finish'), ('tube_index', 'pvsg_tube_index'),
         ('tube_scatter', 'pvsg_tube_scatter'), ('panoptic_select', 'pvsg_panoptic_select'), ('fpn_merge_up2x', 'pvsg_fpn_merge_up2x'),
         ('stem_bn_relu_pool', 'pvsg_stem_bn_relu_pool'), ('nchw_to_tokens', 'pvsg_nchw_to_tokens'),
         ('tokens_to_nchw', 'pvsg_tokens_to_nchw'), ('decoder_kv_inputs', 'pvsg_decoder_kv_inputs'), ('pair_', 'pvsg_pair_score_forward')]


def run_pass(counter, frames, workdir):
    out = os.path.join(workdir, counter)
    cmd = ['rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', out, '--',
           sys.executable, os.path.join(ROOT, 'bench.py'), '--frames', str(frames), '--steps', '1', '--warmup', '1',
           '--cpu-baseline', 'off', '--sub-benchmarks', 'off', '--no-kernel-timing', '--no-flop-count', '--graph', 'off', '--projection', 'off']
    env = dict(os.environ, TMPDIR='/tmp')
    subprocess.run(cmd, cwd='/tmp', env=env, check=True, stdout=subprocess.DEVNULL)
    files = glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise RuntimeError('no counter_collection.csv under ' + out)
    return files[0], ' '.join(cmd[:7]) + ' <dir> -- python bench.py ' + ' '.join(cmd[11:])


def split_entry(fn):
    """C-ABI entry of the split kernels (csrc/token_gemm.hip, conv1x1_split.hip, conv3x3_halo.hip) from their template arguments (round 4: the f16x2 forms)"""
    import re
    m = re.search(r'(\w+_kernel)<([^>]*)>', fn)
    if m:
        name, targs = m.group(1), [t.strip() for t in m.group(2).split(',')]
    else:                                            # rocprofv3 leaves the long instantiations mangled: ...kernelILb1ELb0E...Li128E...E
        m = re.search(r'(gemm_f16x2_t256_kernel|gemm_f16x2_ln128_kernel|gemm_f16x2_dma_kernel|gemm_bf16x3_k32_kernel|conv1x1_bf16x3_k32_kernel|'
                      r'conv3x3_f16x2_halo_kernel|bottleneck_tail64_kernel)I((?:L[bi]\d+E)+)E', fn)
        if not m:
            return None
        name = m.group(1)
        targs = [('true' if v == '1' else 'false') if t == 'b' else v for t, v in re.findall(r'L([bi])(\d+)E', m.group(2))]
    if name == 'gemm_f16x2_t256_kernel':
        return 'pvsg_gemm_f16x2_add_layernorm' if len(targs) > 1 and targs[1] == 'true' else 'pvsg_gemm_f16x2'
    if name == 'gemm_f16x2_ln128_kernel':            # <LN, RELU, KV>: round 5's 128 x 256 tile
        targs += ['false'] * (3 - len(targs))
        return 'pvsg_gemm_f16x2_add_layernorm' if targs[0] == 'true' else ('pvsg_decoder_kv_project_f16x2' if targs[2] == 'true' else 'pvsg_gemm_f16x2')
    if name == 'bottleneck_tail64_kernel':
        return 'pvsg_bottleneck_tail_f16x2'
    if name in ('gemm_f16x2_dma_kernel',):
        return 'pvsg_gemm_f16x2'
    if name == 'conv3x3_f16x2_halo_kernel':
        return 'pvsg_conv3x3_f16x2'
    if name == 'gemm_bf16x3_k32_kernel':
        return 'pvsg_gemm_f16x2' if len(targs) > 1 and targs[1] == 'true' else 'pvsg_gemm_bf16x3'
    if name == 'conv1x1_bf16x3_k32_kernel':          # <RELU, RESIDUAL, IN_NORM, BITS, TM, TAPS, F16>
        targs += ['false', '128', '1', 'false'][len(targs) - 3:] if len(targs) < 7 else []
        sfx = '_f16x2' if targs[6] == 'true' else '_bf16x3'
        if targs[3] == 'true':
            return 'pvsg_attn_mask_bits_packed_f16x2' if sfx == '_f16x2' else 'pvsg_attn_mask_bits' + sfx
        return ('pvsg_conv3x3' if targs[5] == '9' else 'pvsg_conv1x1') + sfx
    return None


def reduce_csv(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter or 'pvsg' not in r['Kernel_Name']:
            continue
        fn = r['Kernel_Name'].replace('void ', '')
        entry = split_entry(fn)
        if entry is None:
            entry = next((e for sub, e in ENTRY if sub in fn), None)
        if entry is not None:
            agg[entry].append((float(r['Counter_Value']), fn.split('(')[0]))
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'pmc_traffic.json'))
    ap.add_argument('--keep-csv', default=None, help='copy the two reduced counter tables here (prefix)')
    ap.add_argument('--workdir', default='/tmp/pvsg_pmc')
    a = ap.parse_args()
    fetch_csv, cmd_f = run_pass('FETCH_SIZE', a.frames, a.workdir)
    write_csv, cmd_w = run_pass('WRITE_SIZE', a.frames, a.workdir)
    fetch, write = reduce_csv(fetch_csv, 'FETCH_SIZE'), reduce_csv(write_csv, 'WRITE_SIZE')
    out = {}
    for entry in sorted(set(fetch) | set(write)):
        f = [v for v, _ in fetch.get(entry, [])]
        w = [v for v, _ in write.get(entry, [])]
        fk = sum(f) / len(f) if f else 0.0
        wk = sum(w) / len(w) if w else 0.0
        fns = sorted({n for _, n in fetch.get(entry, []) + write.get(entry, [])})
        out[entry] = dict(kernel_functions=fns, frames=a.frames, launches_averaged=len(f),
                          fetch_kib_raw=fk, write_kib=wk, fetch_correction='x2 (gfx950, 16 B/lane loads)',
                          hbm_read_bytes_per_launch=fk * 1024 * 2, hbm_write_bytes_per_launch=wk * 1024,
                          hbm_bytes_per_launch=fk * 1024 * 2 + wk * 1024)
    out['_source'] = dict(commands=[cmd_f, cmd_w], note='mean over every launch of the kernel in a 1 warm-up + 1 timed '
                          'step run; kernels with several shapes per step (conv1x1, affine, xattn) are '
                          'per-launch means over those shapes; pvsg_conv1x1_f16x2 / pvsg_conv3x3_f16x2 include their *_stats launches (same kernel function)')
    json.dump(out, open(a.out, 'w'), indent=1)
    if a.keep_csv:
        for tag, agg in (('FETCH_SIZE', fetch), ('WRITE_SIZE', write)):
            with open('%s_%s.csv' % (a.keep_csv, tag), 'w') as fh:
                wr = csv.writer(fh)
                wr.writerow(['entry', 'kernel_function', 'counter', 'mean_kib', 'launches'])
                for entry, vals in sorted(agg.items()):
                    wr.writerow([entry, vals[0][1], tag, round(sum(v for v, _ in vals) / len(vals), 3), len(vals)])
    print(json.dumps({k: round(v['hbm_bytes_per_launch'] / 1e6, 1) for k, v in out.items() if not k.startswith('_')}))


if __name__ == '__main__':
    main()
