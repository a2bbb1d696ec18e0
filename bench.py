#!/usr/bin/env python
"""bench.py -- frames/sec of the 720p 32-frame VPS + relation forward on N MI355X (BASELINE.json).

One "step" = one pass of the hot path over one synthetic clip already resident in HBM:
  ResNet-50 -> MSDeformAttn pixel decoder -> clip-level masked-attention decoder (keys = T*h*w)
  -> last-layer mask logits -> per-frame x4 up-sampling + panoptic fusion -> tube assembly ->
  relation head (object encoders, N^2 pair scorer, top-100 pairs, temporal transformer).
N > 1 (default --scaling strong = BASELINE config 4): ONE 32-frame clip sharded by frame, 32/N frames per GPU; the
attention partials are merged across the ranks every decoder layer and the per-frame segment records are
all-gathered (RCCL) before tube assembly.  `python bench.py --gpus N` without a launcher re-executes itself under
torch.distributed.run with N ranks.  --scaling weak gives every rank its own 32-frame segment of a longer video.

Prints ONE JSON line (rank 0):
  roofline            dominant HAND-WRITTEN kernel (by time), HIP events around its launches inside the timed steps (the table
                      of all entries, `kernels`, is taken the same way during the last warm-up step; --timing all)
  roofline_step       the whole step against the f32 matrix peak: algorithmic FLOP (library ops counted by
                      torch.utils.flop_counter on one extra untimed step + the hand-written kernels' own counts)
                      / 157.3 TF / ms_per_step, with the hand-written / other (library + gaps) time split
  cpu_baseline        the CPU oracle (oracle/, kind "port") on a bounded sample of the same workload on this box's
                      host cores: 1 warm-up + `--cpu-reps` timed repetitions (`--cpu-full`: also the full clip, minutes)
  sub_benchmarks      relation head at N in {32,64,100} tubes and fused post-processing at K in {10,30} kept
                      queries (SURVEY.md section 8d / BASELINE.md section 2)

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TF = 157.3     # f32-input MFMA peak
BF16_MFMA_PEAK_TF = 2516.6   # dense bf16 MFMA peak (16 x the f32 matrix rate)
# the library's switch for the masked cross-attention kernel (csrc/masked_xattn.hip): anything starting with 'f' = f32 MFMA
XATTN_F32 = os.environ.get('PVSG_XATTN', '')[:1] == 'f' or os.environ.get('PVSG_XATTN_LEAN', '')[:1] == '0'
CLS_GAIN = 40.0              # random-init class logits are flat; peaky logits keep a few queries


def host_cores():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box
    exposes 256 logical CPUs but a 16-CPU quota; 256 torch threads then run 50x slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def ops_split_mode():
    from openpvsg_amd import ops
    return ops.split_mode()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mode', default='vps', choices=['vps', 'ips'],
                    help='vps: BASELINE config 3/4/5 flow (clip-level VPS + relation head, the headline); ips: BASELINE config 2 '
                         '(Mask2Former R50 IPS detector on a batch of --frames frames, per-frame decoder, fused panoptic fusion)')
    ap.add_argument('--frames', type=int, default=32)
    ap.add_argument('--height', type=int, default=720)
    ap.add_argument('--width', type=int, default=1280)
    ap.add_argument('--cpu-baseline', default='auto', choices=['auto', 'off'])
    ap.add_argument('--cpu-frames', type=int, default=2, help='frames of the bounded CPU-oracle sample')
    ap.add_argument('--cpu-reps', type=int, default=3, help='timed repetitions of the CPU sample after 1 warm-up')
    ap.add_argument('--cpu-full', action='store_true', help='also time the oracle on the full clip (minutes)')
    ap.add_argument('--head-outputs', default='synthetic', choices=['synthetic', 'model'],
                    help='synthetic: class logits / mask-logit offsets with a controlled keep count (--keep) are applied '
                         'to the head outputs so that ~--keep tubes reach the relation head; model: raw random-init outputs')
    ap.add_argument('--keep', type=int, default=32)
    ap.add_argument('--sub-benchmarks', default='on', choices=['on', 'off'])
    ap.add_argument('--no-flop-count', action='store_true', help='skip the extra untimed flop-counting step (profiling runs)')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--timing', choices=['dominant', 'all'], default='dominant',
                    help='HIP-event pairs inside the timed region: around the dominant C-ABI entry only (the per-kernel table then '
                         'comes from the last warm-up step, where every launch is timed) or around every launch (1.7 ms per step)')
    ap.add_argument('--scaling', default='strong', choices=['weak', 'strong'],
                    help='N>1: strong (default, BASELINE config 4) = ONE 32-frame clip sharded by frame, 32/N frames per '
                         'GPU, attention partials merged across GPUs every decoder layer; weak = one 32-frame segment '
                         'of a longer video per GPU, tube records all-gathered')
    ap.add_argument('--backend', default='nccl', help='nccl (= RCCL over xGMI); gloo only for same-device logic tests')
    ap.add_argument('--graph', choices=('auto', 'on', 'off'), nargs='?', const='on', default='auto',
                    help='replay backbone + head as one hipGraph: auto = clips of <= 8 frames (host-bound)')
    ap.add_argument('--projection', default='on', choices=['on', 'off'],
                    help='N = 1 only: also time the T/N-frame steps (N = 2, 4, 8) and report the projected strong-scaling efficiency')
    ap.add_argument('--checksum', action='store_true', help='add a result checksum (sharding-invariance check)')
    return ap.parse_args()


def make_clip(T, H, W, seed=0):
    """seed-0 uint8 U[0,255] pixels, ImageNet normalisation of the PVSG test pipeline
    (configs/_base_/datasets/pvsg_vps_single_video_test.py:4-6), zero-padded to a multiple of 32."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (T, 3, H, W), generator=g, dtype=torch.uint8).float()
    mean = torch.tensor([123.675, 116.28, 103.53]).view(1, 3, 1, 1)
    std = torch.tensor([58.395, 57.12, 57.375]).view(1, 3, 1, 1)
    img = (img - mean) / std
    Hp, Wp = (H + 31) // 32 * 32, (W + 31) // 32 * 32
    out = torch.zeros(T, 3, Hp, Wp)
    out[:, :, :H, :W] = img
    return out, (Hp, Wp)


def synthetic_head_outputs(T, h4, w4, num_queries=100, num_classes=126, n_keep=32, seed=0, t0=0, T_total=None):
    """Controlled keep-count (random-init weights give flat class logits and noise-like masks,
    so `score > 0.8` + the area filter of mask2former_fusion_head.py:139-169 would leave ~2 segments and the relation
    head would be timed on 2 tubes).  -> class logits (1,Q,classes+1) with `n_keep` confident queries (two thirds
    things, one third stuff) and additive mask-logit offsets (T,Q,h4,w4): +/-40 on a drifting rectangle per kept
    query (some absent in some frames), 0 for the others.  The decoder's own outputs are still computed and the
    offsets are ADDED to its mask logits inside the timed step."""
    g = torch.Generator().manual_seed(1000 + seed)
    T_total = T_total or T
    cls = torch.randn(1, num_queries, num_classes + 1, generator=g)
    cls[..., num_classes] += 8.0                                   # background for everyone ...
    kept = torch.randperm(num_queries, generator=g)[:n_keep]
    off = torch.zeros(T, num_queries, h4, w4)
    gy, gx = 4, (n_keep + 3) // 4
    for i, q in enumerate(kept.tolist()):
        c = int(torch.randint(0, 115, (1,), generator=g)) if i % 3 else int(torch.randint(115, num_classes, (1,), generator=g))
        cls[0, q, c] += 20.0                                       # ... except the kept ones
        ch, cw = h4 // gy, w4 // gx
        y0, x0 = (i // gx) * ch, (i % gx) * cw
        for t in range(T):
            tt = t0 + t
            off[t, q] = -40.0
            if (tt + 3 * i) % 11 == 0 and i % 4 == 1:              # leaves for a frame now and then
                continue
            dy, dx = (tt * (1 + i % 3)) % max(1, ch // 3), (tt * (1 + i % 2)) % max(1, cw // 3)
            off[t, q, y0 + dy // 2: y0 + ch - ch // 6, x0 + dx // 2: x0 + cw - cw // 6] = 40.0
    return cls, off


def build_models(seed=0):
    from openpvsg_amd import backbone, blocks, detectors, fusion, heads  # noqa: F401
    from openpvsg_amd import relation as prel
    from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
    from openpvsg_amd.registry import build_detector
    torch.manual_seed(seed)
    cfg = mask2former_r50_model_cfg(video=True)
    cfg['test_cfg'] = dict(cfg['test_cfg'], instance_on=False)
    det = build_detector(cfg).eval()
    det.inference_mode = 'clip'
    det.panoptic_head.init_weights()
    with torch.no_grad():
        det.panoptic_head.cls_embed.weight.mul_(CLS_GAIN)
        det.panoptic_head.query_feat.weight.mul_(8.0)
    rel = dict(subject_encoder=prel.ObjectEncoder(256), object_encoder=prel.ObjectEncoder(256),
               pair_model=prel.PairProposalNetwork(256, 1024), relation_model=prel.TemporalTransformer(512, 57))
    for m in rel.values():
        m.eval()
    return det, rel


def build_ips_detector(seed=0):
    """BASELINE config 2: Mask2Former R50 IPS detector (configs/mask2former/..._single_video_test.py), panoptic branch"""
    from openpvsg_amd import backbone, blocks, detectors, fusion, heads  # noqa: F401
    from openpvsg_amd.model_zoo import mask2former_r50_model_cfg
    from openpvsg_amd.registry import build_detector
    torch.manual_seed(seed)
    cfg = mask2former_r50_model_cfg(video=False)
    cfg['test_cfg'] = dict(cfg['test_cfg'], instance_on=False)
    det = build_detector(cfg).eval()
    det.panoptic_head.init_weights()
    with torch.no_grad():
        det.panoptic_head.cls_embed.weight.mul_(CLS_GAIN)
        det.panoptic_head.query_feat.weight.mul_(8.0)
    return det


class IPSBatchPipeline(torch.nn.Module):
    """BASELINE config 2 as one step: the IPS detector on a batch of T frames -- backbone, pixel decoder, the PER-FRAME 9-layer
    decoder (keys = h*w of one frame; models/mask2former/mask2former_head.py:397-479), last-layer mask logits, fused x4
    up-sampling + panoptic fusion per frame (mask2former_fusion_head.py:96-171) with the class decision on the device, and the
    one device->host copy of the frames' segment ids the result dictionaries need.  No tube stage: the IPS flavour's
    association is a separate program (tools/prepare_query_tube_ips.py; scripts/ips_pipeline_bench.py measures it)."""

    def __init__(self, det):
        super().__init__()
        self.detector = det
        self.head_override = None
        self.use_graph, self.graph_max_frames, self.relation_graph = False, 0, False

    @torch.no_grad()
    def forward(self, clip, batch_input_shape, img_shape=None, **kw):
        det = self.detector
        head, fusion = det.panoptic_head, det.panoptic_fusion_head
        T = clip.shape[0]
        cls_list, mask_list, q = head._decode(det.extract_feat(clip), T, 1, all_masks=False)
        cls, masks4 = cls_list[-1], mask_list[-1]                    # (T,Q,127), (T,Q,h,w), q (Q,T,C)
        if self.head_override is not None:
            c, m = self.head_override(cls[:1], masks4[None])
            cls, masks4 = c.expand(T, -1, -1), m[0]
        H, W = batch_input_shape
        ih, iw = (img_shape or batch_input_shape)[:2]
        pans, segs = [], []
        for t in range(T):
            pan, seg, sel, _ = fusion.panoptic_fused_device(cls[t], masks4[t:t + 1], (H, W), (ih, iw))
            pans.append(pan[0])
            segs.append(seg[0])
        ids = torch.stack(segs).tolist()                             # the host wait of the step: the frames' segment ids
        self.last_ids = ids
        pans = torch.stack(pans)
        return dict(pan_results=pans, tube_ids=torch.zeros(0, dtype=torch.long, device=clip.device),
                    tube_feats=torch.zeros((0, T, 256), device=clip.device), relation=None, cls=cls, query=q)


def cpu_baseline_and_parity_ips(det_gpu, pipe, args, dev, frames, reps):
    """--mode ips: the oracle's IPS detector (oracle/pipeline.py IPSDetectorOracle, one frame per call as the reference runs it)
    on `frames` frames of the same batch, 1 warm-up + `reps` timed repetitions, and the product's panoptic maps against it."""
    import numpy as np
    import torch.nn.functional as F
    from oracle import pipeline as opipe
    T = frames
    ncpu = host_cores()
    torch.set_num_threads(ncpu)
    clip, (Hp, Wp) = make_clip(T, args.height, args.width)
    o = opipe.IPSDetectorOracle(test_cfg=dict(opipe.DEFAULT_TEST_CFG)).eval()
    o.load_state_dict({k: v.detach().cpu() for k, v in det_gpu.state_dict().items()})
    meta = dict(batch_input_shape=(Hp, Wp), img_shape=(args.height, args.width, 3), ori_shape=(args.height, args.width, 3))
    syn = synthetic_head_outputs(T, Hp // 4, Wp // 4, n_keep=args.keep, T_total=T) if args.head_outputs == 'synthetic' else None

    def run():
        res = []
        with torch.no_grad():
            for t in range(T):
                if syn is not None:
                    up = F.interpolate(syn[1][t:t + 1], size=(Hp, Wp), mode='bilinear', align_corners=False)
                    o.head_override = lambda cls, masks, up=up: (syn[0], masks + up)
                res.append(o.simple_test(clip[t:t + 1], [meta], rescale=True)[0])
        return res
    run()
    times = []
    for _ in range(max(1, reps)):
        t0 = time.perf_counter()
        res = run()
        times.append(time.perf_counter() - t0)
    cpu_s = sum(times) / len(times)
    base = dict(value=T / cpu_s, unit='frames/s', cores=ncpu, kind='port',
                sample='%d frames of the batch, IPS detector one frame per call + panoptic fusion (oracle/, torch CPU fp32, %d threads): '
                       '1 warm-up + %d timed runs, %.1f s each (min %.1f, max %.1f)'
                       % (T, torch.get_num_threads(), len(times), cpu_s, min(times), max(times)), seconds_per_run=times)
    saved = pipe.head_override
    if syn is not None:
        pipe.head_override = make_override(syn, dev)
    try:
        out = pipe(clip.to(dev), (Hp, Wp), (args.height, args.width))
    finally:
        pipe.head_override = saved
    a = out['pan_results'].cpu().numpy()
    b = np.stack([res[t]['pan_results'].numpy() for t in range(T)])
    ids = (set(np.unique(a)) | set(np.unique(b))) - {126}
    inter = sum(((a == i) & (b == i)).sum() for i in ids)
    union = sum(((a == i) | (b == i)).sum() for i in ids)
    seg_equal = all(sorted(set(i for i in pipe.last_ids[t] if i >= 0)) == sorted(res[t]['query_feats'].keys()) for t in range(T))
    return base, dict(frames=T, pixel_mismatch=float((a != b).mean()), mask_iou=float(inter / union) if union else 1.0,
                      segments=len(ids), segment_ids_equal=bool(seg_equal))


def kernel_function(name, a):
    """the kernel function a C-ABI launch runs (csrc dispatch rules), where an entry has more than one"""
    if name == 'pvsg_gemm_f16x2':                      # gemm_split_run: wide layers (N >= 512, N % 256 == 0) on the 128 x 256 tile
        N = a[5]
        wide = N >= 512 and (N % 256 == 0 or os.environ.get('PVSG_W256_RAGGED') == '1')
        return 'gemm_f16x2_ln128_kernel<false> (pvsg_gemm_f16x2, wide layers)' if wide else 'gemm_f16x2_dma_kernel (pvsg_gemm_f16x2)'
    if name == 'pvsg_msda_fused_forward':
        return 'msda_fused_m8d32 (pvsg_msda_fused_forward)'
    if name in ('pvsg_conv1x1_f16x2', 'pvsg_conv1x1_f16x2_stats'):      # conv1x1_bf16x3_k32_kernel<RELU, RESIDUAL, IN_NORM, BITS, TM, 1, F16>
        if name.endswith('_stats'):                                     # (x, wp, scale, shift, residual, y, part, B, Cin, Cout, ..)
            return 'conv1x1_k32<stats> (%s)' % name
        return 'conv1x1_k32<relu=%d, residual=%d, in_norm=%d, TM=%d> (%s)' % (
            int(bool(a[14])), int(a[4] is not None), int(a[5] is not None), 64 if a[10] <= 64 else 128, name)
    if name == 'pvsg_conv3x3_f16x2':                                   # stride 1: the halo kernel <RELU, CT>; stride 2: the TAPS = 9 k32 form
        if a[10] == 1:
            return 'conv3x3_f16x2_halo_kernel<relu=%d, CT=%d> (%s)' % (int(bool(a[11])), 64 if a[7] <= 64 else 128, name)
        return 'conv1x1_k32<TAPS=9> (%s, stride 2)' % name
    return name


class KernelTimer:
    """HIP events around C-ABI launches (same stream the launch goes to): all of them, or those named in `focus`."""

    def __init__(self):
        self.records = []
        self.enabled = False
        self.focus = None          # None: events around every C-ABI launch; a set of entry names: only around those

    def install(self):
        from openpvsg_amd import _lib
        orig = _lib.call
        timer = self

        def timed(name, *args):
            alias = {'pvsg_masked_xattn_partial_strided': 'pvsg_masked_xattn_partial', 'pvsg_conv1x1_f16x2_sliced': 'pvsg_conv1x1_f16x2',
                     'pvsg_conv3x3_f16x2_sliced': 'pvsg_conv3x3_f16x2'}.get(name, name)
            if not timer.enabled or torch.cuda.is_current_stream_capturing() or (timer.focus is not None and alias not in timer.focus):
                return orig(name, *args)          # launches being captured into a hipGraph cannot carry timing events
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            orig(name, *args)
            e.record()
            # accounted under the entry whose work they do: the row-strided attention entry (key / value projections of a level
            # from one GEMM) and the K-sliced convolutions of small maps (their workspace traffic is not algorithmic)
            if name == 'pvsg_masked_xattn_partial_strided':
                name, args = 'pvsg_masked_xattn_partial', args[:13] + args[14:]
            elif name == 'pvsg_conv1x1_f16x2_sliced':
                name, args = 'pvsg_conv1x1_f16x2', args[:5] + (None, None, args[5]) + args[8:]
            elif name == 'pvsg_conv3x3_f16x2_sliced':
                name, args = 'pvsg_conv3x3_f16x2', args[:5] + args[7:]
            timer.records.append((name, args, s, e))
        _lib.call = timed
        import openpvsg_amd.ops as ops
        ops._lib.call = timed

    KEPT_HINT = 32             # kept queries of the synthetic head outputs (--keep): the device-side tail does not tell the host
    SPLIT_F16X2 = ('pvsg_gemm_f16x2', 'pvsg_gemm_f16x2_add_layernorm', 'pvsg_conv1x1_f16x2', 'pvsg_conv3x3_f16x2',
                   'pvsg_mask_logits_f16x2', 'pvsg_attn_mask_bits_f16x2', 'pvsg_attn_mask_bits_packed_f16x2',
                   'pvsg_conv1x1_f16x2_stats', 'pvsg_conv3x3_f16x2_stats', 'pvsg_bottleneck_tail_f16x2',
                   'pvsg_stem7x7_f16x2_bn_relu_pool', 'pvsg_decoder_rows_pre_f16x2', 'pvsg_decoder_rows_post_f16x2')

    @classmethod
    def work(cls, name, a):
        """(algorithmic bytes, flops) of one launch from its scalar arguments."""
        if name == 'pvsg_gemm_f16x2_add_layernorm':     # projection + identity + LayerNorm: reads a, residual; writes the normalised rows
            M, N, K = a[8:11]
            return 4.0 * M * (K + 2 * N) + 4.0 * N * K, 6.0 * M * N * K
        if name == 'pvsg_attn_mask_bits_packed_f16x2':   # embeddings pre-packed by decoder_rows_post: (pack, f, bits, flags, B, T, Q, C, N, ..)
            B, T, Q, C, N = a[4:9]
            return 4.0 * B * T * N * C + 16.0 * B * T * N, 6.0 * B * T * Q * C * N
        if name == 'pvsg_bottleneck_tail_f16x2':        # (mid, w3, sc3, sh3, identity, y, y_s2, w1n, sc1n, sh1n, mid_next, B, Cmid, Cout, Cnext, H, W, ..)
            B, Cmid, Cout, Cnext, H, W = a[11:17]
            by = 4.0 * B * H * W * (Cmid + Cout * (2 if a[4] else 1) + Cnext) + (1.0 * B * H * W * Cout if a[6] else 0.0)
            return by, 6.0 * B * H * W * (Cmid * Cout + (Cout if a[4] else Cmid) * Cnext)
        if name == 'pvsg_stem7x7_f16x2_bn_relu_pool':    # (x, wp, scale, shift, out, N, H, W, ..): flops = the 147 real taps as limb products
            N, H, W = a[5:8]
            hc, wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            return 4.0 * N * (3 * H * W + 64 * ((hc - 1) // 2 + 1) * ((wc - 1) // 2 + 1)), 6.0 * 147 * 64 * N * hc * wc
        if name == 'pvsg_conv1x1_f16x2_stats':          # pvsg_conv1x1_f16x2 + GroupNorm partial sums (x, wp, scale, shift, residual, y, part, B, ..)
            B, Cin, Cout, H, W, st = a[7:13]
            hw = ((H - 1) // st + 1) * ((W - 1) // st + 1)
            return 4.0 * B * hw * (Cin + Cout * (2 if a[4] else 1)) + 6.0 * Cin * Cout, 6.0 * B * Cin * Cout * hw
        if name == 'pvsg_conv3x3_f16x2_stats':          # (x, wp, scale, shift, y, part, B, Cin, Cout, H, W, relu, ..), stride 1
            N, Cin, Cout, H, W = a[6:11]
            return 4.0 * N * H * W * (Cin + Cout) + 54.0 * Cin * Cout, 3 * 18.0 * N * Cin * Cout * H * W
        if name in ('pvsg_decoder_rows_pre_f16x2', 'pvsg_decoder_rows_post_f16x2'):
            # the query-row kernels with their GEMMs on the f16 pipe (same arguments + overflow): 3 limb products per multiply-add
            # (the 100 x 100 self-attention stays on the f32 MFMA: < 2 % of the kernel's flops, counted with the rest)
            by, fl = cls._work(name[:-len('_f16x2')], a)
            return by, 3.0 * fl
        if name in cls.SPLIT_F16X2:
            # two-limb f16 form: same arguments up to the trailing (overflow, stream); flops = the f16 limb products issued,
            # 3 per f32 multiply-add (the bf16 form issues 6)
            by, fl = cls._work(name.replace('_f16x2', '_bf16x3'), a)
            return by, fl / 2.0
        return cls._work(name, a)

    @staticmethod
    def _work(name, a):
        if name == 'pvsg_ms_deform_attn_forward':
            B, S, M, D, Lq, L, P = a[6:13]
            return 4.0 * (B * S * M * D + B * Lq * M * D + 3 * B * Lq * M * L * P), 9.0 * B * Lq * M * L * P * D
        if name == 'pvsg_mask_logits_forward':
            B, T, Q, C, N = a[3:8]
            return 4.0 * B * T * N * (C + Q), 2.0 * B * T * Q * C * N
        if name == 'pvsg_mask_logits_bf16x3':       # flops = bf16 limb products issued (6 per f32 multiply-add)
            B, T, Q, C, N = a[4:9]
            return 4.0 * B * T * N * (C + Q), 12.0 * B * T * Q * C * N
        if name == 'pvsg_attn_mask_bits_bf16x3':
            B, T, Q, C, N = a[5:10]
            return 4.0 * B * T * N * C + 16.0 * B * T * N, 12.0 * B * T * Q * C * N
        if name == 'pvsg_attn_mask_bits_forward':
            B, T, Q, C, N = a[4:9]
            return 4.0 * B * T * N * C + 16.0 * B * T * N, 2.0 * B * T * Q * C * N
        if name == 'pvsg_masked_xattn_partial':
            B, Q, K, M, D, NS = a[7:13]
            limb = 1.0 if XATTN_F32 else 6.0        # default kernel: six bf16 limb products per f32 multiply-add
            return B * K * (2.0 * M * D * 4 + (16 if a[3] else 0)) + 4.0 * B * NS * M * Q * (D + 2), limb * 4.0 * B * Q * M * D * K
        if name == 'pvsg_affine_act_nchw':
            planes, C, HW = a[5:8]
            return 4.0 * planes * HW * (3 if a[3] else 2), 0.0
        if name == 'pvsg_msda_fused_forward':
            B, S, M, D, Lq, L, P = a[9:16]
            return 4.0 * (B * S * M * D + B * Lq * M * D + B * Lq * M * L * P * 3 + Lq * M * L * P * 3), 9.0 * B * Lq * M * L * P * D
        if name == 'pvsg_add_layernorm':
            rows, C = a[6:8]
            return 4.0 * rows * C * (3 if a[1] else 2), 0.0
        if name == 'pvsg_panoptic_fuse':
            T, Q, K, h, w, H, W, ih, iw = a[8:17]
            return 4.0 * T * K * h * w + 5.0 * T * ih * iw + 1.0 * T * ih * iw, 0.0
        if name == 'pvsg_panoptic_fuse_sel':          # kept count decided on the device: priced with the bench's keep count
            T, Q, h, w, H, W, ih, iw = a[6:14]
            return 4.0 * T * KernelTimer.KEPT_HINT * h * w + 5.0 * T * ih * iw + 1.0 * T * ih * iw, 0.0
        if name == 'pvsg_center_downsample':
            planes, H, W = a[4:7]
            return 4.0 * planes * H * W * (1 + 21.0 / 64), 0.0
        if name == 'pvsg_conv1x1_affine':
            B, Cout, Cin, HW = a[6:10]
            return 4.0 * B * HW * (Cin + Cout * (2 if a[4] else 1)), 2.0 * B * HW * Cin * Cout
        if name == 'pvsg_conv3x3_winograd':
            # flops = the multiplications the F(2x2,3x3) form issues on the matrix cores (16 GEMMs over the 2x2 tiles):
            # direct-convolution flops / 2.25 -- the kernel is priced against what it executes, not what it saves
            N, Cin, Cout, H, W = a[5:10]
            return (4.0 * N * H * W * (Cin + Cout) + 64.0 * Cin * Cout,
                    32.0 * N * Cin * Cout * ((H + 1) // 2) * ((W + 1) // 2))
        if name == 'pvsg_gemm_bf16x3':
            # flops = the bf16 limb products the kernel issues (6 per f32 multiply-add), priced against the bf16 roof
            M, N, K = a[4:7]
            return 4.0 * M * (K + N) + 6.0 * N * K, 12.0 * M * N * K
        if name == 'pvsg_conv1x1_bf16x3':
            B, Cin, Cout, H, W, st = a[8:14]
            hw = ((H - 1) // st + 1) * ((W - 1) // st + 1)
            return 4.0 * B * hw * (Cin + Cout * (2 if a[4] else 1)) + 6.0 * Cin * Cout, 12.0 * B * Cin * Cout * hw
        if name == 'pvsg_stem7x7_bn_relu_pool':
            N, H, W = a[5:8]
            hc, wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            return 4.0 * N * (3 * H * W + 64 * ((hc - 1) // 2 + 1) * ((wc - 1) // 2 + 1)), 2.0 * 147 * 64 * N * hc * wc
        if name == 'pvsg_conv3x3s2_affine':
            N, Cin, Cout, H, W = a[5:10]
            ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            return 4.0 * N * (Cin * H * W + Cout * ho * wo) + 36.0 * Cin * Cout, 18.0 * N * Cin * Cout * ho * wo
        if name == 'pvsg_conv3x3_bf16x3':            # flops = bf16 limb products issued (6 per f32 multiply-add)
            N, Cin, Cout, H, W, st = a[5:11]
            ho, wo = (H - 1) // st + 1, (W - 1) // st + 1
            return 4.0 * N * (Cin * H * W + Cout * ho * wo) + 54.0 * Cin * Cout, 6.0 * 18.0 * N * Cin * Cout * ho * wo
        if name == 'pvsg_fpn_merge_up2x':
            planes, h, w = a[5:8]
            return 4.0 * planes * h * w * 9, 0.0                     # lateral 4 + out 4 + top 1 (x h*w cells)
        if name == 'pvsg_stem_bn_relu_pool':
            planes, C, H, W = a[4:8]
            return 4.0 * planes * (H * W + ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1)), 0.0
        if name == 'pvsg_decoder_rows_pre':
            B, Q = a[6:8]
            return 4.0 * (256 * 256 + 768 * 256) + 4.0 * B * Q * 256 * 6, 2.0 * B * Q * 256 * (256 + 768)
        if name == 'pvsg_decoder_rows_post':
            B, Q = a[14:16]
            w_head = 128 * 256 + 3 * 256 * 256 + (256 * 256 if a[2] else 0)
            w_layer = (256 * 256 + 2 * 256 * 2048) if a[0] is not None else 0
            att = 4.0 * B * Q * Q * 256 if a[0] is not None else 0.0
            return 4.0 * (w_head + w_layer) + 4.0 * B * Q * 256 * 6, 2.0 * B * Q * (w_head + w_layer) + att
        if name == 'pvsg_xattn_combine':
            B, Q, M, D, NS = a[3:8]
            return 4.0 * B * NS * M * Q * (D + 2) + 4.0 * B * Q * M * D, 0.0
        if name == 'pvsg_decoder_kv_inputs':
            frames, hw, C = a[5:8]
            return 16.0 * frames * hw * C, 0.0                       # tokens + pe read, value + key written
        if name == 'pvsg_nchw_to_tokens':
            B, C, HW = a[4:7]
            return 8.0 * B * C * HW, 0.0
        if name == 'pvsg_tokens_to_nchw':
            B, C, HW = a[2:5]
            return 8.0 * B * C * HW, 0.0
        if name == 'pvsg_group_norm_affine':
            B, C, G, HW = a[6:10]
            return 4.0 * B * C * HW + 8.0 * B * C, 0.0               # one read of x, scale + shift written
        return 0.0, 0.0

    @classmethod
    def step_flops(cls, name, a):
        """Flops of the launch as the MODEL's arithmetic (what roofline_step sums): the direct-convolution count for the
        Winograd kernel, so that the step total does not depend on which algorithm runs a convolution."""
        if name == 'pvsg_conv3x3_winograd':
            N, Cin, Cout, H, W = a[5:10]
            return 18.0 * N * Cin * Cout * H * W
        if name == 'pvsg_gemm_bf16x3':
            M, N, K = a[4:7]
            return 2.0 * M * N * K
        if name in ('pvsg_conv1x1_bf16x3', 'pvsg_conv3x3_bf16x3', 'pvsg_mask_logits_bf16x3', 'pvsg_attn_mask_bits_bf16x3'):
            return cls.work(name, a)[1] / 6.0
        if name in cls.SPLIT_F16X2:
            return cls.work(name, a)[1] / 3.0
        if name == 'pvsg_masked_xattn_partial' and not XATTN_F32:
            return cls.work(name, a)[1] / 6.0
        return cls.work(name, a)[1]

    @staticmethod
    def mfma_peak(name):
        """(peak TFLOP/s, what the flops of work() count) of the matrix pipe a kernel runs on."""
        if name.startswith(KernelTimer.SPLIT_F16X2):
            return BF16_MFMA_PEAK_TF, 'f16 limb products issued (3 per f32 multiply-add), dense f16 MFMA peak (= the bf16 peak)'
        if name.startswith(('pvsg_gemm_bf16x3', 'pvsg_conv1x1_bf16x3', 'pvsg_conv3x3_bf16x3', 'pvsg_mask_logits_bf16x3',
                            'pvsg_attn_mask_bits_bf16x3')) \
                or (name.startswith('pvsg_masked_xattn_partial') and not XATTN_F32):
            return BF16_MFMA_PEAK_TF, 'bf16 limb products issued (6 per f32 multiply-add), dense bf16 MFMA peak'
        return F32_MFMA_PEAK_TF, 'f32 MFMA'

    def summary(self, records=None):
        agg = {}
        for name, args, s, e in (self.records if records is None else records):
            ms = s.elapsed_time(e)
            by, fl = self.work(name, args)
            key = name
            if name == 'pvsg_masked_xattn_partial':
                key = name + ('[K=%d]' % args[9])
            d = agg.setdefault(key, dict(calls=0, ms=0.0, bytes=0.0, flops=0.0))
            d['calls'] += 1
            d['ms'] += ms
            d['bytes'] += by
            d['flops'] += fl
        return agg


def cpu_baseline_and_parity(det_gpu, rel_gpu, pipe, args, dev, frames, reps, check=True, warmup=True):
    """Oracle (CPU restatement of the reference algorithm, oracle/) on a bounded sample: a `frames`-frame 720p clip
    through the same clip-level flow + relation head, 1 warm-up + `reps` timed repetitions; also compares the
    product's panoptic maps / pair matrix with the oracle's on that sample."""
    import numpy as np
    import torch.nn.functional as F
    from oracle import pipeline as opipe
    from oracle import relation as orel
    T = frames
    ncpu = host_cores()
    torch.set_num_threads(ncpu)
    clip, (Hp, Wp) = make_clip(T, args.height, args.width)
    o = opipe.VPSDetectorOracle(test_cfg=dict(opipe.DEFAULT_TEST_CFG)).eval()
    o.load_state_dict({k: v.detach().cpu() for k, v in det_gpu.state_dict().items()})
    orl = dict(se=orel.ObjectEncoder(256).eval(), oe=orel.ObjectEncoder(256).eval(),
               pp=orel.PairProposalNetwork(256, 1024).eval(), rm=orel.TemporalTransformer(512, 57).eval())
    for k, m in (('se', 'subject_encoder'), ('oe', 'object_encoder'), ('pp', 'pair_model'), ('rm', 'relation_model')):
        orl[k].load_state_dict({n: v.detach().cpu() for n, v in rel_gpu[m].state_dict().items()
                                if n in orl[k].state_dict()})
    meta = dict(batch_input_shape=(Hp, Wp), img_shape=(args.height, args.width, 3),
                ori_shape=(args.height, args.width, 3))
    syn = None
    if args.head_outputs == 'synthetic':
        syn = synthetic_head_outputs(T, Hp // 4, Wp // 4, n_keep=args.keep, T_total=T)

    def run():
        with torch.no_grad():
            cls, masks, q = o.clip_forward(clip[None], (Hp, Wp))          # masks (1,T,Q,H,W): already up-sampled
            if syn is not None:                                          # same controlled head outputs as the product
                cls = syn[0]
                masks = masks + F.interpolate(syn[1], size=(Hp, Wp), mode='bilinear', align_corners=False)[None]
            embds = q.permute(1, 0, 2)
            res = []
            for t in range(T):
                res.append(opipe.heads.fusion_simple_test_with_query(cls, masks[:, t], embds, [meta], o.num_things,
                                                                      o.num_stuff, o.test_cfg, rescale=True)[0])
            # tubes exactly as the product assembles them, then the reference's relation flow
            order, fr = [], []
            for t in range(T):
                fr.append(res[t]['query_feats'])
                for sid in res[t]['query_feats']:
                    if sid not in order:
                        order.append(sid)
            feats = torch.zeros(len(order), T, 256)
            for t in range(T):
                for sid, lst in fr[t].items():
                    feats[order.index(sid), t] = lst[0].reshape(-1)
            rel_out = None
            if len(order) >= 2:
                rel_out = orel.evaluate_video(orl['se'], orl['oe'], orl['pp'], orl['rm'], feats, [], 100)
        return res, order, rel_out

    if warmup:
        run()                                                            # warm-up
    times = []
    for _ in range(max(1, reps)):
        t0 = time.perf_counter()
        res, order, rel_out = run()
        times.append(time.perf_counter() - t0)
    cpu_s = sum(times) / len(times)
    base = dict(value=T / cpu_s, unit='frames/s', cores=ncpu, kind='port',
                sample='%d-frame 720p clip, clip-level VPS forward + fusion + relation head (oracle/, torch CPU fp32, '
                       '%d threads): 1 warm-up + %d timed runs, %.1f s each (min %.1f, max %.1f)'
                       % (T, torch.get_num_threads(), len(times), cpu_s, min(times), max(times)),
                seconds_per_run=times)
    if not check:
        return base, None
    # product on the same sample
    saved = pipe.head_override
    if syn is not None:
        pipe.head_override = make_override(syn, dev)
    try:
        out = pipe(clip.to(dev), (Hp, Wp), (args.height, args.width))
    finally:
        pipe.head_override = saved
    torch.cuda.synchronize()
    a = out['pan_results'].cpu().numpy()
    b = np.stack([res[t]['pan_results'].numpy() for t in range(T)])
    ids = (set(np.unique(a)) | set(np.unique(b))) - {126}
    inter = sum(((a == i) & (b == i)).sum() for i in ids)
    union = sum(((a == i) | (b == i)).sum() for i in ids)
    parity = dict(frames=T, pixel_mismatch=float((a != b).mean()),
                  mask_iou=float(inter / union) if union else 1.0, segments=len(ids),
                  tubes=int(out['tube_feats'].shape[0]), tubes_oracle=len(order))
    if rel_out is not None and out['relation'] is not None:
        pm_a = out['relation']['pred_matrix'].cpu()
        parity['pair_matrix_max_abs_diff'] = float((pm_a - rel_out['pred_matrix']).abs().max()) \
            if pm_a.shape == rel_out['pred_matrix'].shape else None
        pa = out['relation']['pairs'].cpu().tolist()
        parity['top20_pairs_equal'] = pa[:20] == rel_out['pairs'][:20]
    return base, parity


def make_override(syn, dev):
    cls_syn, off = syn[0].to(dev), syn[1].to(dev)

    def override(cls, masks4):
        masks4 = masks4.add_(off[None]) if masks4.shape[1:] == off.shape else masks4 + off[None]
        return cls_syn, masks4
    return override


def fastpath_bit_flip_rate(head, feats, T):
    """Fast path (attention-mask bits from down-sampled features, einsum(E, resize(F))) vs the reference order
    (resize(einsum(E, F)) then threshold) on the LAST layer's mask embeddings, per decoder level, at the bench size."""
    import torch.nn.functional as F
    from openpvsg_amd import ops
    rec = {}
    orig = head._mask_step

    def spy(emb, mf, lows, level, want_logits, need_mask=True, **kw):
        rec['emb'], rec['mf'], rec['lows'] = emb, mf, lows
        return orig(emb, mf, lows, level, want_logits, need_mask, **kw)
    head._mask_step = spy
    try:
        with torch.no_grad():
            head.clip_logits(feats, 1, T)
    finally:
        head._mask_step = orig
    emb, mf, lows = rec['emb'], rec['mf'], rec['lows']
    out = {}
    if lows is None:
        return None
    with torch.no_grad():
        logits = ops.mask_logits(emb, mf)                                  # (1,T,Q,H/4,W/4)
        for lvl in range(3):
            fast = ops.attn_mask_from_lowres_feature(emb, lows[lvl]).bits
            size = tuple(lows[lvl].shape[-2:])
            low = F.interpolate(logits.flatten(0, 1), size, mode='bilinear', align_corners=False).unflatten(0, (1, T))
            exact = ops.attn_mask_pack(low).bits
            x = (fast ^ exact).view(torch.uint8)
            flips = int(sum(int(((x >> k) & 1).sum()) for k in range(8)))
            out['level%d' % lvl] = dict(keys=int(fast.shape[1]), flipped_bits=flips,
                                        rate=flips / (fast.shape[1] * emb.shape[1]))
    return out


def head_only_benchmark(det, pipe, clip_local, step, iters=10):
    """the same step with the ResNet-50 features cached (pixel decoder, decoder, post-processing,
    tubes, relation head only) -- where the hand-written kernels live."""
    with torch.no_grad():
        feats = det.extract_feat(clip_local)
    orig = det.extract_feat
    det.extract_feat = lambda x: feats
    try:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
    finally:
        det.extract_feat = orig
    return dict(ms=ms, frames_per_s=clip_local.shape[0] * 1e3 / ms, note='backbone features cached; everything else as in the timed step')


def sub_benchmarks(det, rel, pipe, args, dev, iters=20):
    """relation head on synthetic tubes N(0,1) [N,T,256] for N in {32,64,100}; fused
    post-processing (x4 up-sampling + panoptic fusion of one clip) with K in {10,30} kept queries."""
    from openpvsg_amd.relation import relation_forward
    from tests.synth_inputs import blob_masks, peaky_cls
    out = {}
    T = args.frames
    g = torch.Generator().manual_seed(7)

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / iters

    with torch.no_grad():
        for N in (32, 64, 100):
            feats = torch.randn(N, T, 256, generator=g).to(dev)
            ms = timeit(lambda: relation_forward(rel['subject_encoder'], rel['object_encoder'], rel['pair_model'],
                                                 rel['relation_model'], feats, 100))
            ms_pair = timeit(lambda: rel['pair_model'](feats, feats))
            out['relation_head_N%d_T%d' % (N, T)] = dict(ms=ms, videos_per_s=1e3 / ms, pair_scorer_us=ms_pair * 1e3)
        fusion = det.panoptic_fusion_head
        Hp, Wp = (args.height + 31) // 32 * 32, (args.width + 31) // 32 * 32
        for K in (10, 30):
            cls, conf = peaky_cls(100, 126, K, 3)
            one = blob_masks(100, Hp // 4, Wp // 4, conf, 5)
            logits4 = one[None].repeat(T, 1, 1, 1).to(dev)
            cls = cls.to(dev)
            ms = timeit(lambda: fusion.panoptic_fused(cls, logits4, (Hp, Wp), (args.height, args.width)))
            kept = int(fusion.panoptic_select(cls)[2].sum())
            out['postprocess_K%d' % K] = dict(ms=ms, frames_per_s=T * 1e3 / ms, kept=kept,
                                              algorithmic_GB=(4.0 * T * kept * (Hp // 4) * (Wp // 4) + 5.0 * T * args.height * args.width) / 1e9)
    return out


def self_spawn(args):
    """`python bench.py --gpus N` with no launcher around it: re-exec under torch.distributed.run with one
    rank per GPU (the reference's launcher does the same for tools/test.py:186-190 `init_dist`)
    and hand its exit code back.  The ranks then find WORLD_SIZE / RANK / LOCAL_RANK in the environment."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_spawn(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('PVSG_ONE_DEVICE') == '1' and world > 1:
        from openpvsg_amd.parallel import isolate_shared_gpu
        isolate_shared_gpu(rank, world)            # ranks sharing GPU 0 get disjoint CU ranges (see its docstring)
    # PVSG_FORCE_COLLECTIVES=1 at N = 1: the process group is created anyway and every exchange of the frame-sharded layout
    # runs over RCCL with a single rank (openpvsg_amd/parallel.py) -- the only way a one-GPU box executes that code path
    force = os.environ.get('PVSG_FORCE_COLLECTIVES', '0') == '1'
    if world > 1 or force:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group(args.backend, init_method='env://', rank=rank, world_size=world)
    if os.environ.get('PVSG_ONE_DEVICE') == '1':   # logic test: every rank on GPU 0 (gloo backend)
        local = 0
    elif torch.cuda.device_count() <= local:
        raise SystemExit('bench.py: rank %d needs cuda:%d but this node shows %d GPUs'
                         % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    torch.backends.cudnn.benchmark = os.environ.get("PVSG_MIOPEN_FIND", "0") == "1"  # exhaustive MIOpen find costs ~5 min per fresh box
    # deterministic MIOpen kernels (no split-K atomics): the step becomes bitwise reproducible run to run
    torch.backends.cudnn.deterministic = os.environ.get('PVSG_DETERMINISTIC', '1') == '1'
    gemm_table = False
    if os.environ.get('PVSG_GEMM_TABLE', 'on') != 'off':
        from openpvsg_amd import tuning
        gemm_table = tuning.enable()      # pre-selected rocBLAS / hipBLASLt solutions for the clip-size GEMMs (load only)

    from openpvsg_amd import build
    if rank == 0 and not os.path.exists(build.lib_path()):
        build.build_hip_lib(verbose=False)
    if world > 1:
        dist.barrier()
    from openpvsg_amd import parallel
    from openpvsg_amd.pipeline import PVSGPipeline

    ips = args.mode == 'ips'
    if ips:
        if world > 1:
            raise SystemExit('bench.py --mode ips: single-GPU line (BASELINE config 2 runs on 1 MI355X)')
        det, rel = build_ips_detector(0).to(dev), {}
        pipe = IPSBatchPipeline(det).eval()
    else:
        det, rel = build_models(0)
        det = det.to(dev)
        rel = {k: m.to(dev) for k, m in rel.items()}
        pipe = PVSGPipeline(det, rel['subject_encoder'], rel['object_encoder'], rel['pair_model'],
                            rel['relation_model'], use_graph={'auto': 'auto', 'on': True, 'off': False}[args.graph]).eval()
        pipe.graph_max_frames = 8      # --graph auto: clips above 8 frames run eagerly here so that their kernels carry HIP events
                                       # (the product default, 64, is measured separately: `product_default_hipgraph`)

    T = args.frames
    weak = world > 1 and args.scaling == 'weak'
    if weak:
        clip, (Hp, Wp) = make_clip(T, args.height, args.width, seed=rank)   # this rank's own segment of the video
        t0, t_local = 0, T
        clip_local = clip.to(dev)
    else:
        clip, (Hp, Wp) = make_clip(T, args.height, args.width)
        t0, t_local = parallel.shard_frames(T, rank, world)
        clip_local = clip[t0:t0 + t_local].to(dev)       # resident in HBM before the timed region
    group = None
    frames_per_step = T * world if weak else T
    if args.head_outputs == 'synthetic':
        syn = synthetic_head_outputs(t_local, Hp // 4, Wp // 4, n_keep=args.keep, seed=rank if weak else 0,
                                     t0=0 if weak else t0, T_total=T)
        pipe.head_override = make_override(syn, dev)

    timer = KernelTimer()
    KernelTimer.KEPT_HINT = args.keep
    if not args.no_kernel_timing and rank == 0:
        timer.install()

    def step():
        return pipe(clip_local, (Hp, Wp), (args.height, args.width), total_frames=T, group=group,
                    shard='segments' if weak else 'frames')

    # Per-kernel table: HIP events around EVERY C-ABI launch cost 1.7 ms per 32-frame step (370 pairs).  They are taken during
    # the LAST WARM-UP step; the timed region carries events only around the entry that step found dominant -- which is what
    # `roofline` reports (measured live inside the timed region).  --timing all (or --warmup 0) keeps every pair in the timed region.
    table_records, table_steps, table_from_eager_step = None, args.steps, False
    for w in range(args.warmup):
        last = w == args.warmup - 1 and args.timing == 'dominant' and not args.no_kernel_timing and rank == 0
        eager_table = (last and world == 1 and getattr(pipe, 'use_graph', False) in (True, 'auto') and
                       (pipe.use_graph is True or t_local <= getattr(pipe, 'graph_max_frames', 0)))
        if last:
            torch.cuda.synchronize()
            timer.enabled = True
        if eager_table:                       # the timed region replays a hipGraph (no per-launch events): table from ONE eager step
            saved_use_graph, pipe.use_graph = pipe.use_graph, False
        out = step()
        if eager_table:
            pipe.use_graph = saved_use_graph
            table_from_eager_step = True
        if last:
            torch.cuda.synchronize()
            timer.enabled = False
            table_records, table_steps, timer.records = timer.records, 1, []
            warm = timer.summary(table_records)
            cand = [k for k in warm if warm[k]['bytes'] > 0]
            if cand:
                timer.focus = {max(cand, key=lambda k: warm[k]['ms']).split('[')[0]}
            else:
                table_records, table_steps = None, args.steps
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    if dist.is_initialized() and rank == 0:
        parallel.EXCHANGE_TIMER = []               # HIP events around every exchange of the timed steps (rank 0)
    start = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - start
    timer.enabled = False
    exchanges, parallel.EXCHANGE_TIMER = parallel.EXCHANGE_TIMER, None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # one extra UNTIMED step under torch's flop counter: algorithmic flops of the library ops (mm / addmm / bmm /
    # convolution); the hand-written kernels' own counts come from KernelTimer.work
    lib_flops = hw_flops_one = None
    try:
        if args.no_flop_count:
            raise RuntimeError('skipped')
        from torch.utils.flop_counter import FlopCounterMode
        n0 = len(timer.records)
        timer.enabled = bool(timer.records)
        focus, timer.focus = timer.focus, None
        with FlopCounterMode(display=False) as fc:
            if world > 1 or force:   # same local compute without the exchanges: no collective runs under the dispatch mode
                pipe(clip_local, (Hp, Wp), (args.height, args.width), total_frames=T, group=group, shard='none')
            else:
                step()
        torch.cuda.synchronize()
        timer.enabled = False
        timer.focus = focus
        hw_flops_one = sum(KernelTimer.step_flops(n, a) for n, a, _, _ in timer.records[n0:])
        del timer.records[n0:]
        lib_flops = float(fc.get_total_flops())
    except Exception:
        timer.enabled = False
        hw_flops_one = None
    if world > 1:
        dist.barrier()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        fps = frames_per_step * args.steps / elapsed
        line = {
            'metric': 'frames/sec for 720p 32-frame VPS+relation forward',
            'value': fps, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'collectives': {'backend': dist.get_backend() if dist.is_initialized() else None,
                            'rccl_ranks': dist.get_world_size() if dist.is_initialized() else 1,
                            'exchanges_run': bool(parallel.is_dist()), 'forced_at_world_1': bool(force and world == 1)},
            'dtype_note': (('every tensor, accumulation and result f32; the token GEMMs, the 1x1 and 3x3 convolutions and the mask '
                            'projections split their f32 operands into ' +
                            ('two f16 limbs (3 limb products per multiply on the f16 MFMA, operands within the f16 range, '
                             'low limbs kept normal by exact power-of-two factors; PVSG_SPLIT=bf16x3 selects the other form)'
                             if ops_split_mode() == 'f16x2' else
                             'three bf16 limbs (6 limb products per multiply on the bf16 MFMA, exact split)') +
                            ', f32 accumulate: error vs f64 at or below the library f32 GEMM (tests/test_gemm_bf16x3.py, '
                            'tests/test_gemm_f16x2.py)') if os.environ.get('PVSG_GEMM', 'bf16x3') != 'lib' else
                           'every tensor, accumulation and result f32 (PVSG_GEMM=lib: library f32 GEMMs)'),
            'config': {'workload': 'Mask2Former-VPS R50 clip-level forward, %d frames %dx%d (padded %dx%d), '
                                   '100 queries, 9 decoder layers over T*h*w keys, panoptic fusion per frame, '
                                   'tube assembly, relation head (TemporalTransformer, top-100 pairs)'
                                   % (T, args.height, args.width, Hp, Wp),
                       'frames': T, 'frames_per_gpu': t_local, 'backbone': 'ResNet-50 (reference ships no Swin-B config)',
                       'weights': 'random init seed 0',
                       'head_outputs': ('synthetic class logits + additive mask-logit offsets, %d confident queries '
                                        '(BASELINE.md section 2: controlled keep count; random-init outputs keep ~2 segments)'
                                        % args.keep) if args.head_outputs == 'synthetic' else
                       'model (cls logits x%g so that some queries pass score>0.8)' % CLS_GAIN,
                       'test_cfg': 'instance_on=False, inference_mode=clip (the shipped configs set instance_on=True and '
                                   'run per-frame; scripts/ips_pipeline_bench.py measures that flavour)',
                       'tubes': int(out['tube_feats'].shape[0]), 'frames_per_step': frames_per_step,
                       'library_gemm_table': 'openpvsg_amd/tuning/gemm_gfx950.csv (load only)' if gemm_table else 'off',
                       'hipgraph': ('backbone + head replayed as one hipGraph' if (pipe.use_graph is True or (
                           pipe.use_graph == 'auto' and t_local <= pipe.graph_max_frames)) and world == 1 else 'eager launches')
                       + ' (--graph %s); relation head graph %s' % (args.graph, 'on' if pipe.relation_graph else 'off'),
                       'parallelism': ('%d x 32-frame segments, all-gather of tube records' % world) if weak
                       else ('frame-shard x%d, attention partials merged per layer' % world)},
        }
        if exchanges:
            by_size = {}
            for nbytes, s_ev, e_ev in exchanges:
                by_size.setdefault(nbytes, []).append(s_ev.elapsed_time(e_ev) * 1e3)
            line['collectives']['per_exchange'] = [dict(bytes_per_rank=k, count_per_step=len(v) / args.steps, avg_us=sum(v) / len(v),
                                                        min_us=min(v), max_us=max(v)) for k, v in sorted(by_size.items())]
            line['collectives']['exchange_us_per_step'] = sum(sum(v) for v in by_size.values()) / args.steps
        if ips:
            line['metric'] = 'frames/sec for the 720p IPS detector forward on a batch of frames (BASELINE config 2)'
            line['config']['workload'] = ('Mask2Former R50 IPS detector, batch of %d frames %dx%d (padded %dx%d): backbone, MSDeformAttn '
                                          'pixel decoder, per-frame 9-layer decoder (100 queries, keys = h*w of one frame), last-layer '
                                          'mask logits, fused x4 up-sampling + panoptic fusion per frame, segment ids to the host'
                                          % (T, args.height, args.width, Hp, Wp))
            line['config']['test_cfg'] = 'instance_on=False (the shipped config sets instance_on=True; scripts/shipped_config_bench.py)'
            line['config']['parallelism'] = 'one GPU'
            line['config'].pop('tubes', None)
        if world == 1 and not ips and args.projection == 'on' and T >= 8:
            # What N GPUs would run on this clip sharded by frame: the step of T/N frames measured here on one GPU (no exchanges) ->
            # the strong-scaling efficiency the frame-independent part of a step allows, before any RCCL latency.  The driver's
            # SCALE record measures the real thing when it has an 8-GPU node.
            proj = {}
            saved_ov, saved_gmf, pipe.graph_max_frames = pipe.head_override, pipe.graph_max_frames, 64   # (the product default)
            timer.enabled = False
            for n in (2, 4, 8):
                if T % n:
                    continue
                tl = T // n
                c = clip[:tl].to(dev)
                if args.head_outputs == 'synthetic':
                    pipe.head_override = make_override(synthetic_head_outputs(tl, Hp // 4, Wp // 4, n_keep=args.keep, T_total=T), dev)
                for _ in range(3):
                    pipe(c, (Hp, Wp), (args.height, args.width), total_frames=T, shard='none')
                torch.cuda.synchronize()
                t0p = time.perf_counter()
                for _ in range(10):
                    pipe(c, (Hp, Wp), (args.height, args.width), total_frames=T, shard='none')
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0p) / 10 * 1e3
                proj[str(n)] = dict(frames_per_gpu=tl, ms_per_step_one_gpu=ms, projected_frames_per_s=T / ms * 1e3,
                                    efficiency_before_collectives=ms_per_step / n / ms)
            pipe.head_override, pipe.graph_max_frames = saved_ov, saved_gmf
            line['projected_strong_scaling'] = dict(
                note='T/N-frame step measured on this GPU (hipGraph replay where the product uses it) vs the %d-frame step / N; '
                     'excludes the 9 + 1 exchanges per step' % T, by_n_gpus=proj)
        if world == 1 and not ips and args.graph == 'auto' and not (pipe.use_graph == 'auto' and T <= pipe.graph_max_frames):
            # the timed region above runs eagerly so that its kernels carry HIP events (`roofline`); the PRODUCT default replays
            # backbone + head as one hipGraph at every clip length: the same step that way, 5 timed steps
            try:
                saved_graph, pipe.use_graph = pipe.use_graph, True
                timer.enabled = False
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t0p = time.perf_counter()
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
                gms = (time.perf_counter() - t0p) / 5 * 1e3
                line['product_default_hipgraph'] = dict(ms_per_step=gms, frames_per_s=frames_per_step / gms * 1e3,
                                                        note='PVSGPipeline(use_graph=True): backbone + head as one hipGraph; no per-kernel events')
            except Exception as e:
                line['product_default_hipgraph'] = dict(failed=repr(e))
            finally:
                pipe.use_graph = saved_graph
        if args.checksum:
            pans = out['pan_results']
            line['checksum'] = dict(pan_local_sum=int(pans.to(torch.int64).sum().item()),
                                    pan_local_frames=int(pans.shape[0]),
                                    tube_ids=out['tube_ids'].tolist(),
                                    tube_feat_sum=float(out['tube_feats'].double().sum().item()),
                                    query_sum=float(out['query'].double().sum().item()),
                                    pair_sum=float(out['relation']['pred_matrix'].double().sum().item())
                                    if out['relation'] is not None else None)
        if timer.records or (table_from_eager_step and table_records):
            live = timer.summary() if timer.records else {}   # the timed region: the dominant entry (or everything: --timing all)
            if table_records is None:
                table_records = timer.records
            agg = timer.summary(table_records)                # every C-ABI entry: last warm-up step, or the timed region
            for k in live:
                agg[k] = dict(live[k])                        # where both exist the timed region's figures are the ones shown
                for f in ('calls', 'ms', 'bytes', 'flops'):
                    agg[k][f] = live[k][f] * table_steps / args.steps
            kern = {}
            for k, d in sorted(agg.items(), key=lambda kv: -kv[1]['ms']):
                per = d['ms'] / d['calls']
                kern[k] = dict(calls_per_step=d['calls'] / table_steps, avg_ms=per,
                               GBps=d['bytes'] / d['calls'] / per / 1e6 if per > 0 else None,
                               TFLOPs=d['flops'] / d['calls'] / per / 1e9 if per > 0 and d['flops'] else None,
                               ms_per_step=d['ms'] / table_steps,
                               measured_in='timed region' if (k in live or table_steps != 1) else 'last warm-up step')
            line['kernels'] = kern
            hw_ms = sum(d['ms'] for d in agg.values()) / table_steps
            if lib_flops is not None and hw_flops_one is not None:
                f0 = lib_flops + hw_flops_one                 # this rank's share of the step
                ideal_ms = f0 / (F32_MFMA_PEAK_TF * 1e12) * 1e3
                line['roofline_step'] = dict(
                    bound='mfma', algorithmic_TFLOP_per_step=f0 * world / 1e12, library_TFLOP_per_gpu=lib_flops / 1e12,
                    handwritten_TFLOP_per_gpu=hw_flops_one / 1e12, peak=F32_MFMA_PEAK_TF, unit='TFLOP/s per GPU',
                    achieved=f0 / (ms_per_step * 1e-3) / 1e12, f32_equivalent_over_f32_peak=ideal_ms / ms_per_step,
                    ideal_ms_per_step=ideal_ms,
                    handwritten_kernel_ms_per_step=hw_ms, other_ms_per_step=ms_per_step - hw_ms,
                    note='rank 0; other = library kernels (MIOpen / rocBLAS / hipBLASLt / ATen) + launch gaps + host syncs')
                # the same step against the pipes its kernels actually issue on: every launch's ISSUED matrix flops (limb
                # products for the split kernels, Winograd's transformed multiplies) over that pipe's dense peak, summed.
                # `frac` above prices the model's f32 arithmetic at the f32 matrix peak although most of it executes as
                # 3 (f16x2) or 6 (bf16x3) limb products on the 16-bit pipe; this is the honest utilisation figure.
                pipe_ms = sum(d['flops'] / table_steps / (KernelTimer.mfma_peak(k.split('[')[0])[0] * 1e12) * 1e3 for k, d in agg.items())
                # `frac` = that utilisation; `f32_equivalent_over_f32_peak` (> 1 is possible: the f32 arithmetic runs on the
                # 16 x faster 16-bit pipe) is kept for comparison with rounds 1-5, where it was called `frac`
                line['roofline_step'].update(ideal_ms_on_pipes_used=pipe_ms, frac_of_pipe_used=pipe_ms / ms_per_step,
                                             frac=pipe_ms / ms_per_step)
            # graph replay in the timed region: the dominant entry and its figures come from the eager warm-up step's events
            pool, pool_steps = (agg, table_steps) if table_from_eager_step else (live, args.steps)
            dom = max((k for k in pool if pool[k]['bytes'] > 0), key=lambda k: pool[k]['ms'])
            d = pool[dom]
            per = d['ms'] / d['calls']
            # the roof that binds = the larger ideal time (bytes / HBM peak vs flops / f32 matrix peak) over its launches
            peak_tf, peak_note = KernelTimer.mfma_peak(dom)
            mfma_bound = d['flops'] / (peak_tf * 1e12) > d['bytes'] / (HBM_PEAK_GBS * 1e9)
            # HBM-side bytes per launch from the PMC passes of scripts/pmc_traffic.py (FETCH_SIZE / WRITE_SIZE in
            # separate rocprofv3 runs of THIS bench at T=32, gfx950 read correction applied), keyed by C-ABI entry;
            # null when the file has no entry for this kernel / clip length
            traffic = None
            tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
            if os.path.exists(tpath):
                ent = json.load(open(tpath)).get(dom.split('[')[0], {})
                if ent.get('frames') == t_local:
                    traffic = ent.get('hbm_bytes_per_launch')
            scope = (('dominant HAND-WRITTEN kernel by time; the timed region replays the step as a hipGraph (no per-launch events), so '
                      'its launches were timed with HIP-event pairs in ONE EAGER warm-up step of the same clip.  ' if table_from_eager_step else
                      'dominant HAND-WRITTEN kernel by time, its launches timed with HIP-event pairs INSIDE the timed region.  ') +
                     ('The other entries of `kernels` were timed the same way during the last warm-up step (events around all '
                      '370 launches cost 1.7 ms per step; --timing all puts them into the timed region).  '
                      if table_steps == 1 else 'HIP-event pairs sit around EVERY C-ABI launch inside the timed region '
                      '(ms_per_step includes their cost).  ') +
                     'No library GEMM / convolution is left on the path (roofline_step = the whole step, time split '
                     'hand-written / other)')
            if mfma_bound:
                ach = d['flops'] / d['calls'] / per / 1e9
                line['roofline'] = dict(kernel=dom, bound='mfma', achieved=ach, peak=peak_tf, flops_counted=peak_note,
                                        unit='TFLOP/s', frac=ach / peak_tf, traffic=traffic,
                                        avg_launch_ms=per, launches_per_step=d['calls'] / pool_steps,
                                        algorithmic_bytes_per_launch=d['bytes'] / d['calls'], scope=scope)
            else:
                ach = d['bytes'] / d['calls'] / per / 1e6
                line['roofline'] = dict(kernel=dom, bound='hbm', achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s',
                                        frac=ach / HBM_PEAK_GBS, traffic=traffic, avg_launch_ms=per,
                                        launches_per_step=d['calls'] / pool_steps,
                                        algorithmic_bytes_per_launch=d['bytes'] / d['calls'], scope=scope)
            # the same for the dominant kernel FUNCTION (a C-ABI entry may dispatch to several: pvsg_gemm_f16x2 runs the wide
            # layers on gemm_f16x2_ln128_kernel and the rest on gemm_f16x2_dma_kernel), from the same HIP-event records
            fagg = {}
            for name, a, s_ev, e_ev in table_records:
                fn = kernel_function(name, a)
                by, fl = KernelTimer.work(name, a)
                fd = fagg.setdefault(fn, dict(calls=0, ms=0.0, bytes=0.0, flops=0.0, entry=name))
                fd['calls'] += 1
                fd['ms'] += s_ev.elapsed_time(e_ev)
                fd['bytes'] += by
                fd['flops'] += fl
            fdom = max((k for k in fagg if fagg[k]['bytes'] > 0), key=lambda k: fagg[k]['ms'])
            fd = fagg[fdom]
            fper = fd['ms'] / fd['calls']
            fpeak_tf, fnote = KernelTimer.mfma_peak(fd['entry'])
            f_mfma = fd['flops'] / (fpeak_tf * 1e12) > fd['bytes'] / (HBM_PEAK_GBS * 1e9)
            ftraffic = None
            if os.path.exists(tpath) and sum(1 for v in fagg.values() if v['entry'] == fd['entry']) == 1:   # PMC traffic is keyed by C-ABI entry
                ent = json.load(open(tpath)).get(fd['entry'], {})
                if ent.get('frames') == t_local:
                    ftraffic = ent.get('hbm_bytes_per_launch')
            fach = fd['flops'] / fd['calls'] / fper / 1e9 if f_mfma else fd['bytes'] / fd['calls'] / fper / 1e6
            fpk = fpeak_tf if f_mfma else HBM_PEAK_GBS
            line['roofline_function'] = dict(
                kernel_function=fdom, c_abi_entry=fd['entry'], bound='mfma' if f_mfma else 'hbm', achieved=fach, peak=fpk,
                unit='TFLOP/s' if f_mfma else 'GB/s', frac=fach / fpk, traffic=ftraffic, avg_launch_ms=fper,
                launches_per_step=fd['calls'] / table_steps, ms_per_step=fd['ms'] / table_steps,
                algorithmic_bytes_per_launch=fd['bytes'] / fd['calls'],
                functions_ms_per_step={k: v['ms'] / table_steps for k, v in sorted(fagg.items(), key=lambda kv: -kv[1]['ms'])[:8]},
                scope='dominant kernel FUNCTION by time over the same HIP-event records as `kernels` (`roofline` is the dominant '
                      'C-ABI entry, whose launches may be spread over several functions)')
            if dom.startswith('pvsg_conv1x1_f16x2'):
                # measured on this kernel itself (round 5): socket power / shader clock while one layer loops, and the timing
                # ablations (no MFMAs / no stores) of the two heaviest shapes
                line['roofline']['limited_by'] = (
                    'mixed over its 41 launches: the HBM-heavy layers (256->64, 64->256 at 184x320) stream 4.7-5.3 TB/s of algorithmic '
                    'bytes (0.58-0.66 of 8 TB/s; 0.74-0.83 of the ~6.3 TB/s a pure streaming kernel reaches here) at 1290-1385 W with '
                    'the shader clock un-throttled (2.35-2.38 GHz) and run no faster with the MFMAs removed; the deep layers '
                    '(512->128, 256->1024) sit at the 1400 W socket limit with the clock down to 1.87-2.05 GHz')
                line['roofline']['limited_by_source'] = 'profiles/r05_power_probe_conv1x1.txt (scripts/lab/r05_power_conv.sh)'
            elif dom.startswith('pvsg_gemm_f16x2'):
                # the token GEMM leads once the layer1 bottleneck tails have left the 1x1-convolution entry (round 5): its time is the
                # encoder's FFN1 (6 x 1.24 ms) and 544-wide projection (6 x 0.76 ms) -- probed in a loop with rocm-smi
                line['roofline']['limited_by'] = (
                    'socket power: in a loop FFN1 (618 240 x 256 -> 1024) draws 1375-1381 W at 2.21 GHz and the 544-wide projection 1392 W '
                    'at 2.11 GHz of the 1400 W limit (2.4 GHz nominal) while issuing 0.26-0.31 of the dense f16 MFMA peak in limb products '
                    '(3 per f32 multiply-add); the aggregate entry is priced against HBM because the 18 small decoder K/V projections '
                    'dominate its launch count, not its time')
                line['roofline']['limited_by_source'] = ('profiles/r05_power_probe_gemm.txt (scripts/lab/power_probe.py ffn1 | proj544), '
                                                         'profiles/r04_power_probe.txt')
            # the four kernels the north-star names, each against its own roof (same HIP-event data)
            named = []
            for key, bound in (('pvsg_gemm_f16x2', 'mfma'), ('pvsg_gemm_bf16x3', 'mfma'), ('pvsg_conv3x3_f16x2', 'mfma'),
                               ('pvsg_msda_fused_forward', 'hbm'), ('pvsg_ms_deform_attn_forward', 'hbm'),
                               ('pvsg_mask_logits_forward', 'mfma'), ('pvsg_mask_logits_bf16x3', 'mfma'),
                               ('pvsg_mask_logits_f16x2', 'mfma'),
                               ('pvsg_attn_mask_bits_forward', 'mfma'), ('pvsg_attn_mask_bits_bf16x3', 'mfma'),
                               ('pvsg_attn_mask_bits_f16x2', 'mfma'), ('pvsg_attn_mask_bits_packed_f16x2', 'mfma'),
                               ('pvsg_gemm_f16x2_add_layernorm', 'mfma'),
                               ('pvsg_masked_xattn_partial', 'mfma'), ('pvsg_pair_score_forward', 'latency')):
                ks = [k for k in agg if k.startswith(key)]
                if not ks:
                    continue
                k = max(ks, key=lambda n: agg[n]['ms'])
                dd = agg[k]
                per_ms = dd['ms'] / dd['calls']
                if bound == 'hbm':
                    a_ = dd['bytes'] / dd['calls'] / per_ms / 1e6
                    named.append(dict(kernel=k, bound='hbm', achieved=a_, peak=HBM_PEAK_GBS, unit='GB/s', frac=a_ / HBM_PEAK_GBS))
                    # the deformable-attention gather is bound by the texture path (TA busy 80-87 %, profiles/r03_msda_pmc.txt),
                    # not by HBM: second entry = measured 64-byte L1 accesses per query x queries of THIS launch / time,
                    # against 256 CUs x 64 B/clk x 2.4 GHz
                    tex = os.path.join(ROOT, 'profiles', 'msda_texture_path.json')
                    kern = {'pvsg_msda_fused_forward': 'msda_fused_m8d32', 'pvsg_ms_deform_attn_forward': 'msda_fwd_m8d32'}.get(k)
                    if kern and os.path.exists(tex):
                        ent = json.load(open(tex)).get(kern)
                        args0 = next(a for n, a, _, _ in table_records if n == k)
                        queries = args0[9] * args0[13] if k == 'pvsg_msda_fused_forward' else args0[6] * args0[10]
                        if ent:
                            tb = ent['l1_accesses_per_query'] * queries * 64.0 / per_ms / 1e9          # TB/s through the L1
                            peak = 256 * 64 * 2.4e9 / 1e12
                            rec = dict(kernel=k, bound='ta_l1', achieved=tb, peak=peak, unit='TB/s (vector-L1 / texture path)',
                                       frac=tb / peak, l1_accesses_per_query=ent['l1_accesses_per_query'],
                                       ta_busy_frac_under_pmc=ent['ta_busy_frac'], source='profiles/msda_texture_path.json')
                            gc = json.load(open(tex)).get('gather_ceiling_GBps_per_cu')
                            if gc:       # what PURE gathers of this shape reach (scripts/lab/vmem_ceiling.hip), blended by the kernel's hit rates
                                h1, h2 = ent['l1_hit_rate'], ent['l2_hit_rate']
                                blend = 1.0 / (h1 / gc['l1'] + (1 - h1) * h2 / gc['l2'] + (1 - h1) * (1 - h2) / gc['beyond_l2'])
                                rec['measured_gather_ceiling_TBps'] = blend * 256 / 1e3
                                rec['frac_of_measured_gather_ceiling'] = tb / (blend * 256 / 1e3)
                                rec['gather_ceiling_source'] = gc['source']
                            named.append(rec)
                elif bound == 'mfma':
                    a_ = dd['flops'] / dd['calls'] / per_ms / 1e9
                    pk, what = KernelTimer.mfma_peak(k.split('[')[0])
                    ent = dict(kernel=k, bound='mfma', achieved=a_, peak=pk, unit='TFLOP/s', frac=a_ / pk, flops_counted=what)
                    # the roof that binds = the larger ideal time, as for `roofline`: the mask projection streams 2.7 GB per launch
                    # through 0.2 TFLOP of limb products -- its matrix fraction says little, its HBM fraction is the bound
                    gbs = dd['bytes'] / dd['calls'] / per_ms / 1e6
                    if dd['bytes'] / (HBM_PEAK_GBS * 1e9) > dd['flops'] / (pk * 1e12):
                        ent = dict(kernel=k, bound='hbm', achieved=gbs, peak=HBM_PEAK_GBS, unit='GB/s', frac=gbs / HBM_PEAK_GBS,
                                   mfma_TFLOPs=a_, mfma_frac=a_ / pk, flops_counted=what)
                    else:
                        ent['hbm_GBps'], ent['hbm_frac'] = gbs, gbs / HBM_PEAK_GBS
                    if k.split('[')[0] in KernelTimer.SPLIT_F16X2 or 'bf16x3' in k:
                        # measured with rocm-smi while one layer loops (scripts/lab/power_probe.py, profiles/r04_power_probe.txt):
                        # the split kernels run at the socket's 1400 W limit with the shader clock throttled to 1.6-2.2 GHz
                        ent['limited_by'] = 'socket power (1400 W), profiles/r04_power_probe.txt, profiles/r05_power_probe_conv1x1.txt'
                    if pk != F32_MFMA_PEAK_TF:      # split kernels: also the model's f32 arithmetic against the f32 matrix roof
                        limb_products = 3.0 if k.split('[')[0] in KernelTimer.SPLIT_F16X2 else 6.0
                        ent['f32_equivalent_TFLOPs'] = a_ / limb_products
                        ent['f32_equivalent_frac_of_f32_mfma_peak'] = a_ / limb_products / F32_MFMA_PEAK_TF
                    named.append(ent)
                else:
                    named.append(dict(kernel=k, bound='launch-latency', avg_launch_us=per_ms * 1e3))
            line['roofline_named_kernels'] = named
        if world == 1 and args.sub_benchmarks == 'on' and not ips:
            try:
                line['sub_benchmarks'] = sub_benchmarks(det, rel, pipe, args, dev)
                line['sub_benchmarks']['head_only'] = head_only_benchmark(det, pipe, clip_local, step)
                feats = det.extract_feat(clip_local)
                line['fastpath_mask_bit_flip_rate'] = fastpath_bit_flip_rate(det.panoptic_head, feats, T)
                del feats
            except Exception as e:
                line['sub_benchmarks'] = dict(failed=repr(e))
        if args.cpu_baseline != 'off' and world == 1 and ips:
            try:
                base, parity = cpu_baseline_and_parity_ips(det, pipe, args, dev, args.cpu_frames, args.cpu_reps)
                line['cpu_baseline'], line['parity_on_cpu_sample'] = base, parity
                line['speedup_vs_cpu_baseline'] = fps / base['value']
            except Exception as e:  # the bench line must still be printed
                line['cpu_baseline'] = dict(value=None, unit='frames/s', cores=host_cores(), kind='port', sample='failed: %r' % (e,))
        elif args.cpu_baseline != 'off' and world == 1:
            try:
                base, parity = cpu_baseline_and_parity(det, rel, pipe, args, dev, args.cpu_frames, args.cpu_reps)
                line['cpu_baseline'] = base
                line['parity_on_cpu_sample'] = parity
                line['speedup_vs_cpu_baseline'] = fps / base['value']
                # the reduced-T sample is reported IN ADDITION to the full clip.  The full
                # 32-frame oracle run takes ~5 min (1 warm-up + 1 timed): `--cpu-full` measures it now, otherwise the line
                # carries the run recorded under profiles/ (same oracle, same box class, file named in `source`)
                if args.cpu_full:
                    full, _ = cpu_baseline_and_parity(det, rel, pipe, args, dev, T, 1, check=False)
                    full['measured'] = 'this run'
                    base['full_clip'] = full
                else:
                    rec = os.path.join(ROOT, 'profiles', 'cpu_full_clip.json')
                    if os.path.exists(rec) and T == 32 and (args.height, args.width) == (720, 1280):
                        full = json.load(open(rec))
                        full['measured'] = 'recorded run (profiles/cpu_full_clip.json), not this run'
                        base['full_clip'] = full
                if 'full_clip' in base:
                    base['full_clip']['speedup_of_this_run'] = fps / base['full_clip']['value']
            except Exception as e:  # the bench line must still be printed
                line['cpu_baseline'] = dict(value=None, unit='frames/s', cores=host_cores(), kind='port',
                                            sample='failed: %r' % (e,))
        print(json.dumps(line))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
