"""Multi-GPU layout of the hot path: one process per GPU, torch.distributed (backend 'nccl' = RCCL
over xGMI on ROCm; 'gloo' in the CPU tests).

The reference is data-parallel only (tools/test.py:244-252).  The north-star shards ONE clip by
frame: backbone, pixel decoder, mask projection and post-processing are per-frame and stay local;
two exchanges exist, both tiny and latency-bound:
  1. clip-level masked attention, ONE message per decoder layer: every rank streams its frames' keys, merges its
     own key ranges locally and publishes one record = the un-normalised partial (o, m, l) per query and head +
     its 128 LOCAL "this query has an unblocked key among my keys" flag bits (108.8 KB); `all_gather`, then the
     merge kernel.  The all-blocked reset of mask2former_head.py:453-454 is a property of the whole clip: a rank
     that sees a query fully blocked attends unmasked, and the merge counts that contribution only if EVERY rank
     reported the query blocked -- so nothing has to be exchanged before the attention kernel runs.
  2. tube assembly before relation scoring: `all_gather` of the per-frame records
     (segment id per query per frame, + per-frame queries/logits in per-frame mode).
Messages are <= a few MB: one direct all-gather over the 7 xGMI links, no ring tuning needed.
"""
import os

import torch
import torch.distributed as dist


# PVSG_FORCE_COLLECTIVES=1: run the exchanges even in a world of ONE rank (all_gather_into_tensor of a single shard, the
# merge kernels on one record).  That is how a one-GPU box exercises the RCCL code path of the N > 1 layout end to end
# (tests/test_parallel_gpu.py::test_rccl_world_size_1_*); results are those of the purely local run.
FORCE_COLLECTIVES = os.environ.get('PVSG_FORCE_COLLECTIVES', '0') == '1'


def is_dist(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or FORCE_COLLECTIVES


KFD_NODES = '/sys/class/kfd/kfd/topology/nodes'


def device_cu_count(device_index=0, default=256):
    """Compute units of GPU `device_index` WITHOUT touching the HIP runtime (HSA_CU_MASK must be set before it starts):
    from the KFD topology in sysfs (`simd_count / simd_per_cu` of the device_index-th node that has SIMDs)."""
    base = KFD_NODES
    try:
        gpus = []
        for node in sorted(os.listdir(base), key=int):
            props = dict(line.split() for line in open(os.path.join(base, node, 'properties')) if len(line.split()) == 2)
            simd = int(props.get('simd_count', 0))
            if simd > 0:
                gpus.append(simd // max(1, int(props.get('simd_per_cu', 4))))
        return gpus[device_index] if device_index < len(gpus) else default
    except Exception:
        return default


def isolate_shared_gpu(slot, slots, device_index=0, cus=None):
    """Several PROCESSES on one MI355X (the gloo logic tests, `bench.py` with PVSG_ONE_DEVICE=1): give each its own
    range of compute units (HSA_CU_MASK), to be called before the process touches the HIP runtime.

    Why it is needed, not just tidy: waves of the split kernels (csrc/token_gemm.hip, conv1x1_split.hip: v_mfma_f32_32x32x16_bf16 at high occupancy) that are
    CO-RESIDENT on a CU with waves of ANOTHER process were observed to corrupt that process's results -- e.g. its
    deformable-attention gather returns wrong values in lanes 48-63 of a wave (heads 6-7) in 1-20 % of launches, on every
    box tried; the same kernel built on the f32 MFMA does not, and neither process is affected once their CU sets are
    disjoint (scripts/coresidency_probe.py; stand-alone reproducer scripts/coresidency_repro.hip: the gather next to a
    register-only v_mfma_f32_16x16x32_bf16 loop, also from a second stream of the SAME process).  The backend runs its
    kernels back to back on one stream, so this never arises in the supported deployment (one process per GPU, one stream)."""
    if slots > 1 and 'HSA_CU_MASK' not in os.environ:
        cus = cus or device_cu_count(device_index)
        per = cus // slots
        os.environ['HSA_CU_MASK'] = '%d:%d-%d' % (device_index, slot * per, (slot + 1) * per - 1)


KFD_PROC = '/sys/class/kfd/kfd/proc'


def other_gpu_processes():
    """PIDs of OTHER processes that hold compute queues on this node's GPUs right now (KFD sysfs; [] where it is not visible)."""
    root = KFD_PROC
    out = []
    try:
        for pid in os.listdir(root):
            if not pid.isdigit() or int(pid) == os.getpid():
                continue
            try:
                if os.listdir(os.path.join(root, pid, 'queues')):
                    out.append(int(pid))
            except OSError:
                continue
    except OSError:
        pass
    return out


def warn_if_gpu_shared():
    """Called once when the backend library is loaded: a second process with live queues on the GPU and no CU partition is
    the one setting in which the bf16 matrix kernels were seen to corrupt a neighbour's results (see isolate_shared_gpu)."""
    if 'HSA_CU_MASK' in os.environ or os.environ.get('PVSG_SHARED_GPU_WARNING', 'on') == 'off':
        return
    others = other_gpu_processes()
    if others:
        import warnings
        warnings.warn('openpvsg_amd: %d other process(es) hold compute queues on this node\'s GPU(s) (pids %s) and HSA_CU_MASK is '
                      'not set.  Waves of the bf16-MFMA kernels co-resident on a CU with another process\'s waves were observed to '
                      'corrupt that process\'s results (DESIGN.md section 3.7): run one process per GPU, or give the processes '
                      'disjoint CU ranges with openpvsg_amd.parallel.isolate_shared_gpu() before HIP starts '
                      '(PVSG_SHARED_GPU_WARNING=off silences this).' % (len(others), others[:8]), RuntimeWarning, stacklevel=3)


def shard_frames(num_frames, rank, world):
    """Contiguous frame range of `rank` (frames must divide evenly: 32 over 1/2/4/8)."""
    if num_frames % world:
        raise ValueError('clip of %d frames does not split over %d ranks' % (num_frames, world))
    per = num_frames // world
    return rank * per, per


EXCHANGE_TIMER = None      # measurements only (bench.py): a list that collects (bytes per rank, start event, end event) per exchange


def all_gather_cat(t, dim=0, group=None):
    """all_gather of equally-shaped tensors, concatenated along `dim` (rank order).  RCCL: one
    `all_gather_into_tensor` straight into the result (no per-rank staging tensors, no torch.cat launch)."""
    if not is_dist(group):
        return t
    world = dist.get_world_size(group)
    t = t.contiguous()
    if dist.get_backend(group) == 'nccl':
        out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        if EXCHANGE_TIMER is not None and t.is_cuda:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            dist.all_gather_into_tensor(out, t, group=group)
            e.record()
            EXCHANGE_TIMER.append((t.numel() * t.element_size(), s, e))
        else:
            dist.all_gather_into_tensor(out, t, group=group)
        if dim == 0:
            return out.flatten(0, 1)
        return torch.cat(list(out.unbind(0)), dim=dim)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return torch.cat(out, dim=dim)


def or_flags(flags, group=None):
    """Bitwise OR of the (B,4) int32 flag words over ranks (NCCL/RCCL has no BOR: gather + fold)."""
    if not is_dist(group):
        return flags
    world = dist.get_world_size(group)
    allf = all_gather_cat(flags[None], 0, group)          # (world, B, 4)
    out = allf[0]
    for r in range(1, world):
        out = out | allf[r]
    return out


def agree_max(count, group=None):
    """MAX over the ranks of a small integer tensor (the f16x2 overflow count before a branch that re-runs the clip: every
    rank must take the same branch, or the re-run's per-layer exchanges pair with another clip's on the peers).  -> int."""
    if not is_dist(group):
        return int(count.reshape(-1)[0].item())
    c = count.reshape(-1)[:1].clone()
    dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
    return int(c.item())


def merge_partials_reference(part_o, part_ml):
    """Pure-torch log-sum-exp merge of attention partials (B,NS,M,Q,D)/(B,NS,M,Q,2): the CPU/gloo
    statement of xattn_combine_kernel, used by the world_size-2 CPU tests."""
    m = part_ml[..., 0]
    l = part_ml[..., 1]
    mstar = m.max(dim=1, keepdim=True).values
    w = torch.where(torch.isinf(m) & (m < 0), torch.zeros_like(m), torch.exp(m - mstar))
    num = (w[..., None] * part_o).sum(1)
    den = (w * l).sum(1)
    out = num / den[..., None]                             # (B,M,Q,D)
    B, M, Q, D = out.shape
    return out.permute(0, 2, 1, 3).reshape(B, Q, M * D)


def pack_record_reference(part_o, part_ml, flags):
    """Pure-torch statement of xattn_merge_local_kernel: (B,NS,M,Q,D)/(B,NS,M,Q,2) + flags (B,4) int32 ->
    packed (B, M*Q*(D+2)+4) float32 (flag words bit-cast), for the CPU/gloo tests."""
    B, NS, M, Q, D = part_o.shape
    m, l = part_ml[..., 0], part_ml[..., 1]
    mstar = m.max(dim=1, keepdim=True).values
    w = torch.where(torch.isinf(m) & (m < 0), torch.zeros_like(m), torch.exp(m - torch.where(torch.isinf(mstar), torch.zeros_like(mstar), mstar)))
    num = (w[..., None] * part_o).sum(1)                    # (B,M,Q,D)
    den = (w * l).sum(1)                                   # (B,M,Q)
    ml = torch.stack([mstar[:, 0], den], -1)               # (B,M,Q,2)
    fl = flags.to(torch.int32).view(torch.float32)
    return torch.cat([num.reshape(B, -1), ml.reshape(B, -1), fl.reshape(B, 4)], 1)


def merge_records_reference(packed, Q, M=8, D=32):
    """Pure-torch statement of xattn_combine_packed_kernel on (R,B,REC) gathered records -> (B,Q,M*D)."""
    R, B, rec = packed.shape
    o = packed[..., :M * Q * D].reshape(R, B, M, Q, D)
    ml = packed[..., M * Q * D:M * Q * (D + 2)].reshape(R, B, M, Q, 2)
    fl = packed[..., M * Q * (D + 2):].contiguous().view(torch.int32)                       # (R,B,4)
    q = torch.arange(Q)
    mine = ((fl[..., q // 32] >> (q % 32)) & 1).bool()                                     # (R,B,Q)
    use = mine | ~mine.any(0, keepdim=True)                                               # reset only if blocked on every rank
    m = torch.where(use[:, :, None, :], ml[..., 0], torch.full_like(ml[..., 0], float('-inf')))
    part_o = o.permute(1, 0, 2, 3, 4)
    part_ml = torch.stack([m, ml[..., 1]], -1).permute(1, 0, 2, 3, 4)
    return merge_partials_reference(part_o, part_ml)


class ClipShard:
    """Wires a Mask2FormerVideoHead for a frame-sharded clip.  One exchange per decoder layer: every rank merges its
    key ranges locally (ops.xattn_merge_local), the 108.8 KB records (partial + the rank's LOCAL mask flags) are
    all-gathered, and ops.xattn_combine_packed applies the clip-wide all-blocked reset while merging -- the attention
    kernel runs on local flags, no flag exchange precedes it."""

    def __init__(self, head, total_frames, group=None):
        self.head, self.group = head, group
        self.rank = dist.get_rank(group) if is_dist(group) else 0
        self.world = dist.get_world_size(group) if is_dist(group) else 1
        self.t0, self.t_local = shard_frames(total_frames, self.rank, self.world)
        head.clip_frame_offset, head.clip_total_frames = self.t0, total_frames
        head.partial_combine = self.combine
        head.mask_sync = None

    def combine(self, part_o, part_ml, mask=None):
        from . import ops
        rec = ops.xattn_merge_local(part_o, part_ml, mask)                 # (B, REC)
        allrec = all_gather_cat(rec[None], 0, self.group)                  # (R, B, REC): the layer's one message
        return ops.xattn_combine_packed(allrec, part_o.shape[3], part_o.shape[2], part_o.shape[4])

    def release(self):
        self.head.partial_combine = None
        self.head.mask_sync = None
        self.head.clip_frame_offset, self.head.clip_total_frames = 0, None
