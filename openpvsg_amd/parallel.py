"""Multi-GPU layout of the hot path: one process per GPU, torch.distributed (backend 'nccl' = RCCL
over xGMI on ROCm; 'gloo' in the CPU tests).

The reference is data-parallel only (tools/test.py:244-252).  The north-star shards ONE clip by
frame: backbone, pixel decoder, mask projection and post-processing are per-frame and stay local;
two exchanges exist, both tiny and latency-bound:
  1. clip-level masked attention: every rank streams its frames' keys and publishes the
     un-normalised partial (o, m, l) per query and head (13.6 KB/head); `all_gather` + the same
     log-sum-exp merge kernel that joins key ranges on one GPU.  The per-query "has an unblocked
     key" flags are OR-ed over ranks first (the reset of mask2former_head.py:453-454 is a global
     property of the clip).
  2. tube assembly before relation scoring: `all_gather` of the per-frame records
     (segment id per query per frame, + per-frame queries/logits in per-frame mode).
Messages are <= a few MB: one direct all-gather over the 7 xGMI links, no ring tuning needed.
"""
import torch
import torch.distributed as dist


def is_dist(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def shard_frames(num_frames, rank, world):
    """Contiguous frame range of `rank` (frames must divide evenly: 32 over 1/2/4/8)."""
    if num_frames % world:
        raise ValueError('clip of %d frames does not split over %d ranks' % (num_frames, world))
    per = num_frames // world
    return rank * per, per


def all_gather_cat(t, dim=0, group=None):
    """all_gather of equally-shaped tensors, concatenated along `dim` (rank order)."""
    if not is_dist(group):
        return t
    world = dist.get_world_size(group)
    t = t.contiguous()
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


class ClipShard:
    """Wires a Mask2FormerVideoHead for a frame-sharded clip: flags OR + partial all-gather."""

    def __init__(self, head, total_frames, group=None):
        self.head, self.group = head, group
        self.rank = dist.get_rank(group) if is_dist(group) else 0
        self.world = dist.get_world_size(group) if is_dist(group) else 1
        self.t0, self.t_local = shard_frames(total_frames, self.rank, self.world)
        head.clip_frame_offset, head.clip_total_frames = self.t0, total_frames
        head.partial_combine = self.combine
        head.mask_sync = self.sync_mask

    def sync_mask(self, mask):
        if mask is not None:
            mask.flags = or_flags(mask.flags, self.group)
        return mask

    def combine(self, part_o, part_ml):
        from . import ops
        return ops.xattn_combine(all_gather_cat(part_o, 1, self.group), all_gather_cat(part_ml, 1, self.group))

    def release(self):
        self.head.partial_combine = None
        self.head.mask_sync = None
        self.head.clip_frame_offset, self.head.clip_total_frames = 0, None
