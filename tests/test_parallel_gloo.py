"""N > 1 path on CPU: world_size-2 gloo processes exercise the frame sharding, the flag OR, the
attention-partial exchange + log-sum-exp merge, and the tube-record all-gather (openpvsg_amd/parallel.py).
The GPU kernels are replaced by their torch statements here; the exchange logic is the product's."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def partial_attention(q, k, v, blocked, honor):
    """torch statement of xattn_partial_kernel for ONE key range: q (B,M,Q,D) scaled, k/v (B,M,K,D),
    blocked (B,Q,K) bool, honor (B,Q) bool -> o (B,1,M,Q,D), ml (B,1,M,Q,2)."""
    s = torch.einsum('bmqd,bmkd->bmqk', q, k)
    eff = blocked & honor[:, :, None]
    s = s.masked_fill(eff[:, None], float('-inf'))
    m = s.max(-1).values
    p = torch.exp(s - torch.where(torch.isinf(m), torch.zeros_like(m), m)[..., None])
    p = torch.where(torch.isinf(m)[..., None], torch.zeros_like(p), p)
    o = torch.einsum('bmqk,bmkd->bmqd', p, v)
    return o[:, None], torch.stack([m, p.sum(-1)], -1)[:, None]


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from openpvsg_amd import parallel
    from openpvsg_amd.pipeline import assemble_tubes
    torch.manual_seed(0)   # same "clip" on every rank
    B, M, Q, D, T, hw = 1, 8, 20, 32, (4 if world == 2 else 8), 6       # world 8: ONE frame per rank (config 4's layout)
    K = T * hw
    q = torch.randn(B, M, Q, D) * 0.3
    k, v = torch.randn(B, M, K, D), torch.randn(B, M, K, D)
    blocked = torch.rand(B, Q, K) < 0.5
    blocked[0, 3] = True                 # blocked everywhere -> reset: attends to every key
    blocked[0, 5] = True
    blocked[0, 5, K - 1] = False         # its only allowed key lives on the LAST rank
    # ---- frame sharding -------------------------------------------------------------------------
    t0, tl = parallel.shard_frames(T, rank, world)
    assert (t0, tl) == (rank * T // world, T // world)
    with pytest.raises(ValueError):
        parallel.shard_frames(5, rank, world)
    ks = slice(t0 * hw, (t0 + tl) * hw)
    # ---- flags: OR over ranks (NCCL has no BOR -> gather + fold) ---------------------------------
    local_has = (~blocked[:, :, ks]).any(-1)                               # (B,Q)
    words = torch.zeros(B, 4, dtype=torch.int32)
    for qq in range(Q):
        if local_has[0, qq]:
            words[0, qq // 32] |= (1 << (qq % 32))
    allw = parallel.or_flags(words)
    honor = torch.tensor([[(int(allw[0, qq // 32]) >> (qq % 32)) & 1 for qq in range(Q)]]).bool()
    assert honor[0, 5] and not honor[0, 3]
    assert torch.equal(honor, (~blocked).any(-1))
    # ---- partial exchange + merge ---------------------------------------------------------------
    po, pml = partial_attention(q, k[:, :, ks], v[:, :, ks], blocked[:, :, ks], honor)
    go, gml = parallel.all_gather_cat(po, 1), parallel.all_gather_cat(pml, 1)
    assert go.shape[1] == world
    merged = parallel.merge_partials_reference(go, gml)                    # (B,Q,M*D)
    eff = blocked & honor[:, :, None]
    s = torch.einsum('bmqd,bmkd->bmqk', q, k).masked_fill(eff[:, None], float('-inf'))
    full = torch.einsum('bmqk,bmkd->bmqd', s.softmax(-1), v).permute(0, 2, 1, 3).reshape(B, Q, M * D)
    assert torch.allclose(merged, full, rtol=1e-5, atol=1e-6)
    # ---- the product's exchange: ONE message per layer, attention on LOCAL flags, reset decided by the merge -------------
    local_honor = torch.tensor([[(int(words[0, qq // 32]) >> (qq % 32)) & 1 for qq in range(Q)]]).bool()
    lo, lml = partial_attention(q, k[:, :, ks], v[:, :, ks], blocked[:, :, ks], local_honor)   # locally blocked -> unmasked
    rec = parallel.pack_record_reference(lo, lml, words)
    allrec = parallel.all_gather_cat(rec[None], 0)                          # (R, B, REC)
    assert allrec.shape == (world, B, M * Q * (D + 2) + 4)
    merged2 = parallel.merge_records_reference(allrec, Q, M, D)
    assert torch.allclose(merged2, full, rtol=1e-5, atol=1e-6)
    # query 5 is blocked on rank 0 only: rank 0 attended unmasked, and that contribution must have been dropped
    if rank == 0:
        assert not local_honor[0, 5] and lml[0, 0, 0, 5, 1] > 0
    # ---- tube records: every rank rebuilds the same tubes -----------------------------------------
    ids_all = torch.tensor([[1005, 120, -1], [1005, -1, 2007], [-1, 120, 2007], [1005, 120, 2007]] * (T // 4))
    feats = torch.arange(3 * 8, dtype=torch.float32).view(3, 8)
    local = ids_all[t0:t0 + tl]
    gathered = parallel.all_gather_cat(local, 0)
    assert torch.equal(gathered, ids_all)
    tube_ids, tubes = assemble_tubes(list(gathered.unbind(0)), [feats] * T, T)
    assert tube_ids.tolist() == [1005, 120, 2007]
    assert tubes.shape == (3, T, 8) and torch.equal(tubes[0, 2], torch.zeros(8)) and torch.equal(tubes[2, 1], feats[2])
    # the device tail's layout (pipeline._tail_device): every rank's id rows + ONE extra row carrying its f16x2 overflow
    # count; after the gather every rank reads the same rows and the same per-rank counts (here: only rank world-1 overflowed)
    rows = torch.full((tl + 1, 128), -1, dtype=torch.int32)
    rows[:tl, :3] = local.to(torch.int32)
    rows[tl, 0] = 7 if rank == world - 1 else 0
    allrows = parallel.all_gather_cat(rows, 0).view(world, tl + 1, 128)
    assert torch.equal(allrows[:, :tl, :3].reshape(T, 3).long(), ids_all)
    assert allrows[:, tl, 0].tolist() == [0] * (world - 1) + [7]
    # the host tail's agreement on the same count (pipeline._forward -> parallel.agree_max): every rank sees the maximum
    assert parallel.agree_max(torch.tensor([3 * rank, 0, 0, 0], dtype=torch.int32)) == 3 * (world - 1)
    torch.save(dict(ok=True, merged=merged), os.path.join(tmp, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4, 8])
def test_two_rank_exchange(tmp_path, world):
    """world 2, 4 and 8 (tools/test.py:186-190's launcher sizes): frame ranges, R-record merge, id rows with one overflow row
    per rank, overflow agreement."""
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), 'r%d.pt' % r)) for r in range(world)]
    assert all(p['ok'] for p in parts) and all(torch.equal(parts[0]['merged'], p['merged']) for p in parts[1:])


def test_single_process_helpers_are_identity():
    from openpvsg_amd import parallel
    x = torch.arange(6).view(2, 3)
    assert parallel.all_gather_cat(x, 0) is x and parallel.or_flags(x) is x
    assert parallel.shard_frames(32, 0, 1) == (0, 32) and parallel.shard_frames(32, 3, 8) == (12, 4)


def test_merge_reference_handles_empty_ranges():
    from openpvsg_amd import parallel
    o = torch.zeros(1, 3, 8, 4, 32)
    ml = torch.zeros(1, 3, 8, 4, 2)
    ml[:, 0, ..., 0] = float('-inf')       # an all-blocked range: m=-inf, l=0, o=0
    o[:, 1] = 2.0
    ml[:, 1, ..., 1] = 4.0                 # m=0, l=4, o=2  -> 0.5
    o[:, 2] = 6.0
    ml[:, 2, ..., 0] = 0.0
    ml[:, 2, ..., 1] = 4.0
    out = parallel.merge_partials_reference(o, ml)
    assert torch.allclose(out, torch.full_like(out, 1.0))


def test_bench_gpus_flag_self_spawns_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` with no launcher around it must re-execute itself under torch.distributed.run
    with N ranks (tools/test.py:186-190 leaves that to its launcher script); with WORLD_SIZE set it must not."""
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, 'call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '2'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4' and cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-4:] == ['--gpus', '4', '--steps', '2'] and seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    # a launcher's WORLD_SIZE that disagrees with --gpus is an error, not a silent 1-GPU run
    monkeypatch.setenv('WORLD_SIZE', '2')
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert 'WORLD_SIZE=2' in str(e.value.code)


def test_cu_mask_is_derived_from_the_kfd_topology(tmp_path, monkeypatch):
    """parallel.isolate_shared_gpu partitions the CUs the device really has (no hard-coded 256): the count comes
    from the KFD topology in sysfs, read without starting the HIP runtime."""
    import os
    from openpvsg_amd import parallel
    for node, simd in (('0', 0), ('1', 1216), ('2', 1024)):                 # node 0 = the CPU, two GPUs with 304 / 256 CUs
        d = tmp_path / node
        d.mkdir()
        (d / 'properties').write_text('cpu_cores_count 64\nsimd_count %d\nsimd_per_cu 4\n' % simd)
    monkeypatch.setattr(parallel, 'KFD_NODES', str(tmp_path))
    assert parallel.device_cu_count(0) == 304 and parallel.device_cu_count(1) == 256 and parallel.device_cu_count(5) == 256
    monkeypatch.delenv('HSA_CU_MASK', raising=False)
    parallel.isolate_shared_gpu(1, 2, device_index=0)
    assert os.environ['HSA_CU_MASK'] == '0:152-303'
    monkeypatch.delenv('HSA_CU_MASK', raising=False)
    monkeypatch.setattr(parallel, 'KFD_NODES', str(tmp_path / 'missing'))
    parallel.isolate_shared_gpu(0, 4)
    assert os.environ['HSA_CU_MASK'] == '0:0-63'
    monkeypatch.delenv('HSA_CU_MASK', raising=False)


def test_second_tenant_on_the_gpu_is_announced(tmp_path, monkeypatch):
    """A second process with live compute queues and no CU partition: the library says so when it is loaded (ADVICE r2,
    co-residency finding), and stays quiet with HSA_CU_MASK set or nobody else around."""
    import os
    import warnings
    from openpvsg_amd import parallel
    (tmp_path / str(os.getpid()) / 'queues' / '0').mkdir(parents=True)      # ourselves: ignored
    (tmp_path / '4242' / 'queues').mkdir(parents=True)                      # a process without queues: ignored
    monkeypatch.setattr(parallel, 'KFD_PROC', str(tmp_path))
    monkeypatch.delenv('HSA_CU_MASK', raising=False)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        parallel.warn_if_gpu_shared()
    (tmp_path / '4243' / 'queues' / '7').mkdir(parents=True)
    assert parallel.other_gpu_processes() == [4243]
    with pytest.warns(RuntimeWarning, match='isolate_shared_gpu'):
        parallel.warn_if_gpu_shared()
    monkeypatch.setenv('HSA_CU_MASK', '0:0-127')
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        parallel.warn_if_gpu_shared()
