"""Latency of a gloo all_gather of a 109 KB CUDA tensor between two processes on ONE GPU (the logic-test setting of
bench.py --backend gloo with PVSG_ONE_DEVICE=1); shows why that setting is a functional check, not a measurement."""
import os, sys, time
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def w(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    x = torch.randn(1, 27204, device='cuda')
    outs = [torch.empty_like(x) for _ in range(world)]
    a = torch.randn(4096, 4096, device='cuda')
    for busy in (False, True):
        for _ in range(3):
            dist.all_gather(outs, x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            if busy:
                for _ in range(4):
                    a @ a
            dist.all_gather(outs, x)
        torch.cuda.synchronize()
        if rank == 0:
            print('busy' if busy else 'idle', 'all_gather+work ms', (time.perf_counter() - t0) / 20 * 1e3)
    dist.destroy_process_group()


if __name__ == '__main__':
    mp.spawn(w, args=(2, 29533), nprocs=2, join=True)
