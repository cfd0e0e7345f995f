"""CPU, world_size 2 over gloo: the multi-rank plumbing bench.py uses (per-rank shards, barrier,
max-over-ranks timing).  The data path itself has no collective."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from uninext_amd import workloads
    dist.barrier()
    slow = bench.max_over_ranks(1.0 + rank, world, device="cpu")          # rank 1 is the slow one
    x = workloads.make_inputs("decoder", "model", batch=1, levels=((4, 5), (2, 3)), num_query=7,
                              seed=100 * rank, device="cpu")                # bench.build_inputs seeding rule
    digest = torch.tensor([float(x["loc"].sum())], dtype=torch.float64)
    both = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(both, digest)
    out.put((rank, slow, [float(t) for t in both]))
    dist.destroy_process_group()


def test_two_rank_timing_and_shards():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, slow, digests in res:
        assert slow == 2.0                       # MAX over ranks, identical on every rank
        assert digests[0] != digests[1]          # ranks hold different frames
    assert res[0][2] == res[1][2]


def test_single_rank_is_identity():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.max_over_ranks(0.25, 1) == 0.25
