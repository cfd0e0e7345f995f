"""BASELINE configs[4] on one GPU box: the DDP training step around the HIP backward kernels.

The reference wraps the model in DistributedDataParallel(broadcast_buffers=False)
(detectron2/engine/defaults.py:380-381 -> create_ddp_model) and shards the batch over ranks; the op's backward
(msda_hip_backward_f32) runs on autograd's thread of every rank and DDP all-reduces (mean) the fp32 gradients.
Here: 2 processes, both on cuda:0, gloo (a 1-GPU box has no second device for RCCL), each with one image of a
two-image batch.  The DDP-averaged parameter gradients must equal a single-process run on the concatenated batch
with the mean loss, to 1e-5 of the gradient scale; the input gradients of each rank must equal the matching half."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEVELS = ((24, 33), (12, 17), (6, 9), (3, 5))    # S = 1065: encoder-style call (Lq == S >= 1024) -> the tiled backward


def _inputs(device):
    from uninext_amd import workloads
    g = torch.Generator().manual_seed(11)
    S = sum(h * w for h, w in LEVELS)
    shapes, lsi = workloads.level_tensors(LEVELS, device)
    src = torch.randn(2, S, 256, generator=g)
    pos = torch.randn(2, S, 256, generator=g) * 0.1
    ref = workloads.encoder_reference_points(LEVELS, "cpu")[None, :, None, :].expand(2, S, len(LEVELS), 2).contiguous()
    wgt = torch.randn(2, S, 256, generator=g)
    return src.to(device), pos.to(device), ref.to(device), wgt.to(device), shapes, lsi


def _module(device):
    from uninext_amd.modules import MSDeformAttn
    torch.manual_seed(5)
    m = MSDeformAttn(256, 4, 8, 4)
    with torch.no_grad():   # non-trivial offsets / logits so that every parameter receives gradient
        m.sampling_offsets.weight.normal_(0, 0.02)
        m.attention_weights.weight.normal_(0, 0.05)
    return m.to(device)


def _loss(m, src, pos, ref, wgt, shapes, lsi):
    out = m(src + pos, ref, src, shapes, lsi, None)
    return (out * wgt).sum() / out.shape[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uninext_amd import _lib
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    src, pos, ref, wgt, shapes, lsi = _inputs(dev)
    m = _module(dev)
    ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0], broadcast_buffers=False)
    s = src[rank:rank + 1].clone().requires_grad_(True)
    loss = _loss(ddp, s, pos[rank:rank + 1], ref[rank:rank + 1], wgt[rank:rank + 1], shapes, lsi)
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().cpu().numpy() for n, p in m.named_parameters()}
    q.put((rank, grads, s.grad.detach().cpu().numpy(), _lib.last_kernel("backward"), float(loss)))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_match_single_process():
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0

    from uninext_amd import _lib
    dev = torch.device("cuda:0")
    src, pos, ref, wgt, shapes, lsi = _inputs(dev)
    m = _module(dev)
    s = src.clone().requires_grad_(True)
    loss = 0.5 * (_loss(m, s[0:1], pos[0:1], ref[0:1], wgt[0:1], shapes, lsi)
                  + _loss(m, s[1:2], pos[1:2], ref[1:2], wgt[1:2], shapes, lsi))
    loss.backward()
    torch.cuda.synchronize()
    assert _lib.last_kernel("backward").startswith("msda_bwd_")
    for rank, grads, sgrad, kern, rloss in res:
        assert kern.startswith("msda_bwd_"), kern          # the HIP backward ran on every rank
        for n, p in m.named_parameters():
            ref_g = p.grad.detach().cpu().numpy()
            scale = max(1e-12, float(np.abs(ref_g).max()))
            err = float(np.abs(grads[n] - ref_g).max()) / scale
            assert err < 1e-5, (rank, n, err)
        # input gradient of this rank's image: the single-process loss carries a factor 1/2
        ref_s = 2.0 * s.grad[rank:rank + 1].detach().cpu().numpy()
        err = float(np.abs(sgrad - ref_s).max()) / max(1e-12, float(np.abs(ref_s).max()))
        assert err < 1e-5, (rank, "src.grad", err)
    # both ranks hold identical (averaged) parameter gradients
    for n in res[0][1]:
        assert np.array_equal(res[0][1][n], res[1][1][n]), n


def _bench_ddp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import bench
    bench.DDP_GRAD_BYTES = 2 * bench.DDP_BUCKET_BYTES + 1          # three buckets: keep the gloo round trips short
    try:
        enc, dec = bench.build_inputs("model", rank)
        sync = bench.RankSync(world)                                # the legs are generators: run_leg meets the other rank at their yields
        res = bench.run_leg(sync, lambda: bench.measure_ddp(enc[:1], dec[:1], world, reps=1))
        st = bench.run_leg(sync, lambda: bench.measure_train_step(enc[:1], dec[:1], world, reps=1))
        q.put((rank, res, st))
    except Exception as e:  # noqa: BLE001  (a worker that dies silently costs the parent its whole queue timeout)
        q.put((rank, {"error": repr(e)}, {"error": repr(e)}))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_ddp_leg_runs_on_two_ranks():
    """bench.py's config-5 legs (`train_step`, `ddp`) with world_size 2: the driver runs them over RCCL on 2-8 GPUs;
    here the same code runs over gloo on one GPU so that a broken leg shows up before the scaling run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bench_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, ddp, st in res:
        assert "error" not in ddp and "error" not in st, (rank, ddp, st)
        assert ddp["buckets"] == 3 and ddp["allreduce_ms"] > 0 and ddp["overlapped_ms"] > 0 and ddp["op_fwd_bwd_ms"] > 0
        assert st["ms_per_step"] > 0
    assert res[0][1]["allreduce_ms"] == res[1][1]["allreduce_ms"]          # MAX over ranks on every rank
