"""CPU, world_size 2 and 8 over gloo: the multi-rank plumbing bench.py uses (per-rank shards, barrier,
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


# ---- bench.main()'s control flow over two gloo ranks, GPU calls stubbed (VERDICT r04 "What's weak" #8) -------------------------
class _FakeLib:
    def load(self):
        return self

    def last_kernel(self, which):
        return "stub_" + which

    def set_variant(self, *a):
        pass


class _FakeEvent:
    def record(self):
        pass

    def elapsed_time(self, other):
        return 0.1


def _main_worker(rank, world, port, fail_rank, fail_where, out):
    import io
    import json
    import contextlib
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import bench

    def init_distributed(n):
        dist.init_process_group("gloo", rank=rank, world_size=world)
        return rank, world

    x = {"value": torch.zeros(2, 30, 8, 32), "loc": torch.zeros(2, 30, 8, 4, 4, 2)}
    calls = {"train_steps": 0}

    def train_step_fn(enc, dec):
        if fail_rank == rank and fail_where == "train_setup":
            raise RuntimeError("injected: set-up of the train step on rank %d" % rank)

        def step():
            calls["train_steps"] += 1
            if fail_rank == rank and fail_where == "train_timed" and calls["train_steps"] > 6:
                raise RuntimeError("injected: a launch of the timed train steps on rank %d" % rank)
        return step

    def build_train_inputs(flavour, r):
        if fail_rank == rank and fail_where == "train_inputs":
            raise MemoryError("injected: training inputs on rank %d" % rank)
        return [x] * 3, [x] * 3

    def rank0_leg():
        assert rank == 0, "a rank-0-only leg ran on rank %d" % rank
        return {"ran_on": rank}

    bench.torch.cuda.is_available = lambda: True
    bench.init_distributed = init_distributed
    bench._lib = _FakeLib()
    bench.build_inputs = lambda flavour, r: ([x] * 6, [x] * 6)
    bench.build_train_inputs = build_train_inputs
    bench.call = lambda *a, **k: None
    bench.run_step = lambda *a, **k: None
    bench.device_sync = lambda: None
    bench.new_event_pairs = lambda n: [(_FakeEvent(), _FakeEvent()) for _ in range(n)]
    bench.measure_flavours = lambda r: {"stub": True}
    bench.measure_forward_kernels = lambda enc: {"far_fraction": {"model": 0.02, "wide": 0.4}}
    bench.measure_backward = lambda e, d: {"stub": True}
    bench.measure_reference_module_on_top = lambda e: {"stub": True}
    bench.train_step_fn = train_step_fn
    bench.measure_model_slice = rank0_leg
    bench.measure_matcher = rank0_leg
    bench.committed_traffic = lambda *a: None
    bench.PREWARM_MS = 1.0
    bench.EXTRA_WARM_MS = 1.0
    bench.DDP_GRAD_BYTES, bench.DDP_BUCKET_BYTES = 4096, 1024
    bench._ddp_buckets = lambda n, n_el: [torch.ones(n_el) for _ in range(n)]

    class _Stream:
        def wait_stream(self, other):
            pass
    bench._side_stream = _Stream
    bench.torch.cuda.current_stream = lambda: _Stream()
    bench.torch.cuda.stream = lambda s: contextlib.nullcontext()

    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "4", "--warmup", "1", "--no-cpu-baseline"])
    line = buf.getvalue().strip()
    out.put((rank, json.loads(line) if line else None, dist.is_initialized()))


def _run_main(fail_rank, fail_where, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + {"none": 0, "train_setup": 1, "train_timed": 2, "train_inputs": 3}[fail_where] + 10 * world
    procs = [ctx.Process(target=_main_worker, args=(r, world, port, fail_rank, fail_where, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r, (line, alive)) for r, line, alive in (q.get(timeout=300) for _ in procs))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0           # nobody hung, nobody died
    for r in range(1, world):
        assert res[r][0] is None         # rank 0 prints the line, alone
    assert not any(res[r][1] for r in range(world))      # the process group was destroyed on every rank (behind the last exchange)
    return res[0][0]


def test_main_two_ranks_all_legs():
    line = _run_main(-1, "none")
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["scaling"] == "weak"
    assert line["train_step"]["ms_per_step"] > 0
    assert line["ddp"]["rccl_ranks"] == 2 and line["busbw"] == line["ddp"]["busbw_GBs"] > 0
    assert line["model_slice"] == {"ran_on": 0} or line["model_slice"]["ran_on"] == 0
    assert line["matcher"]["ran_on"] == 0


@pytest.mark.parametrize("where", ["train_setup", "train_timed", "train_inputs"])
def test_main_survives_a_leg_that_fails_on_one_rank(where):
    """Rank 1 raises inside a leg whose other ranks go on to a collective: every rank abandons the leg at the next
    exchange, the later legs (with real all-reduces: ddp) still run or are skipped on ALL ranks, the line is printed."""
    line = _run_main(1, where)
    assert line["value"] > 0 and line["n_gpus"] == 2
    if where == "train_inputs":
        assert "train_step" not in line and "ddp" not in line and "backward" not in line
    else:
        assert "another rank failed" in line["train_step"]["error"]
        # measure_ddp builds the same step: set-up fails again on rank 1 (skipped everywhere); a failure in the 7th launch
        # of the step surfaces inside ddp's own set-up phase or timed phase -- either way nobody hangs
        assert "ddp" in line
    assert line["matcher"]["ran_on"] == 0


def test_main_eight_ranks_all_legs_and_a_failure_on_the_last_rank():
    """the first real 8-GPU run must not be the first time rank 7 executes.  bench.main() over EIGHT gloo ranks
    (one node's worth), every leg incl. `train_step` and the bucketed `ddp` all-reduce, rank-dependent seeds, derived sites; then
    again with rank 7 raising inside the timed train steps -- every rank abandons the leg at the next exchange, the line is
    printed, nobody hangs."""
    line = _run_main(-1, "none", world=8)
    assert line["n_gpus"] == 8 and line["rccl_ranks"] == 8 and line["scaling"] == "weak"
    assert line["train_step"]["ms_per_step"] > 0
    assert line["ddp"]["rccl_ranks"] == 8 and line["busbw"] == line["ddp"]["busbw_GBs"] > 0
    assert line["matcher"]["ran_on"] == 0
    line = _run_main(7, "train_timed", world=8)
    assert line["value"] > 0 and line["n_gpus"] == 8
    assert "another rank failed" in line["train_step"]["error"] and "ddp" in line
