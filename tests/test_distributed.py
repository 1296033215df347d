"""N > 1 path on CPU: 2 ranks over gloo exercise the shard bounds and the single gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from foldingdiff_amd import distributed as fdist


def test_shard_bounds_cover_and_balance():
    for n in (1, 7, 512, 513, 4096):
        for w in (1, 2, 4, 8):
            b = [fdist.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_shard_by_tokens_manuscript_sweep():
    lengths = [l for l in range(50, 128) for _ in range(10)]
    b = fdist.shard_by_tokens(lengths, 8)
    assert b[0][0] == 0 and b[-1][1] == len(lengths)
    toks = [sum(lengths[lo:hi]) for lo, hi in b]
    assert max(toks) / (sum(toks) / 8) < 1.03


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def run_local(lo, hi):  # stand-in sampler: value encodes the GLOBAL sequence index
            idx = torch.arange(lo, hi, dtype=torch.float32)
            return idx[:, None, None] * torch.ones(hi - lo, 5, 6) + 0.25
        out = fdist.sample_sharded(run_local, n_items)
        if rank == 0:
            q.put(out.numpy())
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 7])
def test_two_rank_gloo_gather(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got.shape == (n_items, 5, 6)
    for i in range(n_items):
        assert (got[i] == i + 0.25).all()


# ------------------------------------------------------------------ sampling.sample under torch.distributed
# The product function itself (sharding by tokens, slicing of the start / step noise, seq_offset, the ONE all-gather,
# trimming, identical result on every rank) with the device sampler replaced at the fd_sample_ex boundary only.
class _StubModel:
    n_inputs = 6
    device = torch.device("cpu")

    def __init__(self):
        self.opts = []

    def prepare(self, betas, is_angle=None):
        return None

    def set_option(self, name, value):
        self.opts.append((name, value))


def _stub_fd_sample(h, x0, lens, t_start, zs, seed, seq_offset, out, full_history):
    """CPU stand-in for libfdmi's fd_sample_ex: a deterministic function of exactly what the device sampler consumes
    (start noise, lengths, the step-noise slice or the Philox seed, the GLOBAL sequence index)."""
    import numpy as np
    B, L, _ = x0.shape
    val = 0.5 * x0
    if zs is not None:
        val = val + 0.01 * zs[1:].sum(axis=0)
    else:
        val = val + 1e-3 * (seed % 97)
    val = val + (np.arange(B) + seq_offset)[:, None, None]
    real = np.arange(L)[None, :, None] < lens[:, None, None]
    out[:] = np.where(real, val, np.nan)[None].astype(np.float32)   # padded positions must never reach a result


def _run_sample(mode, gather="rank0", sweep=(5, 14), n=3, final_only=True, offset=None):
    import numpy as np
    from foldingdiff_amd import datasets, sampling
    sampling._run_fd_sample = _stub_fd_sample
    sampling.NOISE_MODE = mode
    ds = datasets.NoisedAnglesDataset(
        datasets.AnglesEmptyDataset("canonical-full-angles", pad=32, mean_offset=None if offset is None else np.asarray(offset)),
        timesteps=4, beta_schedule="cosine")
    torch.manual_seed(11)
    model = _StubModel()
    res = sampling.sample(model, ds, n=n, sweep_lengths=sweep, batch_size=16, final_only=final_only, gather=gather)
    return res, model.opts


def _sample_worker(rank, world, port, mode, q, kw):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res, opts = _run_sample(mode, **kw)
        q.put((rank, None if res is None else [r.copy() for r in res], opts))
    finally:
        dist.destroy_process_group()


def _two_ranks(mode, **kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sample_worker, args=(r, 2, port, mode, q, kw)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        rank, res, opts = q.get(timeout=180)
        got[rank] = res
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


@pytest.mark.parametrize("mode", ["torch", "philox"])
def test_sample_is_sharded_and_world_size_invariant(mode):
    """Default exchange: ONE gather of the trimmed blocks to rank 0 -- rank 0 holds the complete result,
    identical to the single-process one; the other rank receives nothing and returns None."""
    import numpy as np
    want, opts1 = _run_sample(mode)                     # single process: torch.distributed not initialised
    assert len(want) == 27 and [w.shape for w in want] == [(1, 5 + i // 3, 6) for i in range(27)]
    assert all(np.isfinite(w).all() for w in want)
    assert opts1 == [("varlen", 1), ("varlen", 0)] * 2  # two batches (16 + 11), padded positions not computed
    got = _two_ranks(mode)
    assert got[1] is None
    assert len(got[0]) == 27
    for a, b in zip(got[0], want):
        assert a.shape == b.shape and np.array_equal(a, b)


def test_sample_gather_all_and_none_with_history_and_offsets():
    """gather="all": the complete result on every rank (the opt-in former behaviour); gather="none": every rank keeps its own
    items, together they are the single-process list.  With the stored history (rows > 1) and a float64 mean offset (applied on
    the host copy in numpy's promoted dtype, as the reference's ``s + offset`` does)."""
    import numpy as np
    off = np.array([0.3, -1.2, 3.0, 0.0, 1.9, -2.5], dtype=np.float64)
    kw = dict(final_only=False, offset=off)
    want, _ = _run_sample("philox", **kw)
    assert want[0].shape == (4, 5, 6) and want[0].dtype == np.float64
    got = _two_ranks("philox", gather="all", **kw)
    for rank in (0, 1):
        assert len(got[rank]) == 27
        for a, b in zip(got[rank], want):
            assert a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)
    got = _two_ranks("philox", gather="none", **kw)
    assert 0 < len(got[0]) < 27 and len(got[0]) + len(got[1]) == 27
    # batch 1 = items 0..15, batch 2 = items 16..26; each rank holds a contiguous slice of each
    n0 = [len(got[0]), len(got[1])]
    shapes = sorted(a.shape for r in (0, 1) for a in got[r])
    assert shapes == sorted(w.shape for w in want)
    pool = {w.tobytes() for w in want}
    assert all(a.tobytes() in pool for r in (0, 1) for a in got[r]), n0


@pytest.mark.parametrize("mode", ["torch", "philox"])
def test_sample_with_fewer_items_than_ranks(mode):
    """One item on two ranks: shard_by_tokens gives one rank an empty slice (cuts [0, 0, 1] or [0, 1, 1]); that rank never
    reaches the sampler, still takes part in the exchange, and the result is the single-process one."""
    import numpy as np
    kw = dict(sweep=(9, 10), n=1)
    want, _ = _run_sample(mode, **kw)
    assert len(want) == 1 and want[0].shape == (1, 9, 6)
    got = _two_ranks(mode, **kw)
    assert got[1] is None and len(got[0]) == 1 and np.array_equal(got[0][0], want[0])
    got = _two_ranks(mode, gather="all", **kw)
    assert all(len(got[r]) == 1 and np.array_equal(got[r][0], want[0]) for r in (0, 1))


def _failing_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from foldingdiff_amd import datasets, sampling

        def stub(h, x0, lens, t_start, zs, seed, seq_offset, out, full_history):
            if dist.get_rank() == 1:
                raise ValueError("non-finite prediction on this rank's slice")
            _stub_fd_sample(h, x0, lens, t_start, zs, seed, seq_offset, out, full_history)

        sampling._run_fd_sample = stub
        sampling.NOISE_MODE = "philox"
        ds = datasets.NoisedAnglesDataset(datasets.AnglesEmptyDataset("canonical-full-angles", pad=32), timesteps=4,
                                          beta_schedule="cosine")
        torch.manual_seed(11)
        try:
            sampling.sample(_StubModel(), ds, n=3, sweep_lengths=(5, 9), batch_size=16, final_only=True)
            q.put((rank, "returned"))
        except RuntimeError as e:
            q.put((rank, str(e)))
    finally:
        dist.destroy_process_group()


def test_a_failure_on_one_rank_aborts_every_rank():
    """A rank whose slice fails must not leave the others blocked in the gather: the error flag is all-reduced first and
    every rank raises."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert "this rank" in got[1] and "non-finite" in got[1]
    assert "another rank" in got[0]


@pytest.mark.gpu
def test_two_gpu_rccl_sample_and_c_abi_gather():
    """The N > 1 path on REAL devices: two processes, one per GPU, backend nccl = RCCL over xGMI.  Skipped on a
    one-GPU box; the first multi-GPU lease runs sampling.sample's all-gather, fd_comm_init / fd_gather_dev and (below) bench.py
    --gpus 2 without any new code."""
    import subprocess
    import sys
    from foldingdiff_amd import _binding
    if _binding.load().fd_device_count() < 2:
        pytest.skip("needs two GPUs")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(repo, "tests", "_rccl_worker.py")]
    r = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("RCCL sample + fd_gather_dev OK") == 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--timesteps", "20", "--batch", "64"]
    r = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    import json
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["dist"]["backend"] == "nccl" and line["config"]["global_batch"] == 128


def test_bench_launches_itself_for_n_gpus():
    """`python bench.py --gpus 2` with no launcher around it re-executes itself under torch.distributed.run with one process per
    GPU (the first multi-GPU lease must produce a curve without new code).  Without a GPU every rank stops at the
    device check -- reaching it inside a worker of the elastic agent is the proof that the launcher ran."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("CPU-side check of the launcher (on a GPU box the -m gpu RCCL test runs the real thing)")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=repo, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs torch.distributed.run" not in r.stderr
    # (the elastic agent ends the other rank as soon as one has failed: at least one rank reports, and the agent's summary follows)
    assert r.stderr.count("bench.py needs an MI355X") >= 1 and "local_rank" in r.stderr, r.stderr[-2000:]
