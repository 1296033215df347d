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
