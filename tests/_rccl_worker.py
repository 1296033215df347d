"""Worker of tests/test_distributed.py::test_two_gpu_rccl_* (one process per GPU, launched with torch.distributed.run):
sampling.sample over RCCL must return on rank 0 (gather="all": on EVERY rank) exactly what a single process returns, and the C-ABI collective
(fd_comm_init / fd_gather_dev) must gather device blocks in rank order."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from foldingdiff_amd import _binding, datasets, modelling, sampling  # noqa: E402
from oracle import ref_model  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    kw = dict(hidden_size=192, num_attention_heads=6, intermediate_size=384, num_hidden_layers=2, max_position_embeddings=128,
              position_embedding_type="relative_key")
    oracle = ref_model.synthetic_model(ref_model.OracleConfig(**kw), seed=0)
    pm = modelling.BertForDiffusionBase(modelling.BertConfig(**kw), [True] * 6)
    pm.load_state_dict(oracle.state_dict())
    pm.to(dev)
    ds = datasets.NoisedAnglesDataset(
        datasets.AnglesEmptyDataset("canonical-full-angles", pad=128, mean_offset=np.array([0.3, -1.2, 3.0, 0.0, 1.9, -2.5], np.float32)),
        timesteps=12, beta_schedule="cosine")
    sampling.NOISE_MODE = "philox"
    torch.manual_seed(11)
    got = sampling.sample(pm, ds, n=2, sweep_lengths=(60, 66), batch_size=7, disable_pbar=True)   # sharded: one gather to rank 0 per batch
    torch.manual_seed(11)
    got_all = sampling.sample(pm, ds, n=2, sweep_lengths=(60, 66), batch_size=7, disable_pbar=True, gather="all")
    # the same call as a single process would run it (no process group visible to sample())
    saved = sampling._dist_world
    sampling._dist_world = lambda: (1, 0)
    try:
        torch.manual_seed(11)
        want = sampling.sample(pm, ds, n=2, sweep_lengths=(60, 66), batch_size=7, disable_pbar=True)
    finally:
        sampling._dist_world = saved
    assert (got is None) == (rank != 0)          # default: the result lives on rank 0 only
    for res in ([got] if rank == 0 else []) + [got_all]:
        assert len(res) == len(want) == 12
        for a, b in zip(res, want):
            assert a.shape == b.shape and np.array_equal(a, b)
    # the C-ABI collective: every rank contributes a block of its own size pattern; rank order in the result
    lib = _binding.load()
    h = pm._ensure_handle()
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        _binding.check(lib.fd_comm_unique_id(C.c_void_p(uid.data_ptr())))
    uid_d = uid.to(dev)
    dist.broadcast(uid_d, 0)
    uid = uid_d.cpu()
    _binding.check(lib.fd_comm_init(h, rank, world, C.c_void_p(uid.data_ptr())))
    n = 1000
    loc = torch.full((n,), float(rank + 1), dtype=torch.float32, device=dev)
    out = torch.zeros((world * n,), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    _binding.check(lib.fd_gather_dev(h, C.c_void_p(loc.data_ptr()), n, C.c_void_p(out.data_ptr()), None))
    _binding.check(lib.fd_synchronize(h))
    for r in range(world):
        assert bool((out[r * n:(r + 1) * n] == float(r + 1)).all()), r
    _binding.check(lib.fd_comm_destroy(h))
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: RCCL sample + fd_gather_dev OK", flush=True)


if __name__ == "__main__":
    main()
