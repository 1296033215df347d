"""
Multi-GPU sampling: independent sequences are sharded across the GPUs of one
node, every rank runs the whole reverse process on its slice with NO per-step
communication, and a single collective returns the final angles to rank 0
(RCCL over xGMI when the process group is "nccl"; "gloo" in the CPU tests):
``gather_final`` for equal-shaped [b, L, F] blocks (bench.py), ``gather_ragged`` for the
already trimmed per-item buffers of ``sampling.sample`` (to rank 0, or to every rank on request).

The reference has no multi-GPU sampler (foldingdiff/sampling.py is single
device, :91); this is the MI355X-native extension SURVEY 8(e) specifies.
Sequences are independent (attention is within-sequence; the reference asserts
batch-order equivariance in tests/test_transformer.py:136-162), weights are
57.8 MB and simply replicated.

Why one flat gather and not a ring: the payload is B/world x L x F fp32 per rank
(1.57 MB at B/world = 512, L = 128) once per 1000 timesteps -- latency-bound on
any xGMI topology; ``all_gather_into_tensor`` posts one RCCL collective.
Philox noise is keyed by the GLOBAL sequence index (``seq_offset``), so the
samples do not depend on the world size.
"""
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of ``n_items`` for ``rank`` (first
    ``n_items % world`` ranks get one extra item)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_tokens(lengths: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous slices balanced by total tokens rather than by count (mixed-length
    sweeps such as BASELINE config C3): cut r is placed at the prefix sum nearest to
    r/world of the total.  Returns [lo, hi) per rank (possibly empty)."""
    cum = [0]
    for n in lengths:
        cum.append(cum[-1] + int(n))
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        best = min(range(cuts[-1], len(cum)), key=lambda i: (abs(cum[i] - target), i))
        cuts.append(best)
    cuts.append(len(lengths))
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def gather_final(local: torch.Tensor, counts: Sequence[int], group=None) -> Optional[torch.Tensor]:
    """ONE collective: every rank contributes its [b_r, L, F] block; rank 0 gets the
    concatenation in rank order ([sum b_r, L, F]), other ranks get None.  Ragged
    ``counts`` are padded to the max so a single all_gather_into_tensor suffices."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(counts) == world and local.shape[0] == counts[rank]
    bmax = max(counts)
    pad = local
    if local.shape[0] != bmax:
        pad = torch.zeros((bmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    out = torch.empty((world * bmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    if rank != 0:
        return None
    out = out.view((world, bmax) + tuple(local.shape[1:]))
    return torch.cat([out[r, : counts[r]] for r in range(world)], dim=0)


def all_gather_batches(local: torch.Tensor, counts: Sequence[int], device=None, group=None) -> torch.Tensor:
    """ONE collective: every rank contributes its [b_r, ...] block and EVERY rank gets the concatenation in rank
    order.  ``device``: where the collective runs -- the model's GPU under "nccl" (= RCCL over xGMI), ignored
    under "gloo".  The result comes back on ``local``'s device."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(counts) == world and local.shape[0] == counts[rank]
    home = local.device
    if dist.get_backend(group) == "nccl":
        local = local.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    elif local.is_cuda:   # gloo has no CUDA collectives here
        local = local.cpu()
    bmax = max(counts)
    pad = local
    if local.shape[0] != bmax:
        pad = torch.zeros((bmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    out = torch.empty((world * bmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    out = out.view((world, bmax) + tuple(local.shape[1:]))
    return torch.cat([out[r, : counts[r]] for r in range(world)], dim=0).to(home)


def gather_ragged(local: torch.Tensor, sizes: Sequence[int], device=None, to_all: bool = False, group=None) -> Optional[List[torch.Tensor]]:
    """ONE collective over flat float32 blocks of different sizes: rank r contributes ``sizes[r]`` elements (every rank knows
    every size: they follow from the lengths and the shard bounds).  ``to_all=False``: a gather to rank 0 -- rank 0 gets the
    list of the world's blocks in rank order (views of one receive buffer), every other rank ``None`` and receives nothing;
    ``to_all=True``: an all-gather, every rank gets the list.  Under "nccl" (= RCCL over xGMI) the blocks travel from HBM
    to HBM on ``device``; under "gloo" a CUDA block is taken to the host first (gloo has no CUDA collectives here)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(sizes) == world and local.dim() == 1 and local.numel() == sizes[rank], (local.shape, sizes, rank)
    if dist.get_backend(group) == "nccl":
        local = local.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    elif local.is_cuda:
        local = local.cpu()
    nmax = max(max(sizes), 1)
    pad = local
    if local.numel() != nmax:   # equal-sized slots: one collective instead of world point-to-point messages
        pad = torch.zeros((nmax,), dtype=local.dtype, device=local.device)
        pad[: local.numel()] = local
    pad = pad.contiguous()
    if to_all:
        out = torch.empty((world * nmax,), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, pad, group=group)
    else:
        out = torch.empty((world * nmax,), dtype=local.dtype, device=local.device) if rank == 0 else None
        dist.gather(pad, list(out.view(world, nmax).unbind(0)) if rank == 0 else None, dst=0, group=group)
        if rank != 0:
            return None
    return [out[r * nmax: r * nmax + sizes[r]] for r in range(world)]


def any_rank_failed(failed: bool, device=None, group=None) -> bool:
    """True on every rank iff some rank reports a failure (one scalar all-reduce).  A rank that raised in front of a
    collective would leave the others blocked in it until the backend's timeout."""
    on = device if (dist.get_backend(group) == "nccl" and device is not None) else torch.device("cpu")
    flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=on)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return bool(int(flag.item()))


def sample_sharded(
    run_local: Callable[[int, int], torch.Tensor],
    n_items: int,
    group=None,
) -> Optional[torch.Tensor]:
    """Shard ``n_items`` sequences over the ranks, call ``run_local(lo, hi)`` (which
    must return the final [hi-lo, L, F] tensor for global sequences lo..hi-1, e.g. via
    ``sampling.sample_on_device(..., seq_offset=lo)``), then gather once to rank 0."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = []
    for r in range(world):
        lo, hi = shard_bounds(n_items, world, r)
        counts.append(hi - lo)
    lo, hi = shard_bounds(n_items, world, rank)
    local = run_local(lo, hi)
    assert local.shape[0] == hi - lo
    return gather_final(local, counts, group=group)
