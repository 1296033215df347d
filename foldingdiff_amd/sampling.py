"""
Sampling from the diffusion model on MI355X -- same entry points as
foldingdiff/sampling.py (``p_sample`` :27-75, ``p_sample_loop`` :78-132,
``sample`` :135-224, ``sample_simple`` :227-264), same arguments, same return
types.  The 1000-step loop itself is one C-ABI call (``fd_sample`` /
``fd_sample_dev``): a hipGraph of hand-written kernels replayed per timestep
with the schedule tables and the step counter resident on the device.

Per-step noise (the reference's ``torch.randn_like(x)``, sampling.py:73):
  * ``NOISE_MODE = "torch"`` (default): the draws are taken from torch's global
    CPU generator in the reference's order, so after ``torch.manual_seed(s)`` the
    sampled angles match the reference CPU path.  They are STREAMED: a host thread
    draws ``NOISE_CHUNK`` steps at a time into pinned buffers, a copy stream uploads
    a chunk while the device consumes the previous one (``_StepNoise``,
    ``_run_fd_sample_streamed``) -- the [T, B, L, F] array (1.57 GB at batch 512,
    length 128) is never materialised and the first launch does not wait for it.
  * ``NOISE_MODE = "philox"`` (or env ``FOLDINGDIFF_AMD_NOISE=philox``): noise is
    generated inside the update kernel (Philox4x32-10); a 64-bit seed is drawn
    from torch's CPU generator, so ``torch.manual_seed`` still makes runs
    reproducible, but the stream is not torch's.
"""
import ctypes as C
import json
import logging
import os
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _binding, beta_schedules, utils
from . import datasets as dsets
from . import modelling

NOISE_MODE = os.environ.get("FOLDINGDIFF_AMD_NOISE", "torch")
NOISE_CHUNK = int(os.environ.get("FOLDINGDIFF_AMD_NOISE_CHUNK", "32"))   # reverse steps per uploaded chunk
STREAM_NOISE = os.environ.get("FOLDINGDIFF_AMD_STREAM_NOISE", "1") != "0"


def _as_f32(t: torch.Tensor) -> np.ndarray:
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32, copy=False))


def _lens_array(seq_lens: Sequence[int], B: int, L: int) -> np.ndarray:
    lens = np.ascontiguousarray(np.asarray(seq_lens, dtype=np.int32).reshape(-1))  # a (B, 1) tensor is fine, as in the reference
    assert lens.shape == (B,), f"need one length per batch item, got {lens.shape} for batch {B}"
    if lens.min() < 1 or lens.max() > L:
        raise ValueError(f"sequence lengths must lie in [1, {L}], got [{lens.min()}, {lens.max()}]")
    return lens


@torch.no_grad()
def p_sample(model, x: torch.Tensor, t: torch.Tensor, seq_lens: Sequence[int], t_index, betas: torch.Tensor) -> torch.Tensor:
    """One ancestral step x_t -> x_{t-1} WITHOUT the angular wrap (p_sample_loop wraps).
    As in the reference, ``t`` must be constant over the batch and ``t_index`` is
    ignored in favour of it (sampling.py:46-48)."""
    t_unique = torch.unique(t)
    assert len(t_unique) == 1, f"Got multiple values for t: {t_unique}"
    ti = int(t_unique.item())
    h = model.prepare(betas)
    xs = _as_f32(x)
    B, L, F = xs.shape
    assert F == model.n_inputs, f"{F} features, the model takes {model.n_inputs}"
    lens = _lens_array(seq_lens, B, L)
    z = _as_f32(torch.randn_like(x)) if ti > 0 else None
    out = np.empty_like(xs)
    _binding.check(_binding.load().fd_p_sample_step(
        h, xs.ctypes.data_as(C.c_void_p), ti, lens.ctypes.data_as(C.c_void_p), B, L,
        z.ctypes.data_as(C.c_void_p) if z is not None else None, 0, out.ctypes.data_as(C.c_void_p)))
    return torch.from_numpy(out).to(x.device)


def _draw_philox_seed() -> int:
    """64-bit seed of the on-device generator, drawn from torch's CPU generator (torch.manual_seed reproducible)."""
    return int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())


def _draw_step_noise(T: int, shape) -> np.ndarray:
    """The reference's per-step draws as one [T, B, L, F] array: row i is the
    ``torch.randn_like(x)`` taken at t_index i; drawn for i = T-1 ... 1 in that order,
    one call per step (a single big randn call is NOT the same stream)."""
    noise = np.zeros((T,) + tuple(shape), dtype=np.float32)
    for i in reversed(range(1, T)):
        noise[i] = torch.randn(tuple(shape), dtype=torch.float32).numpy()
    return noise


def _run_fd_sample(h, x0: np.ndarray, lens: np.ndarray, t_start: int, zs: Optional[np.ndarray], seed: int,
                   seq_offset: int, out: np.ndarray, full_history: int) -> None:
    """The one call into libfdmi.so that runs reverse steps t_start .. 0 on a host batch (fd_sample_ex).  Tests of
    the multi-process path replace exactly this function with a CPU stand-in."""
    B, L, _ = x0.shape
    _binding.check(_binding.load().fd_sample_ex(
        h, x0.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), B, L, t_start,
        zs.ctypes.data_as(C.c_void_p) if zs is not None else None, C.c_uint64(seed), C.c_int64(seq_offset),
        out.ctypes.data_as(C.c_void_p), full_history))


class _StepNoise:
    """The reference's per-step draws for reverse steps t = t_start .. 1, produced in the reference's order (one
    ``torch.randn`` of the WHOLE batch shape per step, from torch's global CPU generator) but handed out chunk by
    chunk.  ``rows=(lo, hi)``: keep only that slice of the batch (a rank's shard; the other rows are drawn -- the
    stream must advance exactly as in a single-process run -- and dropped at once, so memory stays bounded)."""

    def __init__(self, t_start: int, shape, rows: Optional[Tuple[int, int]] = None):
        self.t_start, self.shape = t_start, tuple(shape)
        self.rows = rows if rows is not None else (0, self.shape[0])
        self.local_shape = (self.rows[1] - self.rows[0],) + self.shape[1:]

    def chunks(self, n: int):
        """(t_lo, t_hi) of the chunks, first chunk first: steps run from t_start down to 0; a chunk holds the rows
        t_lo .. t_hi (row i = step t_lo + i).  The step t = 0 draws nothing: its row is zero."""
        t_hi = self.t_start
        while t_hi >= 0:
            t_lo = max(t_hi - n + 1, 0)
            yield t_lo, t_hi
            t_hi = t_lo - 1

    def fill(self, t_lo: int, t_hi: int, dst: torch.Tensor) -> None:
        """Draw the steps t_hi .. max(t_lo, 1) (in that order) into dst[t - t_lo]; dst: [>= t_hi - t_lo + 1, b, L, F]."""
        lo, hi = self.rows
        whole = (lo, hi) == (0, self.shape[0])
        for t in range(t_hi, t_lo - 1, -1):
            if t == 0:
                dst[0].zero_()
            elif whole:
                torch.randn(self.shape, dtype=torch.float32, out=dst[t - t_lo])
            else:
                dst[t - t_lo].copy_(torch.randn(self.shape, dtype=torch.float32)[lo:hi])

    def materialize(self) -> np.ndarray:
        """All rows at once, [t_start + 1, b, L, F] (row 0 zero): the array fd_sample_ex takes."""
        out = torch.zeros((self.t_start + 1,) + self.local_shape, dtype=torch.float32)
        self.fill(0, self.t_start, out)
        return out.numpy()


def _run_fd_sample_streamed(model, h, x0: np.ndarray, lens: np.ndarray, t_start: int, noise: _StepNoise, seq_offset: int,
                            out: Optional[np.ndarray], full_history: int, rows: Optional[int] = None) -> torch.Tensor:
    """Reverse steps t_start .. 0 with the step noise streamed to the device (fd_sample_begin_dev / _steps_dev /
    _end_dev): chunk k + 1 is drawn by a host thread and uploaded on a copy stream while the device runs chunk k."""
    import threading
    dev = model.device
    lib = _binding.load()
    B, L, F = x0.shape
    nrow = min(NOISE_CHUNK, t_start + 1)
    key = (nrow, B, L, F, str(dev))
    if key not in _NOISE_BUFFERS:  # (page-locking 2 x 50 MB costs more than a chunk's upload: keep the buffers of the last few shapes)
        while len(_NOISE_BUFFERS) >= 4:
            _NOISE_BUFFERS.pop(next(iter(_NOISE_BUFFERS)))
        _NOISE_BUFFERS[key] = ([torch.empty((nrow, B, L, F), dtype=torch.float32).pin_memory() for _ in range(2)],
                               [torch.empty((nrow, B, L, F), dtype=torch.float32, device=dev) for _ in range(2)],
                               torch.cuda.Stream(dev), torch.cuda.Stream(dev))
    pinned, dbuf, copy_s, run_s = _NOISE_BUFFERS[key] = _NOISE_BUFFERS.pop(key)  # (re-inserted: the dict's order is the order of last use)
    copied = [torch.cuda.Event(), torch.cuda.Event()]
    consumed = [None, None]
    x_d = torch.from_numpy(x0).to(dev)
    lens_d = torch.from_numpy(lens).to(dev)
    rows = out.shape[0] if out is not None else rows
    out_d = torch.empty((rows, B, L, F), dtype=torch.float32, device=dev)
    torch.cuda.current_stream(dev).synchronize()
    chunks = list(noise.chunks(nrow))
    err: List[BaseException] = []

    def draw(k):
        try:
            noise.fill(chunks[k][0], chunks[k][1], pinned[k % 2])
        except BaseException as e:  # surfaced by the main thread
            err.append(e)

    th = threading.Thread(target=draw, args=(0,))
    th.start()
    try:
        _binding.check(lib.fd_sample_begin_dev(h, C.c_void_p(x_d.data_ptr()), C.c_void_p(lens_d.data_ptr()), B, L, t_start,
                                               C.c_uint64(0), C.c_int64(seq_offset), C.c_void_p(out_d.data_ptr()), full_history,
                                               C.c_void_p(run_s.cuda_stream)))
        for k, (t_lo, t_hi) in enumerate(chunks):
            th.join()
            if err:
                raise err[0]
            b = k % 2
            with torch.cuda.stream(copy_s):
                if consumed[b] is not None:
                    copy_s.wait_event(consumed[b])          # the device is done with this buffer's previous content
                dbuf[b][: t_hi - t_lo + 1].copy_(pinned[b][: t_hi - t_lo + 1], non_blocking=True)
                copied[b].record(copy_s)
            if k + 1 < len(chunks):
                if k >= 1:
                    copied[(k + 1) % 2].synchronize()       # its upload has left the pinned buffer the thread writes next
                th = threading.Thread(target=draw, args=(k + 1,))
                th.start()
            run_s.wait_event(copied[b])
            _binding.check(lib.fd_sample_steps_dev(h, t_hi - t_lo + 1, C.c_void_p(dbuf[b].data_ptr()), t_lo,
                                                   C.c_void_p(run_s.cuda_stream)))
            consumed[b] = torch.cuda.Event()
            consumed[b].record(run_s)
        _binding.check(lib.fd_sample_end_dev(h, C.c_void_p(out_d.data_ptr()), C.c_void_p(run_s.cuda_stream)))
        run_s.synchronize()
        _binding.check(lib.fd_check_finite(h))
    except BaseException:
        # a failed call must not leave the draw thread consuming torch's global generator into a cached pinned buffer, nor
        # x_d / lens_d / out_d go back to the allocator under streams that still use them; the buffers of this
        # shape are dropped, the next call starts from fresh ones
        _NOISE_BUFFERS.pop(key, None)
        raise
    finally:
        th.join()
        run_s.synchronize()
        copy_s.synchronize()
    if out is not None:
        out[:] = out_d.cpu().numpy()
    return out_d  # (``out`` None: the stored states stay on the device -- sample() post-processes them there)


def _run_fd_sample_device(model, h, x0: np.ndarray, lens: np.ndarray, t_start: int, zs: Optional[np.ndarray], seed: int,
                          seq_offset: int, rows: int, full_history: int) -> torch.Tensor:
    """fd_sample_dev on device copies of a host batch; the [rows, B, L, F] result stays on the device."""
    dev = model.device
    B, L, F = x0.shape
    x_d, lens_d = torch.from_numpy(x0).to(dev), torch.from_numpy(lens).to(dev)
    z_d = torch.from_numpy(zs).to(dev) if zs is not None else None
    out_d = torch.empty((rows, B, L, F), dtype=torch.float32, device=dev)
    torch.cuda.current_stream(dev).synchronize()
    lib = _binding.load()
    _binding.check(lib.fd_sample_dev(h, C.c_void_p(x_d.data_ptr()), C.c_void_p(lens_d.data_ptr()), B, L, t_start,
                                     C.c_void_p(z_d.data_ptr()) if z_d is not None else None, C.c_uint64(seed),
                                     C.c_int64(seq_offset), C.c_void_p(out_d.data_ptr()), full_history, None))
    _binding.check(lib.fd_synchronize(h))
    _binding.check(lib.fd_check_finite(h))
    return out_d


_run_fd_sample_default = _run_fd_sample
_NOISE_BUFFERS: dict = {}


@torch.no_grad()
def p_sample_loop(
    model,
    lengths: Sequence[int],
    noise: torch.Tensor,
    timesteps: int,
    betas: torch.Tensor,
    is_angle: Union[bool, List[bool]] = [False, True, True, True],
    disable_pbar: bool = False,
    final_only: bool = False,
    history_every: int = 1,
    step_noise: Optional[np.ndarray] = None,
    seed: Optional[int] = None,
    seq_offset: int = 0,
    draw_batch: Optional[Tuple[int, int, int]] = None,
    _device_out: bool = False,
) -> torch.Tensor:
    """Run the whole reverse process from ``noise``.  Returns a CPU tensor of shape
    (timesteps, batch_size, seq_len, n_ft) -- entry j is the state after step
    t = timesteps-1-j, last entry = the sample -- or (1, batch, seq_len, n_ft) holding
    only the final sample when ``final_only`` (extension; skips the history copy), or, with
    ``history_every=k > 1`` (extension), (ceil(timesteps / k), ...): every k-th state
    (j = k-1, 2k-1, ...) with the sample in the last entry -- the history is thinned on the device."""
    assert history_every >= 1
    assert len(betas) == timesteps, f"{len(betas)} betas for {timesteps} timesteps"
    if not isinstance(is_angle, bool):
        assert len(is_angle) == noise.shape[-1]
    assert noise.shape[-1] == model.n_inputs, f"{noise.shape[-1]} features, the model takes {model.n_inputs}"
    h = model.prepare(betas, is_angle)
    x0 = _as_f32(noise)
    B, L, F = x0.shape
    lens = _lens_array(lengths, B, L)
    logging.info(f"Starting from noise {tuple(noise.shape)} with angularity {is_angle} using {model.device}")
    # per-step noise: given by the caller (a slice of a larger batch's draws), else drawn here
    if step_noise is not None:
        zs = np.ascontiguousarray(step_noise, dtype=np.float32)
        assert zs.shape == (timesteps, B, L, F)
        seed = 0
    elif seed is not None:
        zs = None
    elif NOISE_MODE == "torch":
        # the reference's draws, streamed; ``draw_batch = (B_all, lo, hi)``: this call is rows lo..hi of a larger batch
        # whose draws have the shape (B_all, L, F) (multi-rank sample())
        b_all, lo, hi = draw_batch if draw_batch is not None else (B, 0, B)
        assert hi - lo == B
        stream = _StepNoise(timesteps - 1, (b_all, L, F), (lo, hi))
        rows = 1 if final_only else -(-timesteps // history_every)
        on_gpu = _run_fd_sample is _run_fd_sample_default and getattr(getattr(model, "device", None), "type", "") == "cuda"
        if _device_out and on_gpu:  # (private: sample() keeps the stored states on the device for its post-processing)
            if STREAM_NOISE:
                return _run_fd_sample_streamed(model, h, x0, lens, timesteps - 1, stream, seq_offset, None,
                                               0 if final_only else history_every, rows=rows)
            return _run_fd_sample_device(model, h, x0, lens, timesteps - 1, stream.materialize(), 0, seq_offset, rows,
                                         0 if final_only else history_every)
        out = np.empty((rows, B, L, F), dtype=np.float32)
        if STREAM_NOISE and on_gpu:
            _run_fd_sample_streamed(model, h, x0, lens, timesteps - 1, stream, seq_offset, out, 0 if final_only else history_every)
        else:
            _run_fd_sample(h, x0, lens, timesteps - 1, stream.materialize(), 0, seq_offset, out, 0 if final_only else history_every)
        return torch.from_numpy(out)
    elif NOISE_MODE == "philox":
        zs, seed = None, _draw_philox_seed()
    else:
        raise ValueError(f"NOISE_MODE={NOISE_MODE!r}")
    rows = 1 if final_only else -(-timesteps // history_every)
    if _device_out and _run_fd_sample is _run_fd_sample_default and getattr(getattr(model, "device", None), "type", "") == "cuda":
        return _run_fd_sample_device(model, h, x0, lens, timesteps - 1, zs, seed, seq_offset, rows, 0 if final_only else history_every)
    out = np.empty((rows, B, L, F), dtype=np.float32)
    _run_fd_sample(h, x0, lens, timesteps - 1, zs, seed, seq_offset, out, 0 if final_only else history_every)
    return torch.from_numpy(out)


@torch.no_grad()
def sample_on_device(
    model,
    x_init: torch.Tensor,
    lens: torch.Tensor,
    betas: torch.Tensor,
    is_angle=None,
    seed: int = 0,
    seq_offset: int = 0,
    noise: Optional[torch.Tensor] = None,
    full_history: Union[bool, int] = False,
    t_start: Optional[int] = None,
    sync: bool = True,
) -> torch.Tensor:
    """Device-resident variant: ``x_init`` [B, L, F] float32 and ``lens`` [B] int32 are
    CUDA tensors, the result is a CUDA tensor ([B, L, F], or [t_start+1, B, L, F] with
    ``full_history``, or [ceil((t_start+1)/k), B, L, F] with ``full_history=k > 1``: every k-th state plus
    the final one); nothing touches the host.  Under a non-default torch stream the work is
    enqueued on that stream (asynchronous, stream-ordered).  Under torch's default (null)
    stream it runs on the model's own stream: pending torch work is waited for first and,
    unless ``sync=False``, the call returns after the sampler finished."""
    assert x_init.is_cuda and lens.is_cuda and x_init.dtype == torch.float32 and lens.dtype == torch.int32
    x_init, lens = x_init.contiguous(), lens.contiguous()
    h = model.prepare(betas, is_angle)
    B, L, F = x_init.shape
    assert F == model.n_inputs, f"{F} features, the model takes {model.n_inputs}"
    T = len(betas)
    t_start = T - 1 if t_start is None else t_start
    hist = int(full_history)   # 0 final only, 1 every state, k every k-th state
    assert hist >= 0
    shape = (-(-(t_start + 1) // hist), B, L, F) if hist else (B, L, F)
    out = torch.empty(shape, dtype=torch.float32, device=x_init.device)
    nptr = None
    if noise is not None:
        assert noise.is_cuda and noise.dtype == torch.float32 and tuple(noise.shape) == (t_start + 1, B, L, F)
        noise = noise.contiguous()
        nptr = C.c_void_p(noise.data_ptr())
    ts = torch.cuda.current_stream(x_init.device)
    own_stream = ts.cuda_stream == 0
    if own_stream:
        ts.synchronize()
    lib = _binding.load()
    _binding.check(lib.fd_sample_dev(
        h, C.c_void_p(x_init.data_ptr()), C.c_void_p(lens.data_ptr()), B, L, t_start, nptr, C.c_uint64(seed),
        C.c_int64(seq_offset), C.c_void_p(out.data_ptr()), hist,
        None if own_stream else C.c_void_p(ts.cuda_stream)))
    if own_stream and sync:
        _binding.check(lib.fd_synchronize(h))
        _binding.check(lib.fd_check_finite(h))  # FD_E_NONFINITE if the model output went inf / NaN at some step
    return out


def _dist_world():
    """(world size, rank) of the default process group, (1, 0) when torch.distributed is not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def sample(
    model,
    train_dset,
    n: int = 10,
    sweep_lengths: Optional[Tuple[int, int]] = (50, 128),
    batch_size: int = 512,
    feature_key: str = "angles",
    disable_pbar: bool = False,
    trim_to_length: bool = True,
    final_only: bool = False,
    history_every: int = 1,
    gather: str = "rank0",
) -> Optional[List[np.ndarray]]:
    """Sample ``n`` backbones per length in ``range(*sweep_lengths)`` (upper bound
    exclusive) -- or ``n`` backbones with lengths from ``train_dset.sample_length()``
    when ``sweep_lengths`` is None.  Returns one array per backbone of shape
    (timesteps, seq_len, fts); index -1 is the sample.  ``final_only=True`` (extension)
    returns arrays of shape (1, seq_len, fts) and never materialises the history;
    ``history_every=k`` (extension) keeps every k-th state plus the sample (``p_sample_loop``).

    ``train_dset`` needs ``sample_noise``, ``timesteps``, ``alpha_beta_terms``,
    ``feature_is_angular``, ``pad`` (and optionally ``sample_length``,
    ``dset.get_masked_means``), as the reference documents.

    Positions beyond an item's length are cut away below (as in the reference, sampling.py:201-203), so the device
    does not compute them at all (option "varlen": token rows = sum of lengths instead of batch x longest).

    Multi-GPU (extension; the reference is single device): when ``torch.distributed`` is initialised -- e.g.
    ``torchrun bin/sample.py`` with one process per GPU -- every rank draws the same start / step noise (same torch
    seed, as a replicated run of the reference would), runs the reverse process on a token-balanced slice of each
    batch only, cuts / shifts / re-wraps ITS OWN slice on its own device, and ONE collective per batch moves the
    already trimmed (ragged) blocks: ``gather="rank0"`` (default) to rank 0 only -- rank 0 returns the complete list,
    identical to a single-GPU run, every other rank returns ``None`` ("a single RCCL gather ... at the end");
    ``gather="all"`` to every rank (every rank returns the complete list: world x the bytes, e.g. 1.57 GB per 512
    sequences per rank with the full history); ``gather="none"``: no exchange, every rank returns its own items only
    (in batch order)."""
    from . import distributed as fdist

    if gather not in ("rank0", "all", "none"):
        raise ValueError(f"gather={gather!r}: expected 'rank0', 'all' or 'none'")
    if sweep_lengths is not None:
        lo, hi = sweep_lengths
        if not lo < hi:
            raise ValueError(f"Minimum length {lo} must be less than maximum {hi}")
        logging.info(f"Sweeping from {lo}-{hi} with {n} examples at each length")
        lengths = [l for l in range(lo, hi) for _ in range(n)]
    else:
        lengths = [train_dset.sample_length() for _ in range(n)]
    logging.info(f"Sampling {len(lengths)} items in batches of size {batch_size}")
    world, rank = _dist_world()
    T = train_dset.timesteps
    # the training mean offset (datasets.py get_masked_means; sampling.py:208-216), known before the first batch: every batch is
    # cut to its items' lengths, shifted and re-wrapped ON THE DEVICE (fd_shift_trim_dev) and comes back as one ragged copy
    inner = getattr(train_dset, "dset", None)
    offset = None
    if inner is not None and hasattr(inner, "get_masked_means"):
        try:
            offset = inner.get_masked_means()
        except NotImplementedError:
            # AnglesEmptyDataset without training_mean_offset.npy: the reference would
            # raise here (datasets.py:613-617); treat "no offset" as "no shift".
            offset = None
    if offset is not None:
        logging.info(f"Shifting predicted values by original offset: {offset}")
    angular = np.asarray(train_dset.feature_is_angular[feature_key], dtype=bool)
    on_gpu = _run_fd_sample is _run_fd_sample_default and getattr(getattr(model, "device", None), "type", "") == "cuda"
    # every rank finalises the model for THIS schedule and THIS angular mask before the first batch: the trim kernel reads
    # the mask of the finalised model, and a rank whose slice of a batch is empty never reaches p_sample_loop
    model.prepare(train_dset.alpha_beta_terms["betas"], train_dset.feature_is_angular[feature_key])
    rows = 1 if final_only else -(-T // history_every)
    F = model.n_inputs
    results: List[np.ndarray] = []
    for start in range(0, len(lengths), batch_size):
        these = lengths[start : start + batch_size]
        noise = train_dset.sample_noise(torch.zeros((len(these), train_dset.pad, model.n_inputs), dtype=torch.float32))
        if trim_to_length:
            noise = noise[:, : max(these), :]
        B, L, F = noise.shape
        # per-step noise: the reference's draws of the WHOLE batch (identical on every rank; each rank keeps its rows as
        # they are drawn, p_sample_loop(draw_batch=...)), or one Philox seed for the batch
        if NOISE_MODE == "torch":
            seed = None
            if world > 1 and start == 0:
                logging.info("torch-order step noise under torch.distributed: every rank draws the whole batch's stream "
                             "(serial host RNG); FOLDINGDIFF_AMD_NOISE=philox scales with the number of GPUs")
        elif NOISE_MODE == "philox":
            seed = _draw_philox_seed()
        else:
            raise ValueError(f"NOISE_MODE={NOISE_MODE!r}")
        bounds = fdist.shard_by_tokens(these, world) if world > 1 else [(0, B)]
        lo, hi = bounds[rank]
        home = model.device if on_gpu else torch.device("cpu")  # where a batch's stored states live until they are trimmed
        local_error = None
        flat = None
        if hi > lo:
            prev_varlen = model.set_option("varlen", 1)
            try:
                if on_gpu:   # the exact row count of this slice: the library's automatic kernel choices go by it (fd_set_option)
                    model.set_option("rows_hint", sum((int(l) + 7) // 8 * 8 for l in these[lo:hi]))
                traj = p_sample_loop(
                    model=model, lengths=these[lo:hi], noise=noise[lo:hi], timesteps=T,
                    betas=train_dset.alpha_beta_terms["betas"], is_angle=train_dset.feature_is_angular[feature_key],
                    disable_pbar=disable_pbar, final_only=final_only, history_every=history_every,
                    seed=seed, seq_offset=lo, draw_batch=(B, lo, hi), _device_out=on_gpu)
                # this rank's items, cut to their lengths, shifted and re-wrapped where they are (the device), as one ragged buffer
                flat = _shift_trim_flat(model, traj, these[lo:hi], offset, angular)
                del traj
            except Exception as e:  # e.g. FD_E_NONFINITE from this rank's slice: every rank must learn of it before the gather
                if world == 1:
                    raise
                local_error = e
            finally:
                if on_gpu:
                    model.set_option("rows_hint", 0)
                model.set_option("varlen", prev_varlen if prev_varlen is not None else 0)
        elif NOISE_MODE == "torch":  # an empty shard still advances the generator exactly as the other ranks do
            _StepNoise(T - 1, (B, L, F), (0, 0)).materialize()
        if flat is None:
            flat = torch.zeros((0,), dtype=torch.float32, device=home)   # an empty slice, or a failed one (reported just below)
        if world > 1:
            device = getattr(model, "device", torch.device("cpu"))
            failed = fdist.any_rank_failed(local_error is not None, device)   # one tiny all-reduce: all ranks abort together
            if failed:
                raise RuntimeError(f"rank {rank}: sampling failed on " + ("this rank: " + repr(local_error) if local_error else "another rank"))
            if gather != "none":
                # the single exchange of the path: the ragged, already trimmed blocks in rank order = the batch in item order
                # (device-resident blocks stay on the device through the collective: RCCL reads and writes HBM directly)
                sizes = [rows * sum(these[l_:h_]) * F for l_, h_ in bounds]
                flat = fdist.gather_ragged(flat, sizes, device, to_all=(gather == "all"))
                if flat is None:   # gather="rank0" on another rank
                    continue
                lo, hi = 0, B
        results.extend(_split_items(flat, these[lo:hi], rows, F, offset, angular))
    if world > 1 and gather == "rank0" and rank != 0:
        return None
    return results


def _shift_trim_flat(model, traj: torch.Tensor, lengths: Sequence[int], offset: Optional[np.ndarray], angular: np.ndarray) -> torch.Tensor:
    """The tail of the reference's ``sample`` (foldingdiff/sampling.py:200-222) for a block of items, as ONE ragged float32
    buffer on ``traj``'s device: item i = the first ``lengths[i]`` positions of every stored state ([rows, len_i, F],
    contiguous, items in order), plus the training mean offset, angular features re-wrapped to [-pi, pi).  ``traj``:
    [rows, B, L, F].  A CUDA tensor is processed by ``fd_shift_trim_dev`` (one launch).  Only a float32 offset is applied
    here, with the reference's float32 arithmetic (bit-identical); any other offset dtype is applied by ``_split_items`` on
    the host copy in numpy's promoted dtype -- exactly what ``s + offset`` gives in the reference."""
    rows, B, L, F = traj.shape
    assert B == len(lengths)
    sizes = np.array([rows * int(l) * F for l in lengths], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    off32 = offset is not None and np.asarray(offset).dtype == np.float32
    if traj.is_cuda:
        dev = traj.device
        h = model._ensure_handle()
        lens_d = torch.tensor([int(l) for l in lengths], dtype=torch.int32, device=dev)
        item_off = torch.from_numpy(offs[:-1].copy()).to(dev)
        out_d = torch.empty((int(offs[-1]),), dtype=torch.float32, device=dev)
        o32 = np.ascontiguousarray(offset, dtype=np.float32) if off32 else None
        torch.cuda.current_stream(dev).synchronize()
        lib = _binding.load()
        _binding.check(lib.fd_shift_trim_dev(h, C.c_void_p(traj.data_ptr()), rows, B, L, C.c_void_p(lens_d.data_ptr()),
                                             C.c_void_p(item_off.data_ptr()), o32.ctypes.data_as(C.c_void_p) if o32 is not None else None,
                                             C.c_void_p(out_d.data_ptr()), None))
        _binding.check(lib.fd_synchronize(h))
        return out_d
    # host tensors (the CPU stand-in of the multi-process tests): the same ragged buffer, assembled with numpy
    t = traj.numpy()
    flat = np.concatenate([t[:, i, : int(l), :].reshape(-1) for i, l in enumerate(lengths)]) if B else np.zeros((0,), np.float32)
    if off32:
        flat = flat.reshape(-1, F) + np.asarray(offset)
        flat[:, angular] = utils.modulo_with_wrapped_range(flat[:, angular], range_min=-np.pi, range_max=np.pi)
        flat = flat.reshape(-1)
    return torch.from_numpy(np.ascontiguousarray(flat))


def _split_items(flat, lengths: Sequence[int], rows: int, F: int, offset: Optional[np.ndarray], angular: np.ndarray) -> List[np.ndarray]:
    """One ragged device-to-host copy of ``_shift_trim_flat``'s buffer (a tensor, or a list of per-rank pieces in item
    order) and its items as views of the host buffer: [rows, len_i, F] each."""
    sizes = np.array([rows * int(l) * F for l in lengths], dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    if isinstance(flat, (list, tuple)):   # the pieces of a gather: each one copied straight to its place (no device concat)
        host = np.empty((int(offs[-1]),), dtype=np.float32)
        at = 0
        for piece in flat:
            host[at: at + piece.numel()] = piece.cpu().numpy()
            at += piece.numel()
        assert at == host.size
        flat = host
    else:
        flat = flat.cpu().numpy()
    assert flat.size == int(offs[-1])
    if offset is not None and np.asarray(offset).dtype != np.float32:  # e.g. a float64 offset file: numpy promotes, as in the reference
        flat = flat.reshape(-1, F) + np.asarray(offset)
        flat[:, angular] = utils.modulo_with_wrapped_range(flat[:, angular], range_min=-np.pi, range_max=np.pi)
        flat = flat.reshape(-1)
    return [flat[offs[i]: offs[i + 1]].reshape(rows, int(l), F) for i, l in enumerate(lengths)]


def _shift_trim(model, traj: torch.Tensor, lengths: Sequence[int], offset: Optional[np.ndarray], angular: np.ndarray) -> List[np.ndarray]:
    """``_shift_trim_flat`` + ``_split_items``: [rows, B, L, F] stored states -> one [rows, len_i, F] array per item."""
    rows, _, _, F = traj.shape
    return _split_items(_shift_trim_flat(model, traj, lengths, offset, angular), lengths, rows, F, offset, angular)


@torch.no_grad()
def reverse_from(model, x_t: torch.Tensor, lengths: Sequence[int], t_start: int, betas: torch.Tensor,
                 is_angle: Union[bool, List[bool]] = True) -> torch.Tensor:
    """Reverse steps t = t_start, t_start-1, ..., 0 from a partially noised ``x_t`` [B, L, F]; returns the
    final [B, L, F] on the CPU.  This is the loop of the reference's ``get_reconstruction_error``
    (foldingdiff/sampling.py:320-331): ``p_sample`` then ``modulo_with_wrapped_range(img)`` with its
    default bounds, i.e. EVERY feature is wrapped to [-pi, pi) (``is_angle=True``).  Per-step noise: the
    reference's draw order from the global CPU generator (``NOISE_MODE="torch"``) or Philox."""
    T = len(betas)
    assert 0 <= t_start < T, f"t_start={t_start} outside the {T}-step schedule"
    h = model.prepare(betas, is_angle)
    x0 = _as_f32(x_t)
    B, L, F = x0.shape
    assert F == model.n_inputs, f"{F} features, the model takes {model.n_inputs}"
    lens = _lens_array(lengths, B, L)
    if NOISE_MODE == "torch":
        zs, seed = _draw_step_noise(t_start + 1, (B, L, F)), 0
    elif NOISE_MODE == "philox":
        zs, seed = None, _draw_philox_seed()
    else:
        raise ValueError(f"NOISE_MODE={NOISE_MODE!r}")
    out = np.empty((1, B, L, F), dtype=np.float32)
    _run_fd_sample(h, x0, lens, t_start, zs, seed, 0, out, 0)
    return torch.from_numpy(out[0])


@torch.no_grad()
def reconstruct(model, dset, noise_timesteps: int = 250, bs: int = 512):
    """Device part of ``get_reconstruction_error`` (foldingdiff/sampling.py:287-341): forward-noise every
    item of ``dset`` (a ``NoisedAnglesDataset`` over real data) to ``t = noise_timesteps`` and denoise it
    again.  Returns (reconstructed, truth, filenames): per item an array [len_i, F] each."""
    recon, truth, files = [], [], []
    for start in range(0, len(dset), bs):
        idx_batch = list(range(start, min(start + bs, len(dset))))
        items = [dset.__getitem__(idx, use_t_val=noise_timesteps) for idx in idx_batch]
        img = torch.stack([it["corrupted"] for it in items]).clone()
        assert img.ndim == 3
        lengths = [int(torch.as_tensor(it["lengths"]).reshape(-1)[0]) for it in items]
        files.extend(dset.filenames[i] for i in idx_batch)
        # the reference iterates t = noise_timesteps-1 ... 0 whatever t the items were noised at
        final = reverse_from(model, img, lengths, noise_timesteps - 1, dset.alpha_beta_terms["betas"], is_angle=True)
        for i, l in enumerate(lengths):
            recon.append(final[i, :l].numpy())
            truth.append(items[i]["angles"][:l].cpu().numpy())
    return recon, truth, files


def get_reconstruction_error(model, dset, noise_timesteps: int = 250, bs: int = 512, scorer=None):
    """Reference signature + ``scorer``: TM-align (an external binary) and the PDB writer are outside this
    path, so the caller supplies ``scorer(reconst_angles, truth_angles, truth_pdb_file) -> (score,
    score_coord)`` -- the role of the reference's ``_score_angles`` (foldingdiff/sampling.py:266-284).
    Returns (scores, coord_scores) as the reference does."""
    if scorer is None:
        raise NotImplementedError("pass scorer=...: TM-score evaluation is not part of the MI355X hot path; "
                                  "use reconstruct() for the angle sets")
    recon, truth, files = reconstruct(model, dset, noise_timesteps=noise_timesteps, bs=bs)
    scores, coord_scores = zip(*(scorer(r, t, f) for r, t, f in zip(recon, truth, files)))
    return np.array(scores), np.array(coord_scores)


def sample_simple(model_dir: str, n: int = 10, sweep_lengths: Tuple[int, int] = (50, 128)):
    """Load model + dummy dataset from ``model_dir`` and return one DataFrame of final
    angles per sampled backbone."""
    import pandas as pd

    assert os.path.isdir(model_dir), f"{model_dir} is not a local model directory (no network here)"
    with open(os.path.join(model_dir, "training_args.json")) as fh:
        targs = json.load(fh)
    model = modelling.BertForDiffusionBase.from_dir(model_dir).to("cuda:0")
    dummy = dsets.AnglesEmptyDataset.from_dir(model_dir)
    noised = dsets.NoisedAnglesDataset(
        dset=dummy, dset_key="coords" if targs.get("angles_definitions") == "cart-coords" else "angles",
        timesteps=targs["timesteps"], exhaustive_t=False, beta_schedule=targs["variance_schedule"],
        nonangular_variance=1.0, angular_variance=targs["variance_scale"])
    sampled = sample(model, noised, n=n, sweep_lengths=sweep_lengths, disable_pbar=True)
    key = noised.dset_key
    return [pd.DataFrame(s[-1], columns=noised.feature_names[key]) for s in sampled]
