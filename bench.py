#!/usr/bin/env python3
"""
Benchmark of the reverse-diffusion hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher around it: re-executes itself under
                                                            torch.distributed.run, one process per GPU, 127.0.0.1 rendezvous)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): backbones/sec at L=128, T=1000, batch 512 per GPU
(config C2; C4 = the same per-GPU batch on 8 GPUs, i.e. weak scaling).

A "step" = ONE PASS of the hot path over one batch: the whole T=1000 reverse
process for 512 length-128 backbones per GPU (1000 replays of the per-timestep
hipGraph), state + history resident in HBM, Philox noise generated in the update
kernel, then (N > 1) the single RCCL gather of the final angles to rank 0.
Inputs (x_T, lengths, weights, tables) are resident in HBM before the timed
region starts.  Nothing is skipped or cached between passes.

The JSON line also carries
  roofline      the dominant kernel (the fused q|k|v projection + attention kernel, seq_attn16.hip; the q|k|v GEMM where that does not run):
                algorithmic FLOPs per launch /
                average launch duration measured live with hipEvents inside the timed
                region (every 100th timestep is launched eagerly with an event pair per
                kernel instead of replaying the graph), against the dense MFMA peak of the instruction
                actually used (fp16: 2500 TFLOP/s in the default f16x3 mode, fp32: 157.3 in --precision f32).
  cpu_baseline  the reference CPU path (oracle restatement, "port") timed on this box's
                host cores over a bounded sample of the same workload.
  extras.c5     one pass of BASELINE config C5 (L = 512, batch 128, max_position_embeddings = 512).
  extras.small_batch  step time at batch 1 / 8 / 32 (the few-rows GEMM path)
  extras.c3     BASELINE config C3 through sampling.sample (the reference's published setting), Philox and default noise.
  extras.host_entry  p_sample_loop with host buffers in / out at C2, both noise modes.
  extras.nerf   N1: the 780 sampled backbones of C3 through fd_nerf (angles -> coordinates).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

RELEASED = dict(hidden_size=384, num_attention_heads=12, intermediate_size=768, num_hidden_layers=12,
                max_position_embeddings=128, position_embedding_type="relative_key")  # config_jsons/cath_full_angles_cosine.json
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16/bf16 MFMA (2:1-sparsity figures are NOT used)
PEAK_HBM_GBS = 8000.0
HBM_ACHIEVABLE_GBS = 6300.0  # MI355X_MICROARCH.md: measured float4 copy
PRECISION_INFO = {
    "f32": dict(peak=PEAK_F32_MFMA_TFLOPS, dtype="f32",
                kernel="gemm_f32_kernel<2,2,2,2,32,EPI_BIAS> (QKV projection, M=B*L, N=1152, K=384; v_mfma_f32_32x32x2_f32)"),
    "f16x3": dict(peak=PEAK_F16_MFMA_TFLOPS, dtype="f32 (fp16 hi/lo split operands, 3x v_mfma_f32_{32x32x16,16x16x32}_f16 per product, fp32 accumulate)",
                  kernel="gi::gemm_img_kernel<EPI_IMG_QKV> = 128x384-tile LDS-DMA-staged split GEMM on fp16 hi|lo grouped row images (q|k|v "
                         "projection in one launch, M=B*L, N=1152, K=384; algorithmic FLOPs counted once, the 3 MFMAs per product are overhead "
                         "against the dense fp16 peak)"),
}


def flops_per_token(L, d=384, ff=768, layers=12, F=6):
    # N_layers*(8d^2 + 4*d*d_ff + 6*L*d) + 2*F*d + 2*d^2 + 2*d*F
    return layers * (8 * d * d + 4 * d * ff + 6 * L * d) + 2 * F * d + 2 * d * d + 2 * d * F


def cpu_baseline(B, L, T, shape, steps=10, check_batch=8):
    """Reference CPU path (oracle port of foldingdiff/sampling.py p_sample_loop + the restated BertForDiffusion) on the host
    cores of this box, SURVEY 8(d)'s protocol: K = `steps` consecutive reverse steps at the full batch, extrapolated to T
    steps (steps are homogeneous: same kernels, same shapes, t only indexes tables), with the torch thread count probed ON
    THE FULL BATCH (the probe steps are part of the K), plus one complete T-step run at batch `check_batch` as the
    linearity check of that extrapolation."""
    from oracle import ref_model, ref_sampling

    model = ref_model.synthetic_model(ref_model.OracleConfig(**shape), seed=0, perturb=False)
    betas = ref_sampling.beta_schedule("cosine", T)
    torch.manual_seed(0)

    def run(t_hi, nsteps, img, batch):
        lens = [L] * batch
        t0 = time.perf_counter()
        for i in reversed(range(t_hi - nsteps + 1, t_hi + 1)):
            img = ref_sampling.p_sample(model, img, torch.full((batch,), i, dtype=torch.long), lens, betas)
            img = ref_sampling.wrap(img, -torch.pi, torch.pi)
        return time.perf_counter() - t0, img

    ncpu = os.cpu_count() or 1
    cand = sorted({c for c in (16, 32, 64, 128, ncpu) if c <= ncpu})
    img = ref_sampling.initial_noise((B, L, 6), [True] * 6)
    t_next, probe, times = T - 1, {}, []
    for c in cand:  # one full-batch step per candidate thread count (all logical CPUs is rarely the fastest on an SMT host)
        torch.set_num_threads(c)
        dt, img = run(t_next, 1, img, B)
        probe[c] = dt
        t_next -= 1
        if dt > 1.5 * min(probe.values()):  # past the knee: larger counts only get slower (256 threads: 64 s per step here)
            break
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    rest = max(steps - len(probe), 1)
    dt, img = run(t_next, rest, img, B)
    per_step = dt / rest                                  # steps at the chosen thread count only
    # linearity check: a complete T-step run at a small batch against its own first `steps` steps x T / steps
    small = ref_sampling.initial_noise((check_batch, L, 6), [True] * 6)
    k_dt, small = run(T - 1, steps, small, check_batch)
    full_dt, _ = run(T - 1 - steps, T - steps, small, check_batch)
    full_dt += k_dt
    return {
        "value": B / (per_step * T),
        "unit": "backbones/s",
        "cores": cores,
        "kind": "port",
        "sample": f"K={steps} consecutive reverse steps from t={T - 1} at the full batch {B}, L={L} (of T={T}), torch fp32 eval-mode "
                  f"oracle: one step per candidate thread count of {cand} until past the knee ({', '.join(f'{c}: {v:.1f} s' for c, v in probe.items())}), "
                  f"the other {rest} at the best ({cores} threads): {per_step * 1e3:.0f} ms/step, extrapolated x{T}; "
                  f"{ncpu} logical CPUs",
        # the same host at the batch size it runs most efficiently (the full batch thrashes on 403 MB of scores per layer):
        # a complete T-step run, nothing extrapolated
        "best_batch": {"batch": check_batch, "value": check_batch / full_dt, "unit": "backbones/s", "seconds": full_dt,
                       "sample": f"complete T={T} run at batch {check_batch}, L={L}, {cores} threads"},
        "linearity_check": {"batch": check_batch, "full_T_seconds": full_dt, "first_K_steps_seconds": k_dt,
                            "extrapolated_seconds": k_dt * T / steps, "ratio_full_over_extrapolated": full_dt / (k_dt * T / steps),
                            "backbones_per_s": check_batch / full_dt},
    }


class PowerClockSampler:
    """Socket power and shader clock of THIS process's GPU (hwmon files of the card with the device's PCI bus id) sampled every ~10 ms
    while the timed region runs: the path is power-limited (profiles/r06_power_probe.log), so the clock belongs next to the time."""

    def __init__(self, device_index):
        import glob
        import threading
        self.rows, self.on, self.src = [], False, {}
        try:
            pr = torch.cuda.get_device_properties(device_index)
            want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
            for card in sorted(glob.glob("/sys/class/drm/card*/device")):
                if want not in os.path.realpath(card):
                    continue
                for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
                    for name in ("power1_average", "power1_input", "freq1_input", "power1_cap"):
                        if os.path.exists(os.path.join(hw, name)):
                            self.src[name] = os.path.join(hw, name)
                break
        except Exception:  # noqa: BLE001  (telemetry only)
            self.src = {}
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _read(self, name):
        try:
            with open(self.src[name]) as fh:
                return float(fh.read().strip())
        except Exception:  # noqa: BLE001
            return float("nan")

    def _run(self):
        pw = "power1_average" if "power1_average" in self.src else "power1_input"
        while self.on:
            self.rows.append((self._read(pw) / 1e6, self._read("freq1_input") / 1e6))
            time.sleep(0.01)

    def start(self):
        if "freq1_input" in self.src:
            self.on = True
            self.thread.start()

    def stop(self):
        if not self.on:
            return None
        self.on = False
        self.thread.join(timeout=1.0)
        a = np.array(self.rows[len(self.rows) // 5:], dtype=float)   # (the first fifth: clocks still settling)
        if not len(a):
            return None
        med = lambda v: float(np.nanmedian(v))   # noqa: E731
        return {"sclk_mhz_median": round(med(a[:, 1])), "socket_power_w_median": round(med(a[:, 0])),
                "power_cap_w": round(self._read("power1_cap") / 1e6) if "power1_cap" in self.src else None, "samples": int(len(a)),
                "note": "hwmon of this GPU sampled every 10 ms inside the timed region; the shader clock's ceiling is 2400 MHz"}


def measure_traffic(kernel_substr, args):
    """HBM bytes per launch of the dominant kernel, measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need separate
    passes: MI355X_MICROARCH.md, TCC counter slots) over a 3-timestep eager run of this script, counters of the launches whose kernel
    name contains `kernel_substr`; bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB (FETCH_SIZE reports half of wide streaming reads on gfx950,
    same guide).  Returns (bytes or None, provenance)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    means = {}
    tmp = tempfile.mkdtemp(prefix="fdmi_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--timesteps", "3", "--profile-every", "0", "--no-cpu-baseline",
                   "--no-exact-f32", "--no-c5-extra", "--no-user-paths", "--no-traffic"]
            if args.precision:
                cmd += ["--precision", args.precision]
            env = dict(os.environ, FDMI_NO_GRAPH="1", TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
            tot, n = 0.0, 0
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] == counter and kernel_substr in row["Kernel_Name"]:
                            tot += float(row["Counter_Value"])
                            n += 1
            if n == 0:
                return None, f"rocprofv3 --pmc {counter}: no launches of *{kernel_substr}* (rc {r.returncode}): {r.stderr[-200:]}"
            means[counter] = tot / n
        return (2.0 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024.0, (
            f"measured in this run: rocprofv3 --pmc passes over 3 eager timesteps, mean per launch FETCH_SIZE {means['FETCH_SIZE']:.0f} KiB "
            f"(doubled: gfx950 calibration), WRITE_SIZE {means['WRITE_SIZE']:.0f} KiB")
    except Exception as e:  # a profiler problem must not take the benchmark line with it
        return None, f"rocprofv3 passes failed: {e!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed passes (one pass = T reverse steps over the batch)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=["c2", "c5"],
                    help="BASELINE.json configuration: c2 = L 128, batch 512 (the headline metric); c5 = L 512, batch 128, "
                         "max_position_embeddings 512 (long-chain stress)")
    ap.add_argument("--batch", type=int, default=None, help="sequences per GPU (default: the configuration's)")
    ap.add_argument("--length", type=int, default=None)
    ap.add_argument("--no-c5-extra", action="store_true", help="skip the extra BASELINE C5 pass reported in extras (N=1, c2 only)")
    ap.add_argument("--no-user-paths", action="store_true",
                    help="skip extras.c3 (sampling.sample on the manuscript sweep, both noise modes) and extras.host_entry (N=1, c2 only)")
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--profile-every", type=int, default=100)
    ap.add_argument("--no-history", action="store_true", help="do not keep the [T,B,L,F] history in HBM")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic (N=1 only)")
    ap.add_argument("--fuse-ln", type=int, default=-1, help="-1 auto (fused with f16x3), 0 off, 1 on")
    ap.add_argument("--precision", default=None, choices=["f32", "f16x3"], help="GEMM arithmetic (default: library default)")
    ap.add_argument("--no-exact-f32", action="store_true",
                    help="skip the extra single pass in exact-fp32 MFMA mode that is reported next to the headline (N=1 only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # started bare (`python bench.py --gpus 8`): become the launcher -- one process per GPU under torch.distributed.run
            # on this node, rendezvous on 127.0.0.1, same arguments; rank 0 of that job prints the JSON line to this stdout
            import socket
            import subprocess
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            raise SystemExit(subprocess.call(cmd, env=env))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from foldingdiff_amd import _binding, beta_schedules, datasets, modelling, sampling
    from foldingdiff_amd import distributed as fdist

    cfg_B, cfg_L = (512, 128) if args.config == "c2" else (128, 512)
    B, L, T = args.batch or cfg_B, args.length or cfg_L, args.timesteps
    shape = dict(RELEASED, max_position_embeddings=max(128, L))
    torch.manual_seed(0)
    model = modelling.BertForDiffusionBase(modelling.BertConfig(**shape), [True] * 6).to(dev)  # HF init, seed 0
    betas = beta_schedules.cosine_beta_schedule(T)
    if args.precision:
        model.set_precision(args.precision)
    h = model.prepare(betas)
    model.set_option("fuse_ln", args.fuse_ln)
    if os.environ.get("FDMI_NO_GRAPH") == "1":  # e.g. under rocprofv3 --pmc
        model.set_option("use_graph", 0)
    lib = _binding.load()
    ds = datasets.NoisedAnglesDataset(datasets.AnglesEmptyDataset("canonical-full-angles", pad=max(128, L)), timesteps=T,
                                      beta_schedule="cosine")
    torch.manual_seed(7344 + rank)  # bin/sample.py:34-37 default seed
    x_init = ds.sample_noise(torch.zeros(B, max(128, L), 6))[:, :L].contiguous().to(dev)
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)
    counts = [B] * world
    seq_offset = rank * B
    side = torch.cuda.Stream(device=dev)

    def one_pass(seed):
        with torch.cuda.stream(side):
            out = sampling.sample_on_device(model, x_init, lens, betas, seed=seed, seq_offset=seq_offset,
                                            full_history=not args.no_history)
            final = out[-1] if not args.no_history else out
            if world > 1:
                return fdist.gather_final(final.contiguous(), counts)
            return final

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    _binding.check(lib.fd_profile_every(h, 0))
    for w in range(args.warmup):
        one_pass(1000 + w)
    sync_all()
    _binding.check(lib.fd_profile_reset(h))
    _binding.check(lib.fd_profile_every(h, args.profile_every))
    sampler = PowerClockSampler(local_rank if world > 1 else 0) if rank == 0 else None
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    last = None
    for k in range(args.steps):
        last = one_pass(k)
    sync_all()
    elapsed = time.perf_counter() - t0
    telemetry = sampler.stop() if sampler else None
    _binding.check(lib.fd_profile_every(h, 0))
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        assert last is not None and tuple(last.shape) == (B * world, L, 6)
        assert torch.isfinite(last).all() and float(last.abs().max()) <= 3.1415927 + 1e-5

    # per-kernel stats measured inside the timed region
    kernels = {}
    name_p, ms, n, fl, by = C.c_char_p(), C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    for i in range(lib.fd_profile_count(h)):
        _binding.check(lib.fd_profile_get(h, i, C.byref(name_p), C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        if n.value:
            avg_ms = ms.value / n.value
            kernels[name_p.value.decode()] = {
                "avg_ms": avg_ms, "launches": n.value, "tflops": fl.value / (avg_ms * 1e-3) / 1e12,
                "gbs": by.value / (avg_ms * 1e-3) / 1e9, "flops": fl.value, "bytes": by.value}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    n_backbones = B * world * args.steps
    value = n_backbones / elapsed
    flop_per_backbone = flops_per_token(L) * L * T
    # two-sided roofline of one timestep: algorithmic FLOPs against the MFMA peak of the instruction used (x3 MFMAs per
    # product in the split arithmetic) and algorithmic HBM bytes (every activation read / written once per kernel, 4 B per
    # element: the fp16 hi|lo images have the footprint of fp32) against the achievable HBM rate
    per_step_launches = {"embed_ln_time": 1, "gemm_head_dense1": 1, "head_update_wrap": 1, "step_advance": 1}
    hbm_bytes_step = sum(v["bytes"] * per_step_launches.get(k, RELEASED["num_hidden_layers"]) for k, v in kernels.items())
    mfma_mult = 3.0 if model.precision == "f16x3" else 1.0
    ms_step = elapsed / args.steps / T * 1e3
    # the dominant launch: the per-layer kernel with the most time per timestep (round 6: the layer's tail in one launch, ffn16.hip;
    # before it the fused q|k|v projection + attention kernel)
    per_layer = {k: v for k, v in kernels.items() if k not in per_step_launches}
    dom_name = max(per_layer, key=lambda k: per_layer[k]["avg_ms"]) if per_layer else "gemm_qkv"
    dom = kernels.get(dom_name)
    pinfo_key = model.precision  # (the exact-fp32 pass below switches the model)
    pinfo = PRECISION_INFO[pinfo_key]
    legacy_fused = os.environ.get("FDMI_FUSE_ATTN") == "2"
    dom_kernel = pinfo["kernel"]
    sa_desc = ("sa::seq_attn_kernel<12> (round 5: 32-row waves, one per SIMD, 3x v_mfma_f32_32x32x16_f16 per product)" if legacy_fused else
               "s16::seq_attn16_kernel<12> (q|k|v projection + relative_key attention of a whole sequence per workgroup, 16-row waves, "
               "two per SIMD, fp16 hi/lo split, 3x v_mfma_f32_16x16x32_f16 per product)")
    KDESC = {
        "attn_out_ffn_fused": ("ffn16_kernel", "ffn::ffn16_kernel<12, true> (the layer's tail in one launch: attention.output.dense + residual + "
                               "LayerNorm + intermediate.dense + GELU + output.dense + residual + LayerNorm of 128 rows per pass, 16-row waves, two per "
                               "SIMD, fp16 hi/lo split, 3x v_mfma_f32_16x16x32_f16 per product)"),
        "ffn_fused": ("ffn16_kernel", "ffn::ffn16_kernel<12, false> (intermediate.dense + GELU + output.dense + residual + LayerNorm of 128 rows "
                      "per pass, 16-row waves, 3x v_mfma_f32_16x16x32_f16 per product)"),
        "qkv_attention_fused": ("seq_attn", sa_desc),
    }
    sub = "gemm_img_kernel<5" if model.precision == "f16x3" else "gemm_f32_kernel"
    if dom_name in KDESC:
        sub, dom_kernel = KDESC[dom_name]
    traffic, traffic_note = None, "not measured (--no-traffic, N > 1 or another shape)"
    if world == 1 and not args.no_traffic and (B, L) == (512, 128) and dom:
        traffic, traffic_note = measure_traffic(sub, args)
    roofline = None
    if dom:
        roofline = {
            "kernel": dom_kernel,
            "bound": "mfma", "achieved": dom["tflops"], "peak": pinfo["peak"], "unit": "TFLOP/s",
            "frac": dom["tflops"] / pinfo["peak"], "traffic": traffic,
            "traffic_source": traffic_note,
            "avg_launch_ms": dom["avg_ms"], "launches_timed": dom["launches"], "flops_per_launch": dom["flops"],
            "algorithmic_bytes_per_launch": dom["bytes"],
            "profile_name": dom_name,
        }
    # the other fused kernel of a layer, priced the same way (no counter pass)
    roofline_others = [{"profile_name": k, "kernel": KDESC[k][1], "bound": "mfma", "achieved": v["tflops"], "peak": pinfo["peak"],
                        "unit": "TFLOP/s", "frac": v["tflops"] / pinfo["peak"], "avg_launch_ms": v["avg_ms"], "launches_timed": v["launches"]}
                       for k, v in per_layer.items() if k != dom_name and k in KDESC]
    result = {
        "metric": f"backbones/sec (L={L}, T={T}, bs={B})",
        "value": value,
        "unit": "backbones/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": pinfo["dtype"],
        "data": "synthetic",
        "config": {"workload": f"{args.config.upper()}: released foldingdiff_cath shape (d=384,H=12,d_ff=768,12 layers,relative_key"
                               f"{', max_position_embeddings=512' if L > 128 else ''}), "
                               f"L={L}, T={T}, batch {B}/GPU, synthetic HF-init weights, Philox noise, "
                               f"history {'off' if args.no_history else 'in HBM'}; timed through the device-resident entry fd_sample_dev "
                               f"(x_init and the result stay in HBM; extras.host_entry times fd_sample_ex with both PCIe copies)",
                   "global_batch": B * world, "seq_len": L, "timesteps": T, "parallelism": f"batch-shard x{world}",
                   "fuse_ln": (args.fuse_ln if args.fuse_ln >= 0 else int(model.precision == "f16x3")),
                   "gemm_precision": model.precision},
        "whole_step": {"algorithmic_tflops": value * flop_per_backbone / 1e12 / world,
                       "frac_of_mfma_peak": value * flop_per_backbone / 1e12 / world / pinfo["peak"],
                       "ms_per_timestep": ms_step,
                       "hbm_algorithmic_bytes": hbm_bytes_step,
                       "hbm_floor_ms": hbm_bytes_step / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3,
                       "mfma_floor_ms": flops_per_token(L) * L * B * mfma_mult / (pinfo["peak"] * 1e12) * 1e3,
                       "hbm_achieved_gbs": hbm_bytes_step / (ms_step * 1e-3) / 1e9,
                       "note": "floors: HBM at the achievable 6.3 TB/s (8 TB/s peak), MFMA at the dense peak of the instruction "
                               "used (3 MFMAs per product in f16x3; a dense MFMA stream on random operands clocks the chip at ~1.6 GHz, "
                               "profiles/r03_clock_probe.log, so the sustained matrix peak is ~2/3 of the 2.4 GHz figure).  Neither floor shows what "
                               "the probes measured (profiles/r04_ingest_probe.log): a GEMM tile takes loaded bytes / 35 B/clk + STORED bytes / "
                               "9.4 B/clk per CU, not overlapping; with the two fused kernels of a layer (seq_attn16.hip, ffn16.hip) an activation row "
                               "leaves the chip twice per layer (ctx, h) instead of seven times.  The loop runs at the 1400 W socket power cap with "
                               "the shader clock well below 2.4 GHz (`clocks`; profiles/r06_power_probe.log)"},
        "roofline": roofline,
        "roofline_others": roofline_others,
        "clocks": telemetry,
        "kernels": {k: {"avg_ms": round(v["avg_ms"], 5), "tflops": round(v["tflops"], 2), "gbs": round(v["gbs"], 1),
                        "launches": v["launches"]} for k, v in kernels.items()},
        # the HBM-bound row kernels north_star singles out: algorithmic bytes / measured launch time against the 8 TB/s peak
        "row_kernels_hbm": {k: {"avg_us": round(kernels[k]["avg_ms"] * 1e3, 1), "gbs": round(kernels[k]["gbs"], 1),
                                "frac_of_8TBs": round(kernels[k]["gbs"] / PEAK_HBM_GBS, 3)}
                            for k in ("embed_ln_time", "head_update_wrap") if k in kernels},
        "dist": {"backend": dist.get_backend() if world > 1 else None, "world_size": dist.get_world_size() if world > 1 else 1},
    }
    if world == 1 and model.precision != "f32" and not args.no_exact_f32 and L <= 128:
        # the same workload with every contraction on v_mfma_f32_32x32x2_f32 (bitwise-fp32 products), one pass
        model.set_precision("f32")
        model.prepare(betas)
        _binding.check(lib.fd_profile_every(model._handle, 0))
        one_pass(4242)
        sync_all()
        t1 = time.perf_counter()
        one_pass(4243)
        sync_all()
        dt = time.perf_counter() - t1
        result["exact_f32_mode"] = {"value": B / dt, "unit": "backbones/s", "ms_per_step": dt * 1e3,
                                    "frac_of_f32_mfma_peak": (B / dt) * flop_per_backbone / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                    "note": "FD_PREC_F32: all GEMM/attention products on v_mfma_f32_32x32x2_f32, 1 timed pass"}
    if world == 1 and args.config == "c2" and not args.no_c5_extra:
        # BASELINE config C5 (L = 512 long-chain stress, batch 128, max_position_embeddings = 512): one timed pass
        m5 = modelling.BertForDiffusionBase(modelling.BertConfig(**dict(RELEASED, max_position_embeddings=512)), [True] * 6).to(dev)
        m5.set_precision(pinfo_key)
        m5.prepare(betas)
        x5 = ds.sample_noise(torch.zeros(128, 512, 6)).contiguous().to(dev)
        l5 = torch.full((128,), 512, dtype=torch.int32, device=dev)
        with torch.cuda.stream(side):
            sampling.sample_on_device(m5, x5, l5, betas, seed=1, t_start=1)  # workspace + graph
        sync_all()
        t5 = time.perf_counter()
        with torch.cuda.stream(side):
            o5 = sampling.sample_on_device(m5, x5, l5, betas, seed=2)
        sync_all()
        dt5 = time.perf_counter() - t5
        assert torch.isfinite(o5).all()
        result["extras"] = {"c5": {"metric": f"backbones/sec (L=512, T={T}, bs=128)", "value": 128 / dt5, "unit": "backbones/s",
                                   "ms_per_timestep": dt5 / T * 1e3, "passes": 1,
                                   "algorithmic_tflops": 128 / dt5 * flops_per_token(512) * 512 * T / 1e12}}
        del m5
    if world == 1 and args.config == "c2" and not args.no_user_paths and (B, L) == (512, 128):
        # what users call.  (a) BASELINE C3, the reference's own published setting (README.md:100-103, bin/sample.py defaults):
        # sampling.sample(n=10, sweep_lengths=(50, 128), batch_size=512) -- 780 backbones in two chunks, packed rows --
        # with on-device Philox noise and in the DEFAULT mode (the reference's torch.randn order, streamed from the host).
        # Useful tokens only: padded positions are not work.  (b) the host entry the metric is defined on:
        # p_sample_loop with host x_init in / final angles out (fd_sample_ex, PCIe both ways) at C2, both noise modes.
        model.set_precision(pinfo_key)
        model.prepare(betas)
        _binding.check(lib.fd_profile_every(model._handle, 0))
        extras = result.setdefault("extras", {})
        useful = sum(l for l in range(50, 128) for _ in range(10))
        c3 = {"metric": "backbones/sec (lengths 50..127 x 10, T=1000, batch_size 512) through sampling.sample, final_only",
              "backbones": 780, "useful_tokens": useful, "padded_tokens": 512 * 101 + 268 * 127}
        for mode in ("philox", "torch"):
            sampling.NOISE_MODE = mode
            torch.manual_seed(7344)
            if mode == "philox":  # first call of these two (B, L) shapes: workspaces + graph capture, not timed
                sampling.sample(model, ds, n=10, sweep_lengths=(50, 128), batch_size=512, final_only=True)
                torch.manual_seed(7344)
            t3 = time.perf_counter()
            res3 = sampling.sample(model, ds, n=10, sweep_lengths=(50, 128), batch_size=512, final_only=True)
            dt3 = time.perf_counter() - t3
            assert len(res3) == 780 and all(np.isfinite(r).all() for r in res3)
            c3[mode] = {"value": 780 / dt3, "unit": "backbones/s", "seconds": dt3, "passes": 1,
                        "useful_tokens_per_s": useful / dt3}
        # where C3 loses against C2 per token: per-kernel launch times of its two chunks (packed rows, the sweep's real lengths),
        # six eager steps with a hipEvent pair per kernel
        lengths3 = [l for l in range(50, 128) for _ in range(10)]
        per_chunk = {}
        for name, these in (("chunk0", lengths3[:512]), ("chunk1", lengths3[512:])):
            Bc, Lc = len(these), max(these)
            model.set_option("varlen", 1)
            xc = torch.randn(Bc, Lc, 6, device=dev)
            lc = torch.tensor(these, dtype=torch.int32, device=dev)
            with torch.cuda.stream(side):
                sampling.sample_on_device(model, xc, lc, betas, seed=1, t_start=1)
                sync_all()
                _binding.check(lib.fd_profile_reset(model._handle))
                _binding.check(lib.fd_profile_every(model._handle, 1))
                sampling.sample_on_device(model, xc, lc, betas, seed=1, t_start=5)
                sync_all()
                _binding.check(lib.fd_profile_every(model._handle, 0))
            ks, tot = {}, 0.0
            for i in range(lib.fd_profile_count(model._handle)):
                _binding.check(lib.fd_profile_get(model._handle, i, C.byref(name_p), C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
                if n.value:
                    ks[name_p.value.decode()] = round(ms.value / n.value * 1e3, 1)
                    tot += ms.value / 6
            rows = sum((l + 7) // 8 * 8 for l in these)
            per_chunk[name] = {"B": Bc, "L": Lc, "useful_tokens": sum(these), "token_rows": rows, "row_panels": -(-rows // 128),
                               "kernel_us": ks, "ms_per_timestep": round(tot, 3), "useful_tokens_per_us": round(sum(these) / tot / 1e3, 2)}
        model.set_option("varlen", 0)
        _binding.check(lib.fd_profile_reset(model._handle))
        c3["per_chunk"] = per_chunk
        extras["c3"] = c3
        he = {"metric": f"backbones/sec (L={L}, T={T}, bs={B}) through p_sample_loop: host x_init in, final angles out (fd_sample_ex)"}
        x_host = x_init.cpu()
        for mode in ("philox", "torch"):
            sampling.NOISE_MODE = mode
            torch.manual_seed(7344)
            th = time.perf_counter()
            fin = sampling.p_sample_loop(model, [L] * B, x_host, T, betas, is_angle=[True] * 6, final_only=True)
            dth = time.perf_counter() - th
            assert tuple(fin.shape) == (1, B, L, 6) and torch.isfinite(fin).all()
            he[mode] = {"value": B / dth, "unit": "backbones/s", "seconds": dth, "passes": 1}
        sampling.NOISE_MODE = "philox"
        extras["host_entry"] = he
        # (c) few sequences at a time (interactive use): every launch is far below one tile per CU there, and the q | k | v, FFN-up
        # and head-dense1 projections run on the weight-stationary kernel (gemm_ws.hip, launches of <= 12,288 rows)
        sb = {"metric": f"ms per reverse step at L={L}, released architecture, on-device Philox noise, 200 steps timed", "by_batch": {}}
        for bs in (1, 8, 32):
            xs = x_init[:bs].contiguous()
            ls = torch.full((bs,), L, dtype=torch.int32, device=xs.device)
            sampling.sample_on_device(model, xs, ls, betas, seed=1, t_start=3)   # workspace + graph of this shape
            torch.cuda.synchronize()
            ts = time.perf_counter()
            sampling.sample_on_device(model, xs, ls, betas, seed=1, t_start=199)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - ts) * 1e3 / 200
            sb["by_batch"][str(bs)] = {"ms_per_step": round(ms, 4), "backbones_per_s_at_T1000": round(bs / ms, 2)}
        extras["small_batch"] = sb
        # (d) N1: the 780 backbones of C3 from angles to N / CA / C coordinates through fd_nerf (host arrays in and
        # out, one launch, one lane per chain, fp64) -- what bin/sample.py does right behind sampling.sample
        from foldingdiff_amd import nerf as fnerf
        names = ["phi", "psi", "omega", "tau", "CA:C:1N", "C:1N:1CA"]  # canonical-full-angles
        finals = [r[-1] for r in res3]
        fnerf.build_backbones(finals, names)  # (first call: code object load)
        tn = time.perf_counter()
        coords = fnerf.build_backbones(finals, names)
        dtn = time.perf_counter() - tn
        assert len(coords) == 780 and all(np.isfinite(c).all() for c in coords)
        extras["nerf"] = {"metric": "ms for the 780 backbones of C3 (lengths 50..127), angles -> N/CA/C coordinates, host arrays in / out (fd_nerf)",
                          "ms": round(dtn * 1e3, 3), "backbones_per_s": round(780 / dtn, 1)}
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(B, L, T, shape)
    print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
