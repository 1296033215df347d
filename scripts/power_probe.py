#!/usr/bin/env python3
"""Is the reverse-diffusion loop at BASELINE C2 power-limited?  Samples the GPU's socket power and shader clock (hwmon / sysfs, a
reader thread at ~5 ms) while the sampling loop runs in steady state, per fuse_attn setting, and while the chip idles.
A matrix instruction that takes more cycles than its pass count says (16x16x32 f16: 16 nominal, 20.5-21.5 measured in a dense stream)
is what issue throttling under a power cap looks like; this probe puts the power and the clock next to that.   Env: B, L, STEPS."""
import glob
import os
import sys
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from foldingdiff_amd import beta_schedules, modelling, sampling  # noqa: E402


def find_sources():
    """hwmon files of the card this process computes on (the node has eight; the PCI bus id says which)."""
    out = {}
    pr = torch.cuda.get_device_properties(0)
    want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        if not os.path.exists(os.path.join(card, "pp_dpm_sclk")) or want not in os.path.realpath(card):
            continue
        out["card"] = os.path.realpath(card)
        for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp2_input", "power1_cap"):
                p = os.path.join(hw, name)
                if os.path.exists(p):
                    out[name] = p
        out["pp_dpm_sclk"] = os.path.join(card, "pp_dpm_sclk")
        break
    return out


SRC = find_sources()
print("sources:", {k: v for k, v in SRC.items()}, flush=True)


def read(name):
    try:
        with open(SRC[name]) as fh:
            return fh.read()
    except Exception:   # noqa: BLE001
        return None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.on = [], True

    def run(self):
        pw = "power1_average" if "power1_average" in SRC else "power1_input"
        while self.on:
            p, f = read(pw), read("freq1_input")
            self.rows.append((time.perf_counter(), int(p) / 1e6 if p and p.strip().isdigit() else float("nan"),
                              int(f) / 1e6 if f and f.strip().isdigit() else float("nan")))
            time.sleep(0.005)


def summarise(tag, rows):
    import numpy as np
    a = np.array([(p, f) for _, p, f in rows], dtype=float)
    if not len(a):
        print(tag, "no samples")
        return
    q = lambda v: " ".join(f"{x:7.0f}" for x in np.nanpercentile(v, [5, 50, 95]))   # noqa: E731
    print(f"{tag:34s} n={len(a):5d}  power W (p5 p50 p95): {q(a[:, 0])}   sclk MHz (p5 p50 p95): {q(a[:, 1])}", flush=True)


RELEASED = dict(hidden_size=384, num_attention_heads=12, intermediate_size=768, num_hidden_layers=12,
                max_position_embeddings=128, position_embedding_type="relative_key")
B, L, steps = int(os.environ.get("B", 512)), int(os.environ.get("L", 128)), int(os.environ.get("STEPS", 400))
torch.manual_seed(0)
model = modelling.BertForDiffusionBase(modelling.BertConfig(**RELEASED), [True] * 6).to("cuda:0")
betas = beta_schedules.cosine_beta_schedule(1000)
model.prepare(betas)
x = torch.randn(B, L, 6, device="cuda:0")
lens = torch.full((B,), L, dtype=torch.int32, device="cuda:0")
print("power cap W:", (int(read("power1_cap")) / 1e6) if read("power1_cap") else None, " pp_dpm_sclk:", (read("pp_dpm_sclk") or "").replace("\n", " | "))
s = Sampler()
s.start()
time.sleep(1.0)
summarise("idle", s.rows)
for fa in (1, 2, 0, 1):
    model.set_option("fuse_attn", fa)
    sampling.sample_on_device(model, x, lens, betas, seed=1, t_start=20)
    torch.cuda.synchronize()
    n0 = len(s.rows)
    t0 = time.perf_counter()
    sampling.sample_on_device(model, x, lens, betas, seed=1, t_start=steps - 1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rows = s.rows[n0:]
    summarise(f"fuse_attn={fa}: {dt / steps * 1e3:.3f} ms/step", rows[len(rows) // 5:])
if os.environ.get("PREC_F32"):
    pass
s.on = False
