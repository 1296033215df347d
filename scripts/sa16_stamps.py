#!/usr/bin/env python3
"""Cycle anatomy of the 16-row fused projection + attention kernel (seq_attn16.hip) at BASELINE C2: s_memtime stamps of workgroup 0
per head iteration: projection stages | epilogue | K/V barrier | S^T | band | softmax | P V | ctx store."""
import ctypes as C
import os
import sys

os.environ["FDMI_STAMPS"] = "1"
os.environ.setdefault("FDMI_FUSE_ATTN", "1")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from foldingdiff_amd import _binding, beta_schedules, modelling, sampling  # noqa: E402

RELEASED = dict(hidden_size=384, num_attention_heads=12, intermediate_size=768, num_hidden_layers=12,
                max_position_embeddings=128, position_embedding_type="relative_key")
B, L, T = int(os.environ.get("B", 512)), int(os.environ.get("L", 128)), 1000
torch.manual_seed(0)
model = modelling.BertForDiffusionBase(modelling.BertConfig(**RELEASED), [True] * 6).to("cuda:0")
betas = beta_schedules.cosine_beta_schedule(T)
h = model.prepare(betas)
model.set_option("use_graph", 0)
x = torch.randn(B, L, 6, device="cuda:0")
lens = torch.full((B,), L, dtype=torch.int32, device="cuda:0")
sampling.sample_on_device(model, x, lens, betas, seed=1, t_start=2)
torch.cuda.synchronize()
lib = _binding.load()
n0 = 5 * 8 * 64 * 6 + 4 * 64 * 8
n = n0 + 4 * 64 * 16 + 16384
buf = np.zeros(n, dtype=np.uint64)
_binding.check(lib.fd_debug_read(h, b"stamps", buf.ctypes.data_as(C.c_void_p), 2 * n))
a = buf[n0:n0 + 8 * 32 * 16].reshape(8, 32, 16).astype(np.int64)
print("per item: late stages (all waves in step) | epilogue | K/V barrier | region first half | region second half | closing barrier")
print("          region: group 0 (waves 0-3) = attention then the next item's early stages, group 1 the other way round;")
print("          attention pieces: S^T | band | mask, max, exp, sum | P V + store")
for w in (0, 1, 4, 5):
    s = a[w]
    used = [i for i in range(32) if s[i, 0]]
    print(f"wave {w}: {len(used)} items; ticks:  late   epi  KVbar  reg-1  reg-2 close |   item     attention: S^T  band  smax  PV+st")
    rows = []
    for i in used[:30]:
        r = s[i]
        d = [int(r[k + 1] - r[k]) if r[k + 1] and r[k] else 0 for k in range(6)]
        nxt = int(s[i + 1][0] - r[0]) if i + 1 < 32 and s[i + 1][0] else 0
        t0 = r[3] if w < 4 else r[4]   # start of this wave's attention
        t1 = r[4] if w < 4 else r[5]
        at = [int(r[9] - t0), int(r[10] - r[9]), int(r[11] - r[10]), int(t1 - r[11])] if r[9] and r[10] and r[11] else [0, 0, 0, 0]
        rows.append(d + [nxt] + at)
        if i < 6 or i in (11, 12):
            print(f"  item {i:2d}:       " + " ".join(f"{v:5d}" for v in d) + f" | {nxt:6d}                 " + " ".join(f"{v:5d}" for v in at))
    m = np.array(rows[1:11])
    print("  mean 1-10:     " + " ".join(f"{v:5.0f}" for v in m[:, :6].mean(axis=0)) + f" | {m[:, 6].mean():6.0f}                 " + " ".join(f"{v:5.0f}" for v in m[:, 7:].mean(axis=0)))
