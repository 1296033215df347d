#!/usr/bin/env python3
"""Cycle anatomy of the 16-row fused projection + attention kernel (seq_attn16.hip) at BASELINE C2: s_memtime stamps of workgroup 0
per head iteration: projection stages | epilogue | K/V barrier | S^T | band | softmax | P V | ctx store."""
import ctypes as C
import os
import sys

os.environ["FDMI_STAMPS"] = "1"
os.environ.setdefault("FDMI_FUSE_ATTN", "1")
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
names = ["st0", "st1", "st2", "st3", "st4", "st5", "epi", "bar", "S^T", "band", "smax", "PV", "store"]
for w in (0, 4, 7):
    s = a[w]
    used = [i for i in range(32) if s[i, 0]]
    print(f"wave {w}: {len(used)} head iterations; ticks per piece: " + " ".join(f"{v:>5s}" for v in names) + " |   head")
    for i in used[:14]:
        r = s[i]
        d = [int(r[k + 1] - r[k]) if r[k + 1] and r[k] else 0 for k in range(13)]
        print(f"  it {i:2d}: " + " " * 22 + " ".join(f"{v:5d}" for v in d) + f" | {int(r[13] - r[0]):6d}")
    m = np.array([[int(s[i][k + 1] - s[i][k]) for k in range(13)] for i in used[1:11]])
    print("  mean:  " + " " * 22 + " ".join(f"{v:5.0f}" for v in m.mean(axis=0)) + f" | {m.sum(axis=1).mean():6.0f}")
