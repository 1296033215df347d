#!/usr/bin/env python3
"""Cycle anatomy of the fused feed-forward kernel (ffn16.hip) at BASELINE C2: s_memtime stamps of workgroup 0 per pass of 128 rows
(groups of 64 intermediate features, LayerNorm + stores, wait for the next rows) and inside one group (first dense | GELU | second)."""
import ctypes as C
import os
import sys

os.environ["FDMI_STAMPS"] = "1"
os.environ.setdefault("FDMI_FUSE_FFN", "2")
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
n0 = 5 * 8 * 64 * 6 + 4 * 64 * 8 + 4 * 64 * 16 + 16384
n = n0 + 8 * 16 * 16
buf = np.zeros(n, dtype=np.uint64)
_binding.check(lib.fd_debug_read(h, b"stamps", buf.ctypes.data_as(C.c_void_p), 2 * n))
a = buf[n0:].reshape(8, 16, 16).astype(np.int64)
for w in (0, 3, 4, 7):
    s = a[w]
    print(f"wave {w}:")
    for ps in range(15):
        r = s[ps]
        if not r[0]:
            continue
        groups = [int(r[i + 1] - r[i]) for i in range(12) if r[i + 1] and r[i]] if ps == 0 else []
        tail = f"  LayerNorm + stores {int(r[13] - (r[12] if ps == 0 else r[0]))}" + f"  wait {int(r[14] - r[13])}"
        nxt = int(s[ps + 1][0] - r[0]) if ps + 1 < 15 and s[ps + 1][0] else 0
        print(f"  pass {ps}: total {int(r[14] - r[0])} (to next top {nxt})  groups {groups}{tail if ps == 0 else f'  groups + LayerNorm + stores {int(r[13] - r[0])}  wait {int(r[14] - r[13])}'}")
    g = s[15]
    if g[0]:
        print(f"  iteration 1 of pass 0: next group's first dense with this group's GELU inside {int(g[2] - g[1])} | second dense {int(g[3] - g[2])}")
