#!/usr/bin/env python3
"""Where does the fused kernel's ctx differ from the two-kernel path?  One layer, released shape; per (sequence, head, 32-row block)
max |difference| of the attention context, read back through fd_debug_read after 3 launches (embed, q|k|v (+ attention))."""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from foldingdiff_amd import _binding, beta_schedules, modelling  # noqa: E402
from oracle import ref_model, ref_sampling  # noqa: E402

B, L = int(os.environ.get("B", 3)), 128
lens = [128, 100, 128][:B] + [128] * max(0, B - 3)
d, H = 384, 12
kw = dict(hidden_size=d, num_attention_heads=H, intermediate_size=768, num_hidden_layers=1, max_position_embeddings=128,
          position_embedding_type="relative_key")
oracle = ref_model.synthetic_model(ref_model.OracleConfig(**kw), (True,) * 6, "gaussian_fourier", "mlp", seed=3)
pm = modelling.BertForDiffusionBase(modelling.BertConfig(**kw), [True] * 6)
pm.load_state_dict(oracle.state_dict())
pm.to("cuda:0")
h = pm.prepare(beta_schedules.cosine_beta_schedule(100))
lib = _binding.load()
g = torch.Generator().manual_seed(1)
x = ref_sampling.wrap(torch.randn(B, L, 6, generator=g) * 1.5)
mask = torch.zeros(B, L)
for i, n in enumerate(lens):
    mask[i, :n] = 1.0
t = torch.full((B,), 42, dtype=torch.long)
rows_cap = (B * L + 127) // 128 * 128
ctx = {}
for fa in (0, 1):
    pm.set_option("fuse_attn", fa)
    pm.set_option("debug_stop", 3)
    pm(x, t, attention_mask=mask)
    out = np.empty(rows_cap * d, dtype=np.float32)
    _binding.check(lib.fd_debug_read(h, b"ctx", out.ctypes.data_as(C.c_void_p), out.size))
    ctx[fa] = out.reshape(rows_cap, d)[: B * L].reshape(B, 4, 32, H, 32)  # [b, row block, row, head, d]
pm.set_option("debug_stop", 0)
dd = np.abs(ctx[0] - ctx[1])
print("max |ctx fused - ctx two-kernel| =", float(dd.max()), " max|ctx| =", float(np.abs(ctx[0]).max()))
for b in range(B):
    print(f"sequence {b} (len {lens[b]}): rows = heads, columns = 32-row blocks")
    for hh in range(H):
        print(f"   head {hh:2d}: " + " ".join(f"{dd[b, rb, :, hh].max():.2e}" for rb in range(4)))
bad = np.argwhere(dd > 1e-6)
if len(bad):
    b, rb, r, hh, e = bad[0]
    print("first differing element: sequence", b, "row", rb * 32 + r, "head", hh, "d", e, " two-kernel", ctx[0][b, rb, r, hh, e], " fused", ctx[1][b, rb, r, hh, e])
    print("  that row/head, two-kernel:", np.array2string(ctx[0][b, rb, r, hh, :8], precision=5))
    print("  that row/head, fused     :", np.array2string(ctx[1][b, rb, r, hh, :8], precision=5))
