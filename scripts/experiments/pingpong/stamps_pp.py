#!/usr/bin/env python3
"""Position anatomy of the ping-pong GEMM (gemm_pp.hip, FDMI_GEMM_PP=63 FDMI_STAMPS=1) at BASELINE C2 shapes, workgroup 0."""
import ctypes as C
import os
import sys

os.environ["FDMI_STAMPS"] = "1"
os.environ.setdefault("FDMI_GEMM_PP", "63")
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
sampling.sample_on_device(model, x, lens, betas, seed=1, t_start=3)
torch.cuda.synchronize()
lib = _binding.load()
n = 5 * 8 * 64 * 6 + 4 * 64 * 8
buf = np.zeros(n, dtype=np.uint64)
_binding.check(lib.fd_debug_read(h, b"stamps", buf.ctypes.data_as(C.c_void_p), 2 * n))
g = buf[: 5 * 3072].reshape(5, 3072)[:, :3000].reshape(5, 10, 100, 3).astype(np.int64)
names = {0: "GELU (last launch: head, N=384)", 1: "LN (last launch: FFN-down K=768)", 2: "q|k|v (workgroup 0: q columns)"}
NP = int(os.environ.get("NPOS", 60))
for epi in (0, 1, 2):
    s = g[epi]
    print(f"== {names[epi]}: per position  [wave 0 (group 0): work | barrier wait]  [wave 4 (group 1): work | wait]  [loader 8: vmcnt wait | barrier wait]  | position time")
    t0 = s[8, 0, 2]
    for p in range(NP):
        if s[8, p, 2] == 0:
            break
        row = []
        for w in (0, 4):
            a, b, c = s[w, p]
            row.append(f"{(b - a) if a and b else 0:6d} {(c - b) if b and c else 0:6d}")
        a, b, c = s[8, p]
        row.append(f"{b - a:6d} {c - b:6d}")
        nxt = s[8, p + 1, 2] if p + 1 < 100 else 0
        print(f"  {p:3d}  " + "  |  ".join(row) + f"  | {(nxt - c) if nxt else 0:6d}")
