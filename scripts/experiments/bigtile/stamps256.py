#!/usr/bin/env python3
"""Cycle anatomy of gemm_img256.hip (256-row tiles) at BASELINE C2 shapes: FDMI_STAMPS=1, workgroup 0, per stage."""
import ctypes as C
import os
import sys

os.environ["FDMI_STAMPS"] = "1"
os.environ.setdefault("FDMI_GEMM_256", "17")
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
g = buf[: 5 * 8 * 64 * 6].reshape(5, 8, 64, 6).astype(np.int64)
names = {0: "GELU (last launch = head dense1)", 1: "LN (last launch = FFN-down)", 2: "QKV"}
for epi in [int(e) for e in os.environ.get("EPIS", "0").split(",")]:
    for w in (0, 3):
        s = g[epi, w]
        k = s[63]
        if k[3] > k[1]:
            print(f"== epilogue {epi} {names.get(epi)} wave {w}: kernel {k[2]-k[0]} cycles in {(k[3]-k[1]) * 10} ns = {(k[2]-k[0]) / ((k[3]-k[1]) * 10.0):.3f} GHz")
        used = [i for i in range(63) if s[i, 0]]
        print("   slot  pass1+2  wait+barrier  issue+reads  pass3  epilogue | stage total")
        for i in used[:int(os.environ.get("NPER", 50))]:
            r = s[i]
            tot = (s[i + 1, 0] - r[0]) if i + 1 < 63 and s[i + 1, 0] else 0
            print(f"   {i:3d} {r[1]-r[0]:8d} {r[2]-r[1]:12d} {r[3]-r[2]:11d} {r[4]-r[3]:7d} {(r[5]-r[4]) if r[5] else 0:9d} | {tot:8d}")
