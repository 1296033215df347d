#!/usr/bin/env python3
"""Cycle anatomy of the row-image kernels at BASELINE C2 shapes: run a few reverse steps with FDMI_STAMPS=1 and print,
for workgroup 0, the s_memtime deltas of each phase of the GEMM k-loop / the attention position loop."""
import ctypes as C
import os
import sys

os.environ["FDMI_STAMPS"] = "1"
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
if os.environ.get("SPLIT_QKV"):
    model.set_option("split_qkv", int(os.environ["SPLIT_QKV"]))
x = torch.randn(B, L, 6, device="cuda:0")
lens = torch.full((B,), L, dtype=torch.int32, device="cuda:0")
sampling.sample_on_device(model, x, lens, betas, seed=1, t_start=3)
torch.cuda.synchronize()
lib = _binding.load()
n = 5 * 8 * 64 * 6 + 4 * 64 * 8
buf = np.zeros(n, dtype=np.uint64)
_binding.check(lib.fd_debug_read(h, b"stamps", buf.ctypes.data_as(C.c_void_p), 2 * n))
g = buf[: 5 * 8 * 64 * 6].reshape(5, 8, 64, 6).astype(np.int64)
a = buf[5 * 8 * 64 * 6:].reshape(4, 64, 8).astype(np.int64)
os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(REPO, "gpurun_out", os.environ.get("STAMPS_OUT", "stamps_raw") + ".npz"), gemm=g, attn=a)
names = {0: "FFN-up/head GELU", 1: "attn-out / FFN-down LN (last launch = FFN-down K=768)", 2: "QK", 3: "V^T"}
for epi in (2, 3, 1, 0):
    print(f"== GEMM epilogue {epi}: {names[epi]}   [cycles of s_memtime]  wave 0 (group 0) / wave 4 (group 1)")
    for w in (0, 4):
        s = g[epi, w]
        used = np.nonzero(s[:, 0])[0]
        if len(used) == 0:
            continue
        print(f"  wave {w}: periods {len(used)}")
        print("   k-tile   groups 1-5  barrier  fetch  group 6  epilogue  |  k-tile total")
        for i in used[:int(os.environ.get("NPER", 44))]:
            r = s[i]
            tot = (s[i + 1, 0] - r[0]) if i + 1 < 64 and s[i + 1, 0] else 0
            print(f"   {i:3d}   {r[1]-r[0]:9d} {r[2]-r[1]:8d} {r[3]-r[2]:6d} {r[4]-r[3]:8d} {(r[5]-r[4]) if r[5] else 0:9d}  | {tot:8d}")
print("== attention: per position  [A]wait  barrier  S+band  [B]+issue  softmax+[C]wait  barrier  PV  | total")
for w in (0, 3):
    s = a[w]
    used = np.nonzero(s[:, 0])[0]
    print(f"  wave {w}: positions {len(used)}")
    for i in used[:24]:
        r = s[i]
        tot = (s[i + 1, 0] - r[0]) if i + 1 < 64 and s[i + 1, 0] else 0
        print(f"   {i:3d} {r[1]-r[0]:7d} {r[2]-r[1]:7d} {r[3]-r[2]:7d} {r[4]-r[3]:7d} {r[5]-r[4]:7d} {r[6]-r[5]:7d} {r[7]-r[6]:7d} | {tot:8d}")
