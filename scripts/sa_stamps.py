#!/usr/bin/env python3
"""Cycle anatomy of the fused projection + attention kernel (seq_attn.hip) at BASELINE C2: s_memtime stamps of workgroup 0 at the
top of every stage, per iteration (slot): slot 0 = head 0 (projection only), 1-11 fused, 12 = attention of head 11 alone, 13 = the
next sequence's head 0 ..."""
import ctypes as C
import os
import sys

os.environ["FDMI_STAMPS"] = "1"
os.environ.setdefault("FDMI_FUSE_ATTN", "2")   # the 32-row kernel (seq_attn.hip); sa16_stamps.py is the 16-row one
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
a = buf[n0:n0 + 4 * 64 * 16].reshape(4, 64, 16).astype(np.int64)
for w in (0, 3):
    s = a[w]
    used = [i for i in range(64) if s[i, 0]]
    print(f"wave {w}: {len(used)} iterations recorded; cycles per stage (stage k top -> stage k+1 top), last column = whole iteration")
    for i in used[:27]:
        r = s[i]
        nxt = s[i + 1, 0] if i + 1 < 64 and s[i + 1, 0] else 0
        d = [int(r[k + 1] - r[k]) if r[k + 1] and r[k] else 0 for k in range(11)]
        last = int(nxt - r[11]) if nxt and r[11] else 0
        tot = int(nxt - r[0]) if nxt else 0
        print(f"  it {i:2d}: " + " ".join(f"{v:5d}" for v in d) + f" {last:6d} | {tot:7d}")
        if r[12] and r[13] and r[14] and r[15]:  # stage 0 in pieces (main-loop iterations only)
            ss = int(os.environ.get("SUBSTAGE", "0"))  # the stage the library was built to stamp in pieces (-DFDMI_SA_SUBSTAGE)
            end = r[ss + 1] if ss < 11 else nxt
            pcs = [r[15] - r[ss], r[12] - r[15], r[13] - r[12], r[14] - r[13], end - r[14]]
            print("         stage %d: wait+barrier %d | issue, copy-out, slot -1 %d | slots 0-6 %d | slots 7-12 %d | slots 13-17 %d" % ((ss,) + tuple(int(v) for v in pcs)))
