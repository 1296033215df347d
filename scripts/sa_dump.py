#!/usr/bin/env python3
"""Register dump of the fused kernel (library built with FDMI_SA_DUMP=1; FDMI_STAMPS=1 selects the instrumented kernel): the
attention of item 0 (sequence 0 of workgroup 0, head 0), wave 0 (queries 0-31): projection accumulators, raw S^T, S^T with the
band, probabilities, O^T -- each compared with the oracle's float64 values in the MFMA C/D register layout."""
import ctypes as C
import os
import sys

os.environ["FDMI_STAMPS"] = "1"
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from foldingdiff_amd import _binding, beta_schedules, modelling  # noqa: E402
from oracle import ref_model, ref_sampling  # noqa: E402

B, L, d, H = 3, 128, 384, 12
lens = [128, 100, 128]
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
cap = {}
sa = oracle.encoder.layer[0].attention.self
sa.query.register_forward_hook(lambda m, a, o: cap.__setitem__("q", o.detach().double()))
sa.key.register_forward_hook(lambda m, a, o: cap.__setitem__("k", o.detach().double()))
sa.value.register_forward_hook(lambda m, a, o: cap.__setitem__("v", o.detach().double()))
oracle(x, t, attention_mask=mask)
E = sa.distance_embedding.weight.detach().double()
pm.set_option("fuse_attn", 2)   # the 32-row kernel (seq_attn.hip)
pm.set_option("use_graph", 0)
pm(x, t, attention_mask=mask)
n0 = 5 * 8 * 64 * 6 + 4 * 64 * 8 + 4 * 64 * 16
n = n0 + 16384
buf = np.zeros(n, dtype=np.uint64)
_binding.check(lib.fd_debug_read(h, b"stamps", buf.ctypes.data_as(C.c_void_p), 2 * n))
dump = buf[n0:].view(np.float32).reshape(-1, 64)  # [value index][lane]
b, hh = 0, 0
q = cap["q"][b].reshape(L, H, 32)[:, hh].numpy(); k = cap["k"][b].reshape(L, H, 32)[:, hh].numpy(); v = cap["v"][b].reshape(L, H, 32)[:, hh].numpy()
lane = np.arange(64); l31 = lane & 31; half = lane >> 5
rmap = lambda r: (r & 3) + 8 * (r >> 2)  # + 4 half


def ratio_report(name, got, want):
    """got, want: arrays of the same shape; prints the least-squares scale and the worst relative deviation from it"""
    s = float((got * want).sum() / (want * want).sum())
    dev = np.abs(got - s * want).max() / max(np.abs(s * want).max(), 1e-30)
    print(f"{name:28s} scale {s:12.5e}   worst deviation {dev:.2e} of max")
    return s, dev


# projection accumulators of head 0, wave 0: q, k swapped form (lane = token l31, register r -> feature 8 q + 4 half + e), v normal form
for j, (nm, ref) in enumerate((("eo q", q), ("eo k", k))):
    got = dump[16 * j: 16 * j + 16]  # [r][lane]
    want = np.stack([ref[l31, rmap(r) + 4 * half] for r in range(16)])
    ratio_report(nm, got, want)
got = dump[32:48]
want = np.stack([v[rmap(r) + 4 * half, l31] for r in range(16)])  # lane = feature, register -> token
ratio_report("eo v", got, want)
idx = np.arange(L)[:, None] - np.arange(L)[None, :] + 127
rel = np.einsum("ld,lrd->lr", q, E.numpy()[idx])
S0 = q @ k.T
for nm, base, ref in (("raw S^T", 48, S0), ("S^T + band", 112, S0 + rel)):
    for tt in range(4):
        got = dump[base + 16 * tt: base + 16 * tt + 16]
        want = np.stack([ref[l31, 32 * tt + rmap(r) + 4 * half] for r in range(16)])  # [query l31][key]
        ratio_report(f"{nm} tile {tt}", got, want)
Sfull = (S0 + rel) / np.sqrt(32.0)
P = np.exp(Sfull - Sfull.max(1, keepdims=True))
for tt in range(4):
    got = dump[176 + 16 * tt: 176 + 16 * tt + 16]
    want = np.stack([P[l31, 32 * tt + rmap(r) + 4 * half] for r in range(16)])
    ratio_report(f"probabilities tile {tt}", got, want)
O = P @ v  # [query][d]
got = dump[240:256]
want = np.stack([O[l31, rmap(r) + 4 * half] for r in range(16)])  # O^T: lane = query, register -> d
ratio_report("O^T", got, want)
print("l_run / sum P * 1024:", dump[256][:4] / (P.sum(1)[:4] * 1024))

# which k-tiles of the projection are in the dumped accumulators?  least squares over the 12 partial products
hcap = {}
oracle.encoder.layer[0].register_forward_pre_hook(lambda m, a: hcap.__setitem__("h", a[0].detach().double()))
oracle(x, t, attention_mask=mask)
hin = hcap["h"][b].numpy()  # [L, d]
for j, (nm, lin) in enumerate((("q", sa.query), ("k", sa.key), ("v", sa.value))):
    W = lin.weight.detach().double().numpy()[hh * 32: hh * 32 + 32]  # [32, d]
    got = dump[16 * j: 16 * j + 16]
    parts = []
    for kt in range(12):
        part = hin[:, 32 * kt: 32 * kt + 32] @ W[:, 32 * kt: 32 * kt + 32].T  # [L, 32]
        if j < 2:
            parts.append(np.stack([part[l31, rmap(r) + 4 * half] for r in range(16)]).ravel())
        else:
            parts.append(np.stack([part[rmap(r) + 4 * half, l31] for r in range(16)]).ravel())
    A = np.stack(parts, 1)
    coef, *_ = np.linalg.lstsq(A, got.ravel().astype(np.float64), rcond=None)
    print(f"{nm}: per-k-tile coefficients / median:", np.round(coef / np.median(coef), 3))
np.savez_compressed(os.path.join(REPO, "gpurun_out", "r5f", "dump.npz"), dump=dump[:320], q=q, k=k, v=v, hin=hin)
got = dump[0:16].astype(np.float64)
want = np.stack([q[l31, rmap(r) + 4 * half] for r in range(16)])
s = (got * want).sum() / (want * want).sum()
res = np.abs(got - s * want) / np.abs(s * want).max()
print("eo q residual by register (rows) max over lanes:", np.round(res.max(1), 3))
print("eo q residual by lane max over registers (lanes 0-31):", np.round(res.max(0)[:32], 3))
print("eo q residual by lane max over registers (lanes 32-63):", np.round(res.max(0)[32:], 3))
