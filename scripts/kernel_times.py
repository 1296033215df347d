#!/usr/bin/env python3
"""Per-kernel-class average launch time (hipEvents, eager launches) at BASELINE C2 shapes for a few reverse steps.
Results are not checked (usable with the FDMI_*_DBG ablation builds).  Env: B, L, STEPS."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from foldingdiff_amd import _binding, beta_schedules, modelling, sampling  # noqa: E402

RELEASED = dict(hidden_size=384, num_attention_heads=12, intermediate_size=768, num_hidden_layers=12,
                max_position_embeddings=int(os.environ.get("MAXPOS", 128)), position_embedding_type="relative_key")
B, L, steps = int(os.environ.get("B", 512)), int(os.environ.get("L", 128)), int(os.environ.get("STEPS", 6))
torch.manual_seed(0)
model = modelling.BertForDiffusionBase(modelling.BertConfig(**RELEASED), [True] * 6).to("cuda:0")
betas = beta_schedules.cosine_beta_schedule(1000)
h = model.prepare(betas)
if os.environ.get("SPLIT_QKV"):
    model.set_option("split_qkv", int(os.environ["SPLIT_QKV"]))
lib = _binding.load()
x = torch.randn(B, L, 6, device="cuda:0")
lens = torch.full((B,), L, dtype=torch.int32, device="cuda:0")
try:
    sampling.sample_on_device(model, x, lens, betas, seed=1, t_start=1)
except _binding.FdmiError as e:
    print("ignored:", e, file=sys.stderr)
_binding.check(lib.fd_profile_reset(h))
_binding.check(lib.fd_profile_every(h, 1))
try:
    sampling.sample_on_device(model, x, lens, betas, seed=1, t_start=steps - 1)
except _binding.FdmiError as e:  # (ablation builds produce garbage, possibly non-finite: the times still stand)
    print("ignored:", e, file=sys.stderr)
_binding.check(lib.fd_profile_every(h, 0))
name_p, ms, n, fl, by = C.c_char_p(), C.c_double(), C.c_int64(), C.c_double(), C.c_double()
tot = 0.0
out = []
for i in range(lib.fd_profile_count(h)):
    _binding.check(lib.fd_profile_get(h, i, C.byref(name_p), C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
    if n.value:
        out.append(f"{name_p.value.decode()}={ms.value / n.value * 1e3:.1f}")
        tot += ms.value / steps
print(os.environ.get("TAG", ""), " ".join(out), f"| step={tot:.3f} ms")
