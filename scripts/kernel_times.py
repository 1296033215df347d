#!/usr/bin/env python3
"""Per-kernel-class average launch time (hipEvents, eager launches) at BASELINE C2 shapes for a few reverse steps.
Results are not checked.  Env: B, L, STEPS, FDMI_LIB (an experiment build of `python -m foldingdiff_amd.build <variant> ...`).

An experiment build carries build_info.json: its -D flags and the MFMA count of every matrix kernel next to the default build's.  A
library whose counts DIFFER from the default's is an ablation (or hipcc has deleted matrix instructions whose only consumer the
flags removed: round 5 read such a build as "97 % of the matrix peak"): it is refused unless --allow-ablation is given, and its
counts are printed next to its times either way."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from foldingdiff_amd import _binding, beta_schedules, modelling, sampling  # noqa: E402

def _ablation_guard():
    import json
    lib = os.environ.get("FDMI_LIB")
    if not lib:
        return ""
    info = os.path.join(os.path.dirname(lib), "build_info.json")
    if not os.path.exists(info):
        return " [no build_info.json beside FDMI_LIB: MFMA counts unknown]"
    meta = json.load(open(info))
    diff = {k: (meta["mfma_default"].get(k), v) for k, v in meta["mfma"].items() if meta["mfma_default"].get(k) != v}
    diff.update({k: (v, None) for k, v in meta["mfma_default"].items() if k not in meta["mfma"]})
    if not diff:
        return f" [defines {meta['defines']}: MFMA counts = the default build's]"
    short = {k[-60:]: v for k, v in sorted(diff.items())}
    msg = f" [ABLATION defines {meta['defines']}: v_mfma per kernel (default, this build) {short}]"
    if "--allow-ablation" not in sys.argv:
        raise SystemExit("kernel_times.py: refusing to time an ablation build without --allow-ablation:" + msg)
    return msg


GUARD = _ablation_guard()
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
print(os.environ.get("TAG", ""), " ".join(out), f"| step={tot:.3f} ms" + GUARD)
