#!/usr/bin/env python3
"""Per-kernel average launch times at the two chunks of BASELINE C3 (lengths 50..127 x 10, batch_size 512: B=512/L=101 and
B=268/L=127) with the sweep's real ragged lengths and packed rows, as sampling.sample runs them; plus the C2 shape for
reference.  Shows where C3 loses against C2 per token: tile quantization of the GEMMs, ceil8 padding, short sequences in
the attention kernel.  Env: STEPS (default 6)."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from foldingdiff_amd import _binding, beta_schedules, modelling, sampling  # noqa: E402

RELEASED = dict(hidden_size=384, num_attention_heads=12, intermediate_size=768, num_hidden_layers=12,
                max_position_embeddings=128, position_embedding_type="relative_key")
steps = int(os.environ.get("STEPS", 6))
torch.manual_seed(0)
model = modelling.BertForDiffusionBase(modelling.BertConfig(**RELEASED), [True] * 6).to("cuda:0")
betas = beta_schedules.cosine_beta_schedule(1000)
h = model.prepare(betas)
lib = _binding.load()
lengths = [l for l in range(50, 128) for _ in range(10)]
chunks = {"c3 chunk 0": lengths[:512], "c3 chunk 1": lengths[512:], "c3 chunk 0, run of whole rounds": lengths[:437], "c3 chunk 0, rest": lengths[437:512],
          "c3 merged": lengths, "c2": [128] * 512, "c2 half": [128] * 256, "b8": [128] * 8}
if os.environ.get("ONLY"):
    chunks = {k: v for k, v in chunks.items() if any(o in k for o in os.environ["ONLY"].split(","))}
hint = int(os.environ.get("ROWS_HINT", 1))   # 1: tell the library the exact row count of a packed run (as sampling.sample does)
tag = os.environ.get("TAG", "")
for name, these in chunks.items():
    B, L = len(these), max(these)
    packed = 0 if name in ("c2", "c5", "b8") else 1
    model.set_option("varlen", packed)
    model.set_option("rows_hint", sum((l + 7) // 8 * 8 for l in these) if packed and hint else 0)
    x = torch.randn(B, L, 6, device="cuda:0")
    lens = torch.tensor(these, dtype=torch.int32, device="cuda:0")
    sampling.sample_on_device(model, x, lens, betas, seed=1, t_start=1)
    _binding.check(lib.fd_profile_reset(h))
    _binding.check(lib.fd_profile_every(h, 1))
    sampling.sample_on_device(model, x, lens, betas, seed=1, t_start=steps - 1)
    _binding.check(lib.fd_profile_every(h, 0))
    name_p, ms, n, fl, by = C.c_char_p(), C.c_double(), C.c_int64(), C.c_double(), C.c_double()
    tot, out = 0.0, []
    for i in range(lib.fd_profile_count(h)):
        _binding.check(lib.fd_profile_get(h, i, C.byref(name_p), C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
        if n.value:
            out.append(f"{name_p.value.decode()}={ms.value / n.value * 1e3:.1f}")
            tot += ms.value / steps
    rows = sum((l + 7) // 8 * 8 for l in these) if packed else B * L
    tokens = sum(these)
    print(f"{tag} {name}: B={B} L={L} tokens={tokens} rows={rows} panels={-(-rows // 128)} " + " ".join(out)
          + f" | step={tot:.3f} ms  {tokens / tot / 1e3:.2f} useful tokens/us")
model.set_option("varlen", 0)
model.set_option("rows_hint", 0)
