#!/usr/bin/env python3
"""Repeat one small configuration of scripts/ffn16_check.py many times (races show as occasional wrong rows).  Env: D (192 / 384), N."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from foldingdiff_amd import _binding, beta_schedules, modelling  # noqa: E402
from oracle import ref_model, ref_sampling  # noqa: E402

d = int(os.environ.get("D", 192))
B, L, nl = int(os.environ.get("B", 4)), 128, int(os.environ.get("LAYERS", 6))
kw = dict(hidden_size=d, num_attention_heads=d // 32, intermediate_size=2 * d, num_hidden_layers=nl, max_position_embeddings=128,
          position_embedding_type="relative_key")
oracle = ref_model.synthetic_model(ref_model.OracleConfig(**kw), (True,) * 6, "gaussian_fourier", "mlp", seed=3)
pm = modelling.BertForDiffusionBase(modelling.BertConfig(**kw), [True] * 6)
pm.load_state_dict(oracle.state_dict())
pm.to("cuda:0")
pm.prepare(beta_schedules.cosine_beta_schedule(100))
g = torch.Generator().manual_seed(1)
x = ref_sampling.wrap(torch.randn(B, L, 6, generator=g) * 1.5)
mask = torch.ones(B, L)
t = torch.full((B,), 42, dtype=torch.long)
pm.set_option("fuse_attn", 0)
pm.set_option("fuse_ffn", 0)
ref = pm(x, t, attention_mask=mask).detach().cpu()
pm.set_option("fuse_ffn", int(os.environ.get("FUSE_FFN", 2)))
bad = 0
worst = 0.0
for it in range(int(os.environ.get("N", 40))):
    try:
        o = pm(x, t, attention_mask=mask).detach().cpu()
        e = float((o - ref).abs().max())
        worst = max(worst, e if np.isfinite(e) else 1e9)
        if not (e < 1e-5):
            bad += 1
            rows = (o - ref).abs().amax(dim=2)
            print(f"  run {it}: max {e:.3e}; wrong rows (sequence: first..last of 16-row blocks):",
                  {b: sorted(set(int(i) // 16 for i in torch.nonzero(rows[b] > 1e-5).flatten())) for b in range(B) if bool((rows[b] > 1e-5).any())})
    except _binding.FdmiError:
        bad += 1
        print(f"  run {it}: non-finite")
print(os.environ.get("TAG", ""), f"d={d} layers={nl}: {bad} bad runs of {it + 1}; worst finite error {worst:.3e}")
