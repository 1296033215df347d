#!/usr/bin/env python3
"""Fused q|k|v projection + attention kernel (seq_attn.hip) against the two-kernel path and the oracle.
   * released shape (d 384, 12 heads, 12 layers) and the mini shape (d 192, 6 heads), L in {128, 101, 97}, ragged lengths;
   * padded rows and packed rows (fd_set_option varlen);
   * the model output must be BIT-IDENTICAL with fuse_attn 0 / 1 (same arithmetic in the same order) and within 1e-5 of the oracle."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from foldingdiff_amd import beta_schedules, modelling  # noqa: E402
from oracle import ref_model, ref_sampling  # noqa: E402

CASES = [
    dict(hidden=384, heads=12, ff=768, layers=12, B=5, L=128, lens=[128, 128, 77, 50, 1]),
    dict(hidden=384, heads=12, ff=768, layers=2, B=7, L=101, lens=[101, 50, 99, 100, 64, 3, 77]),
    dict(hidden=384, heads=12, ff=768, layers=2, B=3, L=97, lens=[97, 97, 33]),
    dict(hidden=192, heads=6, ff=384, layers=6, B=4, L=128, lens=[128, 50, 33, 100]),
    dict(hidden=384, heads=12, ff=768, layers=2, B=300, L=128, lens=None),
    dict(hidden=384, heads=12, ff=768, layers=1, B=600, L=128, lens=None),
]
bad = 0
for cf in CASES:
    d, H, ff, nl, B, L = cf["hidden"], cf["heads"], cf["ff"], cf["layers"], cf["B"], cf["L"]
    lens = cf["lens"] or [int(v) for v in np.random.RandomState(0).randint(1, L + 1, size=B)]
    kw = dict(hidden_size=d, num_attention_heads=H, intermediate_size=ff, num_hidden_layers=nl, max_position_embeddings=128,
              position_embedding_type="relative_key")
    oracle = ref_model.synthetic_model(ref_model.OracleConfig(**kw), (True,) * 6, "gaussian_fourier", "mlp", seed=3)
    pm = modelling.BertForDiffusionBase(modelling.BertConfig(**kw), [True] * 6)
    pm.load_state_dict(oracle.state_dict())
    pm.to("cuda:0")
    pm.prepare(beta_schedules.cosine_beta_schedule(100))
    g = torch.Generator().manual_seed(1)
    x = ref_sampling.wrap(torch.randn(B, L, 6, generator=g) * 1.5)
    mask = torch.zeros(B, L)
    for i, n in enumerate(lens):
        mask[i, :n] = 1.0
    t = torch.full((B,), 42, dtype=torch.long)
    want = oracle(x, t, attention_mask=mask).detach() if B <= 16 else None
    for packed in (0, 1):
        pm.set_option("varlen", packed)
        outs = {}
        for fa in (0, 1):
            pm.set_option("fuse_attn", fa)
            outs[fa] = pm(x, t, attention_mask=mask).detach().cpu()
        valid = mask.bool()
        a, b = outs[0][valid], outs[1][valid]
        same = bool(torch.equal(a, b))
        dmax = float((a - b).abs().max())
        err = float((outs[1][valid] - want[valid]).abs().max()) if want is not None else float("nan")
        err0 = float((outs[0][valid] - want[valid]).abs().max()) if want is not None else float("nan")
        ok = same and (want is None or err <= 1e-5)
        bad += 0 if ok else 1
        print(f"d={d} layers={nl} B={B} L={L} packed={packed}: fused == two-kernel bitwise: {same} (max|d| {dmax:.3e}); "
              f"vs oracle fused {err:.3e} two-kernel {err0:.3e}  {'OK' if ok else 'FAIL'}", flush=True)
    pm.set_option("varlen", 0)
    pm.set_option("fuse_attn", -1)
print("sa_check:", "ALL OK" if bad == 0 else f"{bad} FAILED")
sys.exit(1 if bad else 0)
