#!/usr/bin/env python3
"""The fused feed-forward kernel (ffn16.hip; fuse_ffn = 1: BertIntermediate + GELU + BertOutput in one launch, 2: BertSelfOutput in front
of them too) against the oracle, next to the GEMM path (0): released shape and the mini shape, L from 7 to 128, ragged lengths, padded and packed rows, with the fused
attention kernel on and off.  Gate: max|d| <= 1e-5 against the fp32 oracle on valid positions (tests/test_gpu_parity.py: FWD_TOL)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from foldingdiff_amd import _binding, beta_schedules, modelling  # noqa: E402
from oracle import ref_model, ref_sampling  # noqa: E402

CASES = [
    dict(hidden=384, heads=12, ff=768, layers=1, B=3, L=128, lens=[128, 128, 77]),
    dict(hidden=384, heads=12, ff=768, layers=12, B=5, L=128, lens=[128, 128, 77, 50, 1]),
    dict(hidden=384, heads=12, ff=768, layers=2, B=7, L=101, lens=[101, 50, 99, 100, 64, 3, 77]),
    dict(hidden=384, heads=12, ff=768, layers=2, B=4, L=64, lens=[64, 17, 33, 48]),
    dict(hidden=384, heads=12, ff=768, layers=2, B=3, L=7, lens=[7, 1, 3]),
    dict(hidden=192, heads=6, ff=384, layers=6, B=4, L=128, lens=[128, 50, 33, 100]),
    dict(hidden=192, heads=6, ff=384, layers=6, B=4, L=64, lens=[64, 50, 33, 64]),
    dict(hidden=384, heads=12, ff=768, layers=2, B=300, L=128, lens=None),
    dict(hidden=384, heads=12, ff=768, layers=1, B=600, L=90, lens=None),
]
only = os.environ.get("CASES")
if only:
    CASES = [CASES[int(i)] for i in only.split(",")]
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
    valid = mask.bool()
    for packed in (0, 1):
        pm.set_option("varlen", packed)
        outs = {}
        for fa in (0, 1, 2, 3):
            pm.set_option("fuse_attn", 1 if fa == 3 else 0)
            pm.set_option("fuse_ffn", (0, 1, 2, 2)[fa])
            try:
                outs[fa] = pm(x, t, attention_mask=mask).detach().cpu()
            except _binding.FdmiError as e:
                print("   case", fa, "->", e)
        sel = valid if packed else torch.ones_like(valid)
        ref = want if want is not None else outs[0]
        errs = {fa: float((o[sel] - ref[sel]).abs().max()) for fa, o in outs.items()}
        nan = {fa: int(torch.isnan(o[sel]).sum()) for fa, o in outs.items()}
        ok = all(errs.get(k, 1.0) <= (1e-5 if want is not None else 2e-5) and nan.get(k, 1) == 0 for k in (1, 2, 3))
        bad += 0 if ok else 1
        print(f"d={d} layers={nl} B={B} L={L} packed={packed}: max|d| vs {'oracle' if want is not None else 'two-kernel'}: "
              + "  ".join(f"{('three GEMMs', 'ffn16', 'tail', 'tail + seq_attn16')[fa]}: {e:.3e}" + (f" ({nan[fa]} NaN)" if nan[fa] else "") for fa, e in errs.items())
              + f"   {'OK' if ok else 'FAIL'}", flush=True)
        if not ok and want is not None and os.environ.get("VERBOSE"):
            dd = (outs[1] - ref).abs()
            for b in range(min(B, 4)):
                print("      seq", b, "len", lens[b], "max err by 16-row block:", [f"{float(dd[b, i:i + 16].max()):.1e}" for i in range(0, L, 16)])
    pm.set_option("varlen", 0)
    pm.set_option("fuse_attn", -1)
    pm.set_option("fuse_ffn", -1)
print("ffn16_check:", "ALL OK" if bad == 0 else f"{bad} FAILED")
sys.exit(1 if bad else 0)
