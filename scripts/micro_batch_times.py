#!/usr/bin/env python3
"""sample_on_device with and without micro-batches (sampling.MICRO_ROWS): C2 (512 x 128) and C5 (128 x 512), released architecture,
on-device Philox noise, final state only.  Env: STEPS (default 30)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from foldingdiff_amd import beta_schedules, modelling, sampling  # noqa: E402

steps = int(os.environ.get("STEPS", 30))
betas = beta_schedules.cosine_beta_schedule(1000)
for B, L in ((512, 128), (128, 512), (1024, 128)):
    cfg = dict(hidden_size=384, num_attention_heads=12, intermediate_size=768, num_hidden_layers=12,
               max_position_embeddings=max(128, L), position_embedding_type="relative_key")
    torch.manual_seed(0)
    model = modelling.BertForDiffusionBase(modelling.BertConfig(**cfg), [True] * 6).to("cuda:0")
    model.prepare(betas)
    x = torch.randn(B, L, 6, device="cuda:0")
    lens = torch.full((B,), L, dtype=torch.int32, device="cuda:0")
    res = {}
    for rows in (0, 32768, 0, 32768):
        sampling.MICRO_ROWS = rows
        sampling.sample_on_device(model, x, lens, betas, seed=1, t_start=3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = sampling.sample_on_device(model, x, lens, betas, seed=1, t_start=steps - 1)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        res.setdefault(rows, []).append((ms, out))
        print(f"B={B} L={L} micro rows {rows:6d}: {ms:.3f} ms per step  {B / ms:.2f} backbones/s at T=1000", flush=True)
    print("   identical:", torch.equal(res[0][0][1], res[32768][0][1]), flush=True)
    del model
