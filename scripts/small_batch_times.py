#!/usr/bin/env python3
"""Step time of the released architecture at small batches (L = 128): where the few-rows GEMM path (gemm_ws.hip) matters.
Env: STEPS (default 20), TAG."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from foldingdiff_amd import beta_schedules, modelling, sampling  # noqa: E402

RELEASED = dict(hidden_size=384, num_attention_heads=12, intermediate_size=768, num_hidden_layers=12,
                max_position_embeddings=128, position_embedding_type="relative_key")
steps = int(os.environ.get("STEPS", 20))
torch.manual_seed(0)
model = modelling.BertForDiffusionBase(modelling.BertConfig(**RELEASED), [True] * 6).to("cuda:0")
betas = beta_schedules.cosine_beta_schedule(1000)
model.prepare(betas)
tag = os.environ.get("TAG", "")
for B in [int(b) for b in os.environ.get("BATCHES", "1,4,8,16,32,64,96,128").split(",")]:
    x = torch.randn(B, 128, 6, device="cuda:0")
    lens = torch.full((B,), 128, dtype=torch.int32, device="cuda:0")
    sampling.sample_on_device(model, x, lens, betas, seed=1, t_start=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sampling.sample_on_device(model, x, lens, betas, seed=1, t_start=steps - 1)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    print(f"{tag} B={B:4d} rows={B * 128:6d}: {ms:.3f} ms/step  {B / ms:.2f} backbones/s at T=1000", flush=True)
