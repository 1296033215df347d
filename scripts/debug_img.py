#!/usr/bin/env python3
"""Stage-by-stage comparison of the row-image kernels (FD_PREC_F16X3) with the CPU oracle.

Runs one forward per stage with option "debug_stop" = n (the step ends after n launches), reads the
intermediate back through fd_debug_read and prints the max abs error against the oracle's activations
captured with torch hooks.  Needs an MI355X.  Usage: python scripts/debug_img.py [small|released|mini]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from foldingdiff_amd import _binding, beta_schedules, modelling  # noqa: E402
from oracle import ref_model, ref_sampling  # noqa: E402

CONFIGS = {
    "small": dict(hidden=64, heads=2, ff=128, maxpos=64, B=3, L=37, lens=[37, 33, 8]),
    "mini": dict(hidden=192, heads=6, ff=384, maxpos=128, B=4, L=64, lens=[64, 50, 33, 64]),
    "released": dict(hidden=384, heads=12, ff=768, maxpos=128, B=5, L=128, lens=[128, 128, 77, 50, 1]),
    "released6": dict(hidden=384, heads=12, ff=768, maxpos=128, B=6, L=128, lens=[128, 128, 77, 50, 1, 100]),
    "ragged": dict(hidden=384, heads=12, ff=768, maxpos=128, B=7, L=101, lens=[101, 50, 99, 100, 64, 3, 77]),
    "long": dict(hidden=64, heads=2, ff=128, maxpos=512, B=2, L=300, lens=[300, 129]),
}


def main(name):
    cf = CONFIGS[name]
    d, H, ff, maxpos, B, L, lens = cf["hidden"], cf["heads"], cf["ff"], cf["maxpos"], cf["B"], cf["L"], cf["lens"]
    ocfg = ref_model.OracleConfig(hidden_size=d, num_attention_heads=H, intermediate_size=ff, num_hidden_layers=1,
                                  max_position_embeddings=maxpos, position_embedding_type="relative_key")
    o32 = ref_model.synthetic_model(ocfg, (True,) * 6, "gaussian_fourier", "mlp", seed=3)
    pcfg = modelling.BertConfig(hidden_size=d, num_attention_heads=H, intermediate_size=ff, num_hidden_layers=1,
                                max_position_embeddings=maxpos, position_embedding_type="relative_key")
    pm = modelling.BertForDiffusionBase(pcfg, [True] * 6)
    pm.load_state_dict(o32.state_dict())
    pm.to("cuda:0")
    pm.set_precision("f16x3")
    T = 100
    h = pm.prepare(beta_schedules.cosine_beta_schedule(T))
    lib = _binding.load()

    g = torch.Generator().manual_seed(1)
    x = ref_sampling.wrap(torch.randn(B, L, 6, generator=g) * 1.5)
    mask = torch.zeros(B, L)
    for i, n in enumerate(lens):
        mask[i, :n] = 1.0
    t = torch.full((B,), 42, dtype=torch.long)

    cap = {}
    layer = o32.encoder.layer[0]
    layer.register_forward_pre_hook(lambda m, a: cap.__setitem__("h", a[0].detach().clone()))
    layer.attention.self.query.register_forward_hook(lambda m, a, o: cap.__setitem__("q", o.detach().clone()))
    layer.attention.self.key.register_forward_hook(lambda m, a, o: cap.__setitem__("k", o.detach().clone()))
    layer.attention.self.value.register_forward_hook(lambda m, a, o: cap.__setitem__("v", o.detach().clone()))
    layer.attention.self.register_forward_hook(
        lambda m, a, o: cap.__setitem__("ctx", (o[0] if isinstance(o, tuple) else o).detach().clone()))
    layer.attention.register_forward_hook(
        lambda m, a, o: cap.__setitem__("a", (o[0] if isinstance(o, tuple) else o).detach().clone()))
    layer.intermediate.register_forward_hook(lambda m, a, o: cap.__setitem__("g", o.detach().clone()))
    layer.register_forward_hook(lambda m, a, o: cap.__setitem__("h_out", (o[0] if isinstance(o, tuple) else o).detach().clone()))
    o32.token_decoder.dense1.register_forward_hook(
        lambda m, a, o: cap.__setitem__("g_head", torch.nn.functional.gelu(o.detach().clone())))
    want_eps = o32(x, t, attention_mask=mask).numpy()

    Lr = (L + 7) // 8 * 8
    rows_cap = (B * Lr + 127) // 128 * 128
    T_ = 4 if L > 128 else (L + 31) // 32
    LTOT = ((L + 127) // 128 if L > 128 else 1) * 32 * T_

    def run(stop):
        pm.set_option("debug_stop", stop)
        return pm(x, t, attention_mask=mask).numpy()

    def read(nm, n):
        out = np.empty(n, dtype=np.float32)
        _binding.check(lib.fd_debug_read(h, nm.encode(), out.ctypes.data_as(C.c_void_p), n))
        return out

    def rows_img(nm, K):
        a = read(nm, rows_cap * K).reshape(rows_cap, K)
        return np.stack([a[b * Lr: b * Lr + L] for b in range(B)])  # [B, L, K]

    def qkv(nm):
        a = read(nm, B * H * LTOT * 32).reshape(B, H, LTOT, 32)[:, :, :L]  # [B,H,L,32]
        return a.transpose(0, 2, 1, 3).reshape(B, L, H * 32)

    def report(tag, got, want, valid_only=True):
        want = want.numpy() if isinstance(want, torch.Tensor) else want
        worst, where = 0.0, None
        for b in range(B):
            n = lens[b] if valid_only else L
            dlt = np.abs(got[b, :n] - want[b, :n])
            if not np.isfinite(got[b, :n]).all():
                print(f"  {tag}: NON-FINITE values in sequence {b}")
            if dlt.size and dlt.max() > worst:
                worst = float(dlt.max())
                where = (b,) + tuple(int(i) for i in np.unravel_index(np.argmax(dlt), dlt.shape))
        scale = float(np.abs(want).max())
        print(f"  {tag:8s} max|d| = {worst:.3e}   (max|want| = {scale:.3e})  at {where}")
        return worst

    split = int(os.environ.get("SPLIT_QKV", "0"))
    pm.set_option("split_qkv", split)
    merged = H % 6 == 0 and not split   # q | k | v in one launch: one stage less
    print(f"== {name}: d={d} H={H} ff={ff} L={L} lens={lens}  ({'one q|k|v launch' if merged else 'q|k and v launches'})")
    run(1); report("h", rows_img("h", d), cap["h"], valid_only=False)
    n = 2
    run(n); report("q", qkv("q"), cap["q"], False); report("k", qkv("k"), cap["k"], False)
    if not merged:
        n += 1
        run(n)
    report("v", qkv("v"), cap["v"], False)
    if os.environ.get("DEBUG_V"):  # where is v wrong: per sequence x head, and the positions of one bad (sequence, head)
        dv = np.abs(qkv("v") - cap["v"].numpy()).reshape(B, L, H, 32)
        bad = dv > 1e-4
        print("    bad fraction per (sequence, head):")
        for b in range(B):
            print("     ", b, " ".join(f"{bad[b, :, hh].mean():.2f}" for hh in range(H)))
        bb, hh = np.unravel_index(np.argmax(bad.mean(axis=(1, 3))), (B, H))
        if bad.any():
            print(f"    sequence {bb} head {hh}: bad positions", np.nonzero(bad[bb, :, hh].any(axis=1))[0].tolist()[:64])
            print(f"    sequence {bb} head {hh}: bad d at the first bad position",
                  np.nonzero(bad[bb, np.nonzero(bad[bb, :, hh].any(axis=1))[0][0], hh])[0].tolist())
    run(n + 1); report("ctx", rows_img("ctx", d), cap["ctx"], False)
    run(n + 2); report("a", rows_img("a", d), cap["a"], False)
    run(n + 3); report("g", rows_img("g", ff), cap["g"], False)
    run(n + 4); report("h_out", rows_img("h_out", d), cap["h_out"], False)
    run(n + 5); report("g_head", rows_img("g_head", d), cap["g_head"], False)
    got = run(0)
    report("eps", got, want_eps, False)


if __name__ == "__main__":
    for nm in (sys.argv[1:] or ["small"]):
        main(nm)
