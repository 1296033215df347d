#!/usr/bin/env python3
"""
Generate the golden fixtures in tests/golden/ by running THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference; the GPU box does not
have it, which is why the outputs are committed):

    python tests/golden/make_golden.py

What is imported from the reference, unchanged: ``foldingdiff.sampling``,
``.beta_schedules``, ``.utils``, ``.datasets``, ``.modelling`` (two stub
modules stand in for the absent ``pytorch_lightning`` / ``biotite``; SURVEY
appendix C).  ``BertForDiffusionBase`` is instantiated with
``position_embedding_type="absolute"`` on the container's transformers 5.15
``BertEncoder``; its ``init_weights`` (broken under 5.15) is replaced by a
no-op and weights are then set from a seeded generator.  The relative_key
variant cannot be run through the reference class (5.15's BERT dropped it), so
for it the fixtures come from the oracle's restated model driven by the
reference's own, unmodified sampler -- and the oracle's relative-key einsum is
cross-checked here against HuggingFace's surviving implementation of the same
term (Wav2Vec2BertSelfAttention).

Fixtures written (all float32 unless noted):
  ref_schedules.npz   beta schedules + compute_alphas terms, T in {10,250,1000}
  ref_wrap.npz        modulo_with_wrapped_range on tensors + the unit-test KATs
  ref_noise.npz       NoisedAnglesDataset.sample_noise under a fixed seed
  ref_time_embed.npz  GaussianFourierProjection / Sinusoidal tables
  ref_abs_model.npz   reference BertForDiffusionBase(absolute): weights, x, t, mask, eps
  ref_abs_traj.npz    reference p_sample_loop on that model: T=10 trajectory
  ref_abs_sample.npz  reference sampling.sample() end to end (with mean offset)
  ref_nerf.npz        NERFBuilder coordinates (raw + centered) for the three angle feature sets, L in {1,2,37,128}
  c1_relkey.npz       BASELINE config C1 (mini relative_key model, L=64, T=10, B=4):
                      oracle model + reference sampler trajectory, fp32 and fp64
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("FD_REFERENCE", "/root/reference")


def import_reference():
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = type("LightningModule", (torch.nn.Module,), {})
    u = types.ModuleType("pytorch_lightning.utilities")
    u.rank_zero_info = lambda *a, **k: None
    u.rank_zero_only = lambda f: f
    pl.utilities = u
    sys.modules.update({"pytorch_lightning": pl, "pytorch_lightning.utilities": u})
    for n in ("biotite", "biotite.structure", "biotite.structure.io",
              "biotite.structure.io.pdb", "biotite.sequence"):
        sys.modules[n] = types.ModuleType(n)
    sys.modules["biotite.structure.io.pdb"].PDBFile = object
    sys.modules["biotite.sequence"].ProteinSequence = object
    sys.path.insert(0, REF)
    from foldingdiff import beta_schedules, datasets, modelling, sampling, utils
    return beta_schedules, datasets, modelling, sampling, utils


def state_to_np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def weight_digest(sd) -> str:
    h = hashlib.sha256()
    for k in sorted(sd.keys()):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k].detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def main():
    sys.path.insert(0, REPO)
    beta_schedules, datasets, modelling, sampling, utils = import_reference()
    from oracle import ref_model, ref_sampling

    # ------------------------------------------------------------ schedules
    out = {}
    for kind in ("cosine", "linear", "quadratic"):
        for T in (10, 250, 1000):
            betas = beta_schedules.get_variance_schedule(kind, T)
            terms = beta_schedules.compute_alphas(betas)
            for k, v in terms.items():
                out[f"{kind}_{T}_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "ref_schedules.npz"), **out)

    # ----------------------------------------------------------------- wrap
    g = torch.Generator().manual_seed(11)
    v = torch.randn(4096, generator=g) * 8.0
    v[:8] = torch.tensor([np.pi, -np.pi, 0.0, 2 * np.pi, -2 * np.pi, 3.0, -3.0, 1e-7])
    w = utils.modulo_with_wrapped_range(v, -np.pi, np.pi)
    w_t = utils.modulo_with_wrapped_range(v, -torch.pi, torch.pi)
    assert torch.equal(w, w_t)
    kat_in = np.array([[3, -2, 2], [5, -2, 2], [-1, -2, 2], [-3, -2, 2], [3, 0, 4], [5, 0, 4], [-1, 0, 4]], dtype=np.float64)
    kat_out = np.array([utils.modulo_with_wrapped_range(a, b, c) for a, b, c in kat_in])
    assert kat_out.tolist() == [-1, 1, -1, 1, 3, 1, 3]  # tests/test_utils.py:11-45
    np.savez_compressed(os.path.join(HERE, "ref_wrap.npz"), v=v.numpy(), w=w.numpy(), kat_in=kat_in, kat_out=kat_out)

    # ---------------------------------------------------------------- noise
    ds = datasets.NoisedAnglesDataset(
        datasets.AnglesEmptyDataset("canonical-full-angles", pad=128, mean_offset=np.zeros(6, "float32")),
        dset_key="angles", timesteps=10, beta_schedule="cosine")
    torch.manual_seed(7344)
    n1 = ds.sample_noise(torch.zeros((3, 128, 6), dtype=torch.float32))
    ds_mixed = datasets.NoisedAnglesDataset(
        datasets.AnglesEmptyDataset("canonical", pad=32, mean_offset=None),
        dset_key="angles", timesteps=10, beta_schedule="linear", angular_variance=0.5)
    torch.manual_seed(99)
    n2 = ds_mixed.sample_noise(torch.zeros((2, 32, 9), dtype=torch.float32))
    np.savez_compressed(os.path.join(HERE, "ref_noise.npz"), full_seed7344=n1.numpy(), mixed_seed99_var05=n2.numpy())

    # ----------------------------------------------------------- time embed
    torch.manual_seed(5)
    gfp = modelling.GaussianFourierProjection(embed_dim=64)
    tt = torch.arange(0, 1000)
    te = gfp(tt)
    sin = modelling.SinusoidalPositionEmbeddings(64)(tt)
    np.savez_compressed(os.path.join(HERE, "ref_time_embed.npz"), W=gfp.W.numpy(), gaussian_fourier=te.numpy(), sinusoidal=sin.numpy())

    # --------------------------------- reference model, absolute positions
    from transformers import BertConfig
    modelling.BertForDiffusionBase.init_weights = lambda self: None  # broken under transformers 5.x
    cfg = BertConfig(max_position_embeddings=64, num_attention_heads=2, hidden_size=64,
                     intermediate_size=128, num_hidden_layers=2, position_embedding_type="absolute",
                     hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, use_cache=False,
                     attn_implementation="eager")
    torch.manual_seed(1234)
    ref = modelling.BertForDiffusionBase(cfg, ft_is_angular=[True] * 6, time_encoding="gaussian_fourier", decoder="mlp")
    g = torch.Generator().manual_seed(4321)
    with torch.no_grad():
        for name, p in ref.named_parameters():
            if name.endswith("LayerNorm.weight") or name.endswith("layer_norm.weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.08 * torch.randn(p.shape, generator=g))
    ref.eval()
    B, L = 4, 48
    lens = [48, 31, 40, 7]
    x = ref_sampling.wrap(torch.randn(B, L, 6, generator=g) * 1.5)
    mask = torch.zeros(B, L)
    for i, n in enumerate(lens):
        mask[i, :n] = 1.0
    fw = {}
    for tval in (0, 1, 5, 9):
        t = torch.full((B,), tval, dtype=torch.long)
        with torch.no_grad():
            fw[f"eps_t{tval}"] = ref(x, t, attention_mask=mask).numpy()
    sd = {k: v for k, v in ref.state_dict().items()}
    np.savez_compressed(
        os.path.join(HERE, "ref_abs_model.npz"),
        x=x.numpy(), mask=mask.numpy(), lens=np.array(lens), **fw,
        **{"sd::" + k: v for k, v in state_to_np(sd).items()})

    # the oracle's restated model must reproduce the reference class
    ocfg = ref_model.OracleConfig(hidden_size=64, num_attention_heads=2, intermediate_size=128,
                                  num_hidden_layers=2, max_position_embeddings=64,
                                  position_embedding_type="absolute")
    om = ref_model.OracleBertForDiffusion(ocfg, [True] * 6)
    om.load_state_dict(sd, strict=True)
    for tval in (0, 1, 5, 9):
        t = torch.full((B,), tval, dtype=torch.long)
        d = (om(x, t, attention_mask=mask) - torch.from_numpy(fw[f"eps_t{tval}"])).abs().max().item()
        print(f"[abs forward] t={tval}: oracle vs reference-class max|d| = {d:.3e}")
        assert d < 5e-6

    # ------------------------ reference p_sample_loop on the reference model
    T = 10
    betas = beta_schedules.get_variance_schedule("cosine", T)
    torch.manual_seed(2024)
    x0 = ds.sample_noise(torch.zeros((B, 128, 6), dtype=torch.float32))[:, :L]
    traj = sampling.p_sample_loop(ref, lens, x0, T, betas, is_angle=[True] * 6, disable_pbar=True)
    # recover the per-step draws (CPU global generator, sequential)
    torch.manual_seed(2024)
    _ = torch.randn(B, 128, 6)
    step_noise = torch.zeros(T, B, L, 6)
    for i in reversed(range(1, T)):
        step_noise[i] = torch.randn(B, L, 6)
    traj_o = ref_sampling.p_sample_loop(ref, lens, x0, T, betas, [True] * 6, step_noise=step_noise)
    assert torch.equal(traj, traj_o), "oracle loop (explicit noise) != reference loop"
    np.savez_compressed(os.path.join(HERE, "ref_abs_traj.npz"), x0=x0.numpy(), lens=np.array(lens),
                        step_noise=step_noise.numpy(), traj=traj.numpy(), T=T)

    # ------------------------------- reference sampling.sample() end to end
    offs = np.array([0.3, -1.2, 3.0, 0.0, 1.9, -2.5], dtype=np.float32)
    ds2 = datasets.NoisedAnglesDataset(
        datasets.AnglesEmptyDataset("canonical-full-angles", pad=64, mean_offset=offs),
        dset_key="angles", timesteps=T, beta_schedule="cosine")
    torch.manual_seed(31337)
    res = sampling.sample(ref, ds2, n=2, sweep_lengths=(40, 44), batch_size=3, disable_pbar=True)
    torch.manual_seed(31337)
    res_o = ref_sampling.sample(ref, 2, (40, 44), 3, 64, T, "cosine", [True] * 6, mean_offset=offs)
    assert len(res) == len(res_o) == 8
    for a, b in zip(res, res_o):
        assert a.shape == b.shape and np.array_equal(a, b)
    np.savez_compressed(os.path.join(HERE, "ref_abs_sample.npz"), offset=offs, n=2, sweep=np.array([40, 44]), batch_size=3,
                        seed=31337, T=T, pad=64, **{f"item{i}": a for i, a in enumerate(res)})

    # ---------------------------------------- relative_key einsum cross-check
    from transformers.models.wav2vec2_bert.configuration_wav2vec2_bert import Wav2Vec2BertConfig
    from transformers.models.wav2vec2_bert.modeling_wav2vec2_bert import Wav2Vec2BertSelfAttention
    maxpos, Hh, dh, Lq = 16, 2, 32, 13
    wc = Wav2Vec2BertConfig(hidden_size=Hh * dh, num_attention_heads=Hh, position_embeddings_type="relative_key",
                            left_max_position_embeddings=maxpos - 1, right_max_position_embeddings=maxpos - 1,
                            attention_dropout=0.0)
    wsa = Wav2Vec2BertSelfAttention(wc).eval()
    ocfg_r = ref_model.OracleConfig(hidden_size=Hh * dh, num_attention_heads=Hh, intermediate_size=64,
                                    num_hidden_layers=1, max_position_embeddings=maxpos,
                                    position_embedding_type="relative_key")
    osa = ref_model.BertSelfAttention(ocfg_r).eval()
    with torch.no_grad():
        for a, b in ((wsa.linear_q, osa.query), (wsa.linear_k, osa.key), (wsa.linear_v, osa.value)):
            a.weight.copy_(torch.randn_like(a.weight) * 0.2); a.bias.copy_(torch.randn_like(a.bias) * 0.1)
            b.weight.copy_(a.weight); b.bias.copy_(a.bias)
        tbl = torch.randn(2 * maxpos - 1, dh) * 0.3
        osa.distance_embedding.weight.copy_(tbl)
        # W2V-BERT indexes by (r - l) + left; BERT 4.11.3 by (l - r) + maxpos - 1  => flipped table
        wsa.distance_embedding.weight.copy_(torch.flip(tbl, dims=(0,)))
        hs = torch.randn(3, Lq, Hh * dh)
        am = torch.zeros(3, 1, 1, Lq); am[1, ..., 9:] = -10000.0
        ctx_w = wsa(hs, attention_mask=am)[0]
        ctx_w = ctx_w  # [B, L, H*dh] before linear_out?  -> handled below
        ctx_o = osa(hs, am)
    # Wav2Vec2BertSelfAttention applies linear_out; undo by comparing pre-projection via a hook-free recompute
    with torch.no_grad():
        ctx_w_pre = torch.linalg.solve(wsa.linear_out.weight.double(), (ctx_w - wsa.linear_out.bias).double().transpose(-1, -2)).transpose(-1, -2)
    dd = (ctx_w_pre.float() - ctx_o).abs().max().item()
    print(f"[relative_key] oracle vs HF Wav2Vec2Bert attention (flipped table): max|d| = {dd:.3e}")
    assert dd < 2e-4, dd

    # ---------------- C1: mini relative_key model + REFERENCE sampler (L=64,T=10,B=4)
    mini = ref_model.OracleConfig(hidden_size=192, num_attention_heads=6, intermediate_size=384,
                                  num_hidden_layers=6, max_position_embeddings=128,
                                  position_embedding_type="relative_key")
    m32 = ref_model.synthetic_model(mini, seed=0)
    digest = weight_digest(m32.state_dict())
    T1, B1, L1 = 10, 4, 64
    lens1 = [64, 64, 64, 64]
    betas1 = beta_schedules.get_variance_schedule("cosine", T1)
    torch.manual_seed(7344)
    x01 = ds.sample_noise(torch.zeros((B1, 128, 6), dtype=torch.float32))[:, :L1]
    traj1 = sampling.p_sample_loop(m32, lens1, x01, T1, betas1, is_angle=[True] * 6, disable_pbar=True)
    torch.manual_seed(7344)
    _ = torch.randn(B1, 128, 6)
    sn1 = torch.zeros(T1, B1, L1, 6)
    for i in reversed(range(1, T1)):
        sn1[i] = torch.randn(B1, L1, 6)
    assert torch.equal(traj1, ref_sampling.p_sample_loop(m32, lens1, x01, T1, betas1, [True] * 6, step_noise=sn1))
    # single forwards (fp32 oracle + fp64 truth sharing the fp32 time table)
    m64 = ref_model.synthetic_model(mini, seed=0).double()
    m64.time_table = m32.time_embed(torch.arange(T1)).double()
    lens_r = [64, 50, 33, 1]
    mask_r = torch.zeros(B1, L1)
    for i, n in enumerate(lens_r):
        mask_r[i, :n] = 1.0
    fwd = {}
    for tval in (0, 4, 9):
        t = torch.full((B1,), tval, dtype=torch.long)
        fwd[f"eps32_t{tval}"] = m32(x01, t, attention_mask=mask_r).numpy()
        fwd[f"eps64_t{tval}"] = m64(x01.double(), t, attention_mask=mask_r.double()).numpy()
        print(f"[c1 forward] t={tval} fp32-vs-fp64 max|d| = {np.abs(fwd[f'eps32_t{tval}'] - fwd[f'eps64_t{tval}']).max():.3e}")
    np.savez_compressed(os.path.join(HERE, "c1_relkey.npz"), weight_seed=0, weight_sha256=digest, x0=x01.numpy(),
                        lens=np.array(lens1), step_noise=sn1.numpy(), traj=traj1.numpy(), T=T1,
                        lens_ragged=np.array(lens_r), **fwd)
    # ------------------------------------------------ NeRF: reference NERFBuilder
    from foldingdiff import nerf as ref_nerf_mod
    from oracle import ref_nerf
    rng = np.random.default_rng(20240924)
    nerf_out = {}
    names9 = ["0C:1N", "N:CA", "CA:C", "phi", "psi", "omega", "tau", "CA:C:1N", "C:1N:1CA"]  # "canonical"
    for tag, names in (("full", names9[3:]), ("minimal", names9[3:7]), ("canonical", names9)):
        for Ln in (1, 2, 37, 128):
            cols = {}
            for nme in names:
                if nme in ("phi", "psi", "omega"):
                    cols[nme] = rng.uniform(-np.pi, np.pi, Ln).astype(np.float32)
                elif ":" in nme and nme.count(":") == 1:
                    cols[nme] = rng.uniform(1.3, 1.6, Ln).astype(np.float32)
                else:
                    cols[nme] = rng.uniform(1.8, 2.2, Ln).astype(np.float32)
            # the keyword mapping of create_new_chain_nerf (angles_and_coords.py:142-173)
            kw = dict(phi_dihedrals=cols["phi"], psi_dihedrals=cols["psi"], omega_dihedrals=cols["omega"])
            if "tau" in cols: kw["bond_angle_ca_c"] = cols["tau"]
            if "CA:C:1N" in cols: kw["bond_angle_c_n"] = cols["CA:C:1N"]
            if "C:1N:1CA" in cols: kw["bond_angle_n_ca"] = cols["C:1N:1CA"]
            if "0C:1N" in cols: kw["bond_len_c_n"] = cols["0C:1N"]
            if "N:CA" in cols: kw["bond_len_n_ca"] = cols["N:CA"]
            if "CA:C" in cols: kw["bond_len_ca_c"] = cols["CA:C"]
            if Ln == 1:  # NERFBuilder squeezes length-1 arrays to 0-d; the seed atoms are the whole answer
                raw = np.array([ref_nerf_mod.N_INIT, ref_nerf_mod.CA_INIT, ref_nerf_mod.C_INIT])
                cen = raw - raw.mean(axis=0)
            else:
                bld = ref_nerf_mod.NERFBuilder(**kw)
                raw, cen = np.asarray(bld.cartesian_coords), np.asarray(bld.centered_cartesian_coords)
                mine = ref_nerf.build(cols["phi"], cols["psi"], cols["omega"], cols.get("tau"), cols.get("CA:C:1N"),
                                      cols.get("C:1N:1CA"), center=False,
                                      len_c_n=cols.get("0C:1N"), len_n_ca=cols.get("N:CA"), len_ca_c=cols.get("CA:C"))
                assert np.array_equal(mine, raw), (tag, Ln, np.abs(mine - raw).max())
            feats = np.stack([cols[n] for n in names], axis=1)
            nerf_out[f"{tag}_{Ln}_feats"] = feats
            nerf_out[f"{tag}_{Ln}_raw"] = raw
            nerf_out[f"{tag}_{Ln}_centered"] = cen
    nerf_out["names_full"] = np.array(names9[3:]); nerf_out["names_minimal"] = np.array(names9[3:7]); nerf_out["names_canonical"] = np.array(names9)
    np.savez_compressed(os.path.join(HERE, "ref_nerf.npz"), **nerf_out)
    print("weights sha256", digest)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f"{f}: {os.path.getsize(os.path.join(HERE, f)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
