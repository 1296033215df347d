"""
Host-side logic of the product package (no GPU): tables, wrap, dataset shells,
model construction / from_dir / state_dict plumbing, argument validation.  The
product's host tables must be bit-identical to the reference's (golden fixtures).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import golden
from foldingdiff_amd import beta_schedules, datasets, modelling, sampling, utils


def test_schedules_match_reference_bitwise():
    g = golden("ref_schedules.npz")
    for kind in ("cosine", "linear", "quadratic"):
        for T in (10, 250, 1000):
            terms = beta_schedules.compute_alphas(beta_schedules.get_variance_schedule(kind, T))
            for k, v in terms.items():
                assert np.array_equal(v.numpy(), g[f"{kind}_{T}_{k}"]), (kind, T, k)
    with pytest.raises(ValueError):
        beta_schedules.get_variance_schedule("nope", 10)


def test_step_coefficients():
    g = golden("ref_schedules.npz")
    c = beta_schedules.step_coefficients(beta_schedules.cosine_beta_schedule(1000)).numpy()
    assert c.shape == (4, 1000) and c.dtype == np.float32
    assert np.array_equal(c[0], (1.0 / torch.sqrt(torch.from_numpy(g["cosine_1000_alphas"]))).numpy())
    assert np.array_equal(c[1], g["cosine_1000_betas"])
    assert np.array_equal(c[2], g["cosine_1000_sqrt_one_minus_alphas_cumprod"])
    # torch.sqrt (what p_sample calls, sampling.py:75) is not numpy's correctly-rounded sqrt on every input
    assert np.array_equal(c[3], torch.sqrt(torch.from_numpy(g["cosine_1000_posterior_variance"])).numpy())
    assert c[3, 0] == 0.0  # t = 0 adds no noise


def test_wrap_kats():
    g = golden("ref_wrap.npz")
    assert np.array_equal(utils.modulo_with_wrapped_range(torch.from_numpy(g["v"]), -np.pi, np.pi).numpy(), g["w"])
    for (a, lo, hi), want in zip(g["kat_in"], g["kat_out"]):
        assert utils.modulo_with_wrapped_range(a, lo, hi) == want
    assert np.allclose(utils.modulo_with_wrapped_range(np.array([2, -2]), -2, 2), [-2, -2])
    with pytest.raises(AssertionError):
        utils.modulo_with_wrapped_range(1.0, 1.0, 2.0)


def test_sample_noise_matches_reference():
    g = golden("ref_noise.npz")
    ds = datasets.NoisedAnglesDataset(datasets.AnglesEmptyDataset("canonical-full-angles", pad=128),
                                      timesteps=10, beta_schedule="cosine")
    torch.manual_seed(7344)
    assert np.array_equal(ds.sample_noise(torch.zeros(3, 128, 6)).numpy(), g["full_seed7344"])
    ds2 = datasets.NoisedAnglesDataset(datasets.AnglesEmptyDataset("canonical", pad=32), timesteps=10,
                                       beta_schedule="linear", angular_variance=0.5)
    torch.manual_seed(99)
    assert np.array_equal(ds2.sample_noise(torch.zeros(2, 32, 9)).numpy(), g["mixed_seed99_var05"])
    assert ds.pad == 128 and ds.timesteps == 10 and ds.feature_is_angular["angles"] == [True] * 6
    with pytest.raises(NotImplementedError):
        ds.dset.get_masked_means()


def test_time_tables_match_reference():
    g = golden("ref_time_embed.npz")
    assert np.array_equal(modelling.gaussian_fourier_table(torch.from_numpy(g["W"]), 1000).numpy(), g["gaussian_fourier"])
    assert np.array_equal(modelling.sinusoidal_table(64, 1000).numpy(), g["sinusoidal"])


def _mini_cfg(**kw):
    base = dict(hidden_size=64, num_attention_heads=2, intermediate_size=128, num_hidden_layers=2,
                max_position_embeddings=64, position_embedding_type="relative_key")
    base.update(kw)
    return modelling.BertConfig(**base)


def test_model_state_dict_names_match_oracle():
    from oracle import ref_model
    for pos in ("relative_key", "absolute"):
        for dec in ("mlp", "linear"):
            m = modelling.BertForDiffusionBase(_mini_cfg(position_embedding_type=pos), [True] * 6, decoder=dec)
            o = ref_model.OracleBertForDiffusion(
                ref_model.OracleConfig(hidden_size=64, num_attention_heads=2, intermediate_size=128,
                                       num_hidden_layers=2, max_position_embeddings=64, position_embedding_type=pos),
                [True] * 6, decoder=dec)
            assert set(m.state_dict().keys()) == set(o.state_dict().keys())
            for k, v in o.state_dict().items():
                assert tuple(m.state_dict()[k].shape) == tuple(v.shape), k
            m.load_state_dict(o.state_dict())  # strict
            with pytest.raises(RuntimeError):
                bad = dict(o.state_dict())
                bad.pop("inputs_to_hidden_dim.bias")
                m.load_state_dict(bad)


def test_from_dir_roundtrip(tmp_path):
    """Reference directory layout: training_args.json, config.json,
    models/best_by_valid/epoch=N-step=M.ckpt with a 'state_dict' key (modelling.py:297-382)."""
    d = tmp_path / "results"
    (d / "models" / "best_by_valid").mkdir(parents=True)
    cfg = _mini_cfg()
    cfg.save_pretrained(d)
    with open(d / "training_args.json", "w") as fh:
        json.dump({"angles_definitions": "canonical-full-angles", "time_encoding": "gaussian_fourier",
                   "decoder": "mlp", "max_seq_len": 64, "timesteps": 10, "variance_schedule": "cosine",
                   "variance_scale": 1.0}, fh)
    src = modelling.BertForDiffusionBase(cfg, [True] * 6)
    for epoch in (3, 12):  # latest epoch must win, idx=-1
        sd = {k: v + epoch for k, v in src.state_dict().items()}
        torch.save({"state_dict": sd, "epoch": epoch}, d / "models" / "best_by_valid" / f"epoch={epoch}-step={epoch * 10}.ckpt")
    m = modelling.BertForDiffusionBase.from_dir(str(d), copy_to=str(tmp_path / "snap"))
    assert m.n_inputs == 6 and m.ft_is_angular == [True] * 6
    assert torch.equal(m.state_dict()["inputs_to_hidden_dim.bias"], src.state_dict()["inputs_to_hidden_dim.bias"] + 12)
    assert os.path.isfile(tmp_path / "snap" / "models" / "best_by_valid" / "epoch=12-step=120.ckpt")
    assert os.path.isfile(tmp_path / "snap" / "config.json") and os.path.isfile(tmp_path / "snap" / "training_args.json")
    m0 = modelling.BertForDiffusionBase.from_dir(str(d), idx=0)
    assert torch.equal(m0.state_dict()["inputs_to_hidden_dim.bias"], src.state_dict()["inputs_to_hidden_dim.bias"] + 3)
    m2 = modelling.BertForDiffusionBase.from_dir(str(tmp_path / "snap"))
    assert torch.equal(m2.state_dict()["time_embed.W"], m.state_dict()["time_embed.W"])
    ds = datasets.AnglesEmptyDataset.from_dir(str(d))
    assert ds.pad == 64


def test_no_cpu_compute_path():
    m = modelling.BertForDiffusionBase(_mini_cfg(), [True] * 6)
    assert next(m.parameters()).device.type == "cpu"
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 8, 6), torch.zeros(1, dtype=torch.long), attention_mask=torch.ones(1, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sampling.p_sample_loop(m, [8], torch.zeros(1, 8, 6), 10, beta_schedules.cosine_beta_schedule(10), [True] * 6)


def test_argument_validation():
    m = modelling.BertForDiffusionBase(_mini_cfg(), [True] * 6)
    ds = datasets.NoisedAnglesDataset(datasets.AnglesEmptyDataset("canonical-full-angles", pad=64), timesteps=10,
                                      beta_schedule="cosine")
    with pytest.raises(ValueError, match="must be less than"):
        sampling.sample(m, ds, n=1, sweep_lengths=(50, 50))
    # masks that are no prefix (or have an all-zero row) are not an error any more: they select the general path (fd_forward_ex)
    assert modelling.BertForDiffusionBase.lengths_from_mask(torch.tensor([[1.0, 0.0, 1.0]])) is None
    assert modelling.BertForDiffusionBase.lengths_from_mask(torch.tensor([[0.0, 0.0, 0.0], [1, 1, 1]])) is None
    assert modelling.BertForDiffusionBase.lengths_from_mask(torch.tensor([[1.0, 1.0, 0.0], [1, 1, 1]])).tolist() == [2, 3]
    with pytest.raises(ValueError):
        modelling.BertForDiffusionBase(_mini_cfg(), [True] * 6, decoder="cnn")
    with pytest.raises(ValueError):
        modelling.BertForDiffusionBase(_mini_cfg(), [True] * 6, time_encoding="learned")
    with pytest.raises(AssertionError):
        sampling.p_sample(m, torch.zeros(2, 8, 6), torch.tensor([1, 2]), [8, 8], 0, beta_schedules.cosine_beta_schedule(10))
    assert not utils.is_huggingface_hub_id(os.getcwd())
    assert utils.is_huggingface_hub_id("wukevin/foldingdiff_cath")


def test_step_noise_draw_order_matches_reference():
    """_draw_step_noise must replicate the reference's sequence of randn_like draws
    (golden: recovered from the reference run in make_golden.py)."""
    g = golden("ref_abs_traj.npz")
    torch.manual_seed(2024)
    _ = torch.randn(4, 128, 6)  # the initial sample_noise draw
    z = sampling._draw_step_noise(int(g["T"]), (4, 48, 6))
    assert np.array_equal(z, g["step_noise"])


def test_streamed_step_noise_is_the_reference_draw_order():
    """_StepNoise (the chunked producer behind the default sampling path) hands out exactly the draws of
    _draw_step_noise -- whole batch, a rank's row slice, any chunking -- and leaves the generator where a
    single-process run leaves it."""
    shape = (5, 7, 6)
    torch.manual_seed(31)
    want = sampling._draw_step_noise(9, shape)
    after = torch.rand(1).item()
    torch.manual_seed(31)
    assert np.array_equal(sampling._StepNoise(8, shape).materialize(), want)
    assert torch.rand(1).item() == after
    torch.manual_seed(31)
    assert np.array_equal(sampling._StepNoise(8, shape, (2, 4)).materialize(), want[:, 2:4])
    assert torch.rand(1).item() == after                      # a shard consumes the whole batch's stream
    torch.manual_seed(31)
    assert sampling._StepNoise(8, shape, (3, 3)).materialize().shape == (9, 0, 7, 6)
    assert torch.rand(1).item() == after                      # and so does an empty shard
    for n in (1, 2, 4, 9, 32):
        torch.manual_seed(31)
        st = sampling._StepNoise(8, shape)
        got = np.full((9,) + shape, np.nan, np.float32)
        seen = []
        for lo, hi in st.chunks(n):
            buf = torch.empty((min(n, 9),) + shape)
            st.fill(lo, hi, buf)
            got[lo:hi + 1] = buf[: hi - lo + 1].numpy()
            seen.append((lo, hi))
        assert seen[0][1] == 8 and seen[-1][0] == 0 and all(a[0] == b[1] + 1 for a, b in zip(seen, seen[1:]))
        assert np.array_equal(got, want), n


def test_gelu_rational_erf_accuracy():
    """The fp16x3 GEMM epilogue evaluates erf as a rational function (csrc/img_common.h:erf_rational,
    coefficients restated here): it must stay in libm-erff's error class against float64 erf, and the
    GELU built on it within torch's own fp32 GELU error (the exact-erf "gelu" of modelling.py:195-196)."""
    import math
    import re
    f = np.float32
    src = open(os.path.join(os.path.dirname(__file__), "..", "foldingdiff_amd", "csrc", "img_common.h")).read()
    body = src[src.index("float erf_rational(float x)"):src.index("float gelu_erf(float x)")]
    consts = [f(c) for c in re.findall(r"(-?\d\.\d+e-\d+)f", body)]
    assert len(consts) == 12, consts            # 7 numerator + 5 denominator coefficients, in evaluation order
    alpha, beta = consts[:7], consts[7:]

    def erf32(x):
        x = np.clip(x.astype(f), f(-4), f(4))
        x2 = x * x
        p = np.full_like(x, alpha[0])
        for a in alpha[1:]:
            p = p * x2 + a
        q = np.full_like(x, beta[0])
        for b in beta[1:]:
            q = q * x2 + b
        return ((p * x) / q).astype(f)

    rng = np.random.default_rng(0)
    xs = np.concatenate([np.linspace(-6, 6, 600001), rng.standard_normal(300000) * 1.5,
                         np.logspace(-8, 0, 50000)]).astype(f)
    erf64 = np.vectorize(math.erf)
    err = np.abs(erf32(xs).astype(np.float64) - erf64(xs.astype(np.float64))).max()
    assert err <= 5e-7, err
    gelu64 = 0.5 * xs.astype(np.float64) * (1 + erf64(xs.astype(np.float64) / math.sqrt(2)))
    h = f(0.5) * xs
    got = (h * erf32(xs * f(0.70710678118654752440)) + h).astype(np.float64)
    torch_err = np.abs(torch.nn.functional.gelu(torch.from_numpy(xs)).double().numpy() - gelu64).max()
    assert np.abs(got - gelu64).max() <= max(2e-6, 1.5 * torch_err)


class _ToyAngles:
    """The synthetic dataset of tests/golden/make_golden_noising.py, rebuilt from the fixture."""
    feature_names = {"angles": ["phi", "psi", "omega", "tau", "CA:C:1N", "C:1N:1CA"]}
    feature_is_angular = {"angles": [True] * 6}

    def __init__(self, g):
        self.angles, self.lengths, self.pad = torch.from_numpy(g["angles"]), g["lengths"].tolist(), int(g["pad"])
        self.filenames = [f"toy_{i}.pdb" for i in range(len(self.lengths))]

    def __len__(self):
        return len(self.lengths)

    def __getitem__(self, index, ignore_zero_center=False):
        l = self.lengths[index]
        mask = torch.zeros(self.pad)
        mask[:l] = 1.0
        return {"angles": self.angles[index].clone(), "attn_mask": mask, "position_ids": torch.arange(self.pad),
                "lengths": torch.tensor(l, dtype=torch.int64)}


def test_forward_noising_matches_reference_bitwise():
    """NoisedAnglesDataset.__getitem__ (q(x_t | x_0)) against the reference's own class under the
    same torch seed: corrupted values, the noise and t are bit-identical (tests/golden/ref_reconstruct.npz)."""
    g = golden("ref_reconstruct.npz")
    ds = datasets.NoisedAnglesDataset(_ToyAngles(g), dset_key="angles", timesteps=int(g["T"]), beta_schedule="cosine")
    assert len(ds) == 5 and ds.filenames[3] == "toy_3.pdb"
    torch.manual_seed(int(g["seed"]))
    for i in range(5):
        it = ds.__getitem__(i, use_t_val=int(g["noise_timesteps"]))
        assert np.array_equal(it["corrupted"].numpy(), g[f"corrupted{i}"])
        assert np.array_equal(it["known_noise"].numpy(), g[f"known_noise{i}"])
        assert np.array_equal(it["t"].numpy(), g[f"t{i}"])
        assert set(it) >= {"angles", "attn_mask", "position_ids", "lengths", "sqrt_alphas_cumprod_t",
                           "sqrt_one_minus_alphas_cumprod_t"}
    # the reference's denoising loop draws its step noise here; skip the same number of values
    nt, B = int(g["noise_timesteps"]), 5
    for _ in range(nt - 1):
        torch.randn(B, int(g["pad"]), 6)
    it = ds[2]
    assert np.array_equal(it["t"].numpy(), g["rand_t"])
    assert np.array_equal(it["corrupted"].numpy(), g["rand_corrupted"])
    with pytest.raises(NotImplementedError):
        sampling.get_reconstruction_error(None, ds)


def test_pdb_writer_fixed_columns(tmp_path):
    """write_coords_to_pdb (foldingdiff/angles_and_coords.py:187-253): PDB ATOM records in the format
    specification's fixed columns, GLY backbone atoms N / CA / C on chain A, round trip within 5e-4 A."""
    from foldingdiff_amd import angles_and_coords as ac
    rng = np.random.default_rng(3)
    coords = rng.standard_normal((3 * 130, 3)) * 40
    f = ac.write_coords_to_pdb(coords, str(tmp_path / "x.pdb"))
    lines = open(f).read().splitlines()
    assert len(lines) == 390 + 2 * 129 and all(len(l) == 80 for l in lines[:390])
    assert lines[390] == "CONECT    3    4" and lines[391] == "CONECT    4    3" and lines[-1] == "CONECT  388  387"
    l = lines[4]  # second residue, CA
    assert l[:6] == "ATOM  " and int(l[6:11]) == 5 and l[12:16] == " CA " and l[17:20] == "GLY" and l[21] == "A"
    assert int(l[22:26]) == 2 and l[54:60] == "  1.00" and l[60:66] == "  5.00" and l[76:78] == " C"
    assert lines[3][12:16] == " N  " and lines[3][76:78] == " N" and lines[389][22:26] == " 130"
    assert np.abs(ac.read_pdb_backbone(f) - coords).max() <= 5.001e-4
    with pytest.raises(AssertionError):
        ac.write_coords_to_pdb(coords[:4], str(tmp_path / "y.pdb"))
    # byte layout pinned against a file written by the reference's own write_coords_to_pdb through biotite
    # (tests/golden/make_golden_pdb.py): same coordinates in -> the same bytes out, ATOM and CONECT records
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_written_backbone.pdb")
    g = ac.write_coords_to_pdb(ac.read_pdb_backbone(gold), str(tmp_path / "g.pdb"))
    assert open(g, "rb").read() == open(gold, "rb").read()
    # column selection rules of create_new_chain_nerf (no device needed until the build)
    vals, keep = ac._select(np.zeros((3, 4), np.float32), ["phi", "psi", "omega", "0C:1N"], None, None)
    assert keep == ["phi", "psi", "omega", "0C:1N"]
    with pytest.raises(ValueError):
        ac._select(np.zeros((3, 4), np.float32), ["phi", "psi", "omega", "chi1"], None, None)
    with pytest.raises(AssertionError):
        ac._select(np.zeros((3, 2), np.float32), ["phi", "psi"], None, None)


def test_f16x3_split_arithmetic_is_fp32_class():
    """The default contraction arithmetic restated in numpy (csrc/gemm_img.hip header, csrc/img_common.h: x*s = hi + lo
    with hi = fp16(x*s), lo = fp16(x*s - hi); a product is a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, exact in fp32, summed in
    fp32): against float64 it must sit in the error class of an fp32 GEMM for the shapes and value ranges of the path,
    an operand must be represented to 2^-22 relative, and the power-of-two scales of api.hip (bound * s <= 30000)
    must keep every hi finite.  No GPU: this pins the CLAIM; tests/test_gpu_parity.py measures the kernel."""
    rng = np.random.default_rng(0)
    f16, f32, f64 = np.float16, np.float32, np.float64

    def scale_for(bound):                       # api.hip: scale_for
        return f32(2.0 ** np.clip(np.floor(np.log2(30000.0 / bound)), -40, 40))

    def split(x, s):
        xs = (x.astype(f32) * s).astype(f32)
        hi = xs.astype(f16)
        lo = (xs - hi.astype(f32)).astype(f16)
        return hi, lo

    def gemm_split(a, w, sa, sw):
        ah, al = split(a, sa)
        wh, wl = split(w, sw)
        acc = np.zeros((a.shape[0], w.shape[0]), f32)
        for k0 in range(0, a.shape[1], 16):     # one v_mfma_f32_32x32x16_f16 step: products exact, fp32 accumulate
            sl = slice(k0, k0 + 16)
            for x, y in ((ah, wh), (ah, wl), (al, wh)):
                acc = (acc + (x[:, sl].astype(f64) @ y[:, sl].astype(f64).T).astype(f32)).astype(f32)
        return acc * f32(1.0 / (sa * sw))

    for K, act_scale, w_std in ((384, 1.0, 0.02), (768, 0.6, 0.02), (384, 4.0, 0.3)):
        a = (rng.standard_normal((64, K)) * act_scale).astype(f32)
        a[3, 5] = 19.0 * act_scale              # LayerNorm outputs reach a large multiple of their typical size
        w = (rng.standard_normal((96, K)) * w_std).astype(f32)
        sa = scale_for(np.abs(a).max() * 1.5)   # any valid bound works; the real one is gamma sqrt(d) + beta
        sw = f32(2.0 ** np.floor(np.log2(16383.0 / np.abs(w).max())))   # api.hip: max|w| * scale in [8192, 16384)
        hi, lo = split(a, sa)
        assert np.isfinite(hi.astype(f32)).all() and np.abs(hi.astype(f32)).max() <= 30000.0 * 1.001
        ref = a.astype(f64) @ w.astype(f64).T
        got = gemm_split(a, w, sa, sw)
        plain = a @ w.T                          # numpy's fp32 GEMM (pairwise / blocked fp32 accumulation)
        rms = np.sqrt(np.mean(ref ** 2))
        err_split = np.sqrt(np.mean((got - ref) ** 2)) / rms
        err_plain = np.sqrt(np.mean((plain.astype(f64) - ref) ** 2)) / rms
        assert err_split <= 4e-7, (K, err_split)                       # measured 1.8e-7 .. 2.4e-7 (GPU kernel: 2.4e-7)
        assert err_split <= 1.5 * err_plain, (K, err_split, err_plain)  # numpy's fp32 GEMM: 3.5e-7
        # representation error of one operand: hi + lo carry 22 bits, rounded to nearest
        back = (hi.astype(f64) + lo.astype(f64)) / f64(sa)
        big = np.abs(a) > np.abs(a).max() * 2.0 ** -10
        assert (np.abs(back - a)[big] / np.abs(a)[big]).max() <= 2.0 ** -22


def test_vt_swizzle_is_conflict_free_for_the_instruction_emitted():
    """The V^T rows' unit swizzle (img_common.h: vt_swz) against the LDS models the counters support.  Lane l31 reads the 8-byte
    unit (ua ^ vt_swz(l31)) of its own 128-byte row.  A lone ds_read_b64 (what the attention kernel emits since round 4) is served
    in 32-lane groups against 64 banks: rows of one parity share 32 banks, so the 16 even and the 16 odd rows of a group must each
    name 16 different units -- (d >> 1) & 15 does, d & 15 names 8 (SQ_LDS_BANK_CONFLICT 1.57 M per launch,
    profiles/r04_attention_lds_conflicts.log).  ds_read2st64_b64 (rounds 2-3: two key tiles' fetches merged) is served in 16-lane
    groups against 32 banks, where it is the other way round (profiles/r03_attention_notes.log)."""
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "foldingdiff_amd", "csrc", "img_common.h")).read()
    m = re.search(r"#else\s*\n__device__ __forceinline__ constexpr int vt_swz\(int d\) \{ return (.*?); \}", src)
    assert m, "vt_swz not found"
    swz = eval("lambda d: " + m.group(1))  # noqa: S307 -- an integer expression of d from our own header
    r3 = lambda d: d & 15  # noqa: E731
    for ua in range(16):
        for parity in range(2):      # ds_read_b64: 32-lane groups, rows of one parity share a bank half
            rows = range(parity, 32, 2)
            assert len({ua ^ swz(d) for d in rows}) == 16
            assert len({ua ^ r3(d) for d in rows}) == 8
        for g in range(2):           # ds_read2st64_b64: 16-lane groups against 32 banks
            lanes = range(16 * g, 16 * g + 16)
            assert len({ua ^ r3(d) for d in lanes}) == 16
            assert len({ua ^ swz(d) for d in lanes}) == 8
    assert sorted(swz(d) for d in range(0, 32, 2)) == list(range(16))


def test_relative_key_band_skew_mapping():
    """The attention kernel's relative_key skew, restated (foldingdiff_amd/csrc/attention_img.hip: op_W / op_G): band tile q is
    computed as R^T = E Q^T with MFMA row i holding band row pi(i), written with ds_write_addtid_b32 (register r of lane L at
    slot + 256 r + 4 L) into scratch slot q & 1, and S^T tile T-1-q gathers ONE dword per score from a lane base plus an
    immediate offset.  Checks, for every pair, lane and register, that the gathered dword is band value (32 q + l - kl + 31, l),
    that every address stays inside the wave's 8 KiB, and that the gathers are LDS bank-conflict free."""
    def pi(i):
        return (i & 24) | ((i & 3) << 1) | ((i >> 2) & 1)

    assert sorted(pi(i) for i in range(32)) == list(range(32))
    scratch = np.full(2048, np.nan)

    def write_tile(q):
        for r in range(16):
            for lane in range(64):
                j, h = lane & 31, lane >> 5
                i = 8 * (r >> 2) + 4 * h + (r & 3)                      # MFMA C/D layout: row of register r in half-wave h
                scratch[((q & 1) * 4096 + r * 256 + lane * 4) // 4] = (32 * q + pi(i)) * 1000 + j

    def gather_pair(q):
        for r in range(16):
            klr = (r & 3) + 8 * (r >> 2)
            for h in (0, 1):
                banks = set()
                for l31 in range(32):
                    kl = klr + 4 * h
                    gb = (l31 - 4 * h + 4) * 128 + 4 * l31
                    base = gb if q % 2 == 0 else (gb - 4096 if l31 > kl else gb + 4096)
                    addr = base + (27 - klr) * 128
                    assert 0 <= addr < 8192
                    assert scratch[addr // 4] == (32 * q + l31 - kl + 31) * 1000 + l31, (q, r, h, l31)
                    banks.add((addr // 4) % 32)
                assert len(banks) == 32                                   # ds_read_b32: one LDS cycle per 32-lane group

    for T in (1, 2, 3, 4):  # M0 M1 W0 W1 M2 | G0 W2 M3 | G1 W3 M4 | G2 W4 | G3   (tiles beyond T do not exist)
        scratch[:] = np.nan
        write_tile(0)
        write_tile(1)
        for q in range(T):
            gather_pair(q)
            if q + 2 <= T:
                write_tile(q + 2)


def test_relative_key_table_rows_in_lds():
    """ELDS layout of the distance table (attention_img.hip): LDS row rho holds table row clamp(rho - esh); a wave's band row
    x of band tile q is LDS row maxpos - LP + esh + 32 wq + 32 q + x.  Every row an ACTIVE wave touches exists (0..255), and
    every (query, key) pair of real positions reads the table row HF BertSelfAttention names: l - r + maxpos - 1."""
    for maxpos in (33, 40, 64, 100, 128):
        for L in range(1, maxpos + 1):
            T = (L + 31) // 32
            LP = 32 * T
            esh = max(0, LP - maxpos)
            for wq in range(T):
                rho0 = maxpos - LP + esh + 32 * wq
                assert rho0 >= 0 and rho0 + 32 * T + 31 <= 255
                for t in range(T):
                    q = T - 1 - t
                    for ql in (0, 13, 31):
                        for kl in (0, 7, 31):
                            l, r = 32 * wq + ql, 32 * t + kl
                            if l >= L or r >= L:
                                continue
                            rho = rho0 + 32 * q + (ql - kl + 31)
                            row = min(max(rho - esh, 0), 2 * maxpos - 2)
                            assert row == l - r + maxpos - 1


def test_gemm_tile_list_with_tail_slices():
    """The GEMM kernel's tile list, restated (foldingdiff_amd/csrc/gemm_img.hip: XCD-aware deal + row slices for an XCD's
    incomplete last round): whatever the tile count and grid, every 32-row block of every (row panel, column tile) is
    computed exactly once, a workgroup's slice is the last tile of its stream, an XCD never hands out more tail items than it has
    workgroups, and the whole-tile loader loop never meets a slice stage."""
    BM = 128

    def tile_list(ntiles, grid, tail=True):
        per = grid // 8
        work = []
        for blk in range(grid):
            xcd, jx = blk & 7, blk >> 3
            tlo, thi = ntiles * xcd // 8, ntiles * (xcd + 1) // 8
            nx = thi - tlo
            nfull, nrem = nx // per, nx % per
            tsplit = (4 if 4 * nrem <= per else 2 if 2 * nrem <= per else 1) if (tail and nrem > 0) else 1
            assert nrem * tsplit <= per
            has_tail = jx < nrem * tsplit
            stream = [(tlo + jx + i * per, 0, BM) for i in range(nfull)]
            if has_tail:
                stream.append((tlo + nfull * per + jx // tsplit, (jx % tsplit) * (BM // tsplit), BM // tsplit))
            work.append((stream, nfull, has_tail and tsplit > 1))
        return work

    for ntiles, grid in [(1, 8), (2, 8), (8, 8), (9, 16), (30, 32), (246, 256), (266, 256), (268, 256), (315, 256), (416, 256),
                         (512, 256), (630, 256), (804, 256), (945, 256), (1536, 256), (100, 304 // 8 * 8)]:
        for tail in (True, False):
            seen = {}
            for stream, nfull, sliced in tile_list(ntiles, grid, tail):
                for k, (tile, r0, rows) in enumerate(stream):
                    assert 0 <= tile < ntiles and rows in (128, 64, 32)
                    assert rows == BM or k == len(stream) - 1          # a slice ends its workgroup's stream
                    for blk in range(r0 // 32, (r0 + rows) // 32):
                        assert (tile, blk) not in seen
                        seen[(tile, blk)] = 1
                nk = 12
                G = len(stream) * nk
                g_main = max(nfull * nk - 2, 0) if sliced else G         # iteration g issues A(g + 2)
                assert all((g + 2) // nk < nfull or not sliced for g in range(g_main))
            assert len(seen) == 4 * ntiles, (ntiles, grid, tail, len(seen))


def test_weight_stationary_gemm_deals_every_group_to_every_slice_exactly_once():
    """Index arithmetic of ws::gemm_ws_kernel (gemm_ws.hip), restated: workgroup b of 256 sits on XCD b % 8; an XCD's 32 workgroups
    form S = 32 // nslice streams of nslice neighbours (workgroups beyond S * nslice idle); the XCD owns groups [NT x / 8, NT (x+1) / 8),
    stream j takes every S-th of them; wave w of slice s owns column block 8 s + w.  Every (32-row group, 32-column block) of the
    output must be produced exactly once, whatever the row count and for column counts that leave a slice partly empty."""
    grid = 256
    for N in (96, 128, 384, 768, 1152, 1280, 3072):
        nb, nslice, per = N // 32, -(-N // 256), grid // 8
        S = per // nslice
        assert S >= 1
        for NT in (1, 4, 7, 31, 64, 257, 984, 2048):   # rows / 32 (rows128 capacity: always a multiple of 4, odd ones for the arithmetic)
            seen = {}
            idle = 0
            for b in range(grid):
                xcd, jx = b & 7, b >> 3
                if jx >= S * nslice:
                    idle += 1
                    continue
                sl, j = jx % nslice, jx // nslice
                tlo, thi = NT * xcd // 8, NT * (xcd + 1) // 8
                cnt = (thi - tlo - j + S - 1) // S if tlo + j < thi else 0
                for i in range(cnt):
                    g = tlo + j + i * S
                    assert tlo <= g < thi
                    for w in range(8):
                        cb = sl * 8 + w
                        if cb < nb:
                            assert (g, cb) not in seen, (N, NT, g, cb)
                            seen[(g, cb)] = b
            assert len(seen) == NT * nb, (N, NT, len(seen))
            assert idle == 8 * (per - S * nslice)


def test_shift_trim_host_path_is_the_reference_arithmetic():
    """N2 on host tensors (sampling._shift_trim's numpy branch: the CPU stand-in of the multi-process tests; the device branch is
    fd_shift_trim_dev, GPU-tested against the same statements): /root/reference/foldingdiff/sampling.py:200-222 restated literally --
    trim every item to its length, `s + offset` in numpy's promoted dtype, modulo_with_wrapped_range on the angular columns."""
    rng = np.random.RandomState(5)
    rows, B, L, F = 3, 5, 17, 6
    traj = torch.from_numpy((rng.randn(rows, B, L, F) * 2.5).astype(np.float32))
    lengths = [17, 1, 9, 16, 4]
    angular = np.array([0, 1, 2, 4])          # a non-angular column in between: bond-angle-like features stay unwrapped
    for offset in (None, (rng.randn(F) * 1.3).astype(np.float32), rng.randn(F) * 1.3):
        got = sampling._shift_trim(None, traj, lengths, offset, angular)
        want = [traj[:, i, :l, :].numpy() for i, l in enumerate(lengths)]                       # sampling.py:201-203
        if offset is not None:                                                                 # sampling.py:218-222
            want = [s + offset for s in want]
            for s in want:
                s[..., angular] = utils.modulo_with_wrapped_range(s[..., angular], range_min=-np.pi, range_max=np.pi)
        assert len(got) == B
        for g, w in zip(got, want):
            assert g.shape == w.shape and g.dtype == w.dtype, (g.dtype, w.dtype)
            assert np.array_equal(g, w)
        if offset is not None:
            assert all(np.all(g[..., angular] >= -np.pi) and np.all(g[..., angular] < np.pi) for g in got)
            assert any(np.abs(g[..., 3]).max() > np.pi for g in got)   # the other column really is left alone


@pytest.mark.parametrize("nkt,nseq", [(12, 1), (12, 2), (12, 3), (6, 1), (6, 4)])
def test_fused_attention_stream_protocol(nkt, nseq):
    """The vector-memory protocol of sa::seq_attn_kernel (seq_attn.hip), restated and simulated for one wave: the weight stream's
    source offsets (wrap test only where a head's k-tile 0 is requested), its ring slots, and the counted stage-top waits.  vmcnt
    retires in issue order, so `s_waitcnt vmcnt(n)` at the top of stage p guarantees everything but the n youngest operations; the
    kernel needs the three LDS-DMA pieces of stage p + 1 landed there (it prefetches that stage's first fragments at the end of stage
    p) and must not wait for more than that: the youngest operations are the pieces of stage p + 2, the four ctx stores of the
    attention slice that holds them, and -- while the next sequence's hidden state replaces the current one in place -- the four loads
    issued at the end of each of the two stages before.  The table restated here is the one in the kernel's item loop."""
    src = open(os.path.join(os.path.dirname(__file__), "..", "foldingdiff_amd", "csrc", "seq_attn.hip")).read()
    for stmt in ("constexpr int ST = (SPS == 1 && (kt == 3 || kt == 4)) ? 4 : 0;",          # the table simulated below IS the kernel's
                 "if (reload) FD_WAIT_VM(3 + ST + (kt == 0 ? 0 : (kt == 1 ? 4 : 8)));",
                 "else FD_WAIT_VM(3 + ST + (kt == 0 ? 8 : (kt == 1 ? 4 : 0)));",
                 "FD_WAIT_VM(3 + ST);", "FD_WAIT_VM(6);", "issue_w(IC<(kt == NKT - 3)>{});",
                 "constexpr int NST = 4;", "constexpr int KT_BYTES = 96 * 128;"):
        assert stmt in src, stmt
    H, NST, KT = nkt, 4, 96 * 128
    sps = 12 // nkt
    store_stage = 2 if sps == 1 else 1          # the stage whose attention slice stores the ctx block of the head before
    nitems = nseq * H
    total = H * nkt * KT
    q = []                                      # issue order: ("w", stage) x3 | ("st", stage) x4 | ("h", stage) x4 | ("h0",) x 4 nkt
    state = dict(w_src=0, w_slot=0, pos_req=0)
    src_of, slot_of = {}, {}

    def issue_w(wrapchk):
        if wrapchk and state["w_src"] == total:
            state["w_src"] = 0
        p = state["pos_req"]
        src_of[p], slot_of[p] = state["w_src"], state["w_slot"]
        q.extend([("w", p)] * 3)
        state["w_src"] += KT
        state["w_slot"] = 0 if state["w_slot"] + KT == NST * KT else state["w_slot"] + KT
        state["pos_req"] = p + 1

    def check_wait(n, pos, exact):
        """after vmcnt(n) at the top of stage `pos`: the pieces of stage pos + 1 have landed; `exact`: and n is not smaller than needed"""
        last = max(i for i, op in enumerate(q) if op == ("w", pos + 1))
        younger = len(q) - 1 - last
        assert n <= younger, (pos, n, younger)   # the wait covers the pieces of stage pos + 1
        if exact:
            assert n == younger, (pos, n, younger)

    q.extend([("h0",)] * (4 * nkt))
    for _ in range(3):
        issue_w(False)
    # FD_WAIT_VM(6): stage 0 (and the hidden state in front of it) landed
    assert len(q) - 1 - max(i for i, op in enumerate(q) if op == ("w", 0)) == 6
    pos = 0
    for kt in range(nkt):                       # iteration 0: the first item's projection alone
        check_wait(3, pos, exact=True)
        issue_w(kt == nkt - 3)
        pos += 1
    head, seq = 1, 0
    for it in range(1, nitems):
        reload = head == H - 1 and seq + 1 < nseq
        after_reload = head == 0
        for kt in range(nkt):
            st = 4 if (sps == 1 and kt in (3, 4)) else 0
            if reload:
                n = 3 + st + (0 if kt == 0 else (4 if kt == 1 else 8))
            elif after_reload:
                n = 3 + st + (8 if kt == 0 else (4 if kt == 1 else 0))
            else:
                n = 3 + st
            # d_model 192 (two slices per stage): the stores leave in stage 1 and the plain count waits for them too -- correct, not exact
            exact = sps == 1 or kt not in (store_stage + 1, store_stage + 2)
            check_wait(n, pos, exact=exact)
            issue_w(kt == nkt - 3)
            if kt == store_stage:
                q.extend([("st", pos)] * 4)
            if reload:
                q.extend([("h", pos)] * 4)
            pos += 1
        head += 1
        if head == H:
            head, seq = 0, seq + 1
    # every requested position is (head, k-tile) of the image in order, wrapping after the last head; a stage's slot is its position mod NST
    for p in range(state["pos_req"]):
        assert src_of[p] == (p % (H * nkt)) * KT, (p, src_of[p])
        assert slot_of[p] == (p % NST) * KT
    # a slot is only overwritten after the stage that read it: position p + NST is requested at the top of stage p + 1 (>= p + 1)
    assert state["pos_req"] == nitems * nkt + 3


def test_fused_attention_weight_image_matches_the_fragment_reads():
    """The contract between api.hip (upload_seq_attn_weights: the q|k|v weights as [head][k-tile][unit 0-7][96 rows q_h | k_h | v_h][16 B],
    a (head, k-tile) = one 12 KiB ring stage byte for byte) and seq_attn.hip's fragment reads (rd_w: lane (l31, half) reads 16 bytes at
    stage + (unit + half) * 1536 + j * 512 + l31 * 16 for accumulator j, unit = 2 c + 4 plane): every read must deliver the eight
    k-consecutive fp16 values W[j d + 32 head + l31][32 kt + 16 c + 8 half + 0..7] of the plane -- the A operand slice of
    v_mfma_f32_32x32x16_f16 for MFMA row l31.  Both formulas restated; the source lines they restate are pinned."""
    root = os.path.join(os.path.dirname(__file__), "..", "foldingdiff_amd", "csrc")
    api, sa = open(os.path.join(root, "api.hip")).read(), open(os.path.join(root, "seq_attn.hip")).read()
    for stmt in ("const int n = (r / 32) * d + h * 32 + (r % 32);", "memcpy(stage + ((size_t)u * 96 + r) * 8, blk + u * 8, 16);"):
        assert stmt in api, stmt
    for stmt in ("const unsigned w_rd = a_Wr + (unsigned)(l31 * 16 + half * 1536);", "lds_u128(wb + (unsigned)(unit * 1536 + j * 512))",
                 "rd_w(Xw, 0);", "rd_w(Yw, 4);", "rd_w(Xw, 2);", "rd_w(Yw, 6);"):
        assert stmt in sa, stmt
    for d in (384, 192):
        nk = H = d // 32
        # the row-major split image (pack_split_weight): per (weight row n, k-tile) 64 halves = units 0-3 hi k 0..31, units 4-7 lo
        # element id = (plane, n, k) so that a read can be checked by value
        def rm_block(n, kt):
            return [(u // 4, n, 32 * kt + 8 * (u % 4) + e) for u in range(8) for e in range(8)]
        for h in (0, 1, H - 1):
            for kt in (0, nk - 1):
                stage = [None] * (96 * 64)
                for r in range(96):
                    n = (r // 32) * d + h * 32 + (r % 32)
                    blk = rm_block(n, kt)
                    for u in range(8):
                        stage[(u * 96 + r) * 8: (u * 96 + r) * 8 + 8] = blk[u * 8: u * 8 + 8]
                assert all(v is not None for v in stage)
                for j in range(3):
                    for l31 in range(32):
                        for half in range(2):
                            for unit in (0, 4, 2, 6):            # hi c0, lo c0, hi c1, lo c1: the four rd_w calls of a stage
                                c, plane = (unit % 4) // 2, unit // 4
                                byte = (unit + half) * 1536 + j * 512 + l31 * 16
                                got = stage[byte // 2: byte // 2 + 8]
                                want = [(plane, j * d + 32 * h + l31, 32 * kt + 16 * c + 8 * half + e) for e in range(8)]
                                assert got == want, (d, h, kt, j, l31, half, unit)



@pytest.mark.parametrize("nkt,tail,passes", [(12, 1, 2), (12, 0, 2), (6, 1, 3), (6, 0, 1)])
def test_layer_tail_stream_protocol_and_weight_order(nkt, tail, passes):
    """ffn::ffn16_kernel (ffn16.hip), its vector-memory protocol and its weight stream restated and simulated for one wave.
    (1) The stream is ONE sequence of 16 KiB stages (two steps of four 2 KiB tiles) in consumption order: attention.output.dense
    (TAIL: NKT^2 / 2 steps) | up(0) | up(1) down(0) | ... | up(NG - 1) down(NG - 2) | down(NG - 1); every block is a whole number of
    turns of the 3-slot ring, so a step's fragment address is a compile-time constant -- checked against the kernel's step order and
    against upload_ffn16_weights' placement formulas.  (2) vmcnt retires in issue order: `s_waitcnt vmcnt(2)` at the stage top inside the
    last step of stage n must cover the two pieces of stage n + 1 and nothing more than needed; the request issued there (stage n + 3)
    lands in the slot of stage n, whose last fragment reads precede the top.  (3) At a pass's end everything is drained once
    (vmcnt(0): the next rows, the stores, two stages in flight)."""
    src = open(os.path.join(os.path.dirname(__file__), "..", "foldingdiff_amd", "csrc", "ffn16.hip")).read()
    api = open(os.path.join(os.path.dirname(__file__), "..", "foldingdiff_amd", "csrc", "api.hip")).read()
    for stmt in ("constexpr int SPS = 2;", "constexpr int NST = 3;", "constexpr int STEP = 4 * TILE;", "FD_WAIT_VM(PPW);",
                 "FD_WAIT_VM(3 * PPW);  // the rows landed", "FD_WAIT_VM(2 * PPW);  // ... and stage 0",
                 "constexpr bool top = s % SPS == SPS - 1;", "constexpr int NSA = TAIL ? NKT * NKT / 2 : 0;",
                 "steps(IC<0>{}, IC<NSA>{}, IC<2>{}, 0u);", "steps(IC<0>{}, IC<NKT>{}, IC<1>{}, abq);", "steps(IC<NKT>{}, IC<SPG>{}, IC<0>{}, 0u);",
                 "constexpr unsigned off = (unsigned)(((s / SPS) % NST) * STAGE + (s % SPS) * STEP + lo * 1024);"):
        assert stmt in src, stmt
    for stmt in ("auto up_at = [&](int G) { return G == 0 ? 0 : nkt + (G - 1) * spg; };",
                 "auto down_at = [&](int G) { return G == ng - 1 ? nkt + (ng - 1) * spg : nkt + G * spg + nkt; };",
                 "uint16_t* tile = timg.data() + ((size_t)(kt * (nkt / 2) + T / 4) * 4 + T % 4) * 1024;"):
        assert stmt in api, stmt
    NG, SPG, SPS, NST, PPW = nkt, 2 * nkt, 2, 3, 2
    nsa = nkt * nkt // 2 if tail else 0
    # ---- (1) the order in which the kernel consumes steps within a pass, as (block, local step): what the stream must hold
    consumed = [("ao", s) for s in range(nsa)] + [("up", 0, s) for s in range(nkt)]
    for G in range(NG):
        if G + 1 < NG:
            consumed += [("up", G + 1, s) for s in range(nkt)]
        consumed += [("down", G, s) for s in range(nkt)]
    up_at = lambda G: 0 if G == 0 else nkt + (G - 1) * SPG                      # noqa: E731  (api.hip)
    down_at = lambda G: nkt + (NG - 1) * SPG if G == NG - 1 else nkt + G * SPG + nkt   # noqa: E731
    placed = {}
    for kt in range(nkt if tail else 0):
        for T in range(2 * nkt):
            placed[kt * (nkt // 2) + T // 4] = ("ao", kt * (nkt // 2) + T // 4)
    for G in range(NG):
        for ks in range(nkt):
            placed[nsa + up_at(G) + ks] = ("up", G, ks)
        for pr in range(2):
            for T in range(2 * nkt):
                placed[nsa + down_at(G) + pr * (nkt // 2) + T // 4] = ("down", G, pr * (nkt // 2) + T // 4)
    assert [placed[i] for i in range(len(consumed))] == consumed and len(placed) == len(consumed) == nsa + NG * SPG
    # every block starts at ring slot 0 (its fragment addresses use the LOCAL step index)
    pos = 0
    for blk_len in ([nsa] if tail else []) + [nkt] * (2 * NG):
        assert (pos // SPS) % NST == 0 and blk_len % (SPS * NST) == 0, (pos, blk_len)
        pos += blk_len
    # ---- (2), (3) one wave's vector-memory queue: the operations still in flight, oldest first; vmcnt(n) retires all but the n youngest
    n_stage_pass = len(consumed) // SPS
    q, issued = [], [0]

    def issue_w():
        q.extend([("w", issued[0])] * PPW)
        issued[0] += 1

    def wait_vm(n):
        del q[:max(0, len(q) - n)]

    q.extend([("rows",)] * (2 * nkt + (2 * nkt if tail else 0)))
    for _ in range(3):
        issue_w()
    wait_vm(3 * PPW)
    assert ("rows",) not in q                      # "the rows landed" (then their lo plane is written to LDS)
    wait_vm(2 * PPW)
    assert ("w", 0) not in q and ("w", 1) in q      # "... and stage 0"; stages 1, 2 may still be in flight
    stage = 0
    for p in range(passes):
        for st in range(n_stage_pass):
            # inside the last step of stage `stage`: FD_WAIT_VM(PPW), barrier, request stage + 3 into the slot of `stage`
            if ("w", stage + 1) in q:
                younger = len(q) - 1 - max(i for i, o in enumerate(q) if o == ("w", stage + 1))
                assert younger == PPW and all(o == ("w", stage + 2) for o in q[-PPW:]), (p, st, q)   # the count is exact: it waits for no more
            wait_vm(PPW)
            assert ("w", stage + 1) not in q, (p, st)
            assert issued[0] == stage + 3 and (stage + 3) % NST == stage % NST   # the request overwrites the slot whose reads are done
            issue_w()
            stage += 1
        q.extend([("rows_next",)] * nkt + [("store",), ("store",), ("lo_next",)] * nkt)
        wait_vm(0)   # once per pass: the next rows, the stores, and with them the two stages in flight
    assert issued[0] == passes * n_stage_pass + 3


def test_layer_tail_weight_stream_and_register_layouts_compose():
    """The layout contract of ffn16.hip, emulated lane by lane in numpy (values in float64, no hi / lo split: the split is orthogonal):
    api.hip's upload_ffn16_weights places the PERMUTED weight rows in the stream; the kernel reads a tile's A operand as lane (c, g) <- unit g,
    row c; v_mfma_f32_16x16x32_f16 computes D[i][j] = sum_k A[i][k] B[k][j] with A[i][k] in lane i + 16 (k / 8), B[k][j] in lane
    j + 16 (k / 8), D[i][j] in lane j + 16 (i / 4), register i % 4 (scripts/probes/mfma16_layout_probe.hip).  With the permutation, what a lane
    holds after a contraction is EXACTLY its operand unit of the next one: attention.output.dense -> (LayerNorm) -> first dense -> (GELU) ->
    second dense -> output image units, with no cross-lane movement.  The emulation follows the kernel's step order and must reproduce
    x -> ((x Wo^T) Wi^T) Wd^T (activations left out: they are elementwise in this layout)."""
    api = open(os.path.join(os.path.dirname(__file__), "..", "foldingdiff_amd", "csrc", "api.hip")).read()
    for stmt in ("const int f = 64 * G + 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3);",
                 "const int o = 32 * (T >> 1) + 8 * (i >> 2) + 4 * (T & 1) + (i & 3);",
                 "put(down_at(G) + pr * (nkt / 2) + T / 4, T % 4, i, rd.data() + ((size_t)o * nkd + 2 * G + pr) * 64);",
                 "put(up_at(G) + ks, t, i, ri.data() + ((size_t)f * nkt + ks) * 64);"):
        assert stmt in api, stmt
    nkt = 2                                  # d_model 64, intermediate 128: the formulas do not depend on the size
    d, ff, NG, SPG = 32 * nkt, 64 * nkt, nkt, 2 * nkt
    rng = np.random.RandomState(0)
    Wo, Wi, Wd = rng.randn(d, d), rng.randn(ff, d), rng.randn(d, ff)
    x = rng.randn(16, d)                     # one wave's sixteen token rows
    # ---- the stream: tiles [step][t] of (16 rows, 32 k), rows permuted (api.hip)
    nsa = nkt * nkt // 2
    stream = np.zeros((nsa + NG * SPG, 4, 16, 32))
    up_at = lambda G: 0 if G == 0 else nkt + (G - 1) * SPG                      # noqa: E731
    down_at = lambda G: nkt + (NG - 1) * SPG if G == NG - 1 else nkt + G * SPG + nkt   # noqa: E731
    for kt in range(nkt):
        for T in range(2 * nkt):
            for i in range(16):
                o = 32 * (T >> 1) + 8 * (i >> 2) + 4 * (T & 1) + (i & 3)
                stream[kt * (nkt // 2) + T // 4, T % 4, i] = Wo[o, 32 * kt:32 * kt + 32]
    for G in range(NG):
        for ks in range(nkt):
            for t in range(4):
                for i in range(16):
                    f = 64 * G + 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3)
                    stream[nsa + up_at(G) + ks, t, i] = Wi[f, 32 * ks:32 * ks + 32]
        for pr in range(2):
            for T in range(2 * nkt):
                for i in range(16):
                    o = 32 * (T >> 1) + 8 * (i >> 2) + 4 * (T & 1) + (i & 3)
                    stream[nsa + down_at(G) + pr * (nkt // 2) + T // 4, T % 4, i] = Wd[o, 64 * G + 32 * pr:64 * G + 32 * pr + 32]

    def mfma(tile, breg, acc):
        """acc[lane][reg] += D, A = tile rows (lane (c, g) reads row c, k 8 g .. 8 g + 7), B = breg[lane][8]: lane (c, g) holds k 8 g .. of token c"""
        B = np.zeros((32, 16))
        for lane in range(64):
            c, g = lane & 15, lane >> 4
            B[8 * g:8 * g + 8, c] = breg[lane]
        D = tile @ B
        for i in range(16):
            for j in range(16):
                acc[j + 16 * (i // 4), i % 4] += D[i, j]

    def operand(rows, kt):   # the stationary operand registers of k32 step kt: lane (c, g) <- features 32 kt + 8 g .. + 7 of token c
        return np.array([rows[lane & 15, 32 * kt + 8 * (lane >> 4):32 * kt + 8 * (lane >> 4) + 8] for lane in range(64)])

    def to_operands(acc_tiles):   # accumulator tiles (2 kt, 2 kt + 1) -> the operand registers of k32 step kt, WITHOUT leaving the lane
        return [np.concatenate([acc_tiles[2 * kt], acc_tiles[2 * kt + 1]], axis=1) for kt in range(len(acc_tiles) // 2)]

    step = [0]

    def next_step():
        step[0] += 1
        return stream[step[0] - 1]

    # attention.output.dense on the context rows: k32 step kt, output tiles 4 q .. 4 q + 3
    Y = [np.zeros((64, 4)) for _ in range(2 * nkt)]
    for kt in range(nkt):
        for q in range(nkt // 2):
            tiles = next_step()
            for t in range(4):
                mfma(tiles[t], operand(x, kt), Y[4 * q + t])
    a_ops = to_operands(Y)                                  # its output IS the next stationary operand
    want_a = x @ Wo.T
    for kt in range(nkt):
        assert np.allclose(a_ops[kt], operand(want_a, kt))
    # the feed-forward in the kernel's order: up(0) | up(1) down(0) | ... | down(NG - 1)
    Y = [np.zeros((64, 4)) for _ in range(2 * nkt)]

    def up():
        U = [np.zeros((64, 4)) for _ in range(4)]
        for ks in range(nkt):
            tiles = next_step()
            for t in range(4):
                mfma(tiles[t], a_ops[ks], U[t])
        return U

    def down(U):
        u_ops = to_operands(U)                              # (the GELU and the split happen here, per register)
        for pr in range(2):
            for q in range(nkt // 2):
                tiles = next_step()
                for t in range(4):
                    mfma(tiles[t], u_ops[pr], Y[4 * q + t])

    U = up()
    for G in range(NG):
        Un = up() if G + 1 < NG else None
        down(U)
        U = Un
    assert step[0] == len(stream)
    want = (want_a @ Wi.T) @ Wd.T
    out_ops = to_operands(Y)                                # = the units of the output image, and of the residual's registers
    for kt in range(nkt):
        assert np.allclose(out_ops[kt], operand(want, kt))
