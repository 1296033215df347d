#!/usr/bin/env python3
"""
Golden fixture for the partial-noise reconstruction path, produced by THE REFERENCE:
  * ``NoisedAnglesDataset.__getitem__(idx, use_t_val=...)`` (foldingdiff/datasets.py:801-886) on a small
    synthetic angle dataset, fixed torch seed;
  * the denoising loop of ``get_reconstruction_error`` (foldingdiff/sampling.py:303-331, statements
    executed verbatim; its TM-align scoring needs an external binary and is not run) on the reference
    ``BertForDiffusionBase`` (absolute positions) whose weights are in ref_abs_model.npz.
Writes tests/golden/ref_reconstruct.npz.  Build container only (needs /root/reference):

    python tests/golden/make_golden_noising.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)

PAD, F, NT, T, SEED = 48, 6, 3, 1000, 777
LENGTHS = [48, 31, 40, 7, 19]


class ToyAngles(torch.utils.data.Dataset):
    """Stand-in for CathCanonicalAnglesDataset: items are dicts with zero-padded [pad, F] angles."""
    feature_names = {"angles": ["phi", "psi", "omega", "tau", "CA:C:1N", "C:1N:1CA"]}
    feature_is_angular = {"angles": [True] * 6}
    pad = PAD

    def __init__(self, angles):
        self.angles = angles
        self.filenames = [f"toy_{i}.pdb" for i in range(len(LENGTHS))]

    def __len__(self):
        return len(LENGTHS)

    def __getitem__(self, index, ignore_zero_center=False):
        l = LENGTHS[index]
        mask = torch.zeros(PAD)
        mask[:l] = 1.0
        return {"angles": self.angles[index].clone(), "attn_mask": mask, "position_ids": torch.arange(PAD),
                "lengths": torch.tensor(l, dtype=torch.int64)}


def main():
    mg.import_reference()
    sys.path.insert(0, mg.REF)
    from foldingdiff import datasets, modelling, sampling, utils
    from torch.utils.data.dataloader import default_collate
    from transformers import BertConfig

    g = torch.Generator().manual_seed(99)
    angles = torch.zeros(len(LENGTHS), PAD, F)
    for i, l in enumerate(LENGTHS):
        angles[i, :l] = utils.modulo_with_wrapped_range(torch.randn(l, F, generator=g) * 1.3, -np.pi, np.pi)
    dset = datasets.NoisedAnglesDataset(ToyAngles(angles), dset_key="angles", timesteps=T, beta_schedule="cosine")

    gm = np.load(os.path.join(HERE, "ref_abs_model.npz"))
    modelling.BertForDiffusionBase.init_weights = lambda self: None  # broken under transformers 5.x
    cfg = BertConfig(max_position_embeddings=64, num_attention_heads=2, hidden_size=64, intermediate_size=128,
                     num_hidden_layers=2, position_embedding_type="absolute", hidden_dropout_prob=0.1,
                     attention_probs_dropout_prob=0.1, use_cache=False, attn_implementation="eager")
    model = modelling.BertForDiffusionBase(cfg, ft_is_angular=[True] * 6, time_encoding="gaussian_fourier", decoder="mlp")
    model.load_state_dict({k[4:]: torch.from_numpy(gm[k]) for k in gm.files if k.startswith("sd::")}, strict=True)
    model.eval()

    out = {"angles": angles.numpy(), "lengths": np.array(LENGTHS), "seed": SEED, "T": T, "noise_timesteps": NT, "pad": PAD}
    torch.manual_seed(SEED)
    # ---- foldingdiff/sampling.py:303-331, verbatim statements (device = cpu, one batch) ----
    device = next(model.parameters()).device
    noise_timesteps = NT
    idx_batch = list(range(len(dset)))
    items = [{k: v.to(device) for k, v in dset.__getitem__(idx, use_t_val=noise_timesteps).items()} for idx in idx_batch]
    for i, it in enumerate(items):
        out[f"corrupted{i}"] = it["corrupted"].numpy().copy()
        out[f"known_noise{i}"] = it["known_noise"].numpy().copy()
        out[f"t{i}"] = it["t"].numpy().copy()
    batch = default_collate(items)
    img = batch["corrupted"].clone()
    assert img.ndim == 3
    with torch.no_grad():
        for i in list(reversed(list(range(0, noise_timesteps)))):
            img = sampling.p_sample(model=model, x=img,
                                    t=torch.full((len(idx_batch),), fill_value=i, dtype=torch.long).to(device),
                                    seq_lens=batch["lengths"], t_index=i, betas=dset.alpha_beta_terms["betas"])
            img = utils.modulo_with_wrapped_range(img)
    for i, l in enumerate(batch["lengths"].squeeze()):
        out[f"reconst{i}"] = img[i, :l].cpu().numpy()
        out[f"truth{i}"] = batch["angles"][i, :l].cpu().numpy()
    # one more item with a random t (the training-time path), continuing the same generator
    it = dset.__getitem__(2)
    out["rand_t"] = it["t"].numpy()
    out["rand_corrupted"] = it["corrupted"].numpy()
    np.savez_compressed(os.path.join(HERE, "ref_reconstruct.npz"), **out)
    print("ref_reconstruct.npz:", os.path.getsize(os.path.join(HERE, "ref_reconstruct.npz")) / 1024, "KiB;",
          "t =", [int(out[f"t{i}"][0]) for i in range(len(LENGTHS))], "rand t =", int(out["rand_t"][0]))


if __name__ == "__main__":
    main()
