"""
The two data-free dataset shells the sampler needs (host-side glue).

``sampling.sample`` only reads ``sample_noise``, ``timesteps``,
``alpha_beta_terms["betas"]``, ``feature_is_angular``, ``pad`` and, optionally,
``dset.get_masked_means()`` from its ``train_dset`` argument
(foldingdiff/sampling.py:150-156, :208-216).  These classes carry exactly that,
with the reference's names and constructor arguments
(foldingdiff/datasets.py:569-623 AnglesEmptyDataset, :685-799 NoisedAnglesDataset).
The reference's own dataset objects can be passed to ``foldingdiff_amd.sampling``
instead -- only the attributes above are used.
"""
import json
import os
from typing import Optional

import numpy as np
import torch

from . import beta_schedules, utils

FEATURE_SET_NAMES_TO_ANGULARITY = {
    "canonical": [False, False, False, True, True, True, True, True, True],
    "canonical-full-angles": [True, True, True, True, True, True],
    "canonical-minimal-angles": [True, True, True, True],
    "cart-coords": [False, False, False],
}
FEATURE_SET_NAMES_TO_FEATURE_NAMES = {
    "canonical": ["0C:1N", "N:CA", "CA:C", "phi", "psi", "omega", "tau", "CA:C:1N", "C:1N:1CA"],
    "canonical-full-angles": ["phi", "psi", "omega", "tau", "CA:C:1N", "C:1N:1CA"],
    "canonical-minimal-angles": ["phi", "psi", "omega", "tau"],
    "cart-coords": ["x", "y", "z"],
}


class AnglesEmptyDataset:
    """Feature-set metadata + pad length + optional training mean offset; no data."""

    def __init__(self, feature_set_key: str, pad: int = 128, mean_offset: Optional[np.ndarray] = None):
        key = "coords" if feature_set_key == "cart-coords" else "angles"
        self.feature_is_angular = {key: list(FEATURE_SET_NAMES_TO_ANGULARITY[feature_set_key])}
        self.feature_names = {key: list(FEATURE_SET_NAMES_TO_FEATURE_NAMES[feature_set_key])}
        self.pad = pad
        self._mean_offset = mean_offset
        if mean_offset is not None:
            assert mean_offset.size == len(self.feature_names[key])

    @classmethod
    def from_dir(cls, dirname: str):
        with open(os.path.join(dirname, "training_args.json")) as fh:
            args = json.load(fh)
        off_file = os.path.join(dirname, "training_mean_offset.npy")
        offset = np.load(off_file) if os.path.isfile(off_file) else None
        return cls(feature_set_key=args["angles_definitions"], pad=args["max_seq_len"], mean_offset=offset)

    def get_masked_means(self) -> np.ndarray:
        if self._mean_offset is None:
            raise NotImplementedError
        return np.copy(self._mean_offset)

    def __len__(self):
        raise NotImplementedError

    def __getitem__(self, index):
        raise NotImplementedError


class NoisedAnglesDataset:
    """Schedule + initial-noise carrier around a (possibly empty) dataset."""

    def __init__(
        self,
        dset,
        dset_key: str = "angles",
        timesteps: int = 250,
        exhaustive_t: bool = False,
        beta_schedule: beta_schedules.SCHEDULES = "linear",
        nonangular_variance: float = 1.0,
        angular_variance: float = 1.0,
    ) -> None:
        assert hasattr(dset, "feature_names") and hasattr(dset, "feature_is_angular")
        assert dset_key in dset.feature_is_angular, f"{dset_key} not in {dset.feature_is_angular}"
        self.dset = dset
        self.dset_key = dset_key
        self.n_features = len(dset.feature_is_angular[dset_key])
        self.nonangular_var_scale = nonangular_variance
        self.angular_var_scale = angular_variance
        self.timesteps = timesteps
        self.schedule = beta_schedule
        self.exhaustive_timesteps = exhaustive_t
        self.alpha_beta_terms = beta_schedules.compute_alphas(
            beta_schedules.get_variance_schedule(beta_schedule, timesteps)
        )

    @property
    def feature_names(self):
        return self.dset.feature_names

    @property
    def feature_is_angular(self):
        return self.dset.feature_is_angular

    @property
    def pad(self):
        return self.dset.pad

    @property
    def filenames(self):
        return self.dset.filenames

    def sample_length(self, *args, **kwargs):
        return self.dset.sample_length(*args, **kwargs)

    def __len__(self) -> int:
        return len(self.dset) if not self.exhaustive_timesteps else int(len(self.dset) * self.timesteps)

    def __getitem__(self, index: int, use_t_val: Optional[int] = None, ignore_zero_center: bool = False):
        """Forward-noise item ``index`` of the wrapped dataset: q(x_t | x_0) =
        sqrt(abar_t) x_0 + sqrt(1 - abar_t) noise, angular columns re-wrapped
        (foldingdiff/datasets.py:801-886; SURVEY 8f N3 -- the input of the partial-noise
        reconstruction path).  ``use_t_val`` fixes t (clipped to [0, timesteps-1]); otherwise t is
        the exhaustive index or one ``torch.randint`` draw, exactly as in the reference."""
        if not 0 <= index < len(self):
            raise AssertionError(f"Index {index} out of bounds for {len(self)}")
        which, step = (divmod(index, self.timesteps) if self.exhaustive_timesteps else (index, None))
        item = self.dset.__getitem__(which, ignore_zero_center=ignore_zero_center)
        x0 = (item[self.dset_key] if self.dset_key is not None else item).clone()
        assert isinstance(x0, torch.Tensor)
        # t: caller-fixed (clipped), the exhaustive index, or ONE draw from the global generator -- the draw
        # comes before the noise draw, as in the reference (the order fixes the random stream)
        if use_t_val is not None:
            assert not self.exhaustive_timesteps, "Cannot use specific t in exhaustive mode"
            t = torch.from_numpy(np.clip(np.array([use_t_val]), 0, self.timesteps - 1)).long()
        elif step is not None:
            t = torch.tensor([step]).long()
        else:
            t = torch.randint(0, self.timesteps, (1,)).long()
        terms = self.alpha_beta_terms
        keep, spread = terms["sqrt_alphas_cumprod"][t.item()], terms["sqrt_one_minus_alphas_cumprod"][t.item()]
        eps = self.sample_noise(x0)                      # shape-only use of x0
        x_t = keep * x0 + spread * eps                    # q(x_t | x_0)
        wrap_cols = np.where(self.dset.feature_is_angular[self.dset_key])[0]
        x_t[:, wrap_cols] = utils.modulo_with_wrapped_range(x_t[:, wrap_cols], -np.pi, np.pi)
        extra = {"corrupted": x_t, "t": t, "known_noise": eps,
                 "sqrt_alphas_cumprod_t": keep, "sqrt_one_minus_alphas_cumprod_t": spread}
        if not isinstance(item, dict):
            return extra
        assert item.keys().isdisjoint(extra.keys())
        item.update(extra)
        return item

    def sample_noise(self, vals: torch.Tensor) -> torch.Tensor:
        """N(0, 1) from the global CPU generator (same draw as the reference, so a
        fixed ``torch.manual_seed`` gives the same start point), per-feature variance
        scale, angular columns wrapped to [-pi, pi)."""
        angular = self.dset.feature_is_angular[self.dset_key]
        noise = torch.randn_like(vals)
        if self.angular_var_scale != 1.0 or self.nonangular_var_scale != 1.0:
            for j in range(noise.shape[-1]):
                noise[..., j] *= self.angular_var_scale if angular[j] else self.nonangular_var_scale
        idx = np.where(angular)[0]
        noise[..., idx] = utils.modulo_with_wrapped_range(noise[..., idx], -np.pi, np.pi)
        return noise
