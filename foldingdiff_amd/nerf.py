"""
NeRF: sampled angles -> Cartesian backbone coordinates on the GPU (one C-ABI call, ``fd_nerf``).

Mirrors the part of foldingdiff/nerf.py that the sampler's post-processing uses
(``NERFBuilder`` :27-142 as driven by ``create_new_chain_nerf``,
foldingdiff/angles_and_coords.py:112-184): N, CA, C positions built from phi / psi / omega, the
bond angles and (if the feature set has them) bond lengths; constants otherwise.  Writing PDB
files (biotite) stays outside this path.
"""
import ctypes as C
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _binding

N_CA_LENGTH = 1.46
CA_C_LENGTH = 1.54
C_N_LENGTH = 1.34
N_INIT = np.array([17.047, 14.099, 3.625])
CA_INIT = np.array([16.967, 12.784, 4.338])
C_INIT = np.array([15.685, 12.755, 5.133])

# feature name -> slot of fd_nerf's feat_idx (create_new_chain_nerf's name mapping, angles_and_coords.py:150-173)
_SLOT = {"phi": 0, "psi": 1, "omega": 2, "tau": 3, "N:CA:C": 3, "CA:C:1N": 4, "C:1N:1CA": 5,
         "0C:1N": 6, "N:CA": 7, "CA:C": 8}


def build_backbones(
    angles: Union[np.ndarray, Sequence[np.ndarray]],
    feature_names: Sequence[str],
    center_coords: bool = True,
    device: int = 0,
) -> List[np.ndarray]:
    """Coordinates for a batch of sampled backbones.

    ``angles``: ``[B, L, F]`` array or a list of ``[len_i, F]`` arrays (e.g. ``[s[-1] for s in sample(...)]``);
    ``feature_names``: the F column names (``dset.feature_names["angles"]``).  Returns one
    ``float64 [3 * len_i, 3]`` array per chain (N, CA, C per residue), centred like
    ``NERFBuilder.centered_cartesian_coords`` unless ``center_coords=False``."""
    chains = [np.asarray(a, dtype=np.float32) for a in (angles if not isinstance(angles, np.ndarray) or angles.ndim != 3 else list(angles))]
    F = len(feature_names)
    assert all(c.ndim == 2 and c.shape[1] == F for c in chains), "each chain must be [len, F]"
    for req in ("phi", "psi", "omega"):
        assert req in feature_names, f"NeRF needs the dihedral {req!r}"
    idx = np.full(9, -1, dtype=np.int32)
    for col, name in enumerate(feature_names):
        if name not in _SLOT:
            raise ValueError(f"Unrecognized feature: {name}")
        idx[_SLOT[name]] = col
    B, L = len(chains), max(len(c) for c in chains)
    feats = np.zeros((B, L, F), dtype=np.float32)
    lens = np.zeros(B, dtype=np.int32)
    for i, c in enumerate(chains):
        feats[i, : len(c)] = c
        lens[i] = len(c)
    out = np.empty((B, 3 * L, 3), dtype=np.float64)
    _binding.check(_binding.load().fd_nerf(
        device, feats.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), B, L, F,
        idx.ctypes.data_as(C.c_void_p), 1 if center_coords else 0, out.ctypes.data_as(C.c_void_p)))
    return [out[i, : 3 * lens[i]].copy() for i in range(B)]


class NERFBuilder:
    """Single-chain builder with the reference's constructor arguments (bond angles / lengths as floats
    or per-residue arrays); ``cartesian_coords`` / ``centered_cartesian_coords`` run on the GPU."""

    def __init__(self, phi_dihedrals, psi_dihedrals, omega_dihedrals,
                 bond_len_n_ca=N_CA_LENGTH, bond_len_ca_c=CA_C_LENGTH, bond_len_c_n=C_N_LENGTH,
                 bond_angle_n_ca=121 / 180 * np.pi, bond_angle_ca_c=109 / 180 * np.pi, bond_angle_c_n=115 / 180 * np.pi,
                 init_coords=None, device: int = 0) -> None:
        if init_coords is not None:
            raise NotImplementedError("custom init_coords: the device builder starts from the reference's 1CRN seed")
        cols, names = [], []
        def add(name, v, default):
            if isinstance(v, (float, int)):
                if abs(float(v) - default) > 1e-12:
                    raise NotImplementedError(f"scalar override of {name} (only the reference default or an array)")
                return
            cols.append(np.asarray(v, dtype=np.float32).squeeze()); names.append(name)
        for n, v in (("phi", phi_dihedrals), ("psi", psi_dihedrals), ("omega", omega_dihedrals)):
            cols.append(np.asarray(v, dtype=np.float32).squeeze()); names.append(n)
        add("tau", bond_angle_ca_c, 109 / 180 * np.pi)
        add("CA:C:1N", bond_angle_c_n, 115 / 180 * np.pi)
        add("C:1N:1CA", bond_angle_n_ca, 121 / 180 * np.pi)
        add("0C:1N", bond_len_c_n, C_N_LENGTH)
        add("N:CA", bond_len_n_ca, N_CA_LENGTH)
        add("CA:C", bond_len_ca_c, CA_C_LENGTH)
        self._feats, self._names, self._device = np.stack(cols, axis=1), names, device

    @property
    def cartesian_coords(self) -> np.ndarray:
        return build_backbones([self._feats], self._names, center_coords=False, device=self._device)[0]

    @property
    def centered_cartesian_coords(self) -> np.ndarray:
        return build_backbones([self._feats], self._names, center_coords=True, device=self._device)[0]
