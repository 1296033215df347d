"""
Output side of the sampler: sampled angles -> backbone PDB files.

Mirrors the functions ``bin/sample.py`` calls after ``sampling.sample``
(``write_preds_pdb_folder`` bin/sample.py:105-128 -> ``create_new_chain_nerf``
foldingdiff/angles_and_coords.py:112-184 -> ``write_coords_to_pdb`` :187-253), with the coordinates
of ALL chains built by one ``fd_nerf`` launch instead of a multiprocessing pool of per-residue
Python loops.  The PDB text is what the reference gets from biotite 0.34's ``PDBFile`` for the structure it
builds (chain A, GLY residues, N / CA / C, occupancy 1.00, B-factor 5.00, every consecutive atom pair bonded):
80-column ATOM records, then one CONECT record per direction for each bond that joins two residues (C of
residue i -- N of residue i + 1; biotite leaves the bonds inside a residue to the residue template).  The byte
layout is pinned against a file the reference's own ``write_coords_to_pdb`` wrote
(``plots/pdb_structures/noising_visualization/fully_noised.pdb`` -> tests/golden/ref_written_backbone.pdb,
tests/test_host.py).
The angle CSVs of bin/sample.py:365-369 are plain ``DataFrame.to_csv`` calls on the arrays
``sampling.sample`` returns and need no counterpart here.
"""
import logging
import os
from typing import List, Optional, Sequence

import numpy as np

from . import nerf

_ATOMS = (("N", "N"), ("CA", "C"), ("C", "C"))  # (atom name, element) per residue, in chain order


def _columns(dists_and_angles):
    """(values [len, F] float32, column names) of a DataFrame or of an (array, names) pair."""
    if hasattr(dists_and_angles, "columns"):
        return np.asarray(dists_and_angles.values, dtype=np.float32), [str(c) for c in dists_and_angles.columns]
    vals, names = dists_and_angles
    return np.asarray(vals, dtype=np.float32), [str(c) for c in names]


def _select(vals, names, angles_to_set, dists_to_set):
    """Column subset the reference would feed to NERFBuilder (angles_and_coords.py:123-173)."""
    if angles_to_set is None and dists_to_set is None:
        keep = list(names)   # auto: every column; one ':' = distance, otherwise angle
    else:
        assert angles_to_set is not None and dists_to_set is not None
        keep = list(angles_to_set) + list(dists_to_set)
    assert all(a in keep for a in ("phi", "psi", "omega")), "phi, psi and omega must be set"
    for c in keep:
        assert c in names, f"{c} not among the columns {names}"
        if c not in nerf._SLOT:
            raise ValueError(f"Unrecognized {'distance' if c.count(':') == 1 else 'angle'}: {c}")
    idx = [names.index(c) for c in keep]
    return vals[:, idx], keep


def write_coords_to_pdb(coords: np.ndarray, out_fname: str) -> str:
    """Write 3N backbone coordinates (N, CA, C per residue) as a PDB file; returns ``out_fname``."""
    coords = np.asarray(coords, dtype=np.float64)
    assert len(coords) % 3 == 0, f"Expected 3N coords, got {len(coords)}"
    lines = []
    for j, (x, y, z) in enumerate(coords):
        name, element = _ATOMS[j % 3]
        atom_field = f" {name:<3}" if len(name) < 4 else name   # one-letter elements start in column 14
        lines.append(
            f"ATOM  {((j % 99999) + 1):>5d} {atom_field} GLY A{((j // 3) % 9999 + 1):>4d}    "
            f"{x:>8.3f}{y:>8.3f}{z:>8.3f}{1.0:>6.2f}{5.0:>6.2f}          {element:>2}  ")
    for c in range(3, len(coords), 3):   # peptide bonds: atom ids c (C) and c + 1 (N of the next residue), both directions
        lines.append(f"CONECT{c:>5d}{c + 1:>5d}")
        lines.append(f"CONECT{c + 1:>5d}{c:>5d}")
    with open(out_fname, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    return out_fname


def read_pdb_backbone(fname: str) -> np.ndarray:
    """[3N, 3] coordinates of the ATOM records of a backbone PDB (fixed columns 31-54)."""
    out = []
    with open(fname) as fh:
        for line in fh:
            if line.startswith("ATOM"):
                out.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
    return np.asarray(out, dtype=np.float64)


def create_new_chain_nerf(out_fname: str, dists_and_angles, angles_to_set: Optional[List[str]] = None,
                          dists_to_set: Optional[List[str]] = None, center_coords: bool = True, device: int = 0) -> str:
    """One chain: angles (DataFrame with the feature names as columns) -> PDB file.  Returns the path,
    or "" when the coordinates contain NaN (as the reference does)."""
    return _write_chains([out_fname], [dists_and_angles], angles_to_set, dists_to_set, center_coords, device)[0]


def write_preds_pdb_folder(final_sampled: Sequence, outdir: str, basename_prefix: str = "generated_",
                           threads: Optional[int] = None, device: int = 0) -> List[str]:
    """``{outdir}/{basename_prefix}{i}.pdb`` for every sampled chain (bin/sample.py:105-128).  ``threads`` is
    accepted for signature compatibility; the coordinates come from a single device launch."""
    os.makedirs(outdir, exist_ok=True)
    logging.info(f"Writing sampled angles as PDB files to {outdir}")
    names = [os.path.join(outdir, f"{basename_prefix}{i}.pdb") for i in range(len(final_sampled))]
    return _write_chains(names, list(final_sampled), None, None, True, device)


def _write_chains(fnames, chains, angles_to_set, dists_to_set, center_coords, device):
    if not chains:
        return []
    groups = {}   # chains sharing a column set go through one fd_nerf call
    for i, ch in enumerate(chains):
        vals, names = _columns(ch)
        vals, keep = _select(vals, names, angles_to_set, dists_to_set)
        groups.setdefault(tuple(keep), []).append((i, vals))
    written = [""] * len(chains)
    for keep, members in groups.items():
        coords = nerf.build_backbones([v for _, v in members], list(keep), center_coords=center_coords, device=device)
        for (i, vals), xyz in zip(members, coords):
            if np.any(np.isnan(xyz)):
                logging.warning(f"Found NaN values, not writing pdb file {fnames[i]}")
                continue
            assert xyz.shape == (3 * len(vals), 3), f"Unexpected shape: {xyz.shape} for input of {len(vals)}"
            written[i] = write_coords_to_pdb(xyz, fnames[i])
    return written
