"""
CPU oracle for the NeRF angles -> backbone-coordinates step (TEST INFRASTRUCTURE ONLY).

Restates ``NERFBuilder.cartesian_coords`` / ``place_dihedral`` of
foldingdiff/nerf.py:27-204 as called by ``create_new_chain_nerf``
(foldingdiff/angles_and_coords.py:112-184) for the sampler's feature sets: per residue
arrays phi, psi, omega and (optionally) the three bond angles, constant bond lengths.
Pinned bit-for-bit against the reference class by tests/golden/make_golden.py.

Arithmetic notes that matter for parity: the angle arrays are float32 (they come from the
sampler), so numpy evaluates cos/sin and ``bond_length * cos(.)`` in float32; the frame
vectors come from the float64 seed coordinates, so everything else is float64.  Quirk kept:
the CA->C placement of residue i+1 uses the N:CA:C angle at index i (nerf.py:104-116).
"""
import numpy as np

N_CA_LENGTH, CA_C_LENGTH, C_N_LENGTH = 1.46, 1.54, 1.34           # nerf.py:17-19
N_INIT = np.array([17.047, 14.099, 3.625])                         # nerf.py:22-24
CA_INIT = np.array([16.967, 12.784, 4.338])
C_INIT = np.array([15.685, 12.755, 5.133])
DEFAULT_ANGLE = {"C:1N:1CA": 121 / 180 * np.pi, "tau": 109 / 180 * np.pi, "CA:C:1N": 115 / 180 * np.pi}  # nerf.py:41-43


def place(a, b, c, bond_angle, bond_length, torsion):
    """nerf.py:145-204 (numpy branch)."""
    unit = lambda x: x / np.linalg.norm(x, axis=-1)
    ab = b - a
    bc = unit(c - b)
    n = unit(np.cross(ab, bc))
    nbc = np.cross(n, bc)
    m = np.stack([bc, nbc, n], axis=-1)
    d = np.stack([
        -bond_length * np.cos(bond_angle),
        bond_length * np.cos(torsion) * np.sin(bond_angle),
        bond_length * np.sin(torsion) * np.sin(bond_angle),
    ], axis=a.ndim - 1)
    return m.dot(d) + c


def build(phi, psi, omega, tau=None, ca_c_n=None, c_n_ca=None, center=True, len_c_n=None, len_n_ca=None, len_ca_c=None):
    """[L] arrays -> [3L, 3] float64 coordinates (N, CA, C per residue)."""
    L = len(phi)
    pick = lambda arr, name, i: DEFAULT_ANGLE[name] if arr is None else arr[i]
    plen = lambda arr, dflt, i: dflt if arr is None else arr[i]
    out = [N_INIT.copy(), CA_INIT.copy(), C_INIT.copy()]
    for i in range(L - 1):
        out.append(place(out[-3], out[-2], out[-1], pick(ca_c_n, "CA:C:1N", i), plen(len_c_n, C_N_LENGTH, i), psi[i]))
        out.append(place(out[-3], out[-2], out[-1], pick(c_n_ca, "C:1N:1CA", i), plen(len_n_ca, N_CA_LENGTH, i), omega[i]))
        out.append(place(out[-3], out[-2], out[-1], pick(tau, "tau", i), plen(len_ca_c, CA_C_LENGTH, i), phi[i + 1]))
    xyz = np.array(out)
    return xyz - xyz.mean(axis=0) if center else xyz
