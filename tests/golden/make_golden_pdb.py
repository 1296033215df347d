#!/usr/bin/env python3
"""Golden fixture for the PDB byte layout.

The reference checkout holds one backbone file written by its own ``write_coords_to_pdb``
(foldingdiff/angles_and_coords.py:187-253, through biotite's PDBFile): the fully noised chain of the noising
illustration, plots/pdb_structures/noising_visualization/fully_noised.pdb (88 residues: GLY, chain A, B-factor
5.00, CONECT records for the peptide bonds).  biotite is not installable here, so this file is the only
output of that writer available; it is copied unchanged.  Run in the build container (needs /root/reference).
"""
import os
import shutil

SRC = "/root/reference/plots/pdb_structures/noising_visualization/fully_noised.pdb"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_written_backbone.pdb")

if __name__ == "__main__":
    shutil.copyfile(SRC, DST)
    n_atom = sum(1 for ln in open(DST) if ln.startswith("ATOM"))
    n_con = sum(1 for ln in open(DST) if ln.startswith("CONECT"))
    print(f"{DST}: {n_atom} ATOM records, {n_con} CONECT records")
