"""
Host-side helpers of the sampler path.

``modulo_with_wrapped_range`` keeps the reference's name, argument meaning and
result (foldingdiff/utils.py:87-121): ``((v - lo) % (hi - lo)) + lo`` evaluated
with python-float bounds, for python scalars, numpy arrays and torch tensors.
(The per-step wrap inside the sampling loop runs on the GPU -- csrc/rowwise.hip
``wrap_pi`` -- this host version is for initial noise and the mean-offset shift.)
"""
import os

import numpy as np


def modulo_with_wrapped_range(vals, range_min: float = -np.pi, range_max: float = np.pi):
    """
    >>> modulo_with_wrapped_range(3, -2, 2)
    -1
    """
    assert range_min <= 0.0
    assert range_min < range_max
    width = range_max - range_min
    return ((vals - range_min) % width) + range_min


def is_huggingface_hub_id(s: str) -> bool:
    """Offline-safe stand-in for foldingdiff/utils.py:15-24 (which does a live HTTP
    GET): an existing local directory is never a hub id; anything else of the form
    ``user/repo`` is reported as a hub id without touching the network."""
    if os.path.isdir(s):
        return False
    parts = s.split("/")
    return len(parts) == 2 and all(parts) and not s.startswith((".", "/"))
