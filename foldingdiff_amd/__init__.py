"""
foldingdiff_amd -- MI355X-native reverse-diffusion sampler for foldingdiff's
protein-backbone angle model.  Python host code mirroring the reference API
(``modelling.BertForDiffusionBase.from_dir``, ``sampling.sample`` ...) over a
thin C-ABI HIP library (include/fdmi.h, csrc/).  No CPU compute path.
"""
__version__ = "0.1.0"

from . import beta_schedules, datasets, modelling, sampling, utils  # noqa: F401
