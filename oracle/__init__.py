"""
oracle/ -- CPU restatement of the foldingdiff reverse-diffusion hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker.  The product path (``foldingdiff_amd``) never
imports anything from here and fails loudly when its HIP library is missing.

What is restated, and from where (paths relative to the reference checkout):

* ``ref_model.OracleBertForDiffusion``  -- ``foldingdiff/modelling.py:42-71``
  (GaussianFourierProjection), ``:74-93`` (SinusoidalPositionEmbeddings),
  ``:132-170`` (BertEmbeddings), ``:173-208`` (AnglesPredictor), ``:239-295``
  (module names => state_dict keys), ``:384-484`` (forward), plus the
  HuggingFace ``transformers==4.11.3`` ``BertEncoder`` semantics
  (``requirements.txt:6``; third-party, NOT vendored in the reference) that
  ``modelling.py:271`` constructs and ``:473-480`` calls.
* ``ref_sampling``  -- ``foldingdiff/sampling.py:27-75`` (p_sample), ``:78-132``
  (p_sample_loop), ``:135-224`` (sample), ``foldingdiff/beta_schedules.py:20-78``,
  ``foldingdiff/utils.py:87-121`` (modulo_with_wrapped_range),
  ``foldingdiff/datasets.py:772-799`` (sample_noise).

Parity pinning status
---------------------
* Schedules, wrap, sample_noise, p_sample / p_sample_loop / sample: PINNED --
  checked bit-for-bit against outputs of the reference's own modules imported
  in the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
* Model forward with ``position_embedding_type="absolute"``: PINNED against the
  reference's own ``BertForDiffusionBase`` class run on the container's
  transformers 5.15 ``BertEncoder`` (``init_weights`` patched to a no-op,
  see make_golden.py), golden outputs committed.
* The ``relative_key`` / ``relative_key_query`` score term: **parity unpinned**
  against BERT itself -- transformers 4.11.3 is not installable here and
  transformers 5.15's BERT dropped the feature.  It is restated from the
  published 4.11.3 algorithm and cross-checked against the one surviving
  HuggingFace implementation of the same einsum
  (``Wav2Vec2BertSelfAttention``, transformers 5.15) in make_golden.py.
"""
