"""CPU oracle for the adaptive-depth U-Net hot path.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy restatement of the maths the
reference executes through TensorFlow/Keras (which is not installed in the build
container nor on the GPU box).  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it; the product package
(``adunet_amd``) never does, and fails loudly when its HIP library is missing.

PARITY PINNING STATUS
---------------------
* Structure (layer graph, every output shape, every parameter count) is pinned
  against the 15 ``model.summary()`` dumps the reference ships
  (``Super_resolution/experiments/*/model_summary/*.txt``), condensed to
  ``tests/golden/model_summaries.json`` by ``tests/golden/make_summary_fixture.py``.
* Depth heuristics are pinned by hand-derived known answers.
* Numerics: **parity unpinned** against TensorFlow itself -- the reference has no
  tests, no golden tensors and no checkpoints, and TF/Keras/OpenCV cannot be
  imported here (ordinary ImportError, nothing was denied).  Every op below is
  instead cross-checked against an independent second implementation
  (PyTorch-CPU) in ``tests/test_oracle_vs_torch.py``; TF-specific deltas
  (LN eps 1e-3, Keras Adam epsilon placement, inclusive clip gradient, HWIO
  kernel layout, float32 ceil for resize sizes, ScaleAndTranslate span rules)
  are restated from the cited reference call sites and the published TF kernels.
"""
