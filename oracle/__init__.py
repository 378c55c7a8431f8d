"""CPU oracle for the RON-320 inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``ron_tensorflow_amd``) never imports this package and raises when its HIP
library is missing.

Contents
--------
``anchors``      numpy restatement of the anchor generators
                 (reference ``nets/ron_vgg_320.py:285-355``).
``np_post``      numpy restatement of the post-processing path of
                 ``nets/np_methods.py:23-242`` plus the objectness gate of
                 ``eval_ron_network.py:227-229``.
``tfe_post``     numpy restatement of the TF post-processing variant
                 (``nets/ssd_common.py:448-589``, ``tf_extended/bboxes.py:60-302``,
                 ``nets/ron_vgg_320.py:196-256``).
``ron_forward``  fp32 NHWC restatement of the conv stack
                 (``nets/ron_vgg_320.py:378-580``) with slim semantics
                 (``nets/ron_vgg_320.py:595-629``).

Pinning status
--------------
* ``anchors`` and ``np_post`` are pinned against outputs of the reference's own
  numpy code, imported in the build container by ``tests/golden/make_golden.py``
  (fixtures ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).
* ``ron_forward`` and ``tfe_post`` restate TensorFlow-1.x graph code.  TensorFlow
  is not installable here and the reference ships no tests, golden vectors or
  checkpoints for it, so for these two: **parity unpinned** (cross-checked only
  against the independent torch-CPU operators and hand-derived cases).
"""
