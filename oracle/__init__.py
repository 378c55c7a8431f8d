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
                 (``nets/ron_vgg_320.py:595-629``); ``ssd_forward``: SSD-512.
``eval_metrics`` GT matching (``tf_extended/bboxes.py:316-450``), streaming TP/FP, P/R,
                 AP VOC07/12 (``tf_extended/metrics.py:100-258``), the numpy PASCAL
                 evaluation of one class (``datasets/voc_eval.py:164-295``).
``preprocess``   ``preprocess_for_eval`` in all resize modes
                 (``preprocessing/ssd_vgg_preprocessing.py:358-425``, ``tf_image.py:141-282``).
``ron_eval_post`` post-processing of the second harness (``ron_eval.py:111-206,369-392``).

Pinning status
--------------
* ``anchors`` and ``np_post`` are pinned against outputs of the reference's own
  numpy code, imported in the build container by ``tests/golden/make_golden.py``
  (fixtures ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).
* ``eval_metrics``: the AP integrals and ``voc_eval_class`` are pinned against the reference's own numpy
  ``voc_ap`` / ``DetectorEvalPascal.voc_eval`` (fixtures g6, g7); the TF matching is **parity unpinned**.
* ``preprocess`` (TF1 bilinear resize restated from its published algorithm) and ``ron_eval_post``: **parity unpinned**.
* ``ron_forward`` / ``ssd_forward``: the VGG-16 backbone (conv1_1 .. conv5_3 with pool1..4; SSD: .. conv7 with pool5 3x3 s1 and the
  rate-6 conv6) and the conv -> inference BatchNorm -> ReLU layer ARE pinned: golden G8 = the reference's only executable conv
  arithmetic, the torch ``VGG16`` / ``vgg()`` of ``convert_pytorch_vgg.py:13-58`` (tests/test_oracle_g8.py).  The rest of the two
  forwards (fc6 / fc7, transposed conv, reverse-connection sum, inception concat + BN, heads, SSD blocks 8-12, L2-norm) and
  ``tfe_post`` restate TensorFlow-1.x graph code with no executable form in the reference: **parity unpinned** (cross-checked
  against the independent torch-CPU operators and hand-derived cases).
"""
