"""Golden G8 (the reference's torch VGG16 run by tests/golden/make_golden.py::g8_vgg_backbone) and the check both the oracle
tests (CPU) and the device tests (GPU) apply to a tensor against it."""
import os

import numpy as np

from oracle import synth

G8 = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g8_vgg_backbone.npz'))


def check_tensor(size, name, a, tol=1e-5, sum_tol=1e-6):
    """a [1,H,W,C] against G8's record of module `name` at input `size`: shape; the sampled values within `tol` of the tensor's
    largest value; the float64 sum and sum of squares over the WHOLE tensor within `sum_tol`, relative (every element enters them:
    a wrong border row or channel anywhere moves them by far more - measured deviations of correct fp32 results: <= 6e-8)."""
    key = '%d/%s' % (size, name)
    assert tuple(G8[key + '/shape']) == a.shape, (key, a.shape)
    iy, ix = synth.g8_sample_index(a.shape[1]), synth.g8_sample_index(a.shape[2])
    want = G8[key + '/sample']
    got = a[:, iy][:, :, ix]
    err = float(np.abs(got - want).max()) / float(np.abs(want).max())
    assert err <= tol, '%s: sampled values off by %.3g of the tensor scale' % (key, err)
    s1, s2 = G8[key + '/sum']
    d1 = abs(a.sum(dtype=np.float64) - s1) / s1
    d2 = abs((a.astype(np.float64) ** 2).sum() - s2) / s2
    assert d1 <= sum_tol and d2 <= 2 * sum_tol, '%s: whole-tensor sums off by %.3g / %.3g' % (key, d1, d2)
    return err
