"""CPU: the C-ABI library loads and exports every symbol include/ron_hip.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'ron_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ron_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_exported_and_bound():
    from ron_tensorflow_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    handle = _lib.lib()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(handle, n), 'libron_hip.so does not export %s' % n
        assert n in _lib.SIGNATURES, 'no ctypes signature for %s' % n
    assert sorted(_lib.SIGNATURES) == names          # nothing bound that the header does not declare
    assert handle.ron_abi_version() == 2          # 2: ron_conv_desc.center_from, RON_DTYPE_F16X3, ron_num_grouped_launches


def test_argument_errors_are_reported_not_raised_in_c():
    """Error behaviour at the boundary: status code + message, mirrored as RonError in Python."""
    import ctypes as C
    from ron_tensorflow_amd import _lib
    handle = _lib.lib()
    rc = handle.ron_anchor_one_layer(320, 320, 0, 5, None, 0, None, 0, 64.0, 0.5, None, None, None, None)
    assert rc == -1
    assert b'bad image / feature shape' in handle.ron_last_error()
    with pytest.raises(_lib.RonError):
        _lib.check(rc)
    cfg = _lib.Config(7, 1, 320, 320, 21, 1, 0, 0)       # unknown variant: rejected before any HIP call
    h = C.c_void_p()
    assert handle.ron_create(C.byref(h), C.byref(cfg)) == -1
    assert b'unknown variant' in handle.ron_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ron_tensorflow_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.RonError):
        _lib.lib()


def test_host_anchor_function_matches_golden(golden_dir):
    import numpy as np
    from ron_tensorflow_amd.nets.ron_vgg_320 import RONNet
    g = np.load(os.path.join(golden_dir, 'g1_anchors_ron320.npz'))
    net = RONNet.__new__(RONNet)                     # anchors() needs no device
    net.params = RONNet.default_params
    for i, (y, x, h, w) in enumerate(net.anchors((320, 320))):
        assert y.shape == (net.params.feat_shapes[i] + (1,))
        for nm, arr in (('y', y), ('x', x), ('h', h), ('w', w)):
            assert np.array_equal(arr, g['%s%d' % (nm, i)]), (nm, i)


def test_host_ssd_anchor_function_matches_golden(golden_dir):
    import numpy as np
    from ron_tensorflow_amd.nets.ssd_vgg_512 import SSDNet
    g = np.load(os.path.join(golden_dir, 'g5_anchors_ssd512.npz'))
    net = SSDNet.__new__(SSDNet)
    net.params = SSDNet.default_params
    for i, (y, x, h, w) in enumerate(net.anchors((512, 512))):
        for nm, arr in (('y', y), ('x', x), ('h', h), ('w', w)):
            assert np.array_equal(arr, g['%s%d' % (nm, i)]), (nm, i)
