"""CPU: TensorFlow V2 checkpoint (tensor bundle) reader / writer and the reference's restore rules
(tf_utils.py:184-243) -- format restated from TensorFlow's published layout, known answers where they exist."""
import os
import struct

import numpy as np
import pytest

from ron_tensorflow_amd import checkpoint as ck


def test_crc32c_known_answers():
    assert ck.crc32c(b'123456789') == 0xE3069283                     # the standard CRC-32C check value
    assert ck.crc32c(b'') == 0
    assert ck.crc32c(bytes(32)) == 0x8A9136AA                          # RFC 3720 B.4: 32 bytes of zeros
    assert ck.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43                 # ... 32 bytes of ones
    assert ck.crc32c(bytes(range(32))) == 0x46DD794E                   # ... incrementing
    big = bytes(range(256)) * 64                                       # >= 4096 bytes: the C routine of libron_hip.so
    ref = 0
    for i in range(0, len(big), 1000):                                 # chained pure-Python pieces (< 4096 bytes each)
        ref = ck.crc32c(big[i:i + 1000], ref)
    assert ck.crc32c(big) == ref
    # LevelDB's mask: leveldb/util/crc32c_test.cc
    c = ck.crc32c(b'foo')
    assert ck.mask_crc(c) != c and ck.unmask_crc(ck.mask_crc(c)) == c
    assert ck.mask_crc(0) == 0xa282ead8


def test_varint_and_entry_roundtrip():
    for v in (0, 1, 127, 128, 300, 2 ** 31, 2 ** 40 + 5):
        assert ck._get_varint(ck._put_varint(v), 0) == (v, len(ck._put_varint(v)))
    assert ck._put_varint(300) == b'\xac\x02'                          # protobuf documentation example
    e = ck.BundleEntry(1, (3, 3, 64, 128), 0, 123456, 3 * 3 * 64 * 128 * 4, 0xDEADBEEF)
    p = ck._parse_entry('x', ck._encode_entry(e))
    assert (p.dtype, p.shape, p.shard_id, p.offset, p.size, p.crc32c) == (1, (3, 3, 64, 128), 0, 123456, 294912, 0xDEADBEEF)


def _tensors(rs):
    t = {'ron_320_vgg/conv1/conv1_1/weights': rs.randn(3, 3, 3, 64).astype(np.float32),
         'ron_320_vgg/conv1/conv1_1/biases': rs.randn(64).astype(np.float32),
         'ron_320_vgg/conv1/conv1_1/weights/Momentum': rs.randn(3, 3, 3, 64).astype(np.float32),
         'global_step': np.array(120000, np.int64),
         'ron_320_vgg/half': rs.randn(5, 7).astype(np.float16),
         'ron_320_vgg/flags': np.array([True, False, True]),
         'ron_320_vgg/empty': np.zeros((0, 4), np.float32)}
    for i in range(300):                                               # enough keys for several 4 KB index blocks
        t['ron_320_vgg/reverse_module/block%d/Conv2d_%03d/BatchNorm/moving_variance' % (i % 4, i)] = rs.rand(17).astype(np.float32)
    return t


def test_write_read_roundtrip_multi_block(tmp_path):
    rs = np.random.RandomState(0)
    t = _tensors(rs)
    prefix = ck.write_checkpoint(str(tmp_path / 'model.ckpt-120000'), t)
    assert os.path.isfile(prefix + '.index') and os.path.isfile(prefix + '.data-00000-of-00001')
    data = open(prefix + '.index', 'rb').read()
    assert struct.unpack('<Q', data[-8:])[0] == 0xdb4775248b80fb57
    r = ck.TensorBundleReader(prefix, verify_crc=True)
    assert sorted(r.keys()) == sorted(t)
    assert r.keys() == sorted(t, key=lambda s: s.encode())             # table order = bytewise key order
    for k, v in t.items():
        got = r.get_tensor(k)
        assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v), k
    assert r.shape_map()['ron_320_vgg/conv1/conv1_1/weights'] == (3, 3, 3, 64)
    assert ck.latest_checkpoint(str(tmp_path)) == prefix
    with pytest.raises(KeyError):
        r.get_tensor('nope')


def test_corruption_is_detected(tmp_path):
    rs = np.random.RandomState(1)
    prefix = ck.write_checkpoint(str(tmp_path / 'm'), {'a': rs.randn(2000).astype(np.float32), 'b': rs.randn(4).astype(np.float32)})
    idx = bytearray(open(prefix + '.index', 'rb').read())
    idx[10] ^= 0x40
    open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(ValueError, match='CRC32C'):
        ck.read_index(prefix)
    idx[10] ^= 0x40
    open(prefix + '.index', 'wb').write(bytes(idx))
    dat = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
    dat[100] ^= 1
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(dat))
    r = ck.TensorBundleReader(prefix, verify_crc=True)
    with pytest.raises(ValueError, match='CRC32C'):
        r.get_tensor('a')
    assert r.get_tensor('b').shape == (4,)
    open(prefix + '.index', 'wb').write(b'not a table' * 10)
    with pytest.raises(ValueError, match='magic'):
        ck.read_index(prefix)


def test_restore_rules_of_the_reference(tmp_path):
    """tf_utils.get_init_fn: exclusion by scope prefix, model-scope remap, missing variables, exact shapes."""
    rs = np.random.RandomState(2)
    ckpt = {'vgg_16/conv1/conv1_1/weights': rs.randn(3, 3, 3, 64).astype(np.float32),
            'vgg_16/conv1/conv1_1/biases': rs.randn(64).astype(np.float32),
            'vgg_16/fc8/weights': rs.randn(1, 1, 4096, 1000).astype(np.float32)}
    prefix = ck.write_checkpoint(str(tmp_path / 'vgg_16.ckpt'), ckpt)
    variables = [('ron_320_vgg/conv1/conv1_1/weights', (3, 3, 3, 64)), ('ron_320_vgg/conv1/conv1_1/biases', (64,)),
                 ('ron_320_vgg/reverse_module/block7/Conv2d_1_3x3/weights', (3, 3, 512, 40))]
    with pytest.raises(KeyError, match='not found in checkpoint'):
        ck.load_checkpoint(prefix, variables, checkpoint_model_scope='vgg_16')
    got = ck.load_checkpoint(prefix, variables, checkpoint_model_scope='vgg_16', ignore_missing_vars=True)
    assert sorted(got) == ['ron_320_vgg/conv1/conv1_1/biases', 'ron_320_vgg/conv1/conv1_1/weights']
    assert np.array_equal(got['ron_320_vgg/conv1/conv1_1/weights'], ckpt['vgg_16/conv1/conv1_1/weights'])
    got = ck.load_checkpoint(str(tmp_path), variables, checkpoint_model_scope='vgg_16',
                             checkpoint_exclude_scopes='ron_320_vgg/reverse_module, ron_320_vgg/conv1/conv1_1/biases')
    assert list(got) == ['ron_320_vgg/conv1/conv1_1/weights']
    with pytest.raises(ValueError, match='shape'):
        ck.load_checkpoint(prefix, [('ron_320_vgg/conv1/conv1_1/biases', (128,))], checkpoint_model_scope='vgg_16')


def test_full_model_roundtrip_through_a_checkpoint(tmp_path):
    """Every variable of ron_net_reducedfc: synthetic weights -> V2 checkpoint -> restore == the dict we started from."""
    from ron_tensorflow_amd import weights as W
    w = W.synthetic_weights('reducedfc', seed=3)
    prefix = ck.write_checkpoint(str(tmp_path / 'model.ckpt-1'), w)
    variables = [(k, v.shape) for k, v in w.items()]
    got = ck.load_checkpoint(prefix, variables, verify_crc=True)
    assert sorted(got) == sorted(w)
    for k in w:
        assert np.array_equal(got[k], w[k])
