"""CPU: the checkpoint reader against files it did NOT write.

`ron_tensorflow_amd/checkpoint.py` restates the tensor-bundle format and was, until round 6, only ever checked against its own
writer.  TensorFlow cannot run here, so this file carries a SECOND, independent encoder of what `tf.train.Saver(write_version=2)`
(ron_net.py:395-398) puts on disk, built from different parts than the package's writer:

  * the two bundle messages are serialised by the `protobuf` runtime from descriptors declared below after TensorFlow's published
    .proto files (tensor_bundle.proto, tensor_shape.proto, versions.proto, types.proto) - not by the package's hand-written varint code;
  * the table is assembled the way tensorflow/core/lib/io/table_builder.cc does it, including the things the package's writer does
    not do: index-block keys that are SHORTENED SEPARATORS between blocks (FindShortestSeparator / FindShortSuccessor) instead of a
    block's last key, long shared prefixes over many keys with restart interval 16, two data shards, a VersionDef with producer and
    min_consumer, an int64 scalar `global_step`, Momentum slot variables, and TF's 256 KB block size next to a small one.

The reader must return every tensor bit-exactly with CRCs verified, and the reference's restore rules (tf_utils.py:186-244) must work
on such a file."""
import os
import struct

import numpy as np
import pytest

from ron_tensorflow_amd import checkpoint as ck

pb = pytest.importorskip('google.protobuf')
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory  # noqa: E402

F = descriptor_pb2.FieldDescriptorProto


def _messages():
    """BundleHeaderProto / BundleEntryProto / TensorShapeProto / VersionDef / TensorSliceProto as published by TensorFlow r1.x."""
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = 'tf_bundle_for_test.proto', 'tftest', 'proto3'

    def msg(name, fields, nested=None):
        m = fd.message_type.add() if nested is None else nested.nested_type.add()
        m.name = name
        for (fname, num, ftype, label, tname) in fields:
            f = m.field.add()
            f.name, f.number, f.type, f.label = fname, num, ftype, label
            if tname:
                f.type_name = tname
        return m

    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    shape = msg('TensorShapeProto', [('dim', 2, F.TYPE_MESSAGE, REP, '.tftest.TensorShapeProto.Dim'), ('unknown_rank', 3, F.TYPE_BOOL, OPT, '')])
    msg('Dim', [('size', 1, F.TYPE_INT64, OPT, ''), ('name', 2, F.TYPE_STRING, OPT, '')], nested=shape)
    msg('VersionDef', [('producer', 1, F.TYPE_INT32, OPT, ''), ('min_consumer', 2, F.TYPE_INT32, OPT, ''), ('bad_consumers', 3, F.TYPE_INT32, REP, '')])
    sl = msg('TensorSliceProto', [('extent', 1, F.TYPE_MESSAGE, REP, '.tftest.TensorSliceProto.Extent')])
    msg('Extent', [('start', 1, F.TYPE_INT64, OPT, ''), ('length', 2, F.TYPE_INT64, OPT, '')], nested=sl)
    msg('BundleHeaderProto', [('num_shards', 1, F.TYPE_INT32, OPT, ''), ('endianness', 2, F.TYPE_INT32, OPT, ''),
                              ('version', 3, F.TYPE_MESSAGE, OPT, '.tftest.VersionDef')])
    msg('BundleEntryProto', [('dtype', 1, F.TYPE_INT32, OPT, ''), ('shape', 2, F.TYPE_MESSAGE, OPT, '.tftest.TensorShapeProto'),
                             ('shard_id', 3, F.TYPE_INT32, OPT, ''), ('offset', 4, F.TYPE_INT64, OPT, ''), ('size', 5, F.TYPE_INT64, OPT, ''),
                             ('crc32c', 6, F.TYPE_FIXED32, OPT, ''), ('slices', 7, F.TYPE_MESSAGE, REP, '.tftest.TensorSliceProto')])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName('tftest.' + n)) for n in ('BundleHeaderProto', 'BundleEntryProto')}


DT = {np.dtype(np.float32): 1, np.dtype(np.int32): 3, np.dtype(np.int64): 9, np.dtype(np.float16): 19}    # types.proto


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _shortest_separator(start, limit):
    """table_builder.cc FindShortestSeparator: shorten `start` to a string in [start, limit)."""
    n = min(len(start), len(limit))
    d = 0
    while d < n and start[d] == limit[d]:
        d += 1
    if d >= n:
        return start                                       # one is a prefix of the other
    b = start[d]
    if b < 0xFF and b + 1 < limit[d]:
        return start[:d] + bytes([b + 1])
    return start


def _short_successor(key):
    """table_builder.cc FindShortSuccessor: the first byte that can be incremented, incremented, the rest dropped."""
    for i, b in enumerate(key):
        if b != 0xFF:
            return key[:i] + bytes([b + 1])
    return key


class _Block(object):
    """One table block, block_builder.cc: entries <shared><non_shared><value_len><key delta><value>, a restart point (shared = 0)
    every `interval` entries, then the restart offsets and their count as fixed32."""

    def __init__(self, interval):
        self.interval, self.body, self.restarts, self.n_since, self.last = interval, bytearray(), [0], 0, b''

    def add(self, key, value):
        shared = 0
        if self.n_since < self.interval:
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.body))
            self.n_since = 0
        self.body += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last = key
        self.n_since += 1

    def size_estimate(self):
        return len(self.body) + 4 * len(self.restarts) + 4

    def finish(self):
        out = bytes(self.body)
        for r in self.restarts:
            out += struct.pack('<I', r)
        return out + struct.pack('<I', len(self.restarts))


def _emit(f_out, contents):
    off = len(f_out)
    trailer = b'\x00'                                                       # kNoCompression
    crc = ck.mask_crc(ck.crc32c(contents + trailer))
    f_out += contents + trailer + struct.pack('<I', crc)
    return _varint(off) + _varint(len(contents))


def write_like_tensorflow(prefix, tensors, num_shards, block_size, shard_of):
    """BundleWriter::Add per tensor in key order (data appended to the shard `shard_of(name)` picks), then Finish(): the index table."""
    M = _messages()
    names = sorted(tensors, key=lambda s: s.encode())
    shard_bytes = [bytearray() for _ in range(num_shards)]
    rows = []
    hdr = M['BundleHeaderProto']()
    hdr.num_shards = num_shards
    hdr.version.producer = 1                                                # kTensorBundleVersion
    hdr.version.min_consumer = 1                                            # (0 in stock TF: proto3 would omit it; 1 is still readable)
    rows.append((b'', hdr.SerializeToString()))
    for name in names:
        a = np.asarray(tensors[name])
        raw = a.astype(a.dtype.newbyteorder('<')).tobytes()
        sh = shard_of(name)
        e = M['BundleEntryProto']()
        e.dtype = DT[a.dtype]
        e.shape.SetInParent()                                               # a scalar's shape is present and empty
        for d in a.shape:
            e.shape.dim.add().size = d
        e.shard_id, e.offset, e.size = sh, len(shard_bytes[sh]), len(raw)
        e.crc32c = ck.mask_crc(ck.crc32c(raw))
        shard_bytes[sh] += raw
        rows.append((name.encode(), e.SerializeToString()))
    out = bytearray()
    index = _Block(1)                                                       # index_block(options with block_restart_interval = 1)
    block = _Block(16)
    pending = None                                                          # (last key of the flushed block, its handle)
    for key, value in rows:
        if pending is not None:
            index.add(_shortest_separator(pending[0], key), pending[1])
            pending = None
        block.add(key, value)
        if block.size_estimate() >= block_size:
            pending = (key, _emit(out, block.finish()))
            block = _Block(16)
    if block.body:
        pending = (block.last, _emit(out, block.finish()))
    if pending is not None:
        index.add(_short_successor(pending[0]), pending[1])
    meta = _emit(out, _Block(16).finish())                                  # empty metaindex block
    idx = _emit(out, index.finish())
    footer = meta + idx
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<II', 0x8b80fb57, 0xdb477524)
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out) + footer)
    for i, b in enumerate(shard_bytes):
        with open('%s.data-%05d-of-%05d' % (prefix, i, num_shards), 'wb') as f:
            f.write(bytes(b))
    return len(names)


def _variables(rs):
    """A checkpoint as a training run of the reference leaves it: model variables, their Momentum slots (train_ron_network.py uses
    MomentumOptimizer), BatchNorm statistics, global_step."""
    t = {'global_step': np.array(120000, np.int64)}
    for b in range(4, 8):
        for part in ('reverse_conv_left', 'objectness', 'reverse_inception1/Branch_0/Conv2d_3x3', 'reverse_inception2/Conv2d_pred_3x3'):
            base = 'ron_320_vgg/reverse_module/block%d_%s' % (b, part)
            t[base + '/weights'] = rs.randn(3, 3, 8, 16).astype(np.float32)
            t[base + '/weights/Momentum'] = rs.randn(3, 3, 8, 16).astype(np.float32)
            for stat in ('beta', 'gamma', 'moving_mean', 'moving_variance'):
                t[base + '/BatchNorm/' + stat] = rs.rand(16).astype(np.float32)
    for i in (1, 2):
        t['ron_320_vgg/conv1/conv1_%d/weights' % i] = rs.randn(3, 3, 3 if i == 1 else 64, 64).astype(np.float32)
        t['ron_320_vgg/conv1/conv1_%d/biases' % i] = rs.randn(64).astype(np.float32)
    t['ron_320_vgg/fc7/weights'] = rs.randn(1, 1, 96, 96).astype(np.float32)
    return t


@pytest.mark.parametrize('block_size', [256 << 10, 1500, 300])              # TF's own, and sizes that force 3 / 40+ data blocks
@pytest.mark.parametrize('num_shards', [1, 2])
def test_reader_on_a_file_written_the_tensorflow_way(tmp_path, num_shards, block_size):
    t = _variables(np.random.RandomState(7))
    prefix = str(tmp_path / 'model.ckpt-120000')
    n = write_like_tensorflow(prefix, t, num_shards, block_size, lambda name: (len(name) + name.count('/')) % num_shards)
    if num_shards == 2:
        assert os.path.getsize(prefix + '.data-00000-of-00002') > 0 and os.path.getsize(prefix + '.data-00001-of-00002') > 0
    r = ck.TensorBundleReader(prefix, verify_crc=True)
    assert r.header['num_shards'] == num_shards and r.header['version'] == {'producer': 1, 'min_consumer': 1}
    assert len(r.keys()) == n and r.keys() == sorted(t, key=lambda s: s.encode())
    for name, a in t.items():
        got = r.get_tensor(name)
        assert got.dtype == a.dtype and got.shape == a.shape and np.array_equal(got, a), name
    g = r.get_tensor('global_step')
    assert g.shape == () and g.dtype == np.int64 and int(g) == 120000
    if num_shards == 2:
        assert {r.entries[k].shard_id for k in r.keys()} == {0, 1}


def test_restore_rules_on_a_tensorflow_layout_file(tmp_path):
    """tf_utils.get_init_fn (tf_utils.py:186-244) on such a file: scope remap, exclusions, slot variables and global_step ignored because
    the model does not ask for them."""
    t = _variables(np.random.RandomState(8))
    renamed = {k.replace('ron_320_vgg', 'vgg_16'): v for k, v in t.items()}
    prefix = str(tmp_path / 'vgg_16.ckpt')
    write_like_tensorflow(prefix, renamed, 2, 700, lambda name: len(name) % 2)
    want = [(k, v.shape) for k, v in t.items() if k != 'global_step' and not k.endswith('/Momentum')]
    got = ck.load_checkpoint(prefix, want, model_name='ron_320_vgg', checkpoint_model_scope='vgg_16',
                             checkpoint_exclude_scopes='ron_320_vgg/reverse_module', verify_crc=True)
    assert sorted(got) == sorted(k for k, _ in want if not k.startswith('ron_320_vgg/reverse_module'))
    for k, a in got.items():
        assert np.array_equal(a, t[k])
    with pytest.raises(KeyError):
        ck.load_checkpoint(prefix, want + [('ron_320_vgg/fc6/weights', (7, 7, 512, 4096))], model_name='ron_320_vgg', checkpoint_model_scope='vgg_16')


def test_a_newer_bundle_version_is_refused(tmp_path):
    """tensor_bundle.cc checks the header's VersionDef against kTensorBundleVersion = 1: a file that asks for a newer consumer is refused."""
    M = _messages()
    h = M['BundleHeaderProto']()
    h.num_shards, h.version.producer, h.version.min_consumer = 1, 3, 2
    blk = _Block(16)
    blk.add(b'', h.SerializeToString())
    out = bytearray()
    handle = _emit(out, blk.finish())
    index = _Block(1)
    index.add(b'\x00', handle)
    meta = _emit(out, _Block(16).finish())
    idx = _emit(out, index.finish())
    footer = meta + idx
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
    prefix = str(tmp_path / 'new.ckpt')
    open(prefix + '.index', 'wb').write(bytes(out) + footer)
    with pytest.raises(NotImplementedError):
        ck.TensorBundleReader(prefix)


def test_separator_helpers_are_the_leveldb_ones():
    assert _shortest_separator(b'abcdefg', b'abzzz') == b'abd'
    assert _shortest_separator(b'abc', b'abcd') == b'abc'                   # a prefix: unchanged
    assert _shortest_separator(b'ab\xff', b'ac') == b'ab\xff'               # nothing to shorten (b'ab' + 1 would equal the limit's byte)
    assert _short_successor(b'\xff\xffabc') == b'\xff\xffb' and _short_successor(b'global_step') == b'h'
