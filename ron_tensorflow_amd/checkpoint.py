"""TensorFlow V2 checkpoint ("tensor bundle") import and export without TensorFlow.

The reference saves with ``tf.train.Saver(write_version=2)`` (ron_net.py:395-398) and restores through
``slim.assign_from_checkpoint_fn`` with an optional scope remap and exclusion list (tf_utils.py:184-243,
eval_ron_network.py:222,346-361).  The weight contract of this repository is a dict {TF variable name: ndarray}
(DESIGN.md 1), so importing a checkpoint is: read the bundle, apply the same remap / exclusion rules, hand the dict to
``RONNet.load_weights``.

TensorFlow is a dependency of the reference that is not vendored in it (r1.x); the format restated here is its published
one (tensorflow/core/util/tensor_bundle + tensorflow/core/lib/io/table, the LevelDB table format):

  <prefix>.index                an SSTable: sorted string keys -> protobuf values, in prefix-compressed blocks, each block
                                followed by a 1-byte compression type and a masked CRC32C; a 48-byte footer holds the
                                handles of the (empty) metaindex block and of the index block and the magic number.
                                key ""   -> BundleHeaderProto {1: num_shards, 2: endianness, 3: version}
                                key name -> BundleEntryProto  {1: dtype, 2: TensorShapeProto, 3: shard_id, 4: offset,
                                                               5: size, 6: crc32c (fixed32, masked), 7: slices}
  <prefix>.data-SSSSS-of-NNNNN  the tensors' raw little-endian bytes at [offset, offset + size)

Only what such checkpoints contain is supported: uncompressed index blocks (what BundleWriter emits), little endian,
dense numeric tensors, any number of data shards, no partitioned-variable slices.  Anything else raises with the reason.
tests/test_checkpoint_tf_layout.py reads files assembled by a second, independent encoder (protobuf runtime + the table builder's
rules, shortened separator keys, two shards): no TensorFlow-written file exists in this environment.
"""
import os
import struct

import numpy as np

_TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_ENUM = {np.dtype(v).newbyteorder('<').str: k for k, v in _DTYPES.items()}


# --------------------------------------------------------------------------- #
# CRC32C (Castagnoli), masked the LevelDB way
# --------------------------------------------------------------------------- #
def _make_crc_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC_TABLE = _make_crc_table()


def crc32c(data, crc=0):
    data = bytes(data)
    if len(data) >= 4096:                     # tensors: the slicing-by-8 C routine of libron_hip.so (host code, no GPU needed)
        from . import _lib
        return int(_lib.lib().ron_crc32c(data, len(data), crc))
    crc ^= 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in bytes(data):
        crc = tab[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def mask_crc(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked):
    rot = (masked - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# --------------------------------------------------------------------------- #
# varints / protobuf wire format (just what the two bundle messages use)
# --------------------------------------------------------------------------- #
def _get_varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError('malformed varint')


def _put_varint(v):
    if v < 0:
        v += 1 << 64
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_message(buf):
    """-> list of (field number, wire type, value); value = int (varint / fixed) or bytes (length delimited)."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        out.append((field, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _parse_shape(buf):
    dims = []
    for field, _, v in _parse_message(buf):
        if field == 2:                                   # TensorShapeProto.Dim
            size = 0
            for f2, _, v2 in _parse_message(v):
                if f2 == 1:
                    size = _signed64(v2)
            dims.append(size)
        elif field == 3 and v:
            raise ValueError('tensor of unknown rank in a checkpoint')
    return tuple(dims)


def _encode_shape(shape):
    out = b''
    for d in shape:
        dim = b'\x08' + _put_varint(int(d))              # field 1, varint
        out += b'\x12' + _put_varint(len(dim)) + dim     # field 2, length delimited
    return out


class BundleEntry(object):
    __slots__ = ('dtype', 'shape', 'shard_id', 'offset', 'size', 'crc32c')

    def __init__(self, dtype, shape, shard_id, offset, size, crc):
        self.dtype, self.shape, self.shard_id, self.offset, self.size, self.crc32c = dtype, shape, shard_id, offset, size, crc


def _parse_entry(name, buf):
    dtype = shard = offset = size = crc = 0
    shape = ()
    for field, _, v in _parse_message(buf):
        if field == 1:
            dtype = v
        elif field == 2:
            shape = _parse_shape(v)
        elif field == 3:
            shard = v
        elif field == 4:
            offset = v
        elif field == 5:
            size = v
        elif field == 6:
            crc = v
        elif field == 7:
            raise NotImplementedError('%s is a partitioned variable (tensor slices): not supported' % name)
    return BundleEntry(dtype, shape, shard, offset, size, crc)


def _encode_entry(e):
    shp = _encode_shape(e.shape)
    out = b'\x08' + _put_varint(e.dtype) + b'\x12' + _put_varint(len(shp)) + shp
    if e.shard_id:
        out += b'\x18' + _put_varint(e.shard_id)
    if e.offset:
        out += b'\x20' + _put_varint(e.offset)
    out += b'\x28' + _put_varint(e.size) + b'\x35' + struct.pack('<I', e.crc32c)
    return out


# --------------------------------------------------------------------------- #
# the SSTable of <prefix>.index
# --------------------------------------------------------------------------- #
def _read_block(data, offset, size, what):
    if offset + size + 5 > len(data):
        raise ValueError('%s block runs past the end of the index file' % what)
    block = data[offset:offset + size]
    ctype = data[offset + size]
    stored = struct.unpack_from('<I', data, offset + size + 1)[0]
    if unmask_crc(stored) != crc32c(data[offset:offset + size + 1]):
        raise ValueError('%s block: CRC32C mismatch (corrupt index file)' % what)
    if ctype == 1:
        raise NotImplementedError('snappy-compressed index block: BundleWriter does not emit these; re-save the checkpoint '
                                  'or convert it with TensorFlow')
    if ctype != 0:
        raise ValueError('%s block: unknown compression type %d' % (what, ctype))
    return block


def _block_entries(block):
    """Prefix-compressed entries of one block -> list of (key bytes, value bytes)."""
    if len(block) < 4:
        raise ValueError('block too small')
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    if end < 0:
        raise ValueError('bad restart array')
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key):
            raise ValueError('bad shared-prefix length')
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def read_index(prefix):
    """<prefix>.index -> (header dict, {tensor name: BundleEntry}) in key order."""
    path = prefix + '.index'
    with open(path, 'rb') as f:
        data = f.read()
    if len(data) < 48:
        raise ValueError('%s: too small for an SSTable footer' % path)
    footer = data[-48:]
    if struct.unpack_from('<Q', footer, 40)[0] != _TABLE_MAGIC:
        raise ValueError('%s: not a TensorFlow V2 checkpoint index (bad table magic)' % path)
    pos = 0
    _, pos = _get_varint(footer, pos)                    # metaindex handle (unused)
    _, pos = _get_varint(footer, pos)
    idx_off, pos = _get_varint(footer, pos)
    idx_size, pos = _get_varint(footer, pos)
    header, entries = None, {}
    for _, handle in _block_entries(_read_block(data, idx_off, idx_size, 'index')):
        off, p = _get_varint(handle, 0)
        size, p = _get_varint(handle, p)
        for key, value in _block_entries(_read_block(data, off, size, 'data')):
            if key == b'':
                header = {f: v for f, _, v in _parse_message(value)}
            else:
                name = key.decode('utf-8')
                entries[name] = _parse_entry(name, value)
    if header is None:
        raise ValueError('%s: no bundle header entry' % path)
    if header.get(2, 0) != 0:
        raise NotImplementedError('big-endian checkpoint')
    # VersionDef {1: producer, 2: min_consumer, 3: bad_consumers}: tensor_bundle.cc refuses a file whose min_consumer is above its
    # kTensorBundleVersion (1), so does this reader (it implements version 1)
    version = {'producer': 0, 'min_consumer': 0}
    if isinstance(header.get(3), bytes):
        for f, _, v in _parse_message(header[3]):
            if f == 1:
                version['producer'] = v
            elif f == 2:
                version['min_consumer'] = v
    if version['min_consumer'] > 1:
        raise NotImplementedError('%s: written for tensor-bundle consumers >= %d, this reader implements version 1' % (path, version['min_consumer']))
    return {'num_shards': header.get(1, 1), 'version': version}, entries


class TensorBundleReader(object):
    """``tf.train.load_checkpoint``-like access: ``keys()``, ``has_tensor``, ``get_tensor``, ``shape_map``."""

    def __init__(self, prefix, verify_crc=False):
        self.prefix, self.verify_crc = prefix, verify_crc
        self.header, self.entries = read_index(prefix)

    def keys(self):
        return list(self.entries)

    def has_tensor(self, name):
        return name in self.entries

    def shape_map(self):
        return {k: e.shape for k, e in self.entries.items()}

    def _data_path(self, shard):
        return '%s.data-%05d-of-%05d' % (self.prefix, shard, self.header['num_shards'])

    def get_tensor(self, name):
        e = self.entries.get(name)
        if e is None:
            raise KeyError('Tensor %s not found in checkpoint %s' % (name, self.prefix))
        if e.dtype not in _DTYPES:
            raise NotImplementedError('%s: dtype enum %d is not a dense numeric type' % (name, e.dtype))
        dt = np.dtype(_DTYPES[e.dtype])
        count = int(np.prod(e.shape, dtype=np.int64)) if e.shape else 1
        if count * dt.itemsize != e.size:
            raise ValueError('%s: %d bytes stored for shape %r of %s' % (name, e.size, e.shape, dt))
        with open(self._data_path(e.shard_id), 'rb') as f:
            f.seek(e.offset)
            raw = f.read(e.size)
        if len(raw) != e.size:
            raise ValueError('%s: data file is truncated' % name)
        if self.verify_crc and unmask_crc(e.crc32c) != crc32c(raw):
            raise ValueError('%s: CRC32C mismatch in the data file' % name)
        return np.frombuffer(raw, dtype=dt.newbyteorder('<')).reshape(e.shape).astype(dt, copy=True)


# --------------------------------------------------------------------------- #
# restore with the reference's rules
# --------------------------------------------------------------------------- #
def latest_checkpoint(model_dir):
    """tf.train.latest_checkpoint: the prefix named by ``model_checkpoint_path`` in <dir>/checkpoint, or None."""
    state = os.path.join(model_dir, 'checkpoint')
    if not os.path.isfile(state):
        return None
    with open(state) as f:
        for line in f:
            if line.startswith('model_checkpoint_path:'):
                p = line.split(':', 1)[1].strip().strip('"')
                return p if os.path.isabs(p) else os.path.join(model_dir, p)
    return None


def load_checkpoint(checkpoint_path, variables, model_name='ron_320_vgg', checkpoint_model_scope=None,
                    checkpoint_exclude_scopes=None, ignore_missing_vars=False, verify_crc=False):
    """The restore of tf_utils.get_init_fn (tf_utils.py:184-243) as a dict for ``RONNet.load_weights``.

    variables: [(model variable name, shape)] (``RONNet.variables()``).  A variable whose name starts with one of
    ``checkpoint_exclude_scopes`` (comma separated string or list) is skipped; the checkpoint key of a variable is its
    name with ``model_name`` replaced by ``checkpoint_model_scope`` (when given); a key missing from the checkpoint is an
    error unless ``ignore_missing_vars``; shapes must match exactly (the Saver is created with reshape=False)."""
    if os.path.isdir(checkpoint_path):
        prefix = latest_checkpoint(checkpoint_path)
        if prefix is None:
            raise IOError('no checkpoint state file in %s' % checkpoint_path)
    else:
        prefix = checkpoint_path
    reader = TensorBundleReader(prefix, verify_crc=verify_crc)
    exclusions = checkpoint_exclude_scopes or []
    if isinstance(exclusions, str):
        exclusions = [s.strip() for s in exclusions.split(',') if s.strip()]
    out = {}
    for name, shape in variables:
        if any(name.startswith(ex) for ex in exclusions):
            continue
        key = name.replace(model_name, checkpoint_model_scope) if checkpoint_model_scope is not None else name
        if not reader.has_tensor(key):
            if ignore_missing_vars:
                continue
            raise KeyError('Tensor %s not found in checkpoint %s' % (key, prefix))
        t = reader.get_tensor(key)
        if tuple(t.shape) != tuple(shape):
            raise ValueError('%s: checkpoint shape %r != model shape %r' % (key, tuple(t.shape), tuple(shape)))
        out[name] = t
    return out


# --------------------------------------------------------------------------- #
# writer (single shard): export of a weight dict, and the round-trip partner of the reader in the tests
# --------------------------------------------------------------------------- #
class _BlockBuilder(object):
    def __init__(self, restart_interval):
        self.interval, self.buf, self.restarts, self.count, self.last = restart_interval, bytearray(), [0], 0, b''

    def add(self, key, value):
        shared = 0
        if self.count % self.interval == 0:
            if self.count:
                self.restarts.append(len(self.buf))
        else:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))


def _append_block(out, contents):
    handle = (len(out), len(contents))
    out += contents + b'\x00' + struct.pack('<I', mask_crc(crc32c(contents + b'\x00')))
    return handle


def write_checkpoint(prefix, tensors, block_size=4096, write_state=True):
    """dict {name: ndarray} -> <prefix>.index + <prefix>.data-00000-of-00001 (+ a ``checkpoint`` state file)."""
    names = sorted(tensors, key=lambda s: s.encode('utf-8'))
    entries, offset = [], 0
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        for name in names:
            a = np.asarray(tensors[name], order='C')          # (ascontiguousarray would turn a scalar into shape (1,))
            code = _DTYPE_ENUM.get(a.dtype.newbyteorder('<').str)
            if code is None:
                raise NotImplementedError('%s: dtype %s cannot be stored' % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder('<'), copy=False).tobytes()
            f.write(raw)
            entries.append((name.encode('utf-8'), _encode_entry(BundleEntry(code, a.shape, 0, offset, len(raw), mask_crc(crc32c(raw))))))
            offset += len(raw)
    header = b'\x08\x01' + b'\x1a\x02\x08\x01'           # num_shards 1, little endian (default), version {producer 1}
    out = bytearray()
    index = _BlockBuilder(1)
    block, last_key = _BlockBuilder(16), None
    for key, value in [(b'', header)] + entries:
        block.add(key, value)
        last_key = key
        if len(block.buf) >= block_size:
            off, size = _append_block(out, block.finish())
            index.add(last_key, _put_varint(off) + _put_varint(size))
            block, last_key = _BlockBuilder(16), None
    if block.count:
        off, size = _append_block(out, block.finish())
        index.add(last_key, _put_varint(off) + _put_varint(size))
    meta = _append_block(out, _BlockBuilder(1).finish())
    idx = _append_block(out, index.finish())
    footer = _put_varint(meta[0]) + _put_varint(meta[1]) + _put_varint(idx[0]) + _put_varint(idx[1])
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', _TABLE_MAGIC)
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out) + footer)
    if write_state:
        with open(os.path.join(os.path.dirname(prefix) or '.', 'checkpoint'), 'w') as f:
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (os.path.basename(prefix), os.path.basename(prefix)))
    return prefix
