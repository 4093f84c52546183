"""CPU: the hand-rolled TensorFlow V2 checkpoint-bundle codec (cgs_vmc_amd/tf_checkpoint.py)
against a SECOND implementation.

TensorFlow is not installed here, so no TF-written file exists to read.  What can be pinned without it:
 * the protobuf payloads: BundleHeaderProto / BundleEntryProto / TensorShapeProto / VersionDef are
   declared below from TensorFlow's published .proto files (tensor_bundle.proto, tensor_shape.proto,
   versions.proto: field numbers and types) and (de)serialised by google.protobuf -- the bytes the
   writer emits must parse with it, and bytes IT serialises must be understood by the reader;
 * the table framing, walked by an independent reader written here (its own varint / block / footer
   code) and produced by an independent writer;
 * CRC32C against the RFC 3720 (iSCSI) appendix B.4 known answers and a bitwise implementation; the
   LevelDB CRC mask; the table magic, which LevelDB defines as the leading 64 bits of
   sha1("http://code.google.com/p/leveldb/\\n").
"""
import hashlib
import struct

import numpy as np
import pytest

from cgs_vmc_amd import tf_checkpoint as tfc

pb = pytest.importorskip('google.protobuf')
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory  # noqa: E402


# --------------------------------------------------------------------------- protobuf schema
def _messages():
  F = descriptor_pb2.FieldDescriptorProto
  fd = descriptor_pb2.FileDescriptorProto(name='cgs_test_tensor_bundle.proto', package='tensorflow', syntax='proto3')

  def msg(name):
    m = fd.message_type.add()
    m.name = name
    return m

  def field(m, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None):
    f = m.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
      f.type_name = type_name
    return f

  shape = msg('TensorShapeProto')                 # tensor_shape.proto
  dim = shape.nested_type.add()
  dim.name = 'Dim'
  field(dim, 'size', 1, F.TYPE_INT64)
  field(dim, 'name', 2, F.TYPE_STRING)
  field(shape, 'dim', 2, F.TYPE_MESSAGE, F.LABEL_REPEATED, '.tensorflow.TensorShapeProto.Dim')
  field(shape, 'unknown_rank', 3, F.TYPE_BOOL)
  ver = msg('VersionDef')                         # versions.proto
  field(ver, 'producer', 1, F.TYPE_INT32)
  field(ver, 'min_consumer', 2, F.TYPE_INT32)
  field(ver, 'bad_consumers', 3, F.TYPE_INT32, F.LABEL_REPEATED)
  head = msg('BundleHeaderProto')                 # tensor_bundle.proto
  field(head, 'num_shards', 1, F.TYPE_INT32)
  field(head, 'endianness', 2, F.TYPE_INT32)      # enum Endianness { LITTLE = 0; BIG = 1; }
  field(head, 'version', 3, F.TYPE_MESSAGE, type_name='.tensorflow.VersionDef')
  ent = msg('BundleEntryProto')
  field(ent, 'dtype', 1, F.TYPE_INT32)            # enum DataType (types.proto): DT_FLOAT = 1, DT_DOUBLE = 2,
  field(ent, 'shape', 2, F.TYPE_MESSAGE, type_name='.tensorflow.TensorShapeProto')   # DT_INT32 = 3, DT_INT64 = 9, DT_BOOL = 10
  field(ent, 'shard_id', 3, F.TYPE_INT32)
  field(ent, 'offset', 4, F.TYPE_INT64)
  field(ent, 'size', 5, F.TYPE_INT64)
  field(ent, 'crc32c', 6, F.TYPE_FIXED32)
  pool = descriptor_pool.DescriptorPool()
  pool.Add(fd)
  get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName('tensorflow.' + n))
  return get('BundleHeaderProto'), get('BundleEntryProto')


Header, Entry = _messages()


# --------------------------------------------------------------------------- independent CRC32C
def crc32c_bitwise(data: bytes) -> int:
  """Reflected CRC-32C (Castagnoli polynomial 0x1EDC6F41), one bit at a time."""
  crc = 0xFFFFFFFF
  for byte in data:
    crc ^= byte
    for _ in range(8):
      crc = (crc >> 1) ^ (0x82F63B78 & -(crc & 1))
  return crc ^ 0xFFFFFFFF


def test_crc32c_rfc3720_known_answers():
  vectors = [(bytes(32), 0x8A9136AA), (b'\xff' * 32, 0x62A8AB43), (bytes(range(32)), 0x46DD794E),
             (bytes(range(31, -1, -1)), 0x113FDB5C), (b'123456789', 0xE3069283), (b'', 0)]
  for data, want in vectors:
    assert tfc.crc32c(data) == want == crc32c_bitwise(data)
  rng = np.random.default_rng(0)
  for n in (1, 7, 64, 1000):
    blob = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    assert tfc.crc32c(blob) == crc32c_bitwise(blob)
  # LevelDB's mask (crc32c.h): rotate right by 15 bits, add 0xa282ead8
  c = 0xE3069283
  assert tfc.masked_crc32c(b'123456789') == (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def test_table_magic_is_leveldbs():
  digest = hashlib.sha1(b'http://code.google.com/p/leveldb/\n').digest()
  assert tfc._MAGIC == int.from_bytes(digest[:8], 'big') == 0xdb4775248b80fb57


# --------------------------------------------------------------------------- independent table walk
def _uvarint(buf, pos):
  value = shift = 0
  while True:
    byte = buf[pos]
    pos += 1
    value |= (byte & 0x7F) << shift
    shift += 7
    if byte < 0x80:
      return value, pos


def _walk_block(blob, offset, size):
  """(key, value) pairs of the block at [offset, offset + size), trailer CRC checked bit by bit."""
  body = blob[offset:offset + size]
  kind = blob[offset + size]
  (stored,) = struct.unpack('<I', blob[offset + size + 1:offset + size + 5])
  c = crc32c_bitwise(blob[offset:offset + size + 1])
  assert stored == (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF and kind == 0
  (n_restarts,) = struct.unpack('<I', body[-4:])
  limit = len(body) - 4 - 4 * n_restarts
  restarts = struct.unpack('<%dI' % n_restarts, body[limit:-4])
  assert n_restarts >= 1 and restarts[0] == 0 and list(restarts) == sorted(restarts)
  pairs, pos, last = [], 0, b''
  while pos < limit:
    if pos in restarts:
      shared, _ = _uvarint(body, pos)
      assert shared == 0                        # a restart point stores its key in full
    shared, pos = _uvarint(body, pos)
    fresh, pos = _uvarint(body, pos)
    vlen, pos = _uvarint(body, pos)
    last = last[:shared] + body[pos:pos + fresh]
    pos += fresh
    pairs.append((last, body[pos:pos + vlen]))
    pos += vlen
  assert pos == limit
  return pairs


def _walk_index(path):
  blob = open(path, 'rb').read()
  footer = blob[-48:]
  assert struct.unpack('<Q', footer[40:])[0] == 0xdb4775248b80fb57
  meta_off, p = _uvarint(footer, 0)
  meta_size, p = _uvarint(footer, p)
  idx_off, p = _uvarint(footer, p)
  idx_size, p = _uvarint(footer, p)
  assert set(footer[p:40]) <= {0}
  assert _walk_block(blob, meta_off, meta_size) == []
  pairs = []
  for sep, handle in _walk_block(blob, idx_off, idx_size):
    off, q = _uvarint(handle, 0)
    size, _ = _uvarint(handle, q)
    block = _walk_block(blob, off, size)
    assert block and block[-1][0] <= sep       # the index key separates this block from the next
    pairs += block
  assert [k for k, _ in pairs] == sorted(k for k, _ in pairs)
  return pairs


def _tensors():
  rng = np.random.default_rng(5)
  return {
      'fully_connected_network/linear/w': rng.standard_normal((100, 256)).astype(np.float32),
      'fully_connected_network/linear/b': np.zeros(256, np.float32),
      'fully_connected_network/linear_1/w': rng.standard_normal((256, 1)).astype(np.float32),
      'fully_connected_network/exp_norm_shift': np.float32(-7.25),
      'num_epochs': np.int32(42),
      'big/int': np.arange(6, dtype=np.int64).reshape(2, 3) * (1 << 40),
      'a/double': rng.standard_normal(5),
      'flags': np.array([True, False, True]),
  }


DT = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9,
      np.dtype(np.bool_): 10}


def test_written_bundle_decodes_with_google_protobuf_and_an_independent_table_reader(tmp_path):
  tensors = _tensors()
  prefix = str(tmp_path / 'model_prior_3_epochs')
  tfc.write_bundle(prefix, tensors)
  pairs = _walk_index(prefix + '.index')
  assert pairs[0][0] == b''
  head = Header()
  head.ParseFromString(pairs[0][1])
  assert head.num_shards == 1 and head.endianness == 0 and head.version.producer == 1
  assert head.SerializeToString() == pairs[0][1]       # canonical field order, nothing unknown
  data = open(prefix + '.data-00000-of-00001', 'rb').read()
  seen, end = {}, 0
  for key, value in pairs[1:]:
    e = Entry()
    e.ParseFromString(value)
    assert e.SerializeToString() == value                  # canonical: nothing unknown, nothing reordered
    name = key.decode()
    want = np.asarray(tensors[name])
    assert e.dtype == DT[want.dtype] and e.shard_id == 0
    assert [d.size for d in e.shape.dim] == list(want.shape) and not e.shape.unknown_rank
    assert e.size == want.nbytes and e.offset == end      # tensors packed back to back in key order
    raw = data[e.offset:e.offset + e.size]
    c = crc32c_bitwise(raw)
    assert e.crc32c == (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF
    seen[name] = np.frombuffer(raw, want.dtype.newbyteorder('<')).reshape(want.shape)
    end = e.offset + e.size
  assert end == len(data) and set(seen) == set(tensors)
  for name, want in tensors.items():
    np.testing.assert_array_equal(seen[name], want)


def _emit(out, body):
  """Independent block writer: body + type byte + masked CRC; returns the BlockHandle bytes."""
  def uv(n):
    b = bytearray()
    while n >= 0x80:
      b.append((n & 0x7F) | 0x80)
      n >>= 7
    b.append(n)
    return bytes(b)
  off = len(out)
  c = crc32c_bitwise(body + b'\x00')
  out += body + b'\x00' + struct.pack('<I', (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF)
  return uv(off) + uv(len(body)), uv


@pytest.mark.parametrize('restart_interval,block_entries', [(16, 100), (1, 100), (2, 3)])
def test_reader_understands_a_bundle_built_by_protobuf_and_an_independent_writer(tmp_path, restart_interval,
                                                                                block_entries):
  """The other direction: entries serialised by google.protobuf (TensorFlow's own serialiser: field
  order by number, zero fields omitted), framed by a table writer written here with prefix-compressed
  keys, several restart points and several data blocks (tf.train.Saver emits those for larger models)."""
  tensors = _tensors()
  prefix = str(tmp_path / 'ref_written')
  names = sorted(tensors, key=lambda s: s.encode())
  data = bytearray()
  pairs = [(b'', Header(num_shards=1, endianness=0, version=dict(producer=1)).SerializeToString())]
  for name in names:
    arr = np.asarray(tensors[name])
    raw = arr.astype(arr.dtype.newbyteorder('<')).tobytes()
    c = crc32c_bitwise(raw)
    e = Entry(dtype=DT[arr.dtype], shard_id=0, offset=len(data), size=len(raw),
              crc32c=(((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF)
    for s in arr.shape:
      e.shape.dim.add(size=int(s))
    pairs.append((name.encode(), e.SerializeToString()))
    data += raw
  open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
  out = bytearray()
  index_pairs = []
  uv = None
  for start in range(0, len(pairs), block_entries):
    chunk = pairs[start:start + block_entries]
    body, restarts, prev = bytearray(), [], b''
    _, uv = _emit(bytearray(), b'')
    for i, (k, v) in enumerate(chunk):
      shared = 0
      if i % restart_interval == 0:
        restarts.append(len(body))
      else:
        while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
          shared += 1
      body += uv(shared) + uv(len(k) - shared) + uv(len(v)) + k[shared:] + v
      prev = k
    body += struct.pack('<%dI' % len(restarts), *restarts) + struct.pack('<I', len(restarts))
    handle, _ = _emit(out, bytes(body))
    index_pairs.append((chunk[-1][0], handle))          # LevelDB: any key >= last key of the block
  meta, _ = _emit(out, struct.pack('<II', 0, 1))
  body = bytearray()
  offs = []
  for k, h in index_pairs:
    offs.append(len(body))
    body += uv(0) + uv(len(k)) + uv(len(h)) + k + h
  body += struct.pack('<%dI' % len(offs), *offs) + struct.pack('<I', len(offs))
  index, _ = _emit(out, bytes(body))
  footer = meta + index
  out += footer + bytes(40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
  open(prefix + '.index', 'wb').write(bytes(out))
  assert tfc.bundle_exists(prefix)
  got = tfc.read_bundle(prefix)
  assert set(got) == set(tensors)
  for name, want in tensors.items():
    np.testing.assert_array_equal(got[name], want)
    assert got[name].dtype == np.asarray(want).dtype and got[name].shape == np.asarray(want).shape
  # and a corrupted tensor byte is caught by the entry CRC
  blob = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read())
  blob[17] ^= 0x40
  open(prefix + '.data-00000-of-00001', 'wb').write(bytes(blob))
  with pytest.raises(ValueError, match='CRC32C'):
    tfc.read_bundle(prefix)
