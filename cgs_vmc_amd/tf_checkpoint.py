"""Pure-Python reader / writer of TensorFlow "V2" checkpoint bundles, so that
variables saved by the reference's `tf.train.Saver` (run_training.py:134-146:
`model_prior_{epoch}_epochs.index` + `.data-00000-of-00001`) can be evaluated here and weights
trained here can be loaded by the reference -- without TensorFlow.

Format (TensorFlow `tensor_bundle` over the LevelDB table format; restated from the published
format descriptions, NOT validated against TensorFlow in this container -- no TF is installed;
`tests/test_host_logic.py` round-trips writer -> reader and checks the framing invariants;
`tests/test_tf_bundle_independent.py` decodes the writer's bytes with google.protobuf messages
declared from TensorFlow's .proto files and an independent table reader, and feeds the reader a
bundle serialised by google.protobuf and framed by an independent writer):

  <prefix>.index   LevelDB table ("sstable"), uncompressed blocks
      block    = entries, restart offsets (uint32 LE each), number of restarts (uint32 LE),
                 then a 5-byte trailer: compression type (0) + masked CRC32C of block + type
      entry    = varint32 shared key bytes, varint32 unshared key bytes, varint32 value bytes,
                 unshared key bytes, value
      footer   = last 48 bytes: metaindex BlockHandle, index BlockHandle (varint64 offset,
                 varint64 size each), zero padding to 40 bytes, magic 0xdb4775248b80fb57 (LE)
      key ""   -> BundleHeaderProto  {1: num_shards, 2: endianness, 3: VersionDef{1: producer}}
      key name -> BundleEntryProto   {1: dtype, 2: TensorShapeProto{2: dim{1: size}}, 3: shard_id,
                                      4: offset, 5: size, 6: crc32c (fixed32, masked)}
  <prefix>.data-00000-of-00001   raw little-endian tensor bytes at [offset, offset + size)
"""
from __future__ import annotations

import os
import struct
from typing import Dict, List, Tuple

import numpy as np

_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_}
_DTYPE_IDS = {np.dtype(v): k for k, v in _DTYPES.items()}


# ------------------------------------------------------------------ CRC32C (Castagnoli)
def _make_table():
  table = []
  for i in range(256):
    c = i
    for _ in range(8):
      c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
    table.append(c)
  return table


_TABLE = _make_table()


def crc32c(data: bytes) -> int:
  c = 0xFFFFFFFF
  for b in data:
    c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
  return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
  c = crc32c(data)
  return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


# ------------------------------------------------------------------ varints / protobuf wire format
def _put_varint(n: int) -> bytes:
  out = bytearray()
  while True:
    b = n & 0x7F
    n >>= 7
    if n:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _get_varint(buf: bytes, pos: int) -> Tuple[int, int]:
  shift = result = 0
  while True:
    b = buf[pos]
    pos += 1
    result |= (b & 0x7F) << shift
    if not b & 0x80:
      return result, pos
    shift += 7


def _parse_proto(buf: bytes) -> Dict[int, list]:
  """field number -> list of raw values (int for varint / fixed, bytes for length-delimited)."""
  out: Dict[int, list] = {}
  pos = 0
  while pos < len(buf):
    key, pos = _get_varint(buf, pos)
    field, wire = key >> 3, key & 7
    if wire == 0:
      v, pos = _get_varint(buf, pos)
    elif wire == 1:
      v = struct.unpack_from('<Q', buf, pos)[0]; pos += 8
    elif wire == 2:
      n, pos = _get_varint(buf, pos)
      v = buf[pos:pos + n]; pos += n
    elif wire == 5:
      v = struct.unpack_from('<I', buf, pos)[0]; pos += 4
    else:
      raise ValueError('unsupported protobuf wire type %d' % wire)
    out.setdefault(field, []).append(v)
  return out


def _field(field: int, wire: int, payload: bytes) -> bytes:
  return _put_varint((field << 3) | wire) + payload


def _signed(v: int) -> int:
  return v - (1 << 64) if v >= (1 << 63) else v


# ------------------------------------------------------------------ LevelDB table
def _read_block(data: bytes, offset: int, size: int, verify: bool) -> bytes:
  contents = data[offset:offset + size]
  ctype = data[offset + size]
  if verify:
    stored = struct.unpack_from('<I', data, offset + size + 1)[0]
    if stored != masked_crc32c(data[offset:offset + size + 1]):
      raise ValueError('checkpoint index block at %d fails its CRC32C' % offset)
  if ctype != 0:
    raise NotImplementedError('compressed (type %d) table blocks are not supported' % ctype)
  return contents


def _block_entries(block: bytes) -> List[Tuple[bytes, bytes]]:
  n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
  end = len(block) - 4 - 4 * n_restarts
  out, pos, key = [], 0, b''
  while pos < end:
    shared, pos = _get_varint(block, pos)
    unshared, pos = _get_varint(block, pos)
    vlen, pos = _get_varint(block, pos)
    key = key[:shared] + block[pos:pos + unshared]
    pos += unshared
    out.append((key, block[pos:pos + vlen]))
    pos += vlen
  return out


def _build_block(entries: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
  out = bytearray()
  restarts, prev = [], b''
  for i, (k, v) in enumerate(entries):
    shared = 0
    if i % restart_interval == 0:
      restarts.append(len(out))
    else:
      while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
        shared += 1
    out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v))
    out += k[shared:] + v
    prev = k
  if not restarts:
    restarts = [0]
  for r in restarts:
    out += struct.pack('<I', r)
  out += struct.pack('<I', len(restarts))
  return bytes(out)


def _emit_block(f, block: bytes) -> bytes:
  """Writes block + trailer; returns its encoded BlockHandle."""
  offset = f.tell()
  trailer = b'\x00'
  f.write(block + trailer + struct.pack('<I', masked_crc32c(block + trailer)))
  return _put_varint(offset) + _put_varint(len(block))


# ------------------------------------------------------------------ public API
def bundle_exists(prefix: str) -> bool:
  return os.path.exists(prefix + '.index')


def read_bundle(prefix: str, verify: bool = True) -> Dict[str, np.ndarray]:
  """All tensors of the checkpoint `prefix` as {variable name: array}."""
  with open(prefix + '.index', 'rb') as f:
    data = f.read()
  if len(data) < 48 or struct.unpack_from('<Q', data, len(data) - 8)[0] != _MAGIC:
    raise ValueError('%s.index is not a TensorFlow checkpoint index (bad table magic)' % prefix)
  footer = data[-48:]
  _, pos = _get_varint(footer, 0)            # metaindex offset
  _, pos = _get_varint(footer, pos)          # metaindex size
  ioff, pos = _get_varint(footer, pos)
  isize, pos = _get_varint(footer, pos)
  entries: List[Tuple[bytes, bytes]] = []
  for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
    boff, p = _get_varint(handle, 0)
    bsize, _ = _get_varint(handle, p)
    entries += _block_entries(_read_block(data, boff, bsize, verify))
  header, tensors = None, {}
  shards: Dict[int, bytes] = {}
  for key, value in entries:
    if key == b'':
      header = _parse_proto(value)
      continue
    e = _parse_proto(value)
    if 7 in e:
      raise NotImplementedError('sliced (partitioned) variables are not supported: %r' % key)
    dtype_id = e.get(1, [0])[0]
    if dtype_id not in _DTYPES:
      raise NotImplementedError('tensor %r has unsupported dtype enum %d' % (key, dtype_id))
    shape = []
    if 2 in e:
      for dim in _parse_proto(e[2][0]).get(2, []):
        shape.append(_signed(_parse_proto(dim).get(1, [0])[0]))
    shard = e.get(3, [0])[0]
    offset = e.get(4, [0])[0]
    size = e.get(5, [0])[0]
    if shard not in shards:
      num = header.get(1, [1])[0] if header else 1
      with open('%s.data-%05d-of-%05d' % (prefix, shard, num), 'rb') as f:
        shards[shard] = f.read()
    raw = shards[shard][offset:offset + size]
    if verify and 6 in e and e[6][0] != masked_crc32c(raw):
      raise ValueError('tensor %r fails its CRC32C' % key)
    arr = np.frombuffer(raw, dtype=np.dtype(_DTYPES[dtype_id]).newbyteorder('<')).reshape(shape)
    tensors[key.decode()] = arr.astype(_DTYPES[dtype_id])
  if header is not None and header.get(2, [0])[0] != 0:
    raise NotImplementedError('big-endian checkpoint bundles are not supported')
  return tensors


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
  """Writes {variable name: array} as a single-shard V2 bundle (`tf.train.Saver` layout)."""
  names = sorted(tensors, key=lambda s: s.encode())
  entries: List[Tuple[bytes, bytes]] = []
  # proto3 serialisers (TensorFlow's) omit zero-valued scalars: endianness LITTLE = 0 is not written
  header = (_field(1, 0, _put_varint(1)) +
            _field(3, 2, _put_varint(2) + _field(1, 0, _put_varint(1))))
  entries.append((b'', header))
  offset = 0
  with open(prefix + '.data-00000-of-00001', 'wb') as f:
    for name in names:
      arr = np.asarray(tensors[name])          # (ascontiguousarray would turn a scalar into [1])
      if not arr.flags.c_contiguous:
        arr = arr.copy(order='C')
      if arr.dtype not in _DTYPE_IDS:
        raise NotImplementedError('dtype %s is not supported' % arr.dtype)
      raw = arr.astype(arr.dtype.newbyteorder('<')).tobytes()
      f.write(raw)
      shape = b''.join(_field(2, 2, (lambda d: _put_varint(len(d)) + d)(_field(1, 0, _put_varint(int(s)))))
                       for s in arr.shape)
      entry = _field(1, 0, _put_varint(_DTYPE_IDS[arr.dtype]))
      entry += _field(2, 2, _put_varint(len(shape)) + shape)
      if offset:
        entry += _field(4, 0, _put_varint(offset))
      entry += _field(5, 0, _put_varint(len(raw)))
      entry += _field(6, 5, struct.pack('<I', masked_crc32c(raw)))
      entries.append((name.encode(), entry))
      offset += len(raw)
  with open(prefix + '.index', 'wb') as f:
    data_handle = _emit_block(f, _build_block(entries))
    meta_handle = _emit_block(f, _build_block([]))
    index_handle = _emit_block(f, _build_block([(entries[-1][0] + b'\x00', data_handle)], 1))
    footer = meta_handle + index_handle
    f.write(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', _MAGIC))
