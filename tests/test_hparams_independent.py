"""CPU: `hparams.pbtxt` against an INDEPENDENT codec and against reference-held values.

(1) TensorFlow's `HParamDef` message (tensorflow/contrib/training/python/training/hparam.proto: a
    map<string, ParamValue> `hparam`; ParamValue = oneof {int64_value, float_value, bytes_value, bool_value,
    int64_list, float_list, bytes_list, bool_list}) is declared programmatically with google.protobuf -- the
    library the reference itself parses the file with (`text_format.Merge`, utils.py:153-166) -- and
      * parses the text `utils.HParams.to_proto()` writes (what run_training.py:100-101 puts on disk),
      * prints a message IT built, which `utils.load_hparams` must read,
      * re-prints the parsed message: byte-identical to the builder's text (map entries sorted by key, float32
        shortest round-trip floats, string escapes).
(2) tests/golden/hparams_defaults.json holds the default VALUES of the reference's `create_hparams` and of both
    drivers' flags, extracted in the build container by ast-parsing the reference sources as text
    (tools/gen_hparams_golden.py; data, not source): all 36 hyper-parameters and all 16 flags must match.
"""
import json
import os

import numpy as np
import pytest

from cgs_vmc_amd import run_energy_evaluation, run_training, utils

pytest.importorskip('google.protobuf')
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory, text_format  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'hparams_defaults.json')


def _hparam_def():
  F = descriptor_pb2.FieldDescriptorProto
  fd = descriptor_pb2.FileDescriptorProto(name='cgs_test_hparam.proto', package='tensorflow', syntax='proto3')
  top = fd.message_type.add()
  top.name = 'HParamDef'

  def nested(name):
    m = top.nested_type.add()
    m.name = name
    return m

  def field(m, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None, oneof=None):
    f = m.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
      f.type_name = '.tensorflow.HParamDef.' + type_name
    if oneof is not None:
      f.oneof_index = oneof
    return f

  for name, ftype in (('BytesList', F.TYPE_BYTES), ('FloatList', F.TYPE_FLOAT), ('Int64List', F.TYPE_INT64),
                      ('BoolList', F.TYPE_BOOL)):
    field(nested(name), 'value', 1, ftype, F.LABEL_REPEATED)
  pv = nested('ParamValue')
  pv.oneof_decl.add().name = 'kind'
  field(pv, 'int64_value', 1, F.TYPE_INT64, oneof=0)
  field(pv, 'float_value', 2, F.TYPE_FLOAT, oneof=0)
  field(pv, 'bytes_value', 3, F.TYPE_BYTES, oneof=0)
  field(pv, 'bool_value', 7, F.TYPE_BOOL, oneof=0)
  field(pv, 'int64_list', 4, F.TYPE_MESSAGE, type_name='Int64List', oneof=0)
  field(pv, 'float_list', 5, F.TYPE_MESSAGE, type_name='FloatList', oneof=0)
  field(pv, 'bytes_list', 6, F.TYPE_MESSAGE, type_name='BytesList', oneof=0)
  field(pv, 'bool_list', 8, F.TYPE_MESSAGE, type_name='BoolList', oneof=0)
  entry = nested('HparamEntry')                     # map<string, ParamValue> hparam = 1;
  entry.options.map_entry = True
  field(entry, 'key', 1, F.TYPE_STRING)
  field(entry, 'value', 2, F.TYPE_MESSAGE, type_name='ParamValue')
  field(top, 'hparam', 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, 'HparamEntry')
  pool = descriptor_pool.DescriptorPool()
  pool.Add(fd)
  return message_factory.GetMessageClass(pool.FindMessageTypeByName('tensorflow.HParamDef'))


HParamDef = _hparam_def()


def _value_of(pv):
  kind = pv.WhichOneof('kind')
  v = getattr(pv, kind)
  if kind.endswith('_list'):
    v = list(v.value)
    return [x.decode() for x in v] if kind == 'bytes_list' else v
  return v.decode() if kind == 'bytes_value' else v


def _same(got, want, name):
  if isinstance(want, float) or (isinstance(want, list) and want and isinstance(want[0], float)):
    np.testing.assert_array_equal(np.asarray(got, np.float32), np.asarray(want, np.float32), err_msg=name)
  else:
    assert got == want and type(got) is type(want), (name, got, want)


OVERRIDES = ('batch_size=64,checkpoint_dir=/tmp/a "quoted" dir\\\\x,learning_rates=[0.01,1e-05,3.3e-07],'
             'num_fc_layers=2,time_evolution_beta=0.125,wavefunction_type=fully_connected')


@pytest.mark.parametrize('overrides', ['', OVERRIDES])
def test_written_text_parses_with_google_protobuf_and_reprints_identically(overrides):
  hp = utils.create_hparams()
  hp.parse(overrides.replace('/tmp/a "quoted" dir\\\\x', '/tmp/run'))
  if overrides:
    hp.set_hparam('checkpoint_dir', '/tmp/a "quoted" dir\\x')       # escapes: quote and backslash
  text = str(hp.to_proto())
  msg = HParamDef()
  text_format.Parse(text, msg)                                       # what utils.py:163 does in the reference
  assert sorted(msg.hparam) == sorted(hp.values())
  for name, want in hp.values().items():
    _same(_value_of(msg.hparam[name]), want, name)
  kinds = {n: msg.hparam[n].WhichOneof('kind') for n in msg.hparam}
  assert kinds['num_sites'] == 'int64_value' and kinds['beta2'] == 'float_value'
  assert kinds['learning_rates'] == 'float_list' and kinds['learning_rate_stops'] == 'int64_list'
  assert kinds['composite_wavefunction_types'] == 'bytes_list' and kinds['optimizer'] == 'bytes_value'
  # protobuf's own printer on the parsed message gives the builder's bytes
  assert text_format.MessageToString(msg) == text


def test_text_printed_by_google_protobuf_loads(tmp_path):
  msg = HParamDef()
  hp = utils.create_hparams()
  want = dict(hp.values(), num_sites=36, beta2=0.75, learning_rates=[0.5, 2e-5], nonlinearity='tanh',
              checkpoint_dir='/x/"y"\\z', learning_rate_stops=[7, 8, 9], composite_wavefunction_types=['a', 'b'])
  for name, v in want.items():
    pv = msg.hparam[name]
    if isinstance(v, list):
      if isinstance(v[0], bool):
        pv.bool_list.value.extend(v)
      elif isinstance(v[0], int):
        pv.int64_list.value.extend(v)
      elif isinstance(v[0], float):
        pv.float_list.value.extend(v)
      else:
        pv.bytes_list.value.extend(x.encode() for x in v)
    elif isinstance(v, bool):
      pv.bool_value = v
    elif isinstance(v, int):
      pv.int64_value = v
    elif isinstance(v, float):
      pv.float_value = v
    else:
      pv.bytes_value = v.encode()
  path = tmp_path / 'hparams.pbtxt'
  path.write_text(text_format.MessageToString(msg))                 # = str(hparams.to_proto()) in the reference
  got = utils.load_hparams(str(path))
  assert sorted(got.values()) == sorted(want)
  for name, v in want.items():
    _same(getattr(got, name), v, name)
  # and the loaded object writes the same text again
  assert str(got.to_proto()) == path.read_text()


def test_all_defaults_match_the_reference_held_values():
  with open(GOLDEN) as f:
    gold = json.load(f)
  hp = utils.create_hparams()
  # the three sr_* entries are this repository's StochasticReconfiguration extension (no reference counterpart;
  # the reference's HParams(hparam_def) accepts a file with extra keys)
  extension = {'sr_diag_shift', 'sr_cg_tolerance', 'sr_cg_max_iterations'}
  assert len(gold['hparams']) == 36 and set(hp.values()) == set(gold['hparams']) | extension
  for name, g in gold['hparams'].items():
    typ, is_list = hp._hparam_types[name]
    assert typ.__name__ == g['type'] and is_list == g['list'], name
    assert getattr(hp, name) == g['value'], name
  for table, key in ((run_training.FLAG_TABLE, 'run_training_flags'),
                     (run_energy_evaluation.FLAG_TABLE, 'run_energy_evaluation_flags')):
    mine = {name: (kind, default) for name, kind, default, _ in table}
    assert sorted(mine) == sorted(gold[key])
    kinds = {'string': str, 'integer': int, 'float': float, 'boolean': bool}
    for name, g in gold[key].items():
      assert mine[name] == (kinds[g['kind']], g['default']), name
  assert len(gold['run_training_flags']) == 12 and len(gold['run_energy_evaluation_flags']) == 4
