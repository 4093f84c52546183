#!/usr/bin/env python3
"""Build-container only: extracts the hyper-parameter and command-line DEFAULT VALUES of the reference by
ast-parsing its sources as text (nothing is imported or executed: TensorFlow 1.x / absl are absent) and writes
them as data to tests/golden/hparams_defaults.json:

  * the keyword arguments of the `tf.contrib.training.HParams(...)` call in cgs_vmc/utils.py:87-148;
  * every `flags.DEFINE_<kind>(name, default, help)` of cgs_vmc/run_training.py:21-68 and
    cgs_vmc/run_energy_evaluation.py:19-37.

This is the one thing in the repository that CAN be pinned to reference-held values.
The GPU box has no /root/reference: the tests read the committed JSON only.

  python tools/gen_hparams_golden.py [/root/reference]"""
import ast
import json
import os
import sys


def literal(node):
  return ast.literal_eval(node)


def hparams_defaults(path):
  tree = ast.parse(open(path).read())
  for node in ast.walk(tree):
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == 'HParams' \
        and node.keywords:
      out = {}
      for kw in node.keywords:
        v = literal(kw.value)
        out[kw.arg] = {'value': list(v) if isinstance(v, (tuple, list)) else v,
                       'type': type(v[0] if isinstance(v, (tuple, list)) else v).__name__,
                       'list': isinstance(v, (tuple, list))}
      return out
  raise SystemExit('no HParams(...) call with keyword defaults in ' + path)


def flag_defaults(path):
  tree = ast.parse(open(path).read())
  out = {}
  for node in ast.walk(tree):
    if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr.startswith('DEFINE_'):
      name, default = literal(node.args[0]), literal(node.args[1])
      out[name] = {'kind': node.func.attr[len('DEFINE_'):], 'default': default}
  return out


def main():
  ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
  pkg = os.path.join(ref, 'cgs_vmc')
  data = {
      '_generated_by': 'tools/gen_hparams_golden.py (ast.literal_eval of the default expressions; no import)',
      '_sources': ['cgs_vmc/utils.py:87-148', 'cgs_vmc/run_training.py:21-68',
                   'cgs_vmc/run_energy_evaluation.py:19-37'],
      'hparams': hparams_defaults(os.path.join(pkg, 'utils.py')),
      'run_training_flags': flag_defaults(os.path.join(pkg, 'run_training.py')),
      'run_energy_evaluation_flags': flag_defaults(os.path.join(pkg, 'run_energy_evaluation.py')),
  }
  dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden',
                     'hparams_defaults.json')
  with open(dst, 'w') as f:
    json.dump(data, f, indent=1, sort_keys=True)
    f.write('\n')
  print('{}: {} hparams, {} + {} flags'.format(dst, len(data['hparams']), len(data['run_training_flags']),
                                               len(data['run_energy_evaluation_flags'])))


if __name__ == '__main__':
  main()
