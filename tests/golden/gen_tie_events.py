"""Searches the sampler's Philox stream (oracle/vmc_oracle.py: step_uniforms, seed 2024) for
mc_steps in which the LARGEST site uniform of a chain is drawn twice, i.e. where the first-index
tie rule of tf.argmax / tf.argmin (graph_builders.py:62-65) decides the proposal, and writes them to
tests/golden/tie_events.json.  Data only: (n_sites, chain id, step, the tied sites, the uniform).

  python tests/golden/gen_tie_events.py        # ~1 minute on one core
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import vmc_oracle as vo  # noqa: E402

SEED = 2024


def search(n_sites, want, batch=32768, max_batches=400):
  found = []
  for bi in range(max_batches):
    # chains [0, batch) at step bi: any (chain, step) pair is reachable through chain_offset /
    # step_counter of the engine
    ids = np.arange(batch, dtype=np.uint32)
    u, _ = vo.step_uniforms(SEED, ids, bi, n_sites)
    top = u.max(axis=1)
    ties = (u == top[:, None]).sum(axis=1)
    for row in np.nonzero(ties >= 2)[0]:
      sites = np.nonzero(u[row] == top[row])[0]
      found.append({'n_sites': n_sites, 'chain': int(row), 'step': int(bi),
                    'sites': [int(s) for s in sites], 'u': float(top[row])})
      if len(found) >= want:
        return found
  return found


if __name__ == '__main__':
  events = search(100, 3) + search(256, 2, batch=16384)
  out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tie_events.json')
  with open(out, 'w') as f:
    json.dump({'seed': SEED, 'events': events}, f, indent=1)
  print(json.dumps(events))
