"""Bond-list helpers (the reference only has the inline 1-D default of
run_training.py:109 and the J.txt reader of run_training.py:103-107)."""
import os

import numpy as np


def chain_bonds(n_sites):
  """run_training.py:109: 1-D periodic chain."""
  return [(i, (i + 1) % n_sites) for i in range(0, n_sites)]


def torus_bonds(size_x, size_y, next_nearest=False):
  """size_x x size_y periodic square lattice, site = x + size_x*y, each bond once."""
  bonds = []
  for y in range(size_y):
    for x in range(size_x):
      s = x + size_x * y
      bonds.append((s, (x + 1) % size_x + size_x * y))
      bonds.append((s, x + size_x * ((y + 1) % size_y)))
  if next_nearest:
    for y in range(size_y):
      for x in range(size_x):
        s = x + size_x * y
        bonds.append((s, (x + 1) % size_x + size_x * ((y + 1) % size_y)))
        bonds.append((s, (x - 1) % size_x + size_x * ((y + 1) % size_y)))
  return bonds


def load_bonds(checkpoint_dir, n_sites):
  """run_training.py:103-109 / run_energy_evaluation.py:51-57: `J.txt` of integer pairs
  (extra columns ignored), else the periodic chain."""
  path = os.path.join(checkpoint_dir, 'J.txt')
  if os.path.exists(path):
    data = np.atleast_2d(np.genfromtxt(path, dtype=int))
    return [[int(bond[0]), int(bond[1])] for bond in data]
  return chain_bonds(n_sites)


def write_bonds(checkpoint_dir, bonds):
  with open(os.path.join(checkpoint_dir, 'J.txt'), 'w') as f:
    for i, j in bonds:
      f.write('{} {}\n'.format(i, j))
