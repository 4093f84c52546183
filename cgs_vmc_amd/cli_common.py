"""Shared plumbing of the two command-line drivers (flag tables, system construction,
parameter broadcast).  The drivers keep the reference's command-line contract
(cgs_vmc/run_training.py:21-68, run_energy_evaluation.py:19-37) but are organised around
these helpers instead of one long main()."""
from __future__ import annotations

import argparse
import os
from typing import Dict, Sequence, Tuple

from . import lattice
from . import operators
from . import parallel
from . import wavefunctions


def _flag_bool(text) -> bool:
  return str(text).strip().lower() in ('1', 'true', 't', 'yes', 'y')


def parser_from_table(description: str, table: Sequence[Tuple[str, type, object, str]]
                      ) -> argparse.ArgumentParser:
  """Builds an absl-like parser: --name=value or --name value; booleans also bare."""
  ap = argparse.ArgumentParser(description=description,
                               formatter_class=argparse.RawTextHelpFormatter)
  for name, kind, default, text in table:
    if kind is bool:
      ap.add_argument('--' + name, type=_flag_bool, nargs='?', const=True, default=default,
                      help=text)
    else:
      ap.add_argument('--' + name, type=kind, default=default, help=text)
  return ap


def heisenberg_system(hparams, directory: str, j_x: float):
  """(ansatz, Hamiltonian) for a run directory: the bond list comes from `J.txt` when the
  directory has one, otherwise the periodic chain of hparams.num_sites sites; j_z is fixed to 1
  as in the reference (run_training.py:112-113)."""
  bonds = lattice.load_bonds(directory, hparams.num_sites)
  ansatz = wavefunctions.build_wavefunction(hparams)
  return ansatz, operators.HeisenbergHamiltonian(bonds, j_x, 1.)


def broadcast_parameters(ansatz):
  """All ranks start from rank 0's freshly initialised parameters."""
  if parallel.world_size() == 1:
    return
  theta = ansatz._get_theta()
  mine = theta if parallel.rank() == 0 else 0.0 * theta
  ansatz._set_theta(parallel.allreduce_array(mine).astype('float32'))


def ensure_directory(path: str):
  if path and not os.path.isdir(path):
    os.makedirs(path, exist_ok=True)


def graph_kwargs(**parts) -> Dict[str, object]:
  """Keyword bundle for build_opt_ops / build_eval_ops (they are called with keywords in the
  reference, run_training.py:120-127)."""
  parts.setdefault('shared_resources', {})
  return parts
