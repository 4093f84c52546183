"""Test helper: N ranks of a sharded job as N THREADS of one process, one VmcEngine (vmc_ctx) each.

BASELINE configs 4 and 5 are 8-rank jobs.  A GPU box of this pool gives one GPU and allows at most 6
processes on it, so eight gloo rank PROCESSES cannot share the card; eight engines in one process can.
Each thread drives its own ctx through the library's multi-rank entry points (`vmc_*_dist`,
`vmc_evaluate`: include/cgsvmc.h says distinct ctxs are independent; ctypes releases the GIL for the
duration of a call) with the host all-reduce hook as transport, and the hooks of the N threads meet
at a barrier: every rank deposits its buffer, the buffers are folded in RANK ORDER -- the same bits on
every rank, as a ring / tree all-reduce guarantees -- and every rank takes the result.  What runs is the
library's real world_size = N code (g_count / N, MAX for update_norm, the float64 evaluation means,
one collective per CG iteration), not a restatement of it.
"""
import threading

import numpy as np

from cgs_vmc_amd import parallel


class Rendezvous:
  """The meeting point of `world` rank threads."""

  def __init__(self, world, timeout=300.0):
    self.world = world
    self.barrier = threading.Barrier(world, timeout=timeout)
    self.slots = [None] * world
    self.calls = [0] * world            # collectives every rank has made (they must agree)

  def collective(self, rank):
    return _ThreadCollective(self, rank)


class _ThreadCollective(parallel.Collective):

  def __init__(self, rv, rank):
    self.rv, self.rank = rv, rank
    super(_ThreadCollective, self).__init__(0, rv.world, use_hook=True)

  def allreduce_host(self, buf, op='sum'):
    rv = self.rv
    rv.slots[self.rank] = buf.copy()
    rv.calls[self.rank] += 1
    rv.barrier.wait()
    assert len(set(rv.calls)) == 1, 'ranks disagree on the number of collectives: {}'.format(rv.calls)
    assert len({s.shape for s in rv.slots}) == 1 and len({s.dtype for s in rv.slots}) == 1
    out = rv.slots[0].copy()
    for r in range(1, rv.world):
      out = np.maximum(out, rv.slots[r]) if op == 'max' else out + rv.slots[r]
    rv.barrier.wait()                   # everybody has read the slots before the next deposit
    buf[...] = out
    return buf


def run_ranks(world, fn):
  """Runs fn(rank, collective) on `world` threads; returns the list of results in rank order and
  re-raises the first failure (a failing rank breaks the barrier, so its peers fail fast too)."""
  rv = Rendezvous(world)
  results, errors = [None] * world, [None] * world

  def work(rank):
    try:
      results[rank] = fn(rank, rv.collective(rank))
    except BaseException as e:  # pylint: disable=broad-except
      errors[rank] = e
      rv.barrier.abort()

  threads = [threading.Thread(target=work, args=(r,), name='rank{}'.format(r)) for r in range(world)]
  for t in threads:
    t.start()
  for t in threads:
    t.join()
  first = [e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)]
  if first or any(errors):
    raise (first or [e for e in errors if e is not None])[0]
  return results, rv
