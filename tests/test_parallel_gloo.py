"""CPU, world_size 2 and 8 over gloo: sharding and accumulator reduction of cgs_vmc_amd.parallel and
the product routing of sharded training epochs (BASELINE configs 4 and 5 are 8-rank jobs: every
world-size dependent line -- parallel.shard, the g_count division, the gathered checks -- runs at
N = 8 here, with the oracle standing in for the kernels)."""
import pytest
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


@pytest.mark.parametrize('world', [2, 8])
def test_sharded_world_over_gloo(world):
  port = _free_port()
  procs = []
  for rank in range(world):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OMP_NUM_THREADS='1' if world > 2 else '2')
    procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', '_gloo_worker.py')],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
  outs = []
  for p in procs:
    try:
      out, _ = p.communicate(timeout=480)
    except subprocess.TimeoutExpired:
      for q in procs:
        q.kill()
      raise
    outs.append(out.decode())
  for rank, (p, out) in enumerate(zip(procs, outs)):
    assert p.returncode == 0, out
    assert 'rank {} ok'.format(rank) in out
