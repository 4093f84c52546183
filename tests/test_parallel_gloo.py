"""CPU, world_size 2 over gloo: sharding and accumulator reduction of cgs_vmc_amd.parallel."""
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


def test_world_size_two_gloo():
  port = _free_port()
  procs = []
  for rank in range(2):
    env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2',
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OMP_NUM_THREADS='2')
    procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', '_gloo_worker.py')],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
  outs = []
  for p in procs:
    try:
      out, _ = p.communicate(timeout=240)
    except subprocess.TimeoutExpired:
      p.kill()
      raise
    outs.append(out.decode())
  for rank, (p, out) in enumerate(zip(procs, outs)):
    assert p.returncode == 0, out
    assert 'rank {} ok'.format(rank) in out
