"""Worker of tests/test_gpu_api.py::test_rccl_single_rank_allreduce_on_library_buffer: a 1-rank
`nccl` (= RCCL) process group on this GPU.  Exercises what the multi-GPU bench path does around
the collective -- zero-copy torch view of the library's accumulator buffer, asynchronous
all_reduce on RCCL's stream while the sampler runs on the library's stream, wait, g_count fix --
with the one thing a 1-GPU box cannot provide (a second rank) removed: SUM over one rank must
leave the buffer bit-identical."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from cgs_vmc_amd import parallel  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402
from oracle import vmc_oracle as vo  # noqa: E402


def main():
  torch.cuda.set_device(0)
  dist.init_process_group(backend='nccl', rank=0, world_size=1)
  n, h, L, b = 16, 32, 2, 64
  rng = np.random.default_rng(0)
  eng = VmcEngine(n, b, L, h, seed=5)
  eng.set_params(vo.init_params(n, h, L, rng))
  eng.set_configs(vo.random_configurations(n, b, np.random.RandomState(1)))
  eng.set_bonds(vo.torus_bonds(4, 4), -1.0, 1.0)
  eng.reset_accumulators()
  eng.accumulate(0)
  before = eng.get_accumulators()
  t = parallel.accumulator_tensor(eng)
  assert t.is_cuda and t.numel() == before.size and t.data_ptr() == eng.accumulators_devptr()[0]
  work = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)   # what allreduce_accumulators_begin does
  eng.mc_steps(3 * n, want_accepted=False)                          # overlapped sampler launch
  work.wait()
  t[t.numel() - 4] /= 1
  torch.cuda.synchronize()
  after = eng.get_accumulators()
  np.testing.assert_array_equal(after, before)
  # the synchronous helper and the scalar reductions on the same group
  parallel.allreduce_accumulators(eng)
  assert parallel.allreduce_max(1.5) == 1.5 and parallel.allreduce_sum(2.0) == 2.0
  np.testing.assert_array_equal(eng.get_accumulators(), before)
  eng.close()
  dist.destroy_process_group()
  print('rccl ok')


if __name__ == '__main__':
  main()
