"""Benchmark of the VMC hot path on MI355X (driver contract: see the task statement).

  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): 10x10 Heisenberg torus (N=100 sites, 200 bonds),
fully-connected ansatz with 3 x 256 hidden units ("256-hidden-unit CGS"),
batch 4096 chains per GPU, synthetic random-init weights and random Sz=0 chains.

One step = one iteration of the reference's training inner loop (training.py:614-617):
  accumulate_gradients  (local energy of every chain: 1 + 200 amplitudes each, plus the two
                         weighted gradient sums) followed by
  one Monte-Carlo sweep (num_sites = 100 exchange mc_steps on every chain),
and, for N > 1, the RCCL all-reduce of the accumulator buffer (2P+8 floats).
`value` = chains x steps / time = chain-level (MC sweep + local-energy evaluation) per
second, whole job.  mc_sweeps_per_sec counts batch sweeps as the reference does (one sweep =
num_sites mc_steps on the whole batch); local_energy_evals_per_sec = chains x steps / time
spent in the local-energy kernels.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense f32 matrix peak
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    # name: (lx, ly, next_nearest, L, H, B per GPU)
    'heisenberg10x10_fc3x256_b4096': (10, 10, False, 3, 256, 4096),
    'heisenberg6x6_fc3x128_b1024': (6, 6, False, 3, 128, 1024),
    'heisenberg16x16j1j2_fc6x256_b1024': (16, 16, True, 6, 256, 1024),
}


def f_amp(n, h, L):
  """flops per amplitude, SURVEY.md 8: 2 (N H + (L-1) H^2 + H)."""
  return 2 * (n * h + (L - 1) * h * h + h)


def torus_bonds(lx, ly, nnn):
  from cgs_vmc_amd import lattice
  return lattice.torus_bonds(lx, ly, nnn)


def make_inputs(n, h, L, b, chain_offset):
  """Synthetic inputs per BASELINE.md: truncated-normal weights (default_rng(1234)), random
  Sz=0 chains keyed by global chain id (default_rng(4321 + global id))."""
  rng = np.random.default_rng(1234)
  parts = []
  fan_in = n
  for l in range(L + 1):
    out = h if l < L else 1
    w = rng.standard_normal((fan_in, out))
    bad = np.abs(w) > 2
    while bad.any():
      w[bad] = rng.standard_normal(int(bad.sum()))
      bad = np.abs(w) > 2
    parts += [(w / np.sqrt(fan_in)).ravel(), np.zeros(out)]
    fan_in = out
  theta = np.concatenate(parts).astype(np.float32)
  cfg = np.ones((b, n), np.float32)
  for i in range(b):
    r = np.random.default_rng(4321 + chain_offset + i)
    cfg[i, r.permutation(n)[:n // 2]] = -1.0
  return theta, cfg


def cpu_baseline(n, h, L, bonds, theta, cfg, seconds_budget=25.0):
  """Times the oracle (numpy restatement with the reference's call structure: one host call
  per mc_step with two forwards, 1 + n_bonds full-batch forwards per local energy, two
  back-prop passes per accumulate) on a bounded sample of the same workload."""
  from oracle import vmc_oracle as vo
  try:
    import threadpoolctl
    info = threadpoolctl.threadpool_info()
    threads = max([i.get('num_threads', 1) for i in info] or [1])
  except Exception:  # pylint: disable=broad-except
    threads = os.cpu_count() or 1
  # bounded sample: a slice of the chains, one full step (accumulate + sweep) on it
  bs = min(cfg.shape[0], 4096)
  sub = cfg[:bs].copy()
  acc = vo.Accumulators(theta.size, np.float32)
  t0 = time.perf_counter()
  vo.energy_gradient_accumulate(acc, theta, sub, bonds, -1.0, 1.0, -10.0, h, L, np.float32)
  t_acc = time.perf_counter() - t0
  t0 = time.perf_counter()
  n_steps = n
  # stop early if the sweep alone would blow the budget
  done = 0
  cur = sub
  while done < n_steps:
    cur, _ = vo.run_sweeps(theta, cur, 10, 2024, done, h, L)
    done += 10
    if time.perf_counter() - t0 > seconds_budget:
      break
  t_sweep = (time.perf_counter() - t0) * (n_steps / done)
  step_s = t_acc + t_sweep
  return {
      'value': bs / step_s, 'unit': 'chain-evals/s', 'cores': int(threads), 'kind': 'port',
      'sample': '{} chains x 1 step (accumulate {:.2f}s + sweep {:.2f}s, {} of {} mc_steps '
                'timed); numpy fp32 restatement of the reference algorithm (TF1 unavailable)'
                .format(bs, t_acc, t_sweep, done, n_steps),
      'mc_sweeps_per_sec_batch{}'.format(bs): 1.0 / t_sweep,
      'local_energy_evals_per_sec': bs / t_acc,
  }


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--workload', default='heisenberg10x10_fc3x256_b4096', choices=sorted(WORKLOADS))
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-timing', action='store_true', help='disable per-kernel HIP events')
  args = ap.parse_args()

  import torch
  from cgs_vmc_amd import _hip, parallel
  from cgs_vmc_amd.engine import VmcEngine

  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world > 1:
    parallel.init_from_env('nccl')
  rank = parallel.rank()
  dev = parallel.local_rank()
  torch.cuda.set_device(dev)

  lx, ly, nnn, L, h, b = WORKLOADS[args.workload]
  n = lx * ly
  bonds = torus_bonds(lx, ly, nnn)
  nb = len(bonds)
  chain_offset = rank * b
  theta, cfg = make_inputs(n, h, L, b, chain_offset)

  eng = VmcEngine(n, b, L, h, device=dev, chain_offset=chain_offset, seed=2024)
  eng.set_params(theta)
  eng.set_configs(cfg)
  j = np.ones(nb, np.float32)
  if nnn:
    j[nb // 2:] = 0.5                               # J1 = 1 on the NN bonds, J2 = 0.5 on the NNN bonds
  eng.set_bonds(bonds, -j, j)
  for _ in range(10):                              # 10 warm-up sweeps, one launch each
    eng.mc_steps(n, want_accepted=False)           # (every k_sweep16 launch of a run is one sweep, so
                                                   # rocprofv3's per-kernel average is per sweep)

  def step():
    # one optimizer-step slice: fresh accumulators (training.py:613 / 758), gradient accumulate,
    # accumulator all-reduce (multi-GPU), one MC sweep
    eng.reset_accumulators()
    eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
    # the sweep does not touch the accumulators: the RCCL all-reduce runs underneath it
    pending = parallel.allreduce_accumulators_begin(eng)
    eng.mc_steps(n, want_accepted=False)
    pending.wait()

  def barrier():
    eng.synchronize()
    torch.cuda.synchronize()
    if world > 1:
      torch.distributed.barrier()
      torch.cuda.synchronize()

  eng.reset_accumulators()
  for _ in range(args.warmup):
    step()
  eng.reset_accumulators()
  # HIP events on the library's stream: inside the timed region only around the two roofline
  # kernels (every recorded event drains the pipeline between two kernels, ~3.5 us); the small
  # kernels are timed in a few extra steps after the timed region
  eng.timing_enable(0 if args.no_timing else 2)
  eng.timing_reset()
  barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step()
  barrier()
  elapsed = time.perf_counter() - t0
  if world > 1:
    elapsed = parallel.allreduce_max(elapsed)
  eng.timing_enable(False)
  main_timings = {name: eng.timing_get(name) for name in ('sweep', 'tail_eloc')}
  if not args.no_timing:
    eng.timing_enable(1)
    eng.timing_reset()
    for _ in range(min(args.steps, 5)):
      step()
    barrier()
    eng.timing_enable(False)

  rows = eng.last_connected_rows()
  eng.local_energy(want_eloc=False)
  rows = eng.last_connected_rows()
  mean_e = eng.mean_energy()

  timings = {}
  for name in ('sweep', 'tail_eloc', 'tail_amp', 'z1', 'bond_list', 'eloc_reduce', 'grad', 'adam'):
    ms, cnt = main_timings[name] if name in main_timings else eng.timing_get(name)
    if cnt:
      timings[name] = {'ms_total': ms, 'launches': cnt, 'ms_avg': ms / cnt}

  if rank == 0:
    fa = f_amp(n, h, L)
    flops_eloc = b * (1 + nb) * fa                  # nominal per E_loc batch
    flops_sweep = b * n * fa                        # nominal per sweep
    exec_per_row = fa - 2 * n * h                   # rank-2 first layer: layer-1 GEMM skipped
    out = {
        'metric': 'mc_sweep+local_energy_evals_per_sec',
        'value': world * b * args.steps / elapsed,
        'unit': 'chain-evals/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': args.workload, 'lattice': '{}x{} torus'.format(lx, ly),
                   'n_sites': n, 'n_bonds': nb, 'ansatz': 'fully_connected {}x{} relu/exp'.format(L, h),
                   'chains_per_gpu': b, 'global_chains': world * b,
                   'step': 'reset + accumulate_gradients (E_loc + grad sums) + 1 MC sweep'
                           + (' + RCCL accumulator all-reduce (overlapped with the sweep)'
                              if world > 1 else ''),
                   'parallelism': 'chains sharded x{}'.format(world)},
        'mean_energy_per_site': mean_e / n,
        'connected_rows_last_eloc': rows,
        'kernels': timings,
    }
    if 'sweep' in timings and 'tail_eloc' in timings:
      ts = timings['sweep']['ms_avg'] * 1e-3
      te = timings['tail_eloc']['ms_avg'] * 1e-3
      # per local-energy call: row list + row kernel + reduction (launch averages; the small
      # kernels come from the extra steps after the timed region)
      t_eloc_call = sum(timings[k]['ms_avg'] for k in ('tail_eloc', 'bond_list', 'eloc_reduce')
                        if k in timings) * 1e-3
      out['mc_sweeps_per_sec'] = world / ts
      out['local_energy_evals_per_sec'] = world * b / t_eloc_call
      dom = 'sweep' if timings['sweep']['ms_total'] >= timings['tail_eloc']['ms_total'] else 'tail_eloc'
      per_kernel = {
          'k_sweep16': {'achieved': flops_sweep / ts / 1e12, 'ms_avg': ts * 1e3,
                        'flops_nominal': flops_sweep,
                        'flops_executed': b * (n + 2) * exec_per_row},
          'k_tail16(eloc)': {'achieved': flops_eloc / te / 1e12, 'ms_avg': te * 1e3,
                             'flops_nominal': flops_eloc, 'flops_executed': rows * exec_per_row},
      }
      for v in per_kernel.values():
        v['frac'] = v['achieved'] / FP32_MFMA_PEAK_TFLOPS
        v['executed_tflops'] = v['flops_executed'] / (v['ms_avg'] * 1e-3) / 1e12
        v['executed_frac'] = v['executed_tflops'] / FP32_MFMA_PEAK_TFLOPS
      key = 'k_sweep16' if dom == 'sweep' else 'k_tail16(eloc)'
      # HBM bytes per launch from the committed rocprofv3 PMC passes of THIS command
      # (profiles/r1_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, separate --pmc runs; see
      # tools/collect_profiles.sh); null when no matching profile is present
      traffic = None
      tpath = os.path.join(ROOT, 'profiles', 'r1_traffic.json')
      if os.path.exists(tpath) and args.workload == 'heisenberg10x10_fc3x256_b4096':
        try:
          prof = json.load(open(tpath))
          for name, rec in prof.items():
            if name.startswith(key.split('(')[0]) and rec.get('hbm_read_bytes') is not None:
              traffic = rec['hbm_read_bytes'] + (rec.get('hbm_write_bytes') or 0)
              per_kernel[key]['pmc'] = {k: rec[k] for k in ('mfma_util', 'clock_ghz', 'median_us')}
        except Exception:  # pylint: disable=broad-except
          traffic = None
      out['roofline'] = {
          'kernel': key, 'bound': 'mfma', 'achieved': per_kernel[key]['achieved'],
          'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': per_kernel[key]['frac'],
          'traffic': traffic,
          'note': 'achieved = nominal algorithmic flops (SURVEY.md 8d) / HIP-event kernel time; '
                  'executed_* counts the flops actually issued (antiparallel bonds only, rank-2 '
                  'first layer); sustained fp32-MFMA rate of the streaming loop skeleton on '
                  'this part is 141-144 TFLOP/s (tools/ubench/mfma_stream.hip, DESIGN.md 4)',
          'per_kernel': per_kernel,
      }
    if not args.no_cpu_baseline:
      try:
        out['cpu_baseline'] = cpu_baseline(n, h, L, bonds, theta, cfg)
      except Exception as e:  # pylint: disable=broad-except
        out['cpu_baseline'] = {'error': repr(e)}
    print(json.dumps(out))
  eng.close()
  if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
