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

Protocol: W warm-up steps, then the timed region of EXACTLY K steps (barrier +
synchronize on both sides, max over ranks) is repeated REPS = 5 times back to back and the
MEDIAN repetition is reported (`repetitions_ms_per_step` lists all five).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts N ranks of
itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1) before it
touches the GPU, relays rank 0's JSON line and exits non-zero if any rank fails.  Under
torch.distributed.run the ranks already exist and `--gpus` must equal WORLD_SIZE.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense f32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2516.6  # dense bf16 matrix peak = 16 x the f32 rate (same guide)
HBM_PEAK_GBS = 8000.0
REPS = 5                        # repetitions of the timed region; the median is reported

WORKLOADS = {
    # name: (lx, ly, next_nearest, L, H, B per GPU[, ansatz, kernel_size])
    # convolutional ansatz types: L = num_conv_layers / num_resnet_blocks,
    # H = num_conv_filters, hparams defaults of utils.py:108-114
    'heisenberg10x10_fc3x256_b4096': (10, 10, False, 3, 256, 4096),
    # the WHOLE batch of config 4 on one GPU (its 8-GPU shard is the line above): 8 rounds of the sampler's
    # one-tile-per-CU grid; not a driver line, a sizing check (32768 chains are 3 % of the HBM)
    'heisenberg10x10_fc3x256_b32768': (10, 10, False, 3, 256, 32768),
    # EXPERIMENT, never the headline line: config 3 with the local-energy row kernel on the
    # BF16 matrix cores -- fp32 operands as three bf16 terms, six products, fp32 accumulate
    # (CGS_VMC_SPLIT_BF16=1, cgs_vmc_amd/csrc/tail_split.hip); sampler and gradient path native fp32
    'heisenberg10x10_fc3x256_b4096_split3xbf16': (10, 10, False, 3, 256, 4096),
    # EXPERIMENT, round 5: the same with the SAMPLER's two H x H layers on the BF16 matrix cores too
    # (CGS_VMC_SPLIT_BF16=2, k_sweep16s in cgs_vmc_amd/csrc/sweep16.hpp / sweep_split.hip); gradient path native fp32
    'heisenberg10x10_fc3x256_b4096_split3xbf16_sampler': (10, 10, False, 3, 256, 4096),
    # config 5's shard with both split kernels (129 .. 256 sites: the split sampler's four-block Philox variant)
    'heisenberg16x16j1j2_fc6x256_b1024_split3xbf16_sampler': (16, 16, True, 6, 256, 1024),
    'heisenberg6x6_fc3x128_b1024': (6, 6, False, 3, 128, 1024),
    'heisenberg16x16j1j2_fc6x256_b1024': (16, 16, True, 6, 256, 1024),
    'heisenberg10x10_fc3x512_b4096': (10, 10, False, 3, 512, 4096),   # 257 .. 512 units: the fused kernels padded to 512
    'heisenberg10x10_fc3x1024_b4096': (10, 10, False, 3, 1024, 4096),  # beyond 512 units: the general path (wide.hip)
    'heisenberg10x10_conv5x16k5_b4096': (10, 10, False, 5, 16, 4096, 'conv_2d', 5),
    'heisenberg10x10_resnet2x16k5_b4096': (10, 10, False, 2, 16, 4096, 'res_net_2d', 5),
    'heisenberg10x10_conv5x32k5_b4096': (10, 10, False, 5, 32, 4096, 'conv_2d', 5),    # two 16-channel blocks
    'heisenberg10x10_conv3x32k3_b4096': (10, 10, False, 3, 32, 4096, 'conv_2d', 3),
    'heisenberg16x16j1j2_conv5x16k5_b1024': (16, 16, True, 5, 16, 1024, 'conv_2d', 5),
    # round 4: four channel blocks (conv64.hip), kernels beyond 7 x 7 (fragments in chunks of <= 25 taps)
    'heisenberg10x10_conv5x64k5_b1024': (10, 10, False, 5, 64, 1024, 'conv_2d', 5),
    'heisenberg10x10_conv3x48k3_b4096': (10, 10, False, 3, 48, 4096, 'conv_2d', 3),
    'heisenberg10x10_conv3x16k9_b4096': (10, 10, False, 3, 16, 4096, 'conv_2d', 9),
    'heisenberg10x10_conv3x32k7_b4096': (10, 10, False, 3, 32, 4096, 'conv_2d', 7),
    # round 5: beyond the fused kernels (more than 64 filters): the general convolution path (conv_general.hip)
    'heisenberg10x10_conv3x128k3_b1024': (10, 10, False, 3, 128, 1024, 'conv_2d', 3),
    'heisenberg10x10_conv3x96k3_b1024': (10, 10, False, 3, 96, 1024, 'conv_2d', 3),     # one column tile, a quarter of it padding
    # a lattice whose feature maps exceed the LDS at the DEFAULT filter count: the general path's weak spot (16 output columns)
    'heisenberg36x36_conv3x16k5_b32': (36, 36, False, 3, 16, 32, 'conv_2d', 5),
    # round 6: the same lattice at 64 filters 3 x 3 (four channel blocks on the band kernel of conv_band.hip)
    'heisenberg36x36_conv3x64k3_b32': (36, 36, False, 3, 64, 32, 'conv_2d', 3),
    # a lattice whose maps still fit the fused kernels' LDS but is wider than the network's reach: fused kernels against the general path's patch kernels (CGS_VMC_CONV_GENERAL=1)
    'heisenberg24x24_conv2x16k5_b256': (24, 24, False, 2, 16, 256, 'conv_2d', 5),
    'heisenberg20x20_conv3x16k3_b256': (20, 20, False, 3, 16, 256, 'conv_2d', 3),
}


def f_amp(n, h, L, ansatz='fully_connected', k=0):
  """flops per amplitude.  Dense: SURVEY.md 8, 2 (N H + (L-1) H^2 + H).  Convolutional: every
  one of the n_conv periodic convolutions is N sites x k^2 taps x Cin x F multiply-adds
  (Cin = 1 for the first), plus the N F-term sum."""
  if ansatz in ('conv_2d', 'res_net_2d'):
    n_conv = L if ansatz == 'conv_2d' else 1 + 2 * L
    return 2 * n * (k * k * h + (n_conv - 1) * k * k * h * h) + n * h
  return 2 * (n * h + (L - 1) * h * h + h)


def mfma_flops_per_amp(n, h, L, ansatz, k):
  """flops one amplitude ISSUES to the matrix cores."""
  if ansatz in ('conv_2d', 'res_net_2d'):
    # 16-channel tiles (filters zero padded to ncb blocks), taps of the first convolution padded to 4
    n_conv = L if ansatz == 'conv_2d' else 1 + 2 * L
    ncb = (h + 15) // 16
    return 2 * n * 16 * ncb * (4 * ((k * k + 3) // 4) + (n_conv - 1) * k * k * 16 * ncb)
  hp = (h + 63) // 64 * 64 if h <= 256 else (h + 127) // 128 * 128   # 257..512 units pad to 384 / 512
  return 2 * (L - 1) * hp * hp


def torus_bonds(lx, ly, nnn):
  from cgs_vmc_amd import lattice
  return lattice.torus_bonds(lx, ly, nnn)


def couplings(n_bonds, nnn):
  """(j_x, j_z) per bond: antiferromagnet in the Marshall-rotated frame (j_x = -J, j_z = J);
  J1 = 1 on the nearest-neighbour bonds, J2 = 0.5 on the next-nearest ones (config 5)."""
  j = np.ones(n_bonds, np.float32)
  if nnn:
    j[n_bonds // 2:] = 0.5
  return -j, j


def make_inputs(n, h, L, b, chain_offset, ansatz='fully_connected', k=0):
  """Synthetic inputs per BASELINE.md: truncated-normal weights (default_rng(1234)), random
  Sz=0 chains keyed by global chain id (default_rng(4321 + global id))."""
  rng = np.random.default_rng(1234)
  parts = []
  if ansatz in ('conv_2d', 'res_net_2d'):     # snt.Conv2D: sigma = 1/sqrt(k k Cin), b = 0
    n_conv = L if ansatz == 'conv_2d' else 1 + 2 * L
    shapes = [(k, k, 1 if l == 0 else h, h) for l in range(n_conv)]
  else:
    shapes = [(n if l == 0 else h, h if l < L else 1) for l in range(L + 1)]
  for shp in shapes:
    fan_in = int(np.prod(shp[:-1]))
    w = rng.standard_normal(shp)
    bad = np.abs(w) > 2
    while bad.any():
      w[bad] = rng.standard_normal(int(bad.sum()))
      bad = np.abs(w) > 2
    parts += [(w / np.sqrt(fan_in)).ravel(), np.zeros(shp[-1])]
  theta = np.concatenate(parts).astype(np.float32)
  cfg = np.ones((b, n), np.float32)
  for i in range(b):
    r = np.random.default_rng(4321 + chain_offset + i)
    cfg[i, r.permutation(n)[:n // 2]] = -1.0
  return theta, cfg


# ----------------------------------------------------------------------------- legs after the timed region
def extra_legs(eng, step, n, b, L, h, soak_seconds):
  """BASELINE config 3 is worded "full SR/SWO training step": after the timed region, on the SAME engine,
    * a soak: the same step for >= soak_seconds back to back (the sustained clock next to the 37 ms bursts
      of the timed repetitions; also what makes the GPU visible to a once-a-second utilisation sampler);
    * a LogOverlapImaginaryTimeSWO batch loop (training.py:731-763: one MC sweep, reset, accumulate with the
      supervisor's local energies and the overlap ratio, Adam) through vmc_epoch_log_overlap, 20 batches;
    * the stochastic-reconfiguration extension: the 50 batches of an epoch at the reference's
      num_batches_per_epoch recorded into the sample store, then 20 CG iterations of (S + 0.01) x = f over
      those 50 x B samples (matrix-free, vmc_sr_solve with tolerance 0 so that every iteration runs);
    * one EnergyGradient epoch at the reference's DEFAULT hparams (utils.py:87-148: 40-site chain, 3 x 80,
      200 chains, 100 equilibration sweeps, 50 batches) through vmc_epoch_energy_gradient + Adam -- the
      reference issues 6,056 session.run calls for it (training.py:608-623).
  Everything here runs after the headline numbers were taken; it changes theta (Adam), so it comes last."""
  from cgs_vmc_amd.engine import VmcEngine
  extra = {}
  # --- soak
  if soak_seconds > 0:
    chunk, steps_done = 250, 0
    eng.synchronize()
    t0 = time.perf_counter()
    while True:
      for _ in range(chunk):
        step()
      eng.synchronize()
      steps_done += chunk
      dt = time.perf_counter() - t0
      if dt >= soak_seconds:
        break
    extra['soak_ms_per_step'] = 1e3 * dt / steps_done
    extra['soak_steps'] = steps_done
    extra['soak_seconds'] = dt
  # --- LogOverlapITSWO batch loop
  k_it = 20
  run = lambda k: eng.epoch_log_overlap(0.12, 0, k, n, 0.0, 1e-3, 0.9, 0.99, 1e-8)
  run(3)                                  # warm-up (also omega <- psi)
  eng.synchronize()
  t0 = time.perf_counter()
  e_it = run(k_it)
  eng.synchronize()
  extra['log_overlap_itswo_ms_per_batch'] = 1e3 * (time.perf_counter() - t0) / k_it
  extra['log_overlap_itswo_batches'] = k_it
  extra['log_overlap_itswo_energy_per_site'] = e_it / n
  # --- SR: record an epoch's batches, then CG iterations
  n_store, k_cg = 50, 20
  eng.sr_reserve(n_store)
  eng.epoch_energy_gradient(0, n_store, n, 0.0)
  eng.sr_solve(0.01, 0.0, 3)              # warm-up
  eng.synchronize()
  t0 = time.perf_counter()
  it, res = eng.sr_solve(0.01, 0.0, k_cg)
  eng.synchronize()
  extra['sr_cg_ms_per_iteration'] = 1e3 * (time.perf_counter() - t0) / max(it, 1)
  extra['sr_cg_iterations'] = it
  extra['sr_samples'] = n_store * b
  extra['sr_rel_residual'] = res
  fa_exec = 2.0 * 2 * (n * h + (L - 1) * h * h + h)      # reverse-mode matvec: 2 F_amp per stored sample
  extra['sr_matvec_frac_of_fp32_mfma_peak'] = (fa_exec * n_store * b / (extra['sr_cg_ms_per_iteration'] * 1e-3)
                                               / 1e12 / FP32_MFMA_PEAK_TFLOPS)
  eng.sr_reserve(0)
  # --- EXPERIMENT (never the headline; its own workload tags carry the full lines): the same step with fp32 results
  # from the BF16 matrix cores -- CGS_VMC_SPLIT_BF16=1: the local-energy row kernel (k_tail16r), =2: the sampler too
  # (k_sweep16s) -- on fresh engines with the headline's inputs, so that the driver's own run carries the numbers
  if h == 256:
    bonds = torus_bonds(int(round(n ** 0.5)), int(round(n ** 0.5)), False)
    theta, cfg = make_inputs(n, h, L, b, 0)
    prev = os.environ.get('CGS_VMC_SPLIT_BF16')
    try:
      for mode, key in (('1', 'split3xbf16_rows'), ('2', 'split3xbf16_rows_and_sampler')):
        os.environ['CGS_VMC_SPLIT_BF16'] = mode
        e2 = VmcEngine(n, b, L, h, device=eng.device, seed=2024)
        e2.set_params(theta); e2.set_configs(cfg); e2.set_bonds(bonds, -1.0, 1.0)
        if e2.kernel_path() != (4 if mode == '1' else 5):
          e2.close()
          continue
        for _ in range(10):
          e2.mc_steps(n, want_accepted=False)

        def step2():
          e2.reset_accumulators(); e2.accumulate(0); e2.mc_steps(n, want_accepted=False)
        for _ in range(5):
          step2()
        e2.synchronize()
        e2.timing_enable(2); e2.timing_reset()
        t0 = time.perf_counter()
        for _ in range(40):
          step2()
        e2.synchronize()
        extra[key + '_ms_per_step'] = 1e3 * (time.perf_counter() - t0) / 40
        e2.timing_enable(False)
        for name, k2 in (('sweep', '_sampler_kernel_ms'), ('tail_eloc', '_row_kernel_ms')):
          ms, cnt = e2.timing_get(name)
          if cnt:
            extra[key + k2] = ms / cnt
        e2.local_energy(want_eloc=False)
        extra[key + '_energy_per_site'] = e2.mean_energy() / n
        e2.close()
    finally:
      if prev is None:
        os.environ.pop('CGS_VMC_SPLIT_BF16', None)
      else:
        os.environ['CGS_VMC_SPLIT_BF16'] = prev
  # --- BASELINE configs 5 (one of its eight shards) and 2 on fresh engines, with sixteen- and with eight-chain sampler
  # tiles (round 6: k_sweep8 -- the same chains bit for bit; vmc_create picks eight-chain tiles for these batches), so
  # that the driver's own run carries their steps; the full lines: --workload heisenberg16x16j1j2_fc6x256_b1024 /
  # heisenberg6x6_fc3x128_b1024 (profiles/r6_config5_*, r6_config2_*)
  for tag, wl in (('config5_shard', 'heisenberg16x16j1j2_fc6x256_b1024'), ('config2', 'heisenberg6x6_fc3x128_b1024')):
    lx, ly, nnn, L2, h2, b2 = WORKLOADS[wl][:6]
    n2 = lx * ly
    bonds2 = torus_bonds(lx, ly, nnn)
    theta2, cfg2 = make_inputs(n2, h2, L2, b2, 0)
    e2 = VmcEngine(n2, b2, L2, h2, device=eng.device, seed=2024)
    e2.set_params(theta2); e2.set_bonds(bonds2, -1.0, 1.0)
    for tile in (16, 8):
      try:
        e2.sweep_tile(tile)
      except Exception:  # pylint: disable=broad-except
        continue
      e2.set_configs(cfg2)
      for _ in range(5):
        e2.mc_steps(n2, want_accepted=False)

      def step3():
        e2.reset_accumulators(); e2.accumulate(0); e2.mc_steps(n2, want_accepted=False)
      for _ in range(4):
        step3()
      e2.synchronize()
      reps3 = 30 if n2 > 100 else 200
      t0 = time.perf_counter()
      for _ in range(reps3):
        step3()
      e2.synchronize()
      extra['{}_ms_per_step_tile{}'.format(tag, tile)] = 1e3 * (time.perf_counter() - t0) / reps3
    e2.close()
  # --- one epoch at the reference's default hparams (a second, small engine)
  dn, dh, dL, db, dnb = 40, 80, 3, 200, 50
  theta, cfg = make_inputs(dn, dh, dL, db, 0)
  small = VmcEngine(dn, db, dL, dh, device=eng.device)
  small.set_params(theta); small.set_configs(cfg)
  small.set_bonds([(i, (i + 1) % dn) for i in range(dn)], -1.0, 1.0)

  def epoch():
    small.epoch_energy_gradient(100 * dn, dnb, dn, 1e10)
    small.apply_adam(0, 1e-3)
  for _ in range(3):
    epoch()
  small.synchronize()
  t0 = time.perf_counter()
  reps = 10
  for _ in range(reps):
    epoch()
  small.synchronize()
  extra['epoch_ms_reference_default_hparams'] = 1e3 * (time.perf_counter() - t0) / reps
  extra['epoch_reference_session_runs'] = 100 * dn + dnb * (1 + dn) + 6    # training.py:608-623: 6,056
  small.close()
  return extra


# ----------------------------------------------------------------------------- CPU baseline
def _blas_info():
  try:
    import threadpoolctl
    info = threadpoolctl.threadpool_info()
    blas = [i for i in info if i.get('user_api') == 'blas'] or info
    return (max([i.get('num_threads', 1) for i in blas] or [1]),
            ', '.join(sorted({'{} {}'.format(i.get('internal_api', '?'), i.get('version', ''))
                              for i in blas})))
  except Exception:  # pylint: disable=broad-except
    return os.cpu_count() or 1, 'unknown'


def _cpu_step_seconds(vo, theta, cfg, bonds, jx, jz, h, L, n, mc_steps_timed, ansatz='fully_connected'):
  """One step of the reference's structure on `cfg`: accumulate_gradients + one sweep (the
  sweep is timed on `mc_steps_timed` of its n mc_steps and scaled).  `h` is the layer size, or
  the oracle's geometry tuple for the convolutional ansatz types."""
  acc = vo.Accumulators(theta.size, np.float32)
  t0 = time.perf_counter()
  vo.energy_gradient_accumulate(acc, theta, cfg, bonds, jx, jz, -10.0, h, L, np.float32, ansatz=ansatz)
  t_acc = time.perf_counter() - t0
  t0 = time.perf_counter()
  vo.run_sweeps(theta, cfg, mc_steps_timed, 2024, 0, h, L, ansatz=ansatz)
  t_sweep = (time.perf_counter() - t0) * (n / float(mc_steps_timed))
  return t_acc, t_sweep


def usable_cpus():
  """CPUs this process may actually use: the scheduler affinity mask capped by the cgroup CPU
  quota (a GPU box hands a 1-GPU job a share of the host, not all of os.cpu_count())."""
  try:
    n = len(os.sched_getaffinity(0))
  except AttributeError:
    n = os.cpu_count() or 1
  quota = None
  for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    try:
      with open(path) as f:
        fields = f.read().split()
      if path.endswith('cpu.max'):
        if fields[0] != 'max':
          quota = float(fields[0]) / float(fields[1])
      else:
        q = float(fields[0])
        if q > 0:
          with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
            quota = q / float(f.read().split()[0])
      break
    except (OSError, ValueError, IndexError):
      continue
  if quota is not None:
    n = max(1, min(n, int(quota + 0.5)))
  return n, quota


def cpu_worker_main(argv):
  """`bench.py --cpu-worker WORKLOAD INDEX P T_GO SECONDS`: one single-threaded process of the
  all-core CPU baseline.  Takes its own disjoint shard of the chains (64, or 16 for the
  convolutional types), waits for the common start time, then repeats the reference-structured
  step -- accumulate_gradients + one sweep (timed on a few of its mc_steps and scaled) -- until
  SECONDS have passed, and prints its median rate.  No GPU, no torch."""
  workload, index, nproc, t_go, seconds = argv[0], int(argv[1]), int(argv[2]), float(argv[3]), float(argv[4])
  from oracle import vmc_oracle as vo
  lx, ly, nnn, L, h, b = WORKLOADS[workload][:6]
  ansatz, ksz = (WORKLOADS[workload][6:] + ('fully_connected', 0))[:2]
  conv = ansatz in ('conv_2d', 'res_net_2d')
  n = lx * ly
  bonds = torus_bonds(lx, ly, nnn)
  jx, jz = couplings(len(bonds), nnn)
  bs = 16 if conv else 64
  theta, cfg = make_inputs(n, h, L, bs, index * bs, ansatz, ksz)      # this worker's own chains
  shape = (h, ksz, ly, lx) if conv else h
  steps = max(2, min(n, 4 if conv else 10))
  _cpu_step_seconds(vo, theta, cfg[:8], bonds, jx, jz, shape, L, n, 1, ansatz)   # page in, first touch
  late = time.time() - t_go
  while time.time() < t_go:
    time.sleep(0.005)
  t_start = time.time()
  rates = []
  while True:
    t_acc, t_sw = _cpu_step_seconds(vo, theta, cfg, bonds, jx, jz, shape, L, n, steps, ansatz)
    rates.append(bs / (t_acc + t_sw))
    if time.time() - t_start >= seconds:
      break
  rates.sort()
  print(json.dumps({'index': index, 'chains': bs, 'rate': rates[len(rates) // 2], 'reps': len(rates),
                    'late_s': max(0.0, late), 'busy_s': time.time() - t_start}))


def cpu_all_cores_leg(workload, seconds=6.0):
  """All-core CPU baseline: chains are independent, so the honest many-core
  figure is P single-threaded processes on DISJOINT chain shards -- not one threaded BLAS call on a
  small batch.  P = the CPUs this job may use (usable_cpus; half of them when nothing limits the
  job, i.e. one per physical core).  Aggregate = sum of the workers' rates while all of them run."""
  cpus, quota = usable_cpus()
  limited = quota is not None or cpus < (os.cpu_count() or 1)
  nproc = max(1, cpus if limited else cpus // 2)
  env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1',
             NUMEXPR_NUM_THREADS='1')
  for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  t_go = time.time() + 4.0 + 0.02 * nproc           # interpreter + numpy start-up of every worker
  procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', workload, str(i),
                             str(nproc), repr(t_go), repr(seconds)], env=env, stdout=subprocess.PIPE,
                            stderr=subprocess.DEVNULL) for i in range(nproc)]
  res = []
  for pr in procs:
    try:
      out, _ = pr.communicate(timeout=120 + 4 * seconds)
      res.append(json.loads(out.decode().strip().splitlines()[-1]))
    except Exception:  # pylint: disable=broad-except
      pr.kill()
  if not res:
    return {'error': 'no CPU worker finished'}
  total = sum(r['rate'] for r in res)
  return {'value': total, 'unit': 'chain-evals/s', 'cores': len(res), 'processes_started': nproc,
          'per_core': total / len(res), 'usable_cpus': cpus, 'cgroup_cpu_quota': quota,
          'host_cpu_count': os.cpu_count(), 'late_workers': sum(1 for r in res if r['late_s'] > 0),
          'sample': '{} single-threaded processes x {} chains each (disjoint shards), every process repeating '
                    '1 step = accumulate_gradients + sweep (scaled from a few mc_steps) for {:.0f} s from a common '
                    'start; sum of the per-process median rates; numpy fp32 restatement of the reference '
                    'algorithm (TensorFlow 1.x unavailable)'.format(len(res), res[0]['chains'], seconds)}


def cpu_baseline(n, h, L, bonds, jx, jz, theta, cfg, ansatz='fully_connected', k=0, lx=0, ly=0, workload=None):
  """Times the oracle (numpy fp32 restatement with the reference's call structure: one host
  call per mc_step with two forwards, 1 + n_bonds full-batch forwards per local energy, two
  back-prop passes per accumulate; no rank-2 update, no bond skipping) on a bounded sample of
  the same workload: median of 3 repetitions, with all BLAS threads and single-threaded."""
  from oracle import vmc_oracle as vo
  threads, blas = _blas_info()
  flops_per_chain_step = (1 + len(bonds) + 2 * n) * f_amp(n, h, L, ansatz, k)
  conv = ansatz in ('conv_2d', 'res_net_2d')
  shape = (h, k, ly, lx) if conv else h        # oracle geometry: (filters, kernel, size_x, size_y)
  out = {'unit': 'chain-evals/s', 'kind': 'port', 'cores': int(threads), 'blas': blas,
         'host_cpu_count': os.cpu_count()}

  def measure(bs, mc_steps_timed, reps):
    sub = np.ascontiguousarray(cfg[:bs])
    runs = [_cpu_step_seconds(vo, theta, sub, bonds, jx, jz, shape, L, n, mc_steps_timed, ansatz)
            for _ in range(reps)]
    runs.sort(key=lambda r: r[0] + r[1])
    return runs[len(runs) // 2]

  # all cores: a batch large enough for the BLAS to thread (2048 chains on a many-core host,
  # 512 on a small one); ~10-20 s in total
  bs_all = min(cfg.shape[0], (2048 if threads >= 32 else 512) // (8 if conv else 1))
  steps_all = max(4, min(n, 20)) // (4 if conv else 1)
  measure(min(bs_all, 64), 2, 1)                       # page in BLAS, first-touch
  t_acc, t_sweep = measure(bs_all, steps_all, 3)
  out['value'] = bs_all / (t_acc + t_sweep)
  out['mc_sweeps_per_sec_per_chain'] = 1.0 / t_sweep * bs_all
  out['local_energy_evals_per_sec'] = bs_all / t_acc
  out['sample'] = ('median of 3: {} chains x 1 step = accumulate_gradients ({:.2f} s) + sweep '
                   '({:.2f} s scaled from {} of {} mc_steps); numpy fp32 restatement of the '
                   'reference algorithm (TensorFlow 1.x unavailable), {} BLAS threads'
                   .format(bs_all, t_acc, t_sweep, steps_all, n, threads))
  # single thread: 64 chains
  try:
    import threadpoolctl
    with threadpoolctl.threadpool_limits(limits=1):
      bs_1 = min(cfg.shape[0], 16 if conv else 64)
      steps_1 = max(2, min(n, 4 if conv else 10))
      a1, s1 = measure(bs_1, steps_1, 3)
    out['single_thread'] = {
        'value': bs_1 / (a1 + s1), 'unit': 'chain-evals/s', 'cores': 1,
        'gflops': flops_per_chain_step * bs_1 / (a1 + s1) / 1e9,
        'sample': 'median of 3: {} chains x 1 step (accumulate {:.2f} s + sweep {:.2f} s scaled '
                  'from {} of {} mc_steps)'.format(bs_1, a1, s1, steps_1, n)}
  except Exception as e:  # pylint: disable=broad-except
    out['single_thread'] = {'error': repr(e)}
  out['all_threads'] = {'value': out['value'], 'cores': int(threads), 'sample': out['sample']}
  st = out['single_thread']
  if 'value' in st and st['value'] > out['value']:
    # small batches that stay in one core's cache beat the threaded BLAS on this host: the
    # reported baseline is the faster of the two configurations
    out['value'], out['cores'], out['sample'] = st['value'], 1, st['sample'] + '; numpy fp32 restatement of the reference algorithm (TensorFlow 1.x unavailable)'
  # every core the job may use, one single-threaded process per core on its own chain shard
  if workload is not None:
    try:
      out['all_cores'] = cpu_all_cores_leg(workload)
      ac = out['all_cores']
      if 'value' in ac:
        ac['gflops'] = flops_per_chain_step * ac['value'] / 1e9
        if ac['value'] > out['value']:
          out['value'], out['cores'], out['sample'] = ac['value'], ac['cores'], ac['sample']
    except Exception as e:  # pylint: disable=broad-except
      out['all_cores'] = {'error': repr(e)}
  out['gflops'] = flops_per_chain_step * out['value'] / 1e9
  if not conv:
    try:
      import torch
      default_threads = torch.get_num_threads()
      variants = {}
      for nt in (1, default_threads):      # one core and torch's default intra-op pool
        torch.set_num_threads(nt)
        variants['threads_{}'.format(nt)] = torch_cpu_baseline(
            n, h, L, bonds, jx, jz, theta, cfg, bs=min(cfg.shape[0], 64 if nt == 1 else 256),
            mc_steps_timed=max(2, min(n, 10)))
      torch.set_num_threads(default_threads)
      out['torch_cpu'] = variants
    except Exception as e:  # pylint: disable=broad-except
      out['torch_cpu'] = {'error': repr(e)}
  return out


def torch_cpu_baseline(n, h, L, bonds, jx, jz, theta, cfg, bs=256, mc_steps_timed=10):
  """The same reference structure on torch-CPU tensors (fully_connected only): 1 + n_bonds
  full-batch forwards and two autograd backward passes per accumulate (training.py:542-547), two
  forwards per mc_step (graph_builders.py:54-55, 74); median of 3 on `bs` chains."""
  import torch
  torch.manual_seed(0)
  th = torch.tensor(theta)
  shapes, off, params = [], 0, []
  fan = n
  for l in range(L + 1):
    out = h if l < L else 1
    w = th[off:off + fan * out].reshape(fan, out).clone().requires_grad_(True); off += fan * out
    b = th[off:off + out].clone().requires_grad_(True); off += out
    params += [w, b]; fan = out
  def logit(x):
    a = x
    for l in range(L):
      a = torch.relu(a @ params[2 * l] + params[2 * l + 1])
    return (a @ params[2 * L] + params[2 * L + 1])[:, 0]
  x0 = torch.tensor(np.ascontiguousarray(cfg[:bs]))
  ij = torch.tensor(np.asarray(bonds, np.int64))
  hx = torch.tensor(0.5 * np.asarray(jx, np.float32)); qz = torch.tensor(0.25 * np.asarray(jz, np.float32))
  def accumulate():
    with torch.no_grad():
      psi = torch.exp(logit(x0) + 10.0)
      diag = torch.zeros(bs); offd = torch.zeros(bs)
      for k in range(ij.shape[0]):
        i, j = int(ij[k, 0]), int(ij[k, 1])
        si, sj = x0[:, i], x0[:, j]
        upd = x0.clone(); upd[:, i] = sj; upd[:, j] = si
        sz = si * sj
        offd += hx[k] * (sz < 0).float() * torch.exp(logit(upd) + 10.0)
        diag += qz[k] * sz
      eloc = diag + offd / psi
    lg = logit(x0)
    g1 = torch.autograd.grad(lg.sum(), params, retain_graph=True)
    g2 = torch.autograd.grad((lg * eloc).sum(), params)
    return g1, g2
  def sweep(steps):
    x = x0.clone()
    rows = torch.arange(bs)
    with torch.no_grad():
      for _ in range(steps):
        u = torch.rand(bs, n)
        sc = x * u
        i_up, i_dn = sc.argmax(1), sc.argmin(1)
        psi = torch.exp(logit(x) + 10.0)
        upd = x.clone(); upd[rows, i_dn] += 2.0; upd[rows, i_up] -= 2.0
        ratio = torch.exp(logit(upd) + 10.0).abs() / psi.abs()
        acc = ratio > torch.rand(bs).sqrt()
        x[acc] = upd[acc]
  runs = []
  for _ in range(3):
    t0 = time.perf_counter(); accumulate(); t_acc = time.perf_counter() - t0
    t0 = time.perf_counter(); sweep(mc_steps_timed); t_sw = (time.perf_counter() - t0) * (n / float(mc_steps_timed))
    runs.append((t_acc, t_sw))
  runs.sort(key=lambda r: r[0] + r[1])
  t_acc, t_sw = runs[1]
  return {'value': bs / (t_acc + t_sw), 'unit': 'chain-evals/s', 'cores': int(torch.get_num_threads()),
          'sample': 'median of 3: {} chains x 1 step (accumulate {:.2f} s + sweep {:.2f} s scaled from {} of {} '
                    'mc_steps); torch {} CPU tensors, {} threads'.format(bs, t_acc, t_sw, mc_steps_timed, n,
                                                                        torch.__version__, torch.get_num_threads())}


# ----------------------------------------------------------------------------- rank launcher
def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def spawn_ranks(n_ranks, script=None):
  """Starts `n_ranks` copies of this script (or `script`: tools/itswo_bench.py, tools/sr_bench.py), one
  per GPU, and relays rank 0's output.  Runs before this process has made any GPU / HIP call: the
  children are fresh processes."""
  port = os.environ.get('MASTER_PORT') or str(_free_port())
  procs = []
  for r in range(n_ranks):
    env = dict(os.environ)
    env.update({'RANK': str(r), 'LOCAL_RANK': str(r), 'WORLD_SIZE': str(n_ranks),
                'LOCAL_WORLD_SIZE': str(n_ranks), 'MASTER_ADDR': '127.0.0.1',
                'MASTER_PORT': port, 'HSA_ENABLE_IPC_MODE_LEGACY': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0')})
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(script or __file__)] + sys.argv[1:],
                                  env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
  out0, _ = procs[0].communicate()
  codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
  sys.stdout.write(out0.decode())
  sys.stdout.flush()
  bad = [(r, c) for r, c in enumerate(codes) if c != 0]
  if bad:
    sys.stderr.write('bench.py: ranks failed (rank, exit code): {}\n'.format(bad))
    return 1
  return 0


def device_identity(dev):
  """(PCI bus id, name, uuid or None) of HIP device `dev` -- what tells two ranks' devices apart."""
  import ctypes as C
  import torch
  from cgs_vmc_amd import _hip
  # asked of the ONE HIP runtime the process already has (the library's own hip* symbols), never of a
  # libamdhip64 opened by soname: that would map a second runtime next to torch's
  lib = _hip.load()
  bus = None
  buf = C.create_string_buffer(64)
  if lib.vmc_device_pci_bus_id(int(dev), buf, 64) == 0:
    bus = buf.value.decode() or None
  props = torch.cuda.get_device_properties(dev)
  uuid = getattr(props, 'uuid', None)
  return {'ordinal': int(dev), 'pci_bus_id': bus, 'name': props.name, 'uuid': None if uuid is None else str(uuid),
          'hip_runtime': lib.vmc_hip_runtime_path().decode(), 'hip_runtimes_mapped': mapped_libraries('libamdhip64')}


def mapped_libraries(stem):
  """Distinct files whose name contains `stem` mapped into this process (/proc/self/maps)."""
  found = set()
  try:
    with open('/proc/self/maps') as f:
      for line in f:
        path = line.split(None, 5)[-1].strip() if line.count('/') else ''
        if stem in os.path.basename(path):
          found.add(os.path.realpath(path))
  except OSError:
    pass
  return sorted(found)


def prove_collectives(eng, world, rank, dev):
  """N > 1, BEFORE the timed region: evidence in the JSON line that the job really ran on `world`
  DISTINCT devices and that the collective the timed step makes reduces correctly --
  torch.distributed's process group on the library's accumulator buffer: every float == N(N+1)/2
  and g_count divided back by N.  Under the `nccl` backend two ranks on one device are refused."""
  import torch.distributed as dist
  from cgs_vmc_amd import parallel
  ident = dict(device_identity(dev), rank=rank, host=socket.gethostname(), pid=os.getpid())
  idents = [None] * world
  dist.all_gather_object(idents, ident)
  backend = dist.get_backend()
  keys = [(i['host'], i['pci_bus_id'] or i['ordinal']) for i in idents]
  distinct = len(set(keys)) == world
  if backend == 'nccl' and not distinct:
    raise SystemExit('bench.py: {} ranks but only {} distinct devices {}: refusing to report an N-GPU number'
                     .format(world, len(set(keys)), sorted(set(keys))))
  nacc = 2 * eng.num_params + 8
  eng.set_accumulators(np.full(nacc, rank + 1.0, np.float32))
  parallel.allreduce_accumulators(eng)
  acc = eng.get_accumulators()
  tot = world * (world + 1) / 2
  expect = np.full(nacc, tot, np.float32)
  expect[nacc - 4] = np.float32(tot) / np.float32(world)
  pg_ok = bool(np.array_equal(acc, expect))
  eng.reset_accumulators()
  if not pg_ok:
    raise SystemExit('bench.py: all-reduce check of the accumulator buffer failed on rank {}'.format(rank))
  return {'backend': backend, 'devices': idents, 'distinct_devices': distinct,
          'checked_allreduce': {'process_group_on_accumulators': pg_ok, 'expected_sum_of_ranks': world * (world - 1) / 2}}


TRANSPORT_NAMES = {'torch': 'device hook: torch.distributed all-reduce on the library buffer, in stream',
                   'rccl': "the library's own RCCL communicator (in stream)", 'host': 'host hook (pinned staging)'}


def prove_library_transport(eng, world, rank, timeout_s=90.0):
  """N > 1, BEFORE the timed region: the transport of the library's multi-rank entry points
  (cgs_vmc_amd/parallel.py Collective: the device hook over torch's process group by default, the
  library's own RCCL communicator under CGS_VMC_TRANSPORT=rccl, the host hook under a non-RCCL
  backend) on a device buffer of the library: sum of rank ids == N(N-1)/2, sum of ones == N, max of
  rank ids == N-1, and the same sums in float64 (the evaluation means).  It runs on a watchdog thread:
  creating a second RCCL communicator has never run on more than one GPU, and a failure or a timeout
  is REPORTED (and the timed step falls back to torch's process group) instead of costing the run."""
  import threading
  from cgs_vmc_amd import parallel
  out = {'library_transport': 'timeout after {:.0f} s'.format(timeout_s), 'library_ok': None}

  def work():
    try:
      coll = parallel.collective()
      got = eng.debug_allreduce(coll, np.array([float(rank), 1.0], np.float32), 'sum')
      got_max = eng.debug_allreduce(coll, np.array([float(rank)], np.float32), 'max')
      out.update({'library_transport': TRANSPORT_NAMES.get(coll.transport, coll.transport),
                  'library_sum_of_ranks': float(got[0]), 'library_sum_of_ones': float(got[1]),
                  'library_max_of_ranks': float(got_max[0]),
                  'library_ok': bool(got[0] == world * (world - 1) / 2 and got[1] == world and got_max[0] == world - 1)})
    except Exception as e:  # pylint: disable=broad-except
      out.update({'library_transport': 'unavailable', 'library_error': repr(e), 'library_ok': False})

  t = threading.Thread(target=work, daemon=True)
  t.start()
  t.join(timeout_s)
  return dict(out), t.is_alive()


# ----------------------------------------------------------------------------- main
def main():
  if len(sys.argv) > 1 and sys.argv[1] == '--cpu-worker':
    return cpu_worker_main(sys.argv[2:])
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=100)
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--workload', default='heisenberg10x10_fc3x256_b4096', choices=sorted(WORKLOADS))
  ap.add_argument('--warm-sweeps', type=int, default=10, help='equilibration sweeps before the warm-up steps (BASELINE.md: 10; '
                  'fewer only for rocprofv3 --pmc passes of the general paths, whose sweeps are hundreds of launches)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-timing', action='store_true', help='disable per-kernel HIP events')
  ap.add_argument('--no-extra', action='store_true',
                  help='skip the legs after the timed region (soak, LogOverlapITSWO batch loop, SR CG loop, '
                       'default-hparams epoch); they only run at N = 1 on the headline workload anyway')
  ap.add_argument('--soak-seconds', type=float, default=10.0, help='length of the sustained-clock soak leg')
  ap.add_argument('--reps', type=int, default=REPS, help='repetitions of the timed region (median reported)')
  ap.add_argument('--collective', choices=['library', 'torch'], default=None,
                  help="N > 1: 'library' (default) = the step is ONE call of vmc_epoch_energy_gradient_dist, the "
                       "entry training.run_optimization_epoch uses, all-reduce issued in stream by the library "
                       "(transport: CGS_VMC_TRANSPORT); 'torch' = op by op with torch.distributed's async "
                       "all-reduce overlapped with the sweep")
  ap.add_argument('--spawn', action='store_true', help='start the rank processes from here even for --gpus 1 '
                  '(checks that the launcher path costs nothing)')
  args = ap.parse_args()

  if args.gpus < 1:
    ap.error('--gpus must be >= 1')
  env_world = os.environ.get('WORLD_SIZE')
  if env_world is None and (args.gpus > 1 or args.spawn):
    sys.exit(spawn_ranks(args.gpus))          # no GPU call has been made in this process
  world = int(env_world or '1')
  if world != args.gpus:
    sys.stderr.write('bench.py: --gpus {} but WORLD_SIZE={}; start {} ranks (or drop WORLD_SIZE and '
                     'let bench.py start them)\n'.format(args.gpus, world, args.gpus))
    sys.exit(2)

  import torch
  from cgs_vmc_amd import _hip, parallel
  from cgs_vmc_amd.engine import VmcEngine

  if world > 1:
    parallel.init_from_env('nccl')
  rank = parallel.rank()
  dev = parallel.local_rank()
  torch.cuda.set_device(dev)

  split_sweep = args.workload.endswith('_split3xbf16_sampler')
  split = args.workload.endswith('_split3xbf16') or split_sweep
  if split:
    os.environ['CGS_VMC_SPLIT_BF16'] = '2' if split_sweep else '1'
  lx, ly, nnn, L, h, b = WORKLOADS[args.workload][:6]
  ansatz, ksz = (WORKLOADS[args.workload][6:] + ('fully_connected', 0))[:2]
  conv = ansatz in ('conv_2d', 'res_net_2d')
  n = lx * ly
  bonds = torus_bonds(lx, ly, nnn)
  nb = len(bonds)
  jx, jz = couplings(nb, nnn)
  chain_offset = rank * b
  theta, cfg = make_inputs(n, h, L, b, chain_offset, ansatz, ksz)

  # lattice.torus_bonds: site = x + lx * y, so the reshape [-1, size_x, size_y, 1] of the
  # convolutional ansatz (wavefunctions.py:596) has size_x = ly, size_y = lx
  eng = VmcEngine(n, b, L, h, device=dev, chain_offset=chain_offset, seed=2024, ansatz=ansatz,
                  kernel_size=ksz, size_x=ly if conv else 0, size_y=lx if conv else 0)
  eng.set_params(theta)
  eng.set_configs(cfg)
  eng.set_bonds(bonds, jx, jz)
  if split and eng.kernel_path() != (5 if split_sweep else 4):
    raise SystemExit('bench.py: the split workload did not get the split kernels')
  proof = prove_collectives(eng, world, rank, dev) if world > 1 else None
  lib_proof, lib_stuck = prove_library_transport(eng, world, rank) if world > 1 else ({}, False)
  if world > 1 and lib_stuck:
    # a rank is stuck inside the transport check (communicator creation): no further collective can be
    # trusted to line up -- say so and stop
    if rank == 0:
      print(json.dumps({'metric': 'mc_sweep+local_energy_evals_per_sec', 'value': None, 'n_gpus': world,
                        'error': 'library transport check timed out', 'rccl': dict(proof, **lib_proof)}))
      sys.stdout.flush()
    os._exit(3)
  collective = args.collective or 'library'
  collective_fallback = None
  if world > 1:
    # every rank must take the same path: fall back together when ANY rank's check failed
    bad = parallel.allreduce_max(0.0 if lib_proof.get('library_ok') is True else 1.0) > 0.5
    if bad and collective == 'library':
      collective, collective_fallback = 'torch', 'library transport check failed: {}'.format(
          lib_proof.get('library_error', lib_proof.get('library_transport')))
  coll = parallel.collective() if (world > 1 and collective == 'library') else None
  for _ in range(args.warm_sweeps):                # 10 warm-up sweeps, one launch each
    eng.mc_steps(n, want_accepted=False)           # (every k_sweep16 launch of a run is one sweep, so
                                                   # rocprofv3's per-kernel average is per sweep)

  def step_library():
    # the same slice as ONE host call of the entry training.run_optimization_epoch uses for sharded
    # chains (vmc_epoch_energy_gradient_dist: reset, [accumulate, sweep] x 1, accumulator all-reduce
    # issued in stream by the library; no equilibration, no update_norm)
    eng.epoch_energy_gradient_dist(coll, 0, 1, n, 0.0)

  def step_torch():
    # one optimizer-step slice: fresh accumulators (training.py:613 / 758), gradient accumulate,
    # accumulator all-reduce (multi-GPU), one MC sweep
    eng.reset_accumulators()
    eng.accumulate(_hip.VMC_MODE_ENERGY_GRADIENT)
    # the sweep does not touch the accumulators: the RCCL all-reduce runs underneath it
    pending = parallel.allreduce_accumulators_begin(eng)
    eng.mc_steps(n, want_accepted=False)
    pending.wait()

  step = step_library if coll is not None else step_torch

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
  # HIP events on the library's streams: inside the timed region only around the two roofline
  # kernels (every recorded event drains the pipeline between two kernels, ~3.5 us); the small
  # kernels are timed in a few extra steps after the timed region
  eng.timing_enable(0 if args.no_timing else 2)
  eng.timing_reset()
  rep_s, rep_local = [], []
  for _ in range(max(1, args.reps)):
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
      step()
    eng.synchronize()
    torch.cuda.synchronize()
    rep_local.append(time.perf_counter() - t0)     # this rank alone: a straggler shows up here
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
      elapsed = parallel.allreduce_max(elapsed)
    rep_s.append(elapsed)
  elapsed = sorted(rep_s)[len(rep_s) // 2]
  local_ms = 1e3 * sorted(rep_local)[len(rep_local) // 2] / args.steps
  rank_ms = [local_ms]
  if world > 1:
    rank_ms = [None] * world
    torch.distributed.all_gather_object(rank_ms, local_ms)
  eng.timing_enable(False)
  main_timings = {name: eng.timing_get(name) for name in ('sweep', 'tail_eloc')}
  if not args.no_timing:
    eng.timing_enable(1)
    eng.timing_reset()
    for _ in range(min(args.steps, 20)):     # (5 samples were few for the 3 % slice they time)
      step()
    barrier()
    eng.timing_enable(False)

  # the RCCL all-reduce alone (blocking), for the record
  allreduce_ms = None
  if world > 1:
    barrier()
    t0 = time.perf_counter()
    for _ in range(10):
      parallel.allreduce_accumulators(eng)
    barrier()
    allreduce_ms = parallel.allreduce_max(1e3 * (time.perf_counter() - t0) / 10)

  # N > 1: the two ways of taking the step -- op by op with torch.distributed's all-reduce, and the
  # library's one-call entry with the all-reduce issued in stream -- from the same chains and step counter
  paths_agree = None
  if world > 1:
    start_cfg, start_step = eng.get_configs(), eng.step_counter
    eng.set_configs(start_cfg)          # both start from chains just loaded (no sampler hand-over of activations)
    step_torch(); acc_t = eng.get_accumulators()
    eng.set_configs(start_cfg); eng.step_counter = start_step
    try:
      lib_coll = parallel.collective()
      eng.epoch_energy_gradient_dist(lib_coll, 0, 1, n, 0.0)
      acc_l = eng.get_accumulators()
      scale = float(np.abs(acc_t).max())
      paths_agree = {'bitwise': bool(np.array_equal(acc_t, acc_l)),
                     'max_abs_diff_over_max_abs': float(np.abs(acc_t - acc_l).max() / scale) if scale > 0 else 0.0,
                     'g_count': [float(acc_t[-4]), float(acc_l[-4])], 'e_count': [float(acc_t[-7]), float(acc_l[-7])]}
    except Exception as e:  # pylint: disable=broad-except
      paths_agree = {'error': repr(e)}

  eng.local_energy(want_eloc=False)
  rows = eng.last_connected_rows()
  mean_e = eng.mean_energy()

  timings = {}
  for name in ('sweep', 'tail_eloc', 'tail_amp', 'z1', 'bond_list', 'eloc_reduce', 'grad', 'adam'):
    ms, cnt = main_timings[name] if name in main_timings else eng.timing_get(name)
    if cnt:
      timings[name] = {'ms_total': ms, 'launches': cnt, 'ms_avg': ms / cnt}

  # the legs BASELINE config 3's wording asks for beside the EnergyGradient slice (N = 1, the dense
  # headline workload only; they change theta, so they come after every headline reading)
  extra = None
  if world == 1 and not args.no_extra and args.workload == 'heisenberg10x10_fc3x256_b4096':
    try:
      extra = extra_legs(eng, step, n, b, L, h, args.soak_seconds)
    except Exception as e:  # pylint: disable=broad-except
      extra = {'error': repr(e)}

  if rank == 0:
    fa = f_amp(n, h, L, ansatz, ksz)
    hp = (h + 63) // 64 * 64
    n_hh = L - 1
    mfma_per_row = mfma_flops_per_amp(n, h, L, ansatz, ksz)   # H x H layers / convolutions of one amplitude, on MFMA
    p = eng.num_params
    flops_eloc = b * (1 + nb) * fa                  # nominal per E_loc batch
    flops_sweep = b * n * fa                        # nominal per sweep
    # executed on the matrix cores: one row per mc_step + the exact refresh of the final
    # chains (sweep); one row per antiparallel bond (local energy); delta chain + the dual
    # weight-gradient GEMMs (gradient sums)
    exec_sweep = b * (n + 1) * mfma_per_row
    exec_eloc = rows * mfma_per_row
    exec_grad = b * (n_hh * 2 * hp * hp + 4 * (n * h + n_hh * h * h + h))
    if conv:      # taped forward + transposed convolutions + the two correlation sums
      exec_sweep = b * n * mfma_per_row
      exec_grad = 4 * b * mfma_per_row
    ms_step = 1e3 * elapsed / args.steps
    out = {
        'metric': 'mc_sweep+local_energy_evals_per_sec',
        'value': world * b * args.steps / elapsed,
        'unit': 'chain-evals/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': ('f32 via 3xbf16 split, fp32 accumulate (E_loc row kernel and sampler; gradients native f32)' if split_sweep else
                  'f32 via 3xbf16 split, fp32 accumulate (E_loc row kernel only; sampler and gradients native f32)' if split else 'f32'),
        'data': 'synthetic',
        'config': {'workload': args.workload, 'lattice': '{}x{} torus'.format(lx, ly),
                   'n_sites': n, 'n_bonds': nb, 'ansatz': ('{} {} x {} filters, kernel {}, relu/exp'.format(ansatz, L, h, ksz) if conv
                              else 'fully_connected {}x{} relu/exp'.format(L, h)),
                   'chains_per_gpu': b, 'global_chains': world * b,
                   'step': 'reset + accumulate_gradients (E_loc + grad sums) + 1 MC sweep'
                           + ('' if world == 1 else
                              ' + accumulator all-reduce in stream (one vmc_epoch_energy_gradient_dist call)'
                              if coll is not None else
                              ' + RCCL accumulator all-reduce (torch.distributed, overlapped with the sweep)'),
                   'parallelism': 'chains sharded x{}'.format(world)},
        'repetitions': len(rep_s), 'statistic': 'median',
        'repetitions_ms_per_step': [1e3 * t / args.steps for t in rep_s],
        'mean_energy_per_site': mean_e / n,
        'connected_rows_last_eloc': rows,
        'kernels': timings,
    }
    if world > 1:
      proof['checked_allreduce'].update(lib_proof)
      proof['checked_allreduce']['ok'] = bool(lib_proof.get('library_ok') is True)
      out['rccl'] = dict(proof, library_transport=lib_proof['library_transport'], ranks=world,
                         timed_collective=('library entry (vmc_epoch_energy_gradient_dist), transport: '
                                           + lib_proof['library_transport']) if coll is not None
                         else 'torch.distributed async all-reduce on the accumulator buffer',
                         collective_fallback=collective_fallback, paths_agree=paths_agree,
                         allreduce_floats=2 * p + 8, allreduce_ms_blocking=allreduce_ms,
                         ms_per_step_ranks={'min': min(rank_ms), 'max': max(rank_ms), 'all': rank_ms})
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
      k_sweep, k_eloc = ('k_conv_sweep', 'k_conv_rows(eloc)') if conv else ('k_sweep16', 'k_tail16(eloc)')
      if not conv and h <= 256 and eng.sweep_tile() == 8:     # eight-chain tiles (csrc/sweep8.hip): <= half of the CUs otherwise
        k_sweep = 'k_sweep8'
      if not conv and h > 256:        # 257..512 relu units: the LDS-operand row kernel (tail_co.hip)
        k_eloc = 'k_tail_lds(eloc)' if h <= 512 else 'wide GEMM rows(eloc)'
      if conv and eng.kernel_path() == 6:     # the general convolution path (conv_general.hip)
        k_sweep, k_eloc = 'k_cgen_im2col + GEMM(sampler)', 'k_cgen_im2col + GEMM(eloc)'
        kw = ksz if ly > 1 else 1
        if h <= 64 and 2 <= ksz <= 7 and ksz * kw * 4 * ((h + 15) // 16) <= 208 and os.environ.get('CGS_VMC_CONV_BAND', '1') != '0':    # conv_band.hip (round 6)
          k_sweep, k_eloc = 'k_cgen_band(sampler)', 'k_cgen_band(eloc)'
        if eng.conv_patch(n):       # the patch sampler (csrc/conv_patch.hip): per step the two boxes of every convolution, not the lattice
          k_sweep = 'k_cgen_patch_sweep'
          taps = ksz * kw
          per_step = 2 * ((taps + 15) // 16) * ((taps + 3) // 4) * 2048       # first convolution: the taps over the MFMA's k index
          for l in range(1, L):
            s1, s2 = (l + 1) * (ksz - 1) + 1, (l + 1) * (kw - 1) + 1
            per_step += 2 * ((s1 * s2 + 15) // 16) * taps * 4 * 2048           # 16 channels x 16 positions x 4 channels per MFMA
          exec_sweep = b * n * per_step
          if rows >= 4 * b:         # ... and the local energies' rows through its ELOC form (cgen_forward: the same boxes around the exchanged bond)
            k_eloc = 'k_cgen_patch_sweep<ELOC>(eloc)'
            # a bond's two sites are neighbours: ONE box per convolution, e1 x e2 sites larger than s1 x s2 (the displacement);
            # the torus has as many bonds along either axis, and with J2 as many diagonal ones as nearest-neighbour ones
            kinds = [(1, 0), (0, 1)] + ([(1, 1), (1, 1)] if nnn else [])
            if ly == 1: kinds = [(1, 0)]
            per_row = 0.0
            for e1, e2 in kinds:
              mf = ((ksz + e1) * (kw + e2) + 15) // 16 * ((taps + 3) // 4)
              for l in range(1, L):
                s1, s2 = (l + 1) * (ksz - 1) + 1 + e1, (l + 1) * (kw - 1) + 1 + e2
                mf += (s1 * s2 + 15) // 16 * taps * 4
              per_row += mf * 2048 / len(kinds)
            exec_eloc = rows * per_row
      if not conv and h > 512:        # the general path: per mc_step one k_wide_step launch + the H x H layers as GEMMs
        k_sweep = 'k_wide_step + k_gemm_ring(sampler)'
      per_kernel = {
          k_sweep: {'ms_avg': ts * 1e3, 'flops_executed': exec_sweep, 'flops_nominal': flops_sweep},
          k_eloc: {'ms_avg': te * 1e3, 'flops_executed': exec_eloc, 'flops_nominal': flops_eloc},
      }
      if split_sweep:
        per_kernel[k_sweep]['kernel'] = 'k_sweep16s (3 x bf16 split)'
        per_kernel[k_sweep]['flops_executed_bf16'] = 6 * exec_sweep
        per_kernel[k_sweep]['f32_equivalent_tflops'] = exec_sweep / ts / 1e12
      if split:       # six bf16 products per fp32 product, priced against the BF16 peak
        per_kernel[k_eloc]['kernel'] = 'k_tail16r (3 x bf16 split, weights through an LDS-DMA ring)'
        per_kernel[k_eloc]['flops_executed_bf16'] = 6 * exec_eloc
        per_kernel[k_eloc]['f32_equivalent_tflops'] = exec_eloc / te / 1e12
      for v in per_kernel.values():
        t = v['ms_avg'] * 1e-3
        v['achieved'] = v['flops_executed'] / t / 1e12         # TFLOP/s issued to the matrix cores
        v['frac'] = v['achieved'] / FP32_MFMA_PEAK_TFLOPS
        if 'flops_executed_bf16' in v:
          v['achieved'] = v['flops_executed_bf16'] / t / 1e12
          v['peak'] = BF16_MFMA_PEAK_TFLOPS
          v['frac'] = v['achieved'] / BF16_MFMA_PEAK_TFLOPS
        v['algorithmic_tflops'] = v['flops_nominal'] / t / 1e12   # SURVEY 8d count / time; not a roofline fraction
      key = k_sweep if dom == 'sweep' else k_eloc
      # HBM bytes per launch from the committed rocprofv3 PMC passes of THIS command
      # (profiles/*_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, separate --pmc runs; see
      # tools/collect_profiles.sh); null when no matching profile is present
      traffic = None
      pmc = None
      suffix = {'heisenberg10x10_fc3x256_b4096': '', 'heisenberg10x10_fc3x256_b4096_split3xbf16': '_split', 'heisenberg10x10_fc3x256_b4096_split3xbf16_sampler': '_splits',
                'heisenberg16x16j1j2_fc6x256_b1024': '_config5', 'heisenberg16x16j1j2_fc6x256_b1024_split3xbf16_sampler': '_config5_splits',
                'heisenberg6x6_fc3x128_b1024': '_config2', 'heisenberg10x10_fc3x1024_b4096': '_fc3x1024',
                'heisenberg10x10_conv3x128k3_b1024': '_conv_general', 'heisenberg36x36_conv3x16k5_b32': '_conv_general_36x36', 'heisenberg36x36_conv3x64k3_b32': '_conv_general_36x36x64',
                'heisenberg10x10_conv5x16k5_b4096': '_conv', 'heisenberg16x16j1j2_conv5x16k5_b1024': '_conv16',
                'heisenberg10x10_fc3x512_b4096': '_fc3x512', 'heisenberg10x10_conv5x32k5_b4096': '_conv32'}.get(args.workload)
      for rnd in ('r6', 'r5', 'r4', 'r3', 'r2'):        # the newest committed profile of this workload
        tag = None if suffix is None else rnd + suffix
        tpath = os.path.join(ROOT, 'profiles', '{}_traffic.json'.format(tag))
        if tag and os.path.exists(tpath):
          break
        tag = None
      if tag:
        try:
          prof = json.load(open(tpath))
          collected_at = prof.pop('_collected_at_source_hash', None)
          for name, rec in prof.items():
            if key.startswith('k_cgen_patch_sweep'):      # one template, two forms: <K, KW, waves, ELOC>
              hit = name.startswith('k_cgen_patch_sweep') and name.endswith('true>' if 'ELOC' in key else 'false>')
            else:
              hit = name.startswith(('k_tail16r', 'k_tail16s') if (split and key == k_eloc) else (('k_sweep16s',) if (split_sweep and key == k_sweep) else (key.split('(')[0],)))
            if hit and rec.get('hbm_read_bytes') is not None:
              traffic = rec['hbm_read_bytes'] + (rec.get('hbm_write_bytes') or 0)
              pmc = {k: rec[k] for k in ('mfma_util', 'clock_ghz', 'median_us') if k in rec}
              pmc['source'] = 'profiles/{}_traffic.json'.format(tag)
              # counters are collected in separate rocprofv3 --pmc passes (tools/collect_profiles.sh),
              # not by this run: say whether the kernels have changed since
              pmc['collected_at_source_hash'] = collected_at
              pmc['stale'] = None if collected_at is None else bool(collected_at != _hip.source_hash())
          # The general paths (more than 512 units; convolutions beyond the fused kernels) keep their activations /
          # feature maps in HBM and run the products on k_gemm_ring with one launch shape per role: the counters of the
          # dominant role's shape (tools/summarise_profiles.py: "k_gemm_ring<...> @ grid <threads>"), per launch,
          # next to the bytes that product has to move: A rows x K in (a convolution's implicit gather reads the map,
          # rows x F, once), the weights, rows x columns out.
          general = (not conv and h > 512) or (conv and eng.kernel_path() == 6 and not key.startswith(('k_cgen_band', 'k_cgen_patch')))
          if general and traffic is None:
            shapes = {k2: v2 for k2, v2 in prof.items() if k2.startswith('k_gemm_ring') and ' @ grid ' in k2
                      and v2.get('hbm_read_bytes') is not None}
            if shapes:
              if dom == 'sweep':       # the sampler's shape is the one launched most often
                k2, rec = max(shapes.items(), key=lambda kv: kv[1].get('launches', 0))
              else:                    # the local energies' row blocks: the largest grid
                k2, rec = max(shapes.items(), key=lambda kv: int(kv[0].rsplit(' ', 1)[1]))
              cols = h if not conv else ((h + 127) // 128) * 128
              col_tiles = (cols + 127) // 128
              rows = int(k2.rsplit(' ', 1)[1]) // 512 // col_tiles * 128
              kdim = h if not conv else h * ksz * (ksz if ly > 1 else 1)
              a_bytes = rows * (h if conv else kdim) * 4           # the map / activation rows the product reads once
              algorithmic = a_bytes + kdim * h * 4 + rows * h * 4
              traffic = rec['hbm_read_bytes'] + (rec.get('hbm_write_bytes') or 0)
              pmc = {k3: rec[k3] for k3 in ('mfma_util', 'clock_ghz', 'median_us', 'launches') if k3 in rec}
              pmc.update(source='profiles/{}_traffic.json'.format(tag), launch_shape=k2, rows_per_launch=rows,
                         hbm_read_bytes=rec['hbm_read_bytes'], hbm_write_bytes=rec.get('hbm_write_bytes'),
                         algorithmic_bytes=algorithmic, traffic_over_algorithmic=traffic / algorithmic,
                         collected_at_source_hash=collected_at,
                         stale=None if collected_at is None else bool(collected_at != _hip.source_hash()))
        except Exception:  # pylint: disable=broad-except
          traffic = None
      nominal = per_kernel[key]['algorithmic_tflops']
      out['roofline'] = {
          'kernel': key, 'bound': 'mfma', 'achieved': per_kernel[key]['achieved'],
          'peak': per_kernel[key].get('peak', FP32_MFMA_PEAK_TFLOPS), 'unit': 'TFLOP/s', 'frac': per_kernel[key]['frac'],
          'traffic': traffic,
          'nominal_achieved': nominal, 'nominal_frac': nominal / FP32_MFMA_PEAK_TFLOPS,
          'step_frac': (exec_sweep + exec_eloc + exec_grad) / (ms_step * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
          'pmc': pmc,
          'note': 'achieved / frac = flops ISSUED to the matrix cores (antiparallel bonds only, '
                  'rank-2 first layer) / HIP-event kernel time / peak at the 2.4 GHz peak clock; it '
                  'equals rocprofv3 MFMA-busy x (measured clock / 2.4 GHz).  nominal_* = SURVEY.md '
                  '8d algorithmic count (includes the first-layer GEMM and the masked parallel '
                  'bonds that are never executed) / the same time: a throughput, not a utilisation. '
                  'step_frac = executed flops of the whole step / ms_per_step / peak.',
          'per_kernel': per_kernel,
      }
    if extra is not None:
      out['extra'] = extra
      out['extra_keys'] = sorted(extra)
    if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (rank 0's host cores)
      try:
        out['cpu_baseline'] = cpu_baseline(n, h, L, bonds, jx, jz, theta, cfg, ansatz, ksz, lx, ly, args.workload)
      except Exception as e:  # pylint: disable=broad-except
        out['cpu_baseline'] = {'error': repr(e)}
    print(json.dumps(out))
    sys.stdout.flush()
  eng.close()
  if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
