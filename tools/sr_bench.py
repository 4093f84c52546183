"""Times the stochastic-reconfiguration solve at BASELINE config 3 (10x10 torus, FC 3x256,
4096 chains): one epoch slice of `n_store` accumulate calls, then CG iterations.
Usage: python tools/sr_bench.py [n_store] [cg_iters] [conv]   (conv: the 5 x 16-filter k 5 network)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cgs_vmc_amd.engine import VmcEngine  # noqa: E402

n_store = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
conv = len(sys.argv) > 3 and sys.argv[3] == 'conv'
n, h, L, b = 100, 256, 3, 4096
if conv:
  h, L = 16, 5
  theta, cfg = bench.make_inputs(n, h, L, b, 0, 'conv_2d', 5)
  eng = VmcEngine(n, b, L, h, ansatz='conv_2d', kernel_size=5, size_x=10, size_y=10)
else:
  theta, cfg = bench.make_inputs(n, h, L, b, 0)
  eng = VmcEngine(n, b, L, h)
eng.set_params(theta); eng.set_configs(cfg)
eng.set_bonds(bench.torus_bonds(10, 10, False), -1.0, 1.0)
eng.sr_reserve(n_store)
eng.mc_steps(10 * n)
eng.reset_accumulators()
for _ in range(n_store):
  eng.accumulate(0)
  eng.mc_steps(n)
eng.sr_solve(0.01, 0.0, 3)          # warm-up
eng.timing_enable(True); eng.timing_reset()
eng.synchronize(); t0 = time.perf_counter()
it, res = eng.sr_solve(0.01, 0.0, iters)
eng.synchronize(); t1 = time.perf_counter()
ms, launches = eng.timing_get('sr_matvec')
f_amp = 2 * (n * h + (L - 1) * h * h + h) if not conv else 2 * n * 25 * (h + (L - 1) * h * h)
# per stored sample and CG iteration the reverse-mode form executes 2 F_amp (t_b = sum_l delta_l .
# (a_{l-1} V_l): 1 F_amp; u = sum_b t_b O_b: 1 F_amp); the forward-mode tangent chain of round 1
# needed 3 F_amp, which is the count its 50 TFLOP/s figure was quoted on
flops = 2.0 * f_amp * b * n_store
print(json.dumps({
    'n_store': n_store, 'samples': n_store * b, 'cg_iters': it, 'rel_residual': res,
    'wall_ms_per_iter': (t1 - t0) * 1e3 / max(it, 1),
    'matvec_ms': ms / max(launches, 1), 'matvec_ms_per_batch': ms / max(launches, 1) / n_store,
    'matvec_tflops_executed': flops / (ms / max(launches, 1) * 1e-3) / 1e12,
    'matvec_frac_of_fp32_mfma_peak': flops / (ms / max(launches, 1) * 1e-3) / 1e12 / 157.3,
    'matvec_tflops_round1_count': 1.5 * flops / (ms / max(launches, 1) * 1e-3) / 1e12,
}))
