set -e
cd /root/repo
mkdir -p gpurun_out/r2g
timeout -k 10 700 python -m pytest tests/test_gpu_conv.py -q -m gpu -x > gpurun_out/r2g/pytest_conv.log 2>&1 || true
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r2g/pytest_conv.log | tail -20
for wl in heisenberg10x10_conv5x16k5_b4096 heisenberg16x16j1j2_conv5x16k5_b1024; do
  timeout -k 10 300 python bench.py --workload $wl --steps 5 --warmup 1 --reps 3 --no-cpu-baseline > gpurun_out/r2g/bench_$wl.json 2> gpurun_out/r2g/bench_$wl.err || { echo "bench $wl failed"; tail -5 gpurun_out/r2g/bench_$wl.err; exit 1; }
  python -c "
import json,sys
d=json.load(open('gpurun_out/r2g/bench_$wl.json')); print('$wl', round(d['ms_per_step'],3), {k:round(v['ms_avg'],4) for k,v in d['kernels'].items()}, {k:round(v['frac'],3) for k,v in d['roofline']['per_kernel'].items()})
"
done
