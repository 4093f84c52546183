set -e
cd /root/repo
mkdir -p gpurun_out/r2n
for v in 0 1; do
for wl in heisenberg10x10_conv5x16k5_b4096 heisenberg16x16j1j2_conv5x16k5_b1024; do
  CGS_VMC_CONV_WG_PER_CU=$v timeout -k 10 300 python bench.py --workload $wl --steps 5 --warmup 1 --reps 3 --no-cpu-baseline > gpurun_out/r2n/bench_${v}_$wl.json 2> gpurun_out/r2n/bench_${v}_$wl.err
  python -c "
import json,sys
d=json.load(open('gpurun_out/r2n/bench_${v}_$wl.json')); print('one_per_cu=$v', '$wl', round(d['ms_per_step'],3), {k:round(v['ms_avg'],4) for k,v in d['kernels'].items()}, {k:round(v['frac'],3) for k,v in d['roofline']['per_kernel'].items()})
"
done; done
