set -e
cd /root/repo
mkdir -p gpurun_out/r2l
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r2l/pytest.log 2>&1 || true
tail -4 gpurun_out/r2l/pytest.log
timeout -k 10 300 python bench.py --workload heisenberg10x10_conv5x16k5_b4096 --steps 5 --warmup 1 --reps 3 --no-cpu-baseline > gpurun_out/r2l/bench_conv.json 2> gpurun_out/r2l/bench_conv.err
python -c "
import json
d=json.load(open('gpurun_out/r2l/bench_conv.json')); print(round(d['ms_per_step'],3), {k:round(v['ms_avg'],4) for k,v in d['kernels'].items()})"
