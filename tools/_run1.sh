set -x
cd /root/repo
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_engine.py -x -q -m gpu > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout 600 python tools/sweep_variants.py heisenberg10x10_fc3x256_b4096 4096 8192 > gpurun_out/r2a/variants.log 2>&1
timeout 300 python tools/sweep_variants.py heisenberg16x16j1j2_fc6x256_b1024 1024 2048 >> gpurun_out/r2a/variants.log 2>&1
cat gpurun_out/r2a/variants.log
timeout 300 python bench.py --workload heisenberg16x16j1j2_fc6x256_b1024 --no-cpu-baseline > gpurun_out/r2a/bench_c5.json 2> gpurun_out/r2a/bench_c5.err
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2a/bench_c3.json 2> gpurun_out/r2a/bench_c3.err
python -c "
import json
for f in ('gpurun_out/r2a/bench_c5.json','gpurun_out/r2a/bench_c3.json'):
    d=json.load(open(f)); print(f, d['ms_per_step'], {k:v['ms_avg'] for k,v in d['kernels'].items()})
"
