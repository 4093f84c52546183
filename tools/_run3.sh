cd /root/repo
mkdir -p gpurun_out/r2c
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/pytest.log
tail -5 gpurun_out/r2c/pytest.log
timeout 600 python bench.py > gpurun_out/r2c/bench_c3.json 2> gpurun_out/r2c/bench_c3.err
timeout 300 python bench.py --workload heisenberg16x16j1j2_fc6x256_b1024 --no-cpu-baseline > gpurun_out/r2c/bench_c5.json 2> gpurun_out/r2c/bench_c5.err
python -c "
import json
for f in ('bench_c3','bench_c5'):
    d=json.load(open('gpurun_out/r2c/%s.json'%f)); print(f, d['ms_per_step'], d['repetitions_ms_per_step'], {k:round(v['ms_avg'],4) for k,v in d['kernels'].items()}); print(json.dumps(d['roofline'])[:1500]); print(d.get('cpu_baseline'))
"
