set -e
cd /root/repo
mkdir -p gpurun_out/r2s
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -q -m gpu -x > gpurun_out/r2s/pytest.log 2>&1 || true
tail -3 gpurun_out/r2s/pytest.log
timeout -k 10 200 python bench.py --no-cpu-baseline --reps 5 > gpurun_out/r2s/bench.json 2> gpurun_out/r2s/bench.err
python -c "
import json
d=json.load(open('gpurun_out/r2s/bench.json')); print(d['ms_per_step'], d['repetitions_ms_per_step'], {k:round(v['ms_avg'],4) for k,v in d['kernels'].items()})"
