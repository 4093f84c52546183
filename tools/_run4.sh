cd /root/repo
mkdir -p gpurun_out/r2d
timeout 1800 python -m pytest tests/test_gpu_activations.py -x -q -m gpu > gpurun_out/r2d/pytest_act.log 2>&1; echo "rc=$?" >> gpurun_out/r2d/pytest_act.log
tail -30 gpurun_out/r2d/pytest_act.log
timeout 1800 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_activations.py > gpurun_out/r2d/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2d/pytest.log
tail -8 gpurun_out/r2d/pytest.log
timeout 300 python bench.py --no-cpu-baseline --reps 3 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:round(v['ms_avg'],4) for k,v in d['kernels'].items()})"
