cd /root/repo
mkdir -p gpurun_out/r2b
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -5 gpurun_out/r2b/pytest.log
for ov in 0 1; do
CGS_VMC_OVERLAP=$ov timeout 300 python bench.py --workload heisenberg16x16j1j2_fc6x256_b1024 --no-cpu-baseline > gpurun_out/r2b/bench_c5_ov$ov.json 2> gpurun_out/r2b/bench_c5_ov$ov.err
done
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r2b/bench_c3.json 2> gpurun_out/r2b/bench_c3.err
python -c "
import json
for f in ('bench_c5_ov0','bench_c5_ov1','bench_c3'):
    d=json.load(open('gpurun_out/r2b/%s.json'%f)); print(f, d['ms_per_step'], d['mean_energy_per_site'], {k:round(v['ms_avg'],4) for k,v in d['kernels'].items()})
"
