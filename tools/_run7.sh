cd /root/repo
mkdir -p gpurun_out/r2i
for ov in 1 2; do
CGS_VMC_OVERLAP=$ov timeout -k 10 200 python bench.py --no-cpu-baseline --reps 3 > gpurun_out/r2i/bench_ov$ov.json 2> gpurun_out/r2i/bench_ov$ov.err || { echo fail; tail -3 gpurun_out/r2i/bench_ov$ov.err; exit 1; }
python -c "
import json
d=json.load(open('gpurun_out/r2i/bench_ov$ov.json')); print('overlap=$ov', d['ms_per_step'], d['repetitions_ms_per_step'], {k:round(v['ms_avg'],4) for k,v in d['kernels'].items()})"
done
