set -e
cd /root/repo
mkdir -p gpurun_out/r2j
timeout -k 10 600 python -m pytest tests/test_gpu_sr.py -x -q -m gpu > gpurun_out/r2j/pytest_sr.log 2>&1 || true
grep -E "^FAILED|^ERROR|passed|failed|Error" gpurun_out/r2j/pytest_sr.log | tail -20
timeout -k 10 300 python tools/sr_bench.py 50 10 > gpurun_out/r2j/sr_bench_50.json 2> gpurun_out/r2j/sr_bench_50.err
cat gpurun_out/r2j/sr_bench_50.json
timeout -k 10 300 python tools/sr_bench.py 8 20 > gpurun_out/r2j/sr_bench_8.json 2> gpurun_out/r2j/sr_bench_8.err
cat gpurun_out/r2j/sr_bench_8.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r2j/prof -- python3 /root/repo/tools/sr_bench.py 50 10 > /root/repo/gpurun_out/r2j/prof.log 2>&1
cd /root/repo
f=$(find gpurun_out/r2j/prof -name "*kernel_stats.csv" | head -1)
test -n "$f" && head -14 "$f" | cut -c1-200 > gpurun_out/r2j/sr_kernel_stats.txt
cat gpurun_out/r2j/sr_kernel_stats.txt
rm -rf gpurun_out/r2j/prof
