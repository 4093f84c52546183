set -e
mkdir -p /root/repo/gpurun_out/r2j
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r2j/prof -- python3 /root/repo/tools/sr_bench.py 50 10 > /root/repo/gpurun_out/r2j/prof.log 2>&1
cd /root/repo
f=$(find gpurun_out/r2j/prof -name "*kernel_stats.csv" | head -1)
test -n "$f" && head -12 "$f" | cut -c1-220 > gpurun_out/r2j/sr_kernel_stats.txt
cat gpurun_out/r2j/sr_kernel_stats.txt
rm -rf gpurun_out/r2j/prof
