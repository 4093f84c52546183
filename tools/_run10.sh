set -e
mkdir -p /root/repo/gpurun_out/r2k
cd /tmp && export TMPDIR=/tmp
for v in 2 3 4; do
SR_ROWDOT_PIPE=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r2k/prof$v -- python3 /root/repo/tools/sr_bench.py 50 10 > /root/repo/gpurun_out/r2k/sr_bench_$v.json 2>&1
f=$(find /root/repo/gpurun_out/r2k/prof$v -name "*kernel_stats.csv" | head -1)
test -n "$f" && grep -E "k_sr_rowdot" "$f" | cut -c1-160
rm -rf /root/repo/gpurun_out/r2k/prof$v
done
