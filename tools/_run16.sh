set -e
mkdir -p /root/repo/gpurun_out/r2q
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --steps 5 --warmup 1 --reps 1 --no-cpu-baseline --no-timing"
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --output-format csv -d /root/repo/gpurun_out/r2q/p1 -- $B > /root/repo/gpurun_out/r2q/p1.log 2>&1
cd /root/repo
python3 tools/pmc_summary.py gpurun_out/r2q/p1 > gpurun_out/r2q/pmc_fc.txt 2>&1
rm -rf gpurun_out/r2q/p1
head -24 gpurun_out/r2q/pmc_fc.txt
