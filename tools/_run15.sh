set -e
mkdir -p /root/repo/gpurun_out/r2p
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --workload heisenberg16x16j1j2_conv5x16k5_b1024 --steps 3 --warmup 1 --reps 1 --no-cpu-baseline --no-timing"
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU --output-format csv -d /root/repo/gpurun_out/r2p/p1 -- $B > /root/repo/gpurun_out/r2p/p1.log 2>&1
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d /root/repo/gpurun_out/r2p/p2 -- $B > /root/repo/gpurun_out/r2p/p2.log 2>&1
cd /root/repo
python3 tools/pmc_summary.py gpurun_out/r2p/p1 gpurun_out/r2p/p2 > gpurun_out/r2p/pmc_conv16.txt 2>&1
rm -rf gpurun_out/r2p/p1 gpurun_out/r2p/p2
grep -A9 "k_conv_rows" gpurun_out/r2p/pmc_conv16.txt | head -40
