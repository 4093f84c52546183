#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace stats + separate PMC passes of
# the default bench workload.  Output under gpurun_out/r1_prof/; summarise with
# tools/summarise_profiles.py and copy the summaries into profiles/.
set -u
OUT=/root/repo/gpurun_out/r1_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-timing"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
python /root/repo/bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
ls -R $OUT | head -30
