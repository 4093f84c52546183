#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel-trace stats + separate PMC passes of one
# bench workload.  Usage: tools/collect_profiles.sh <tag> [workload]
# Output under gpurun_out/<tag>_prof/; summarise with tools/summarise_profiles.py <tag> and the
# summaries land in profiles/<tag>_*.
set -euo pipefail
TAG=${1:-r4}
WL=${2:-heisenberg10x10_fc3x256_b4096}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${TAG}_prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the program itself follows `--` (no env / bash -c hop under the profiler)
# PMC_STEPS / PMC_WARMUP / STATS_STEPS / STATS_WARMUP: smaller counts for the workloads of many launches per step (the
# general paths: a PMC pass serialises every dispatch; the default counts outran the box's 7-minute silence limit)
B="python3 $ROOT/bench.py --workload $WL --steps ${PMC_STEPS:-10} --warmup ${PMC_WARMUP:-2} --warm-sweeps ${PMC_WARM_SWEEPS:-10} --reps 1 --no-cpu-baseline --no-extra --no-timing"
# the stats pass runs the bench's own step count, so that its per-kernel average is the steady
# state the bench line reports (the first launches of a process run ~10 % slower: clocks, caches)
BS="python3 $ROOT/bench.py --workload $WL --steps ${STATS_STEPS:-100} --warmup ${STATS_WARMUP:-10} --reps 1 --no-cpu-baseline --no-extra --no-timing"
# a heartbeat under gpurun_out/ while the passes run (a long pass prints nothing)
( while true; do date +%s >> "$OUT/heartbeat"; sleep 45; done ) &
HB=$!
trap "kill $HB 2>/dev/null || true" EXIT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BS > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $B > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B > $OUT/pmc_write.log 2>&1
# summarise first and put the fresh traffic file where bench.py looks for it, so that the bench line
# kept with the profile carries these counters (roofline.pmc.stale = false), then add the line
cd "$ROOT" && python3 tools/summarise_profiles.py "$TAG"
cp "$ROOT/gpurun_out/${TAG}_summary/${TAG}_traffic.json" "$ROOT/profiles/"
# (BENCH_ARGS: e.g. "--steps 5 --warmup 1 --warm-sweeps 2 --reps 3 --no-extra --no-cpu-baseline" for a 36 x 36 lattice, whose CPU leg alone outruns the box)
python3 $ROOT/bench.py --workload $WL ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err
python3 tools/summarise_profiles.py "$TAG"
# raw counter dumps are large; the summaries under gpurun_out/<tag>_summary are what is kept.
# Reached only when every pass and the summary succeeded (set -e): a failed run keeps its dumps.
if [ -z "${KEEP_RAW:-}" ]; then rm -rf "$OUT/pmc_sq" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/stats"; fi
