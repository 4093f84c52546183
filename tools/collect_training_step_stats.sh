#!/bin/bash
# GPU box helper: rocprofv3 --kernel-trace --stats of the two "full training step" legs at config 3 that the bench
# line carries in `extra` -- the SR CG loop (tools/sr_bench.py 50 20: 50 recorded batches = 204,800 samples, 20 CG
# iterations) and the LogOverlapITSWO batch loop (tools/itswo_bench.py).   usage: tools/collect_training_step_stats.sh <tag>
set -uo pipefail
T=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${T}_training_step; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sr -- python3 $ROOT/tools/sr_bench.py 50 20 > $OUT/sr.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/itswo -- python3 $ROOT/tools/itswo_bench.py > $OUT/itswo.log 2>&1
cp "$(find $OUT/sr -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/${T}_sr_kernel_stats.csv
cp "$(find $OUT/itswo -name '*kernel_stats.csv' | head -1)" $ROOT/gpurun_out/${T}_itswo_kernel_stats.csv
rm -rf $OUT/sr $OUT/itswo
head -8 $ROOT/gpurun_out/${T}_sr_kernel_stats.csv | cut -c1-150
head -8 $ROOT/gpurun_out/${T}_itswo_kernel_stats.csv | cut -c1-150
