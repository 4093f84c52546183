#!/bin/bash
ROOT=$(pwd)
mkdir -p gpurun_out/r4/dbg
for D in 0 1 2 3 4 5 7; do
 for S in 0 ; do
  (cd /tmp && export TMPDIR=/tmp && CGS_VMC_WGRAD_DBG=$D CGS_VMC_WGRAD_SLICES=$S rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r4/dbg/d${D}_s$S -o p -- python3 $ROOT/bench.py --steps 20 --warmup 5 --reps 2 --no-cpu-baseline --no-timing > /dev/null 2>&1)
  echo "dbg=$D slices=$S: $(grep k_wgrad gpurun_out/r4/dbg/d${D}_s$S/p_kernel_stats.csv | cut -d, -f2-4)"
 done
done
for S in 2 3 4 6 8 10; do
  (cd /tmp && export TMPDIR=/tmp && CGS_VMC_WGRAD_SLICES=$S rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r4/dbg/s$S -o p -- python3 $ROOT/bench.py --steps 20 --warmup 5 --reps 2 --no-cpu-baseline --no-timing > /dev/null 2>&1)
  echo "slices=$S: $(grep k_wgrad gpurun_out/r4/dbg/s$S/p_kernel_stats.csv | cut -d, -f2-4)"
done
