#!/bin/bash
# GPU box helper: the LogOverlapITSWO batch and the SR CG iteration at config 3 through the device-resident entries,
# with one rank and with two gloo ranks sharing the one GPU (functional timings of the library's in-stream
# collective path, not scaling numbers).  Usage: tools/collect_library_collectives.sh <tag>   ->  gpurun_out/<tag>_*
set -uo pipefail
cd "$(dirname "$0")/.."
T=${1:-r5}
F=gpurun_out/${T}_itswo_bench.txt
{
  echo "# tools/itswo_bench.py: LogOverlapITSWO batch at config 3 through the device-resident entries (all lines at the sources of this commit)"
  echo "# 1 rank:"
  timeout -k 10 200 python tools/itswo_bench.py 2>&1 | grep -v "amdgpu.ids"
  echo
  echo "# 2 gloo ranks sharing ONE GPU (functional timing of the library's in-stream collective path, not a scaling number): device hook (CGS_VMC_TRANSPORT=torch)"
  CGS_VMC_DIST_BACKEND=gloo CGS_VMC_TRANSPORT=torch timeout -k 10 300 python tools/itswo_bench.py --gpus 2 2>&1 | grep -v "amdgpu.ids"
  echo
  echo "# the same on the host hook"
  CGS_VMC_DIST_BACKEND=gloo CGS_VMC_TRANSPORT=host timeout -k 10 300 python tools/itswo_bench.py --gpus 2 2>&1 | grep -v "amdgpu.ids"
} > $F
F=gpurun_out/${T}_sr_bench.txt
{
  echo "# tools/sr_bench.py 50 20: one CG iteration over 204,800 stored samples, 1 rank"
  timeout -k 10 300 python tools/sr_bench.py 50 20 2>&1 | grep -v "amdgpu.ids"
  echo
  echo "# tools/sr_bench.py 25 20 --gpus 2: the same 204,800 samples sharded over 2 gloo ranks sharing ONE GPU (device hook; functional timing, not a scaling number)"
  CGS_VMC_DIST_BACKEND=gloo CGS_VMC_TRANSPORT=torch timeout -k 10 300 python tools/sr_bench.py 25 20 --gpus 2 2>&1 | grep -v "amdgpu.ids"
} > $F
tail -2 gpurun_out/${T}_itswo_bench.txt gpurun_out/${T}_sr_bench.txt
