#!/bin/bash
# GPU box helper (round 5): parity tests of the 3 x bf16 split kernels, then the bench line of the split workload
# with the per-wave weight stream (CGS_VMC_SPLIT_RING=0, k_tail16s) and with the LDS-DMA ring (k_tail16r).
set -uo pipefail
cd "$(dirname "$0")/.."
timeout -k 10 400 python -m pytest tests/test_gpu_split.py -x -q -m gpu > gpurun_out/r5_split_t.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5_split_t.log
for r in 0 1; do
  CGS_VMC_SPLIT_RING=$r timeout -k 10 200 python bench.py --workload heisenberg10x10_fc3x256_b4096_split3xbf16 --steps 100 --warmup 10 --no-cpu-baseline --no-extra > gpurun_out/r5_split_ring$r.json 2> gpurun_out/r5_split_ring$r.err
  echo "ring=$r rc=$?"
  python - <<PY
import json
try:
  d = json.load(open("gpurun_out/r5_split_ring$r.json"))
  print(d["ms_per_step"], {k: round(v["ms_avg"], 4) for k, v in d["kernels"].items()})
except Exception as e:
  print("no line:", e)
PY
done
