#!/bin/bash
# GPU box helper (round 5): the split SAMPLER experiment -- its parity tests, then the bench lines of the three tags
# (native, row kernel split, row kernel + sampler split).
set -uo pipefail
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests/test_gpu_split.py -x -q -m gpu -k "sampler" > gpurun_out/r5_splits_t.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5_splits_t.log
for w in heisenberg10x10_fc3x256_b4096 heisenberg10x10_fc3x256_b4096_split3xbf16 heisenberg10x10_fc3x256_b4096_split3xbf16_sampler; do
  timeout -k 10 200 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-extra > gpurun_out/r5_ab_$w.json 2> gpurun_out/r5_ab_$w.err
  echo "$w rc=$?"
  python - <<PY
import json
try:
  d = json.load(open("gpurun_out/r5_ab_$w.json"))
  print(d["ms_per_step"], {k: round(v["ms_avg"], 4) for k, v in d["kernels"].items()}, d["mean_energy_per_site"])
except Exception as e:
  print("no line:", e)
PY
done
