#!/bin/bash
# GPU box helper: the 3 x bf16 split experiment -- parity tests, the bench line of its workload tag and a
# rocprofv3 kernel-trace summary.  Usage: tools/split_bench.sh <tag>
set -uo pipefail
OUT=gpurun_out/r4/$1; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_split.py -q -m gpu -s > $OUT/tests.log 2>&1; echo rc=$? >> $OUT/tests.log
grep -E "max error|passed|failed|Error|rc=" $OUT/tests.log | head -12
python bench.py --workload heisenberg10x10_fc3x256_b4096_split3xbf16 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], {k: round(v["ms_avg"] * 1e3, 1) for k, v in d["kernels"].items()})
r = d["roofline"]["per_kernel"]["k_tail16(eloc)"]
print({k: r[k] for k in ("achieved", "frac", "f32_equivalent_tflops")})
PY
