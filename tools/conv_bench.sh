#!/bin/bash
# The convolutional bench workloads, one JSON line each (no CPU baseline leg): regression check of the
# 16 / 32-filter kernels and the first numbers of the 48 / 64-filter and 8 / 9-tap instantiations.
# usage (GPU box): tools/conv_bench.sh [out_dir]
out=${1:-gpurun_out/r4/conv_bench}
mkdir -p "$out"
for w in heisenberg10x10_conv5x16k5_b4096 heisenberg10x10_conv5x32k5_b4096 heisenberg10x10_conv3x32k3_b4096 \
         heisenberg10x10_conv5x64k5_b1024 heisenberg10x10_conv3x48k3_b4096 heisenberg10x10_conv3x16k9_b4096 \
         heisenberg10x10_conv3x32k7_b4096 heisenberg10x10_resnet2x16k5_b4096; do
  python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > "$out/$w.json" 2> "$out/$w.err" || { echo "FAILED $w"; tail -5 "$out/$w.err"; }
  python - "$out/$w.json" <<'PY'
import json, sys
try:
  j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  r = j.get('roofline', {})
  print(j['config']['workload'], 'ms/step', round(j['ms_per_step'], 2), 'kernel', r.get('kernel'), 'frac', r.get('frac'), {k: v for k, v in j.items() if k.endswith('_ms') or 'frac' in k})
except Exception as e:
  print('unreadable', sys.argv[1], e)
PY
done
