set -e
O=gpurun_out
PMC_STEPS=2 PMC_WARMUP=1 PMC_WARM_SWEEPS=1 STATS_STEPS=5 STATS_WARMUP=1 BENCH_ARGS="--steps 5 --warmup 1 --warm-sweeps 2 --reps 3 --no-extra --no-cpu-baseline" timeout -k 10 500 tools/collect_profiles.sh r6_conv_general_36x36 heisenberg36x36_conv3x16k5_b32
tools/conv_groups.sh heisenberg10x10_conv3x128k3_b1024 5 > $O/r6_conv_groups.txt
tools/conv_groups.sh heisenberg10x10_conv3x96k3_b1024 5 1 2 >> $O/r6_conv_groups.txt
tools/conv_groups.sh heisenberg36x36_conv3x64k3_b32 3 1 2 >> $O/r6_conv_groups.txt
CGS_VMC_CONV_PATCH=0 tools/conv_groups.sh heisenberg36x36_conv3x16k5_b32 3 1 2 >> $O/r6_conv_groups.txt
timeout -k 10 600 python3 tools/conv_patch_bench.py 36 36 3 16 5 32 256 1024 2>/dev/null > $O/r6_conv_patch_bench.txt
CGS_VMC_CONV_PATCH_PROF=1 timeout -k 10 250 python3 bench.py --workload heisenberg36x36_conv3x16k5_b32 --steps 2 --warmup 1 --warm-sweeps 1 --reps 1 --no-extra --no-cpu-baseline 2>$O/prof.err >/dev/null
grep k_cgen_patch $O/prof.err | tail -1 >> $O/r6_conv_patch_bench.txt
timeout -k 10 250 python3 bench.py --workload heisenberg36x36_conv3x64k3_b32 --steps 5 --warmup 1 --warm-sweeps 2 --reps 3 --no-extra --no-cpu-baseline 2>/dev/null > $O/r6_conv_general_36x36x64_bench.json
: > $O/r6_conv_routed.txt
for w in heisenberg24x24_conv2x16k5_b256 heisenberg20x20_conv3x16k3_b256; do for gen in 0 ""; do
  if [ -n "$gen" ]; then export CGS_VMC_CONV_GENERAL=$gen; else unset CGS_VMC_CONV_GENERAL; fi
  timeout -k 10 250 python3 bench.py --workload $w --steps 10 --warmup 2 --warm-sweeps 2 --reps 3 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print(d['config']['workload'], 'CGS_VMC_CONV_GENERAL=${gen:-unset}:', '%.2f ms per step;' % d['ms_per_step'], ', '.join('%s %.2f' % (n, v['ms_avg']) for n, v in k.items()), '; rows', d['connected_rows_last_eloc'], '; roofline kernel', d['roofline']['kernel'])" >> $O/r6_conv_routed.txt
done; done
unset CGS_VMC_CONV_GENERAL
timeout -k 10 200 python3 tools/step_host.py heisenberg6x6_fc3x128_b1024 2000 2>/dev/null > $O/r6_step_host.txt
timeout -k 10 200 python3 tools/step_host.py heisenberg10x10_fc3x256_b4096 500 2>/dev/null >> $O/r6_step_host.txt
cat $O/r6_conv_routed.txt $O/r6_conv_patch_bench.txt
