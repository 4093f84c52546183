set -e
cd /root/repo
mkdir -p gpurun_out/r2t
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r2t/pytest.log 2>&1 || true
tail -3 gpurun_out/r2t/pytest.log
bash tools/collect_profiles.sh r2_conv heisenberg10x10_conv5x16k5_b4096 > gpurun_out/collect_r2_conv.log 2>&1
bash tools/collect_profiles.sh r2_conv16 heisenberg16x16j1j2_conv5x16k5_b1024 > gpurun_out/collect_r2_conv16.log 2>&1
timeout -k 10 300 python bench.py --workload heisenberg10x10_resnet2x16k5_b4096 --steps 20 --warmup 2 > gpurun_out/r2t/bench_resnet.json 2> gpurun_out/r2t/bench_resnet.err
timeout -k 10 300 python bench.py > gpurun_out/r2t/bench_c3.json 2> gpurun_out/r2t/bench_c3.err
ls gpurun_out/*_summary
