set -e
cd /root/repo
bash tools/collect_profiles.sh r2 heisenberg10x10_fc3x256_b4096 > gpurun_out/collect_r2.log 2>&1
bash tools/collect_profiles.sh r2_config5 heisenberg16x16j1j2_fc6x256_b1024 > gpurun_out/collect_r2_config5.log 2>&1
bash tools/collect_profiles.sh r2_conv heisenberg10x10_conv5x16k5_b4096 > gpurun_out/collect_r2_conv.log 2>&1
bash tools/collect_profiles.sh r2_conv16 heisenberg16x16j1j2_conv5x16k5_b1024 > gpurun_out/collect_r2_conv16.log 2>&1
ls gpurun_out/*_summary
