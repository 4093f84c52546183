set -e
cd /root/repo
mkdir -p gpurun_out/r2o
timeout -k 10 500 python tools/train_10x10.py EnergyGradient 150 conv_2d > gpurun_out/r2o/train_conv.txt 2>&1 || true
tail -3 gpurun_out/r2o/train_conv.txt
timeout -k 10 400 python tools/train_10x10.py EnergyGradient 150 res_net_2d > gpurun_out/r2o/train_resnet.txt 2>&1 || true
tail -3 gpurun_out/r2o/train_resnet.txt
timeout -k 10 200 python tools/train_10x10.py StochasticReconfiguration 60 > gpurun_out/r2o/train_sr.txt 2>&1 || true
tail -3 gpurun_out/r2o/train_sr.txt
timeout -k 10 200 python tools/train_10x10.py EnergyGradient 300 > gpurun_out/r2o/train_eg.txt 2>&1 || true
tail -3 gpurun_out/r2o/train_eg.txt
