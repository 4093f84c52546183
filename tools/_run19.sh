set -e
cd /root/repo
mkdir -p gpurun_out/r2v
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2v/smoke.log 2>&1; tail -2 gpurun_out/r2v/smoke.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r2v/pytest.log 2>&1 || true
tail -3 gpurun_out/r2v/pytest.log
