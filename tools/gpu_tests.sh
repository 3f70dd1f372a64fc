# Runs the GPU test suite on the box: tools/gpu_tests.sh [pytest -k expression]
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K="${1:-}"
if [ -n "$K" ]; then
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread -k "$K" > gpurun_out/t_gpu.log 2>&1 || { tail -60 gpurun_out/t_gpu.log; exit 1; }
else
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > gpurun_out/t_gpu.log 2>&1 || { tail -60 gpurun_out/t_gpu.log; exit 1; }
fi
tail -40 gpurun_out/t_gpu.log
