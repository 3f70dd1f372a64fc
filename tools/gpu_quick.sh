# Quick GPU check: selected parity tests + timing script. Usage: tools/gpu_quick.sh [pytest -k expr]
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
K="${1:-guided or furnace}"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "$K" > gpurun_out/t_quick.log 2>&1 || { tail -40 gpurun_out/t_quick.log; exit 1; }
tail -2 gpurun_out/t_quick.log
timeout -k 10 300 python -u tools/time_workloads.py
