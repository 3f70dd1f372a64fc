# Guided tests + timing, binned record pass (default) against the direct splat (PTGPU_DIRECT_SPLAT=1).
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -k "guided or furnace" > gpurun_out/t_quick.log 2>&1 || { tail -40 gpurun_out/t_quick.log; exit 1; }
tail -2 gpurun_out/t_quick.log
timeout -k 10 300 python -u tools/time_workloads.py 2>&1 | head -1
PTGPU_DIRECT_SPLAT=1 timeout -k 10 300 python -u tools/time_workloads.py 2>&1 | head -1
timeout -k 10 100 python -u tools/time_guided.py
PTGPU_DIRECT_SPLAT=1 timeout -k 10 100 python -u tools/time_guided.py
