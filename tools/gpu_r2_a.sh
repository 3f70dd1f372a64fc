# Round-2 first pass: the no-alloc entry point's test alone (fresh process), then the whole GPU suite.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > gpurun_out/t_noalloc.log 2>&1 || { tail -40 gpurun_out/t_noalloc.log; exit 1; }
tail -5 gpurun_out/t_noalloc.log
timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > gpurun_out/t_gpu.log 2>&1 || { tail -60 gpurun_out/t_gpu.log; exit 1; }
tail -8 gpurun_out/t_gpu.log
