# SD-tree parity tests on the box
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_sdtree.py -v --timeout 300 --timeout-method thread ${1:+-k "$1"} > gpurun_out/t_sdtree.log 2>&1 || { tail -80 gpurun_out/t_sdtree.log; exit 1; }
tail -15 gpurun_out/t_sdtree.log
