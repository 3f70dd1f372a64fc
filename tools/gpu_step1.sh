set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -v --timeout 120 --timeout-method thread -k "bvh_layout or coplanar or empty_batch" > gpurun_out/t_extend.log 2>&1 || { tail -40 gpurun_out/t_extend.log; exit 1; }
tail -15 gpurun_out/t_extend.log
