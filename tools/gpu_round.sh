# Full GPU pass: tests, smoke, bench (with CPU baseline + census), rocprofv3 kernel stats. Usage: tools/gpu_round.sh <tag>
set -e
cd $GRAFT_REPO_ROOT
TAG="${1:-r01}"
mkdir -p gpurun_out
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t_gpu_$TAG.log 2>&1 || { tail -60 gpurun_out/t_gpu_$TAG.log; exit 1; }
tail -3 gpurun_out/t_gpu_$TAG.log
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_$TAG.log 2>&1 || { tail -30 gpurun_out/smoke_$TAG.log; exit 1; }
tail -2 gpurun_out/smoke_$TAG.log
timeout -k 10 300 python3 bench.py --census > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -20 gpurun_out/bench_$TAG.err; exit 1; }
cat gpurun_out/bench_$TAG.json
bash tools/gpu_profile.sh $TAG
