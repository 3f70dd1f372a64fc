# One bench line per BASELINE config (GPU only, no CPU baseline). Usage: tools/gpu_workloads.sh <tag>
cd $GRAFT_REPO_ROOT
TAG="${1:-r01}"
OUT=gpurun_out/workloads_$TAG.jsonl
: > $OUT
timeout -k 10 300 python -u -m pytest tests/test_golden.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
for W in path veach-mis lightpath guided guided-large; do
  timeout -k 10 400 python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline >> $OUT 2> gpurun_out/workloads_$TAG.err || { tail -5 gpurun_out/workloads_$TAG.err; exit 1; }
  tail -1 $OUT | cut -c1-600
done
