# bench.py on the box for the listed workloads (default: the headline): one JSON line each into gpurun_out/bench_<tag>.jsonl
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG="$1"; shift
WL="${@:-guided}"
: > gpurun_out/bench_$TAG.jsonl
for W in $WL; do
  timeout -k 10 400 python3 bench.py --workload $W --steps 5 --warmup 2 >> gpurun_out/bench_$TAG.jsonl 2> gpurun_out/bench_$TAG.err || { tail -20 gpurun_out/bench_$TAG.err; exit 1; }
done
cat gpurun_out/bench_$TAG.jsonl
