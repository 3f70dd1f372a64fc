# rocprofv3 kernel trace + stats of the default bench command. Usage: tools/gpu_profile.sh <tag>
set -e
cd $GRAFT_REPO_ROOT
TAG="${1:-r01}"
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
find $OUT -name "*stats*" | head
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$TAG.csv && cat "$F"
# the full per-dispatch trace is large; keep only the stats
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
