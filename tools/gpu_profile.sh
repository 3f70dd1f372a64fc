# rocprofv3 kernel trace + stats of the default bench command. Usage: tools/gpu_profile.sh <tag> [--serial] [bench args]
# --serial: the record pass on the walk's stream (PTGPU_NO_OVERLAP=1): per-kernel times without a co-running kernel.
set -e
cd $GRAFT_REPO_ROOT
TAG="${1:-r01}"; shift || true
if [ "$1" = --serial ]; then export PTGPU_NO_OVERLAP=1; shift; fi
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && cp "$F" $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$TAG.csv
# the full per-dispatch trace is large; keep only the stats
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
python3 - $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$TAG.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0][-40:]
    print(f"{n:42s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:9.1f} max_us {float(r['MaxNs'])/1e3:9.1f}")
PY
