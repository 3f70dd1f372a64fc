# One rocprofv3 --pmc pass over a short bench run. Usage: tools/gpu_pmc.sh <tag> "<counters>" [bench args]
set -e
cd $GRAFT_REPO_ROOT
TAG="$1"; CTRS="$2"; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o pmc -- python3 bench.py --steps 1 --warmup 0 --timed-only "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
F=$(find $OUT -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} total {v:.4g}  per-dispatch {v / calls[(k, c)]:.4g}  dispatches {calls[(k, c)]}")
PY
find $OUT -name "*counter_collection.csv" -size +30M -delete || true
