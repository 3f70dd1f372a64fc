# HBM-side traffic of every kernel of the default bench workload: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE
# cannot share one), summarised per kernel into gpurun_out/traffic_<tag>.json. Usage: tools/gpu_traffic.sh <tag> [bench args]
set -e
cd $GRAFT_REPO_ROOT
TAG="$1"; shift
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/traffic_${TAG}_$C
  mkdir -p $OUT
  timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $OUT -o pmc -- python3 bench.py --steps 1 --warmup 0 --timed-only "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
done
python3 - "$TAG" <<'PY'
import csv, sys, json, glob, collections, os
tag = sys.argv[1]
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out"
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{root}/traffic_{tag}_{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0]
        agg[k] += float(r["Counter_Value"]); n[k] += 1
    for k in agg:
        out[k][c + "_KB_total"] = agg[k]; out[k]["dispatches"] = n[k]
    os.remove(f)
for k, d in out.items():
    # MI355X_MICROARCH.md "HBM": rocprofv3 reports KB; on gfx950 FETCH_SIZE counts 128-B read requests as 64 B -> x2
    rd = 2.0 * d.get("FETCH_SIZE_KB_total", 0.0) * 1024.0; wr = d.get("WRITE_SIZE_KB_total", 0.0) * 1024.0
    d["hbm_bytes_per_launch"] = (rd + wr) / max(1, d["dispatches"]); d["read_bytes_per_launch"] = rd / max(1, d["dispatches"]); d["write_bytes_per_launch"] = wr / max(1, d["dispatches"])
json.dump({"tag": tag, "correction": "FETCH_SIZE x2 (gfx950), KB->bytes x1024", "kernels": out}, open(f"{root}/traffic_{tag}.json", "w"), indent=1)
for k, d in out.items(): print(k, {a: round(b) for a, b in d.items()})
PY
