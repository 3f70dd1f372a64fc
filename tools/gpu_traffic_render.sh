# HBM-side traffic per kernel of ONE render of the headline workload and nothing else (tools/render_rank.py 1 1):
# two rocprofv3 --pmc passes. Usage: tools/gpu_traffic_render.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1
for C in FETCH_SIZE WRITE_SIZE; do
  OUT=/tmp/tr_$C; rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $OUT -o pmc -- python3 tools/render_rank.py 1 1 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
done
python3 - $TAG <<'PY'
import csv, glob, collections, sys
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for ci, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    f = glob.glob(f"/tmp/tr_{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0].replace("void pt::", "").replace("pt::", "")
        tot[k][ci] += float(r["Counter_Value"]) * 1024.0 * (2.0 if ci == 0 else 1.0)
        if ci == 0: tot[k][2] += 1
for k, (rd, wr, n) in sorted(tot.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    print(f"{k:40s} launches {n:5d}  read {rd / 1e9:8.2f} GB  write {wr / 1e9:8.2f} GB  per launch {(rd + wr) / max(1, n) / 1e6:9.1f} MB")
PY
