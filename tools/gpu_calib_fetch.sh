# FETCH_SIZE calibration on scattered reads (tools/calib_fetch.hip; built beforehand by `make calib`).
# Per case: a timed run, one --pmc pass for FETCH_SIZE and one for the raw request counters. Summary:
# gpurun_out/fetch_calibration.json (copied to profiles/r03_fetch_calibration.json).
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
BIN=$GRAFT_REPO_ROOT/path-tracer_amd/lib/calib_fetch
OUT=$GRAFT_REPO_ROOT/gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
# mode, table MiB, reads (2^20): a 41 MiB table is the headline's sampling arena (behind 4 MiB of L2 per XCD, inside the
# 256 MiB Infinity Cache); 8192 MiB is past every cache
CASES="0:41:0 0:8192:0 32:41:64 64:41:64 128:41:64 32:8192:256 64:8192:256 128:8192:256"
for C in $CASES; do
  M=${C%%:*}; R=${C#*:}; T=${R%%:*}; N=${R#*:}
  timeout -k 10 60 $BIN $M $T $N 3 > $OUT/time_${M}_${T}.json
  cat $OUT/time_${M}_${T}.json
  timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f_${M}_${T} -o pmc -- $BIN $M $T $N 1 > /dev/null 2> $OUT/f_${M}_${T}.err || { tail -5 $OUT/f_${M}_${T}.err; exit 1; }
  timeout -s KILL 60 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/r_${M}_${T} -o pmc -- $BIN $M $T $N 1 > /dev/null 2> $OUT/r_${M}_${T}.err || { tail -5 $OUT/r_${M}_${T}.err; exit 1; }
done
python3 - <<'PY'
import csv, glob, json, os
root = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/calib"
rows = []
for tf in sorted(glob.glob(root + "/time_*.json")):
    t = json.load(open(tf))
    key = f"{t['mode']}_{t['table_mib']}"
    c = {}
    for d in ("f_", "r_"):
        for f in glob.glob(f"{root}/{d}{key}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Kernel_Name"].startswith(("k_stream", "k_scatter", "void k_scatter")):
                    c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    req = t["requested_bytes_per_launch"]
    fetch = c.get("FETCH_SIZE", 0.0) * 1024.0
    t.update({"FETCH_SIZE_bytes": fetch, "TCC_EA0_RDREQ": c.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_RDREQ_32B": c.get("TCC_EA0_RDREQ_32B_sum"),
              "fetch_over_requested": fetch / req if req else None,
              "fetch_bytes_per_read": fetch / t["reads"] if t["mode"] else None})
    rows.append(t)
json.dump(rows, open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/fetch_calibration.json", "w"), indent=1)
for t in rows:
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items()})
PY
