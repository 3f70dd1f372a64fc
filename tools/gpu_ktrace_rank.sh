cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=/tmp/kt_rank; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 240 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o kt -- python3 tools/render_rank.py 8 2 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cat $OUT/run.log | tail -3
ls $OUT
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$F" gpurun_out/kt_rank8.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void pt::", "").replace("pt::", "")
        f.write(f'{name};{(int(r["Start_Timestamp"]) - t0) / 1e3:.1f};{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.1f}\n')
PY
