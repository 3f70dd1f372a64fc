# Durations of one kernel (name prefix $1) launch by launch in the second repetition of one rank's share of an
# 8-GPU guided render. Usage: tools/gpu_ktrace_kernel.sh k_guide_splat_bins [world]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=/tmp/kt_kernel; rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python3 tools/render_rank.py ${2:-8} 2 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
F=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$F" "$1" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
half = len(d) // 2
print(sys.argv[2], "us:", [round(x) for x in d[half:]], "sum", round(sum(d[half:])))
PY
