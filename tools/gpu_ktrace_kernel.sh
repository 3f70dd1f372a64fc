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
d = d[half:]
if len(d) > 40:
    big = [x for x in d if x >= 1000]; mid = [x for x in d if 100 <= x < 1000]; small = [x for x in d if x < 100]
    print(sys.argv[2], f"launches {len(d)} sum {sum(d) / 1e3:.1f} ms; >=1 ms: {len(big)} sum {sum(big) / 1e3:.1f}; 0.1-1 ms: {len(mid)} sum {sum(mid) / 1e3:.1f}; <0.1 ms: {len(small)} sum {sum(small) / 1e3:.1f}")
else:
    print(sys.argv[2], "us:", [round(x) for x in d], "sum", round(sum(d)))
PY
