# Per-launch kernel durations of one guided render, for each listed library variant ("default" = in-tree).
# Usage: tools/gpu_ktrace2.sh tag lib1 lib2 ...   -> gpurun_out/kt_<tag>_<lib>.csv (kernel name, start, duration)
cd $GRAFT_REPO_ROOT
TAG=$1; shift
export TMPDIR=/tmp
for L in "$@"; do
  if [ "$L" = default ]; then P=$GRAFT_REPO_ROOT/path-tracer_amd/lib/libptgpu.so; else P=$GRAFT_REPO_ROOT/path-tracer_amd/lib/variants/$L; fi
  OUT=/tmp/kt_$L; rm -rf $OUT; mkdir -p $OUT
  export PTGPU_LIB=$P PTGPU_NO_OVERLAP=1
  timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python3 tools/render_once.py 512 2 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
  F=$(find $OUT -name "*kernel_trace.csv" | head -1)
  python3 - "$F" gpurun_out/kt_${TAG}_$L.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void pt::", "").replace("pt::", "")
        f.write(f'{name},{(int(r["Start_Timestamp"]) - t0) / 1e3:.1f},{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.1f}\n')
PY
done
