# rocprofv3 kernel traces of guided renders on the box.
#   tools/gpu_ktrace.sh render <tag> [lib ...]     per-launch (name, start us, duration us) of two 512-spp renders, per library
#                                                  variant ("default" = in-tree, else path-tracer_amd/lib/variants/<lib>),
#                                                  record pass serialised -> gpurun_out/kt_<tag>_<lib>.csv
#   tools/gpu_ktrace.sh rank [world]               the same for one rank of `world` (tools/render_rank.py) -> gpurun_out/kt_rank<world>.csv
#   tools/gpu_ktrace.sh kernel <name> [world]      launch-by-launch durations of one kernel in that rank's second repetition
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
MODE=$1; shift
to_csv() {
python3 - "$1" "$2" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void pt::", "").replace("pt::", "")
        f.write(f'{name};{(int(r["Start_Timestamp"]) - t0) / 1e3:.1f};{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.1f}\n')
PY
}
case $MODE in
render)
  TAG=$1; shift
  for L in "${@:-default}"; do
    if [ "$L" = default ]; then P=$GRAFT_REPO_ROOT/path-tracer_amd/lib/libptgpu.so; else P=$GRAFT_REPO_ROOT/path-tracer_amd/lib/variants/$L; fi
    OUT=/tmp/kt_$L; rm -rf $OUT; mkdir -p $OUT
    PTGPU_LIB=$P PTGPU_NO_OVERLAP=1 timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python3 tools/time_render.py --spp 512 --reps 2 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
    to_csv "$(find $OUT -name '*kernel_trace.csv' | head -1)" gpurun_out/kt_${TAG}_$L.csv
  done ;;
rank|kernel)
  if [ $MODE = kernel ]; then K=$1; shift; fi
  W=${1:-8}
  OUT=/tmp/kt_rank; rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python3 tools/render_rank.py $W 2 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
  tail -2 $OUT/run.log
  F=$(find $OUT -name "*kernel_trace.csv" | head -1)
  if [ $MODE = rank ]; then to_csv "$F" gpurun_out/kt_rank$W.csv; else
python3 - "$F" "$K" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
d = d[len(d) // 2:]
if len(d) > 40:
    big = [x for x in d if x >= 1000]; mid = [x for x in d if 100 <= x < 1000]; small = [x for x in d if x < 100]
    print(sys.argv[2], f"launches {len(d)} sum {sum(d) / 1e3:.1f} ms; >=1 ms: {len(big)} sum {sum(big) / 1e3:.1f}; 0.1-1 ms: {len(mid)} sum {sum(mid) / 1e3:.1f}; <0.1 ms: {len(small)} sum {sum(small) / 1e3:.1f}")
else:
    print(sys.argv[2], "us:", [round(x) for x in d], "sum", round(sum(d)))
PY
  fi ;;
*) echo "usage: tools/gpu_ktrace.sh render <tag> [lib ...] | rank [world] | kernel <name> [world]"; exit 2 ;;
esac
