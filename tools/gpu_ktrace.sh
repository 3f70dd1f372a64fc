# Kernel trace of one guided render (per-launch durations). Usage: tools/gpu_ktrace.sh <tag>
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_$1
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d $OUT -o kt -- python3 tools/render_once.py 512 2 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
tail -2 $OUT/run.log; find $OUT -name "*.csv" | head
