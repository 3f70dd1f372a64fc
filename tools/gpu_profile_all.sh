# rocprofv3 kernel stats of bench.py for every workload. Usage: tools/gpu_profile_all.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=$1
export TMPDIR=/tmp
for W in path veach-mis lightpath guided guided-large; do
  OUT=/tmp/prof_${TAG}_$W; rm -rf $OUT; mkdir -p $OUT
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_${TAG}_$W.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
  F=$(find $OUT -name "*kernel_stats.csv" | head -1)
  cp "$F" gpurun_out/kernel_stats_${TAG}_$W.csv
  echo "== $W"; head -5 "$F" | cut -c1-60,200-330
done
