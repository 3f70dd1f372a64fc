# HBM-side traffic per kernel for every bench workload (two --pmc passes each). Usage: tools/gpu_traffic_all.sh <tag>
cd $GRAFT_REPO_ROOT
TAG=$1
for W in guided path veach-mis lightpath guided-large; do
  echo "== $W"
  bash tools/gpu_traffic.sh ${TAG}_$W --workload $W > gpurun_out/traffic_${TAG}_$W.log 2>&1 || { tail -5 gpurun_out/traffic_${TAG}_$W.log; exit 1; }
  tail -3 gpurun_out/traffic_${TAG}_$W.log | cut -c1-200
done
