# A/B of one environment switch on the headline render: tools/gpu_ab_env.sh VAR=value [reps]   (alternates off / on)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for E in "PT_AB_OFF=1" "$1"; do
    echo "== $E"
    env $E timeout -k 10 200 python -u tools/time_render.py --spp 512 --reps 3 --stages 2>&1 | tail -1 | cut -c1-300 || exit 1
  done
done
