# A/B of the fused guided walker (k_walk_guided) against separate extend/occlude launches: headline step, one rank's
# share of an 8-GPU job, and the guided parity tests on the fused path.
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_sdtree.py -m gpu -x -q --timeout 120 --timeout-method thread -k "guided or sdtree" || exit 1
for V in hybrid fused separate; do
  echo "== $V"
  if [ $V = separate ]; then export PTGPU_WALKS=1; fi; if [ $V = fused ]; then export PTGPU_WALKS=2; fi
  timeout -k 10 200 python -u tools/time_guided_stages.py || exit 1
  timeout -k 10 200 python -u tools/render_rank.py 8 || exit 1
done
unset PTGPU_WALKS
timeout -k 10 200 python -u tools/tree_sizes.py 1 || exit 1
