# A/B of library variants on the headline workload, stage times serialised (PTGPU_NO_OVERLAP=1) and overlapped.
# Usage: tools/gpu_variants3.sh lib1.so lib2.so ...  ("default" = the in-tree library)
cd $GRAFT_REPO_ROOT
for L in "$@"; do
  echo "== $L"
  if [ "$L" = default ]; then P=$GRAFT_REPO_ROOT/path-tracer_amd/lib/libptgpu.so; else P=$GRAFT_REPO_ROOT/path-tracer_amd/lib/variants/$L; fi
  PTGPU_LIB=$P PTGPU_NO_OVERLAP=1 timeout -k 10 200 python -u tools/time_guided_stages.py || exit 1
  PTGPU_LIB=$P timeout -k 10 200 python -u tools/time_guided_stages.py || exit 1
done
