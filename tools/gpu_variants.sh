# Times tools/time_workloads.py with several builds of libptgpu (PTGPU_LIB). Usage: tools/gpu_variants.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
for L in "$@"; do
  echo "== $L"
  PTGPU_LIB=$GRAFT_REPO_ROOT/path-tracer_amd/lib/$L timeout -k 10 300 python -u tools/time_workloads.py || exit 1
done
