# Stage times (record pass on the main stream, no co-running kernels) for several builds. Usage: tools/gpu_variants_serial.sh lib1.so ...
cd $GRAFT_REPO_ROOT
export PTGPU_NO_OVERLAP=1
for L in "$@"; do
  echo "== $L"
  PTGPU_LIB=$GRAFT_REPO_ROOT/path-tracer_amd/lib/$L timeout -k 10 300 python -u tools/time_workloads.py 2>&1 | head -1 || exit 1
done
