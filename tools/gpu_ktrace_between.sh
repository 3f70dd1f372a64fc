# Time of the between-wave kernels (refine / push / scan / top) launch by launch in the second repetition of a render
# (world $1), per library variant. Usage: tools/gpu_ktrace_between.sh <world> lib1.so ...
cd $GRAFT_REPO_ROOT
W=$1; shift
for L in "$@"; do
  if [ "$L" = default ]; then export PTGPU_LIB=$GRAFT_REPO_ROOT/path-tracer_amd/lib/libptgpu.so; else export PTGPU_LIB=$GRAFT_REPO_ROOT/path-tracer_amd/lib/variants/$L; fi
  echo "== $L (world $W)"
  for K in k_refine_sums k_refine_emit_count k_refine_write k_push_lazy; do bash tools/gpu_ktrace_kernel.sh $K $W || exit 1; done
done
