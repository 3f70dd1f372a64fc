# A/B of libptgpu builds (PTGPU_LIB) on the box. Usage: tools/gpu_variants.sh [--stages] lib1 lib2 ...
# "default" = the in-tree library, anything else = path-tracer_amd/lib/variants/<lib>. Without --stages: the six standard
# workloads of tools/time_render.py per library; with it: the headline workload's stage times, record pass serialised
# (PTGPU_NO_OVERLAP=1: each kernel alone on the GPU) and overlapped.
cd $GRAFT_REPO_ROOT
STAGES=0; if [ "$1" = --stages ]; then STAGES=1; shift; fi
for L in "$@"; do
  echo "== $L"
  if [ "$L" = default ]; then export PTGPU_LIB=$GRAFT_REPO_ROOT/path-tracer_amd/lib/libptgpu.so; else export PTGPU_LIB=$GRAFT_REPO_ROOT/path-tracer_amd/lib/variants/$L; fi
  if [ $STAGES = 1 ]; then
    PTGPU_NO_OVERLAP=1 timeout -k 10 200 python -u tools/time_render.py --spp 512 --stages || exit 1
    timeout -k 10 200 python -u tools/time_render.py --spp 512 --stages || exit 1
  else
    timeout -k 10 300 python -u tools/time_render.py || exit 1
  fi
done
