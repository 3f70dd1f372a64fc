# Usage: tools/gpu_variants2.sh script.py lib1.so lib2.so ...  ("default" = the in-tree library)
cd $GRAFT_REPO_ROOT
S="$1"; shift
for L in "$@"; do
  echo "== $L"
  if [ "$L" = default ]; then timeout -k 10 300 python -u $S || exit 1
  else PTGPU_LIB=$GRAFT_REPO_ROOT/path-tracer_amd/lib/variants/$L timeout -k 10 300 python -u $S || exit 1; fi
done
