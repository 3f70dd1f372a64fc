cd $GRAFT_REPO_ROOT
for L in "$@"; do
  echo "== $L"
  PTGPU_LIB=$GRAFT_REPO_ROOT/path-tracer_amd/lib/$L timeout -k 10 300 python -u tools/dbg2.py || exit 1
done
