# guided timing only, with and without the record/walk overlap
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u tools/dbg2.py && PTGPU_NO_OVERLAP=1 timeout -k 10 300 python -u tools/dbg2.py
