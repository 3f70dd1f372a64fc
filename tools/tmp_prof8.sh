set -e
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_rank8
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
export PTGPU_NO_OVERLAP=1
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o trace -- python3 tools/time_rank_of_8.py > $OUT/out.txt 2>&1 || { tail -20 $OUT/out.txt; exit 1; }
grep "rank 0 of" $OUT/out.txt | cut -c1-120
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# split the timeline into renders at k_pack_film dispatches
renders = [[]]
for r in rows:
    renders[-1].append(r)
    if 'k_pack_film' in r['Kernel_Name']: renders.append([])
for idx in (2, 4):   # render 2 = "rank 0 of 1" (after the two warm renders), render 4 = second "rank 0 of 8"
    if idx >= len(renders): continue
    agg = collections.Counter()
    for r in renders[idx]:
        agg[r['Kernel_Name'].split('(')[0][-34:]] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    print('render', idx, {k: round(v, 1) for k, v in agg.most_common(14)})
PY
