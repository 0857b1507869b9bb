set -e
ROOT=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_b2; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o build --output-format csv -- python3 $ROOT/tools/quick_build2.py > $OUT/run.log 2>&1
cat $OUT/run.log | grep single
python3 - $OUT <<'PY'
import csv,sys,collections,numpy as np
rows=list(csv.DictReader(open(sys.argv[1]+'/build_kernel_trace.csv')))
d=collections.defaultdict(list)
for r in rows:
    if 'ndt::' in r['Kernel_Name']:
        d[(r['Kernel_Name'].split('(')[0][-28:],r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,x in d.items(): print(k,len(x),round(float(np.median(x)),2))
PY
