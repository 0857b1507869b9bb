set -eo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/sq3"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --kernel-trace -d "$OUT/sq" -o b3 --output-format csv -- python3 $ROOT/tools/quick_batch3d.py 256 4 > "$OUT/log.txt" 2>&1
python3 - <<PY
import csv, glob, collections
f=glob.glob("$OUT/sq/**/*counter_collection.csv", recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    if 'k_batch3' in r['Kernel_Name']:
        agg['k_batch3'][r['Counter_Name']]+=float(r['Counter_Value']); 
        if r['Counter_Name']=='SQ_WAVES': cnt['k_batch3']+=1
for k,v in agg.items():
    print(k, cnt[k], 'dispatches')
    for c,x in v.items(): print('  ',c, x/cnt[k])
PY
