#!/bin/bash
# Per-kernel times of the grid build (tools/quick_build.py) under rocprofv3 --kernel-trace --stats; run on the GPU box.
set -eo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
VARIANTS="${1:-1,2}"
OUT="$ROOT/gpurun_out/prof_build"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/tools/quick_build.py" "$VARIANTS"
rocprofv3 --kernel-trace --stats -d "$OUT" -o build --output-format csv -- python3 "$ROOT/tools/quick_build.py" "$VARIANTS" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
for fn in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        print(r['Name'][:78].ljust(78), r['Calls'].rjust(6), ('%.2f' % (float(r['AverageNs']) / 1e3)).rjust(9), ('%.2f' % (float(r['MinNs']) / 1e3)).rjust(9),
              ('%.2f' % (float(r['MaxNs']) / 1e3)).rjust(9))
PY
# per call: the 1M-point set_target and the 100k-point update launch the same kernels with different grids
python3 - "$OUT" <<'PY' > "$OUT/build_per_call.txt"
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1] + '/build_kernel_trace.csv')))
d = collections.defaultdict(list)
for r in rows:
    if 'ndt::' in r['Kernel_Name']:
        d[(r['Kernel_Name'].split('(')[0].replace('void ', ''), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']))].append(
            (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print("kernel | workgroups x | y | dispatches | median us | min us")
for k, x in sorted(d.items()):
    print(f"{k[0]} | {k[1]} | {k[2]} | {len(x)} | {statistics.median(x):.2f} | {min(x):.2f}")
PY
cat "$OUT/build_per_call.txt"
find "$OUT" -name "*.csv" -size +2M -delete
