#!/bin/bash
# Per-kernel times of tools/quick_build_order.py (random against bearing order) under rocprofv3 --kernel-trace; GPU box.
set -eo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof_build_order"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT" -o bo --output-format csv -- python3 "$ROOT/tools/quick_build_order.py" > "$OUT/run.log" 2>&1
cat "$OUT/run.log" | tail -2
python3 - "$OUT" <<'PY'
import csv, sys, statistics
rows = [r for r in csv.DictReader(open(sys.argv[1] + '/bo_kernel_trace.csv')) if 'ndt::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
half = len(rows) // 2
for label, part in (("random order", rows[:half]), ("rooms / scan by bearing", rows[half:])):
    d = {}
    for r in part:
        k = (r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ndt::', ''), int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']))
        d.setdefault(k, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print(label)
    for k, v in sorted(d.items()):
        print(f"  {k[0]:28s} {k[1]:5d} x {k[2]}  n={len(v):3d}  median {statistics.median(v):7.2f} us")
PY
find "$OUT" -name "*.csv" -size +2M -delete
