#!/bin/bash
# Per-kernel times of the 3D voxel-grid build (tools/quick_build3d.py) under rocprofv3 --kernel-trace --stats; GPU box.
set -eo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof_build3d"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/tools/quick_build3d.py"
rocprofv3 --kernel-trace --stats -d "$OUT" -o b --output-format csv -- python3 "$ROOT/tools/quick_build3d.py" > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1] + '/b_kernel_stats.csv')):
    if 'ndt::' in r['Name']:
        print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), ('%.2f' % (float(r['AverageNs']) / 1e3)).rjust(8), ('%.2f' % (float(r['MinNs']) / 1e3)).rjust(8), ('%.2f' % (float(r['MaxNs']) / 1e3)).rjust(8))
PY
find "$OUT" -name "*.csv" -size +2M -delete
