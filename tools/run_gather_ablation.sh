#!/bin/bash
# tools-only: k_tile_gather with parts of its point loop compiled out (NDT_EXP_GATHER), timed by rocprofv3
set -eo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-0 1 2 3}; do
  OUT="$ROOT/gpurun_out/prof_gather$v"; rm -rf "$OUT"; mkdir -p "$OUT"
  if [ $v != 0 ]; then export NDT_HIP_LIB="$ROOT/tools/bin/libndt_gather$v.so"; fi
  rocprofv3 --kernel-trace --stats -d "$OUT" -o build --output-format csv -- python3 "$ROOT/tools/quick_build.py" 1 > "$OUT/run.log" 2>&1
  echo "variant NDT_EXP_GATHER=$v"
  python3 - "$OUT" <<'PY'
import csv, glob, sys
for fn in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'ndt::' in r['Name']:
            print('  ', r['Name'][:60].ljust(60), r['Calls'].rjust(6), ('%.2f' % (float(r['AverageNs']) / 1e3)).rjust(9), ('%.2f' % (float(r['MinNs']) / 1e3)).rjust(9), ('%.2f' % (float(r['MaxNs']) / 1e3)).rjust(9))
PY
  find "$OUT" -name "*.csv" -size +1M -delete
done
