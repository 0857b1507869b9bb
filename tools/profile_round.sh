#!/bin/bash
# Profiles of the bench command for profiles/ (run on the GPU box, from the repo root):
#   1. rocprofv3 --kernel-trace --stats        -> per-kernel time
#   2. rocprofv3 --pmc FETCH_SIZE              -> HBM-side reads   (own pass, kernel trace only)
#   3. rocprofv3 --pmc WRITE_SIZE              -> HBM-side writes  (own pass)
#   4. the same two counters on tools/pmc_calib (1 GiB streamed reads / writes) for the unit check
#   5. rocprofv3 --pmc SQ_* (one pass, 8 slots)  -> VALU / LDS activity of the hot kernels
#   6. rocprofv3 --marker-trace --kernel-trace --stats -> the library's roctx ranges
# then tools/pmc_summarise.py folds them into gpurun_out/prof/{kernel_stats.csv,pmc_traffic.json,sq_summary.json}.
set -eo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof"
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --headline-only --steps 10 --warmup 2"
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o "$OUT/pmc_calib" "$ROOT/tools/pmc_calib.hip"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- $BENCH > "$OUT/bench_stats.log" 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o bench --output-format csv -- $BENCH > "$OUT/bench_fetch.log" 2>&1
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o bench --output-format csv -- $BENCH > "$OUT/bench_write.log" 2>&1
echo "WRITE_SIZE pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/calib_fetch" -o calib --output-format csv -- "$OUT/pmc_calib" > "$OUT/calib_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/calib_write" -o calib --output-format csv -- "$OUT/pmc_calib" > "$OUT/calib_write.log" 2>&1
echo "calibration passes done"
# where the issue slots of the two hot kernels go (SQ block: 8 slots, one pass)
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --kernel-trace -d "$OUT/sq" -o bench --output-format csv -- $BENCH > "$OUT/bench_sq.log" 2>&1
echo "SQ pass done"
# the library's roctx ranges (grid build / Gauss-Newton loop / batch / gather) over the same command
rocprofv3 --marker-trace --kernel-trace --stats -d "$OUT/markers" -o bench --output-format csv -- $BENCH > "$OUT/bench_markers.log" 2>&1
echo "marker pass done"
python3 "$ROOT/tools/pmc_summarise.py" "$OUT"
rm -f "$OUT/pmc_calib"
find "$OUT" -name "*.csv" -size +2M -delete     # keep the merge-back small; the summaries are what is judged
