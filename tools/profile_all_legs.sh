set -eo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof_all"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench --output-format csv -- python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 > "$OUT/bench.log" 2>&1
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
cp "$f" "$OUT/kernel_stats.csv"
head -25 "$OUT/kernel_stats.csv" | cut -c1-150
