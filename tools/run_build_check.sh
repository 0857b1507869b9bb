#!/bin/bash
# grid-build regression + timing on the GPU box: build tests, event timing, rocprof per-kernel stats
set -eo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$ROOT"
python3 -m pytest tests/test_gpu_ndt2d.py tests/test_gpu_single_sync_build.py tests/test_gpu_scan_sequence.py tests/test_gpu_map_io.py tests/test_gpu_ndt3d.py tests/test_gpu_scan_sequence3d.py -x -q 2>&1 | tail -3
python3 tools/quick_build_events.py "${1:-1,2}"
bash tools/run_build_profile.sh "${1:-1,2}" | grep "ndt::"
