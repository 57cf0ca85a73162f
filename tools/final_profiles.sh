#!/bin/bash
# The round's closing kernel traces, run through gpurun on the
# HEAD build:  gpurun -- 'bash tools/final_profiles.sh r6'
#   gpurun_out/<tag>_final/{bench,icp_slam_vga,icp_slam_720p}/..._results.db
# rocprofv3 7.x writes a rocpd SQLite database; tools/kstats.py turns it into
# the per-kernel CSV that is committed under profiles/.
TAG=${1:-r6}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
O=$ROOT/gpurun_out/${TAG}_final; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/bench" -o bench -- \
  python "$ROOT/bench.py" --no-secondary --no-cpu-baseline --no-pmc > "$O/bench.json" 2> "$O/bench.err"
timeout 300 rocprofv3 --kernel-trace --stats -d "$O/icp_slam_vga" -o vga -- \
  "$ROOT/examples/icp_slam" 60 640 480 > "$O/icp_slam_vga.json" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$O/icp_slam_720p" -o hd -- \
  "$ROOT/examples/icp_slam" 60 1280 720 > "$O/icp_slam_720p.json" 2>&1
cd "$ROOT"
for i in 1 2 3 4 5; do timeout 60 examples/icp_slam 60 640 480; done > "$O/icp_slam_vga_untraced.jsonl" 2>&1
for i in 1 2 3 4 5; do timeout 60 examples/icp_slam 60 1280 720; done > "$O/icp_slam_720p_untraced.jsonl" 2>&1
tail -c 600 "$O/bench.json"; echo
grep -o '"frames_per_s": [0-9.]*' "$O"/icp_slam_*_untraced.jsonl | tr '\n' ' '
