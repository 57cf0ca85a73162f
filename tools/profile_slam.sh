#!/bin/bash
# rocprofv3 kernel-trace stats of the ICP tracking loop (tools/bench_slam.py --mode slam).
set -u
TAG=${1:-r1_slam}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- \
  python $ROOT/tools/bench_slam.py --mode slam "$@" > "$OUT/trace.log" 2>&1
tail -1 "$OUT/trace.log" > "$SUM/${TAG}_bench.json"
python $ROOT/tools/summarize_profile.py "$OUT" "$SUM" "$TAG"
find "$OUT" -name '*.csv' -size +8M -delete
find "$OUT" -name '*.db' -delete
