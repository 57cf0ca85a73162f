#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats and PMC passes of bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
# Raw output goes to gpurun_out/prof_<tag>/, summaries to gpurun_out/profiles_<tag>/ (copied
# into profiles/ by hand after the call).
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline $*"
echo "== kernel trace ==" 
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1
tail -2 "$OUT/trace.log"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C =="
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o pmc -- $BENCH --steps 2 --warmup 1 > "$OUT/pmc_$C.log" 2>&1
  tail -1 "$OUT/pmc_$C.log"
done
python $ROOT/tools/summarize_profile.py "$OUT" "$SUM" "$TAG"
# keep the merged-back payload small
find "$OUT" -name '*.csv' -size +8M -delete
find "$OUT" -name '*.db' -delete
