#!/bin/bash
# Runs on the GPU box (via gpurun): the VALU issue-rate calibration
# (tools/calib_valu.hip) alone and under the SQ counters bench.py reads.
# Usage: tools/profile_valu.sh <tag>   ->  gpurun_out/profiles_<tag>/<tag>_valu_calibration.json
set -u
TAG=${1:-r5}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/valu_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
cd /tmp
[ -x $ROOT/tools/calib_valu ] || hipcc --offload-arch=gfx950 -O3 $ROOT/tools/calib_valu.hip -o $ROOT/tools/calib_valu
timeout 120 $ROOT/tools/calib_valu > "$OUT/alone.json" 2> "$OUT/alone.err"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/sq" -o pmc -- $ROOT/tools/calib_valu > "$OUT/sq.log" 2>&1
# which VALU counters this rocprofv3 knows, and a pass with the cycle-weighted
# ones (absent counters make the pass fail: it is optional)
timeout 120 rocprofv3 --list-avail 2>&1 | grep -o "SQ_[A-Z_0-9]*VALU[A-Z_0-9]*" | sort -u > "$OUT/valu_counters.txt"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/sq2" -o pmc -- $ROOT/tools/calib_valu > "$OUT/sq2.log" 2>&1
python $ROOT/tools/summarize_valu.py "$OUT" "$SUM/${TAG}_valu_calibration.json"
find "$OUT" -name '*.db' -delete
