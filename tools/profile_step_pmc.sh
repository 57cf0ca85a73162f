#!/bin/bash
# Runs on the GPU box (via gpurun): PMC passes of bench.py for the dominant
# kernel (FrameStepKernel) and the FETCH_SIZE / WRITE_SIZE calibration binary.
# Each counter set is its own rocprofv3 run (--pmc with --kernel-trace only).
# Usage: tools/profile_step_pmc.sh <tag> [env assignments for bench, e.g. O3DMI_EXACT_DIV=1]
# Summaries land in gpurun_out/profiles_<tag>/ (copied into profiles/ by hand).
set -u
TAG=${1:-r2}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
SUM=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$SUM"
export TMPDIR=/tmp
cd /tmp
for kv in "$@"; do export "$kv"; done
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-pmc --steps 2 --warmup 1 --batch 200 --no-secondary"
declare -A SETS
SETS[sq1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE GRBM_COUNT"
SETS[sq2]="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH"
SETS[sq3]="SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_BUSY_CU_CYCLES"
SETS[fetch]="FETCH_SIZE"
SETS[write]="WRITE_SIZE"
SETS[l2]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
SETS[ea]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
SETS[tcp]="TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"
for S in ${PMC_SETS:-sq1 sq2 sq3 fetch write l2 ea tcp}; do
  echo "== pmc $S: ${SETS[$S]} =="
  timeout 400 rocprofv3 --pmc ${SETS[$S]} --kernel-trace --output-format csv -d "$OUT/$S" -o pmc -- $BENCH > "$OUT/$S.log" 2>&1
  tail -1 "$OUT/$S.log" | cut -c1-200
done
python $ROOT/tools/summarize_pmc.py "$OUT" "$SUM/${TAG}_step_pmc.json" FrameStepKernel > /dev/null
python - <<PY
import json
d=json.load(open("$SUM/${TAG}_step_pmc.json"))
for k,v in d.items():
    print(k[:60]); print("  ", {a: round(b) for a,b in v.items()})
PY
if [ -x $ROOT/tools/calib_hbm ] && [ "${CALIB:-1}" = "1" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/calib_$C" -o pmc -- $ROOT/tools/calib_hbm > "$OUT/calib_$C.log" 2>&1
  done
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/calib_trace" -o trace -- $ROOT/tools/calib_hbm > "$OUT/calib_trace.log" 2>&1
  python $ROOT/tools/summarize_calib.py "$OUT" "$SUM/${TAG}_hbm_calibration.json"
fi
find "$OUT" -name '*.csv' -size +8M -delete
find "$OUT" -name '*.db' -delete
