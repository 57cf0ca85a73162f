#!/bin/bash
# configs[4] integrate leg on the GPU box: launch-by-launch event timings, the
# same command under rocprofv3 --kernel-trace (fused and un-fused launches) and
# under the counter passes (one block per pass, kernel trace only beside it).
#   bash tools/c4_profile.sh <tag> [extra c4_probe.py args]
# writes gpurun_out/<tag>/*; tools/c4_summarize.py turns it into profiles/.
set -u
TAG=${1:-r4_c4}; shift || true
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
PROBE="python $PWD/tools/c4_probe.py $*"

run_trace() {  # name, env...
  local name=$1; shift
  local d=/tmp/c4_$name
  rm -rf "$d"
  (cd /tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --stats \
      --output-format csv -d "$d" -o t -- $PROBE --passes 1 --stride 0 \
      > "$OUT/${name}_probe.json" 2> "$OUT/${name}.err")
  python tools/c4_summarize.py trace "$d" > "$OUT/${name}_trace.json" 2>> "$OUT/${name}.err"
  f=$(find "$d" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && head -12 "$f" > "$OUT/${name}_kernel_stats.csv"
}

run_pmc() {  # name, counters...
  local name=$1; shift
  local d=/tmp/c4_pmc_$name
  rm -rf "$d"
  (cd /tmp && timeout 900 rocprofv3 --pmc "$@" --kernel-trace \
      --output-format csv -d "$d" -o p -- $PROBE --passes 1 --stride 0 \
      > "$OUT/pmc_${name}_probe.json" 2> "$OUT/pmc_${name}.err")
  python tools/c4_summarize.py pmc "$d" > "$OUT/pmc_${name}.json" 2>> "$OUT/pmc_${name}.err"
}

timeout 600 $PROBE --passes 2 --out "$OUT/events.json" > "$OUT/events_brief.json" 2> "$OUT/events.err"
run_trace fused
run_trace nofuse O3DMI_NO_FUSE=1
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
if [ "${C4_EXTRA_PMC:-0}" = 1 ]; then
  run_pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  run_pmc tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum
fi
ls -la "$OUT"
