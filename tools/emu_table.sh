#!/bin/bash
# GPU side of an A/B over builds of the library (run through gpurun):
#   bash tools/emu_table.sh <tag> [worlds="2 4 8"] [forms="0 1"]
# For the in-tree library and every _ab/<name>/libo3d_mi355x.so (built HERE, on
# the CPU, by tools/build_variant.sh -- compiling on the GPU box costs GPU
# minutes) it runs one rank's share of an N-rank job (bench.py
# --emulate-world N) in the records (0) and raw (1) form of the chunk launch
# and prints frames/s of the whole job and the chunk launch's ms. About 4 s per
# cell on a warm box. Writes gpurun_out/<tag>/emu_<lib>_w<N>_raw<F>.json.
set -u
TAG=${1:-emu}; WORLDS=${2:-"2 4 8"}; FORMS=${3:-"0 1"}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/$TAG; mkdir -p "$O"
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary --no-configs4"
libs="default"
for d in _ab/*/; do
  n=$(basename "$d"); [ "$n" = r3lib ] && continue
  [ -f "$d/libo3d_mi355x.so" ] && libs="$libs $n"
done
for lib in $libs; do
  for w in $WORLDS; do
    for raw in $FORMS; do
      if [ "$lib" = default ]; then L="O3DMI_EMU_TABLE=1"; else L="O3DMI_LIB=$PWD/_ab/$lib/libo3d_mi355x.so"; fi
      env $L O3DMI_SLICED_RAW=$raw timeout 90 $B --emulate-world $w \
        > "$O/emu_${lib}_w${w}_raw${raw}.json" 2> "$O/emu_${lib}_w${w}_raw${raw}.err"
    done
  done
done
python - "$O" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "emu_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-44s %9d frames/s   chunk launch %.4f ms"
              % (os.path.basename(f)[4:-5], round(d["value"]),
                 d["roofline"].get("avg_kernel_ms") or 0))
    except Exception as e:  # noqa: BLE001
        print("%-44s failed: %s" % (os.path.basename(f), e))
PY
