#!/bin/bash
# A/B builds of the library with other compile-time constants of
# vbg_stream.hip:  tools/build_variant.sh <name> -DO3DMI_RAW_CHUNK=8 ...
# -> _ab/<name>/libo3d_mi355x.so (use with O3DMI_LIB=...). The other objects
# come from the in-tree build (run python -m open3d_amd.build first).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p _ab/$name
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Wall -Wno-unused-function -Iinclude -Iopen3d_amd/csrc"
/opt/rocm/bin/hipcc $FLAGS "$@" -Rpass-analysis=kernel-resource-usage -c open3d_amd/csrc/vbg_stream.hip -o _ab/$name/vbg_stream.o 2> _ab/$name/resources.txt
objs=$(ls open3d_amd/lib/obj/*.o | grep -v vbg_stream.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _ab/$name/vbg_stream.o -lz -o _ab/$name/libo3d_mi355x.so
python - "$name" <<'PY'
import re,sys
t=open('_ab/%s/resources.txt'%sys.argv[1]).read()
for m in re.finditer(r"Function Name: (\S*ChunkIntegrateKernel\S*).*?VGPRs: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?ScratchSize \[bytes/lane\]: (\d+)", t, re.S):
    n=m.group(1)
    if 'Lb1ELb0E' in n or 'Lb0ELb0E' in n:   # raw/records, not pipelined
        print(n[-40:], 'vgpr', m.group(2), 'occ', m.group(3), 'scratch', m.group(4))
PY
