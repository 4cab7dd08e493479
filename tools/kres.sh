#!/bin/bash
# kernel resource usage (and optionally the ISA) of one HIP source of tlab_amd/csrc:
#     tools/kres.sh htile.hip [grep pattern on the mangled kernel names]      -> VGPRs, spills, scratch, occupancy per kernel
#     tools/kres.sh -S htile.hip /tmp/htile.s                                 -> device ISA
cd "$(dirname "$0")/../tlab_amd/csrc" || exit 1
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
if [ "$1" = "-S" ]; then
    hipcc $F -S --cuda-device-only "$2" -o "$3" 2>&1 | grep -v "hip-link" ; exit 0
fi
hipcc $F -c "$1" -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|VGPRs Spill|ScratchSize|Occupancy" | paste - - - - - | sed 's/[a-z_0-9]*\.hip:[0-9:]* remark: *//g;s/\[-Rpass-analysis=kernel-resource-usage\]//g' | cut -c1-300 | grep -E "${2:-.}"; rm -f /tmp/kres_$$.o
