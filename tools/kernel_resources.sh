#!/bin/bash
# Register / LDS / spill figures of the kernels in one translation unit (compiler view, -Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_resources.sh crowdstep.hip [name-filter (regex on the demangled name)]
R=$(cd "$(dirname "$0")/.." && pwd)
src=${1:-crowdstep.hip}
filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -I "$R/include" \
    -Rpass-analysis=kernel-resource-usage -c "$R/social_navigation_pyenvs_amd/csrc/$src" -o /dev/null 2>&1 |
  awk '/Function Name:/{name=$(NF-1)} / VGPRs:/{v=$(NF-1)} /AGPRs:/{ag=$(NF-1)} /TotalSGPRs:/{s=$(NF-1)} /ScratchSize/{sc=$(NF-1)} /Occupancy/{oc=$(NF-1)} /VGPRs Spill/{sp=$(NF-1)} /LDS Size/{print name, "vgpr="v, "sgpr="s, "scratch="sc, "spill="sp, "occ="oc, "lds="$(NF-1)}' |
  c++filt | sed 's/void (anonymous namespace):://; s/(cstep::KArgs)//' | grep -E "$filt"
