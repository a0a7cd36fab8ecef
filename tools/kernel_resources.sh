#!/bin/bash
# Prints VGPR / AGPR / scratch / LDS / occupancy per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
# Usage: tools/kernel_resources.sh correrender_amd/csrc/kernels_rank.hip [filter-regex]
F=$1; FILTER=${2:-.}
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950 $RFLAGS \
  -Rpass-analysis=kernel-resource-usage -c "$F" -o /tmp/kres.o 2>&1 | \
  grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size| SGPRs:" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | \
  awk '/Function Name/{if(n)print n, v, a, s, o, l; n=$3} /^VGPRs:/{v="vgpr="$2} /^AGPRs:/{a="agpr="$2} /ScratchSize/{s="scratch="$3} /Occupancy/{o="occ="$3} /LDS Size/{l="lds="$4} END{print n, v, a, s, o, l}' | \
  sed 's/_ZN3crf//' | grep -E "$FILTER"
