#!/bin/bash
# tools/kernel_resources.sh [source.hip] [kernel name regex] -- registers, scratch and occupancy the compiler reports for the
# kernels of one translation unit (no GPU needed: hipcc cross-compiles for gfx950).  A kernel that spills shows scratch > 0.
SRC=${1:-femto_amd/csrc/femto_amd_api.hip}; RX=${2:-count_direct_kernel}
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -Wno-unused-result -Wno-cuda-compat --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage -c "$SRC" -o /tmp/kernel_resources.o 2>&1 \
  | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | paste - - - - | grep -E "$RX" \
  | sed -E 's/.*Function Name: ([^ ]*).*VGPRs: ([0-9]+).*ScratchSize \[bytes\/lane\]: ([0-9]+).*Occupancy \[waves\/SIMD\]: ([0-9]+).*/\1 vgprs=\2 scratch=\3 waves_per_simd=\4/' \
  | sed -E 's/_ZN9femto_amd[0-9]+//; s/EEvNS_8DevIndex[^ ]*//' | cut -c1-160
