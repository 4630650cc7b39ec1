#!/bin/bash
# Builds libcrowdnav.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   build.sh            the product library
#   build.sh timing     ONLY the profiling sibling libcrowdnav_timing.so (stage time stamps, PMC calibration kernels,
#                       device-math test kernels; never loaded by the product)
#   build.sh all        both
#   -ffp-contract=off : no implicit FMA contraction; every fma() in the sources is explicit, which is
#                       what makes the simulator bit-reproducible against the CPU oracle
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-builtin-pow -Wall -Wno-unused-function"
WHAT="${1:-product}"
if [ "$WHAT" = "product" ] || [ "$WHAT" = "all" ]; then
  # build next to the target and rename: a process that already mapped the old file keeps it, nobody maps a partial one
  "$HIPCC" $FLAGS ${CN_EXTRA_FLAGS:-} -shared -o "$OUT/.libcrowdnav.so.$$" "$HERE/crowdnav_kernel.hip" "$HERE/crowdnav_abi.hip"
  mv -f "$OUT/.libcrowdnav.so.$$" "$OUT/libcrowdnav.so"
  echo "built $OUT/libcrowdnav.so"
fi
if [ "$WHAT" = "timing" ] || [ "$WHAT" = "all" ]; then
  "$HIPCC" $FLAGS -DCN_TIMING -shared -o "$OUT/.libcrowdnav_timing.so.$$" "$HERE/crowdnav_kernel.hip" "$HERE/crowdnav_abi.hip"
  mv -f "$OUT/.libcrowdnav_timing.so.$$" "$OUT/libcrowdnav_timing.so"
  echo "built $OUT/libcrowdnav_timing.so (stage time stamps; profiling only)"
fi
