#!/bin/bash
# Builds libcrowdnav.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   build.sh            the product library
#   build.sh timing     ONLY the profiling sibling libcrowdnav_timing.so (stage time stamps, PMC calibration kernels,
#                       device-math test kernels; never loaded by the product)
#   build.sh all        both
#   -ffp-contract=off : no implicit FMA contraction; every fma() in the sources is explicit, which is
#                       what makes the simulator bit-reproducible against the CPU oracle
# crowdnav_kernel.hip is compiled as four units: CN_TU=1 (every one-step kernel), CN_TU=2 (the sequence / policy kernels of the
# plain simulator), CN_TU=3 (the sequence kernels of the social-force / wheel-ramp simulators), CN_TU=4 (their policy kernels + the
# 720-ray one), CN_TU=5 (round 6: the sequence / policy kernels of the contact ticks and of the two older observation layouts)
# -- units 2-5 with -mllvm -disable-machine-licm: see the note above the kernel definitions.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-builtin-pow -Wall -Wno-unused-function"
WHAT="${1:-product}"
build_lib() {   # $1 = output name, $2.. = extra flags
  local name="$1"; shift
  local T; T="$(mktemp -d)"
  trap 'rm -rf "$T"' RETURN
  local pids=()
  "$HIPCC" $FLAGS ${CN_EXTRA_FLAGS:-} "$@" -DCN_TU=1 -c -o "$T/k1.o" "$HERE/crowdnav_kernel.hip" & pids+=($!)
  "$HIPCC" $FLAGS ${CN_EXTRA_FLAGS:-} "$@" -DCN_TU=2 -mllvm -disable-machine-licm -c -o "$T/k2.o" "$HERE/crowdnav_kernel.hip" & pids+=($!)
  "$HIPCC" $FLAGS ${CN_EXTRA_FLAGS:-} "$@" -DCN_TU=3 -mllvm -disable-machine-licm -c -o "$T/k3.o" "$HERE/crowdnav_kernel.hip" & pids+=($!)
  "$HIPCC" $FLAGS ${CN_EXTRA_FLAGS:-} "$@" -DCN_TU=4 -mllvm -disable-machine-licm -c -o "$T/k4.o" "$HERE/crowdnav_kernel.hip" & pids+=($!)
  "$HIPCC" $FLAGS ${CN_EXTRA_FLAGS:-} "$@" -DCN_TU=5 -mllvm -disable-machine-licm -c -o "$T/k5.o" "$HERE/crowdnav_kernel.hip" & pids+=($!)
  "$HIPCC" $FLAGS ${CN_EXTRA_FLAGS:-} "$@" -c -o "$T/abi.o" "$HERE/crowdnav_abi.hip" & pids+=($!)
  "$HIPCC" $FLAGS ${CN_EXTRA_FLAGS:-} "$@" -c -o "$T/td3.o" "$HERE/crowdnav_td3.hip" & pids+=($!)
  local failed=0 pid
  for pid in "${pids[@]}"; do wait "$pid" || failed=1; done     # a bare `wait` returns 0 whatever the jobs returned
  if [ "$failed" != 0 ]; then echo "build.sh: a compile of $name failed" >&2; return 1; fi
  # link next to the target and rename: a process that already mapped the old file keeps it, nobody maps a partial one
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/.$name.$$" "$T/k1.o" "$T/k2.o" "$T/k3.o" "$T/k4.o" "$T/k5.o" "$T/abi.o" "$T/td3.o"
  mv -f "$OUT/.$name.$$" "$OUT/$name"
}
if [ "$WHAT" = "product" ] || [ "$WHAT" = "all" ]; then
  build_lib libcrowdnav.so
  echo "built $OUT/libcrowdnav.so"
fi
if [ "$WHAT" = "timing" ] || [ "$WHAT" = "all" ]; then
  build_lib libcrowdnav_timing.so -DCN_TIMING
  echo "built $OUT/libcrowdnav_timing.so (stage time stamps; profiling only)"
fi
