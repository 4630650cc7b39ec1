#!/bin/bash
# tools/profc/build.sh -> <pkg>/lib/prof/libcrowdnav_profc.so + /tmp/profc/k1.out (the counted code object) + tools/profc/symbols.json
# The region-counter build of translation unit 1 (every one-step kernel); the other units are the product's.  Profiling only.
#   1. device code of unit 1 -> optimised LLVM IR with clang's region counters (atomic updates) and coverage mapping
#   2. tools/profc/rewrite_ir.py: every counter update -> cn_prof_hit() (per-wavefront + per-lane counts)
#   3. IR -> code object -> offload bundle; host side of unit 1 compiled against that bundle; link with the other units
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
PKG="$ROOT/drl-based-mapless-crowd-navigation-with-perceived-risk_amd"; SRC="$PKG/csrc"; W="${CN_PROFC_TMP:-/tmp/profc}"; mkdir -p "$W" "$PKG/lib/prof"
LL=/opt/rocm/lib/llvm/bin; HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-builtin-pow -Wno-unused-function"
INJ="-DCN_TU=1 -DCN_PROFC -include $ROOT/tools/profc/profc_block.h"
pids=()
for u in 2 3 4 5; do $HIPCC $FLAGS -DCN_TU=$u -mllvm -disable-machine-licm -c -o "$W/k$u.o" "$SRC/crowdnav_kernel.hip" & pids+=($!); done
$HIPCC $FLAGS -c -o "$W/abi.o" "$SRC/crowdnav_abi.hip" & pids+=($!)
$HIPCC $FLAGS -c -o "$W/td3.o" "$SRC/crowdnav_td3.hip" & pids+=($!)
if [ ! -f "$W/k1.ll" ] || [ "$SRC/crowdnav_kernel.hip" -nt "$W/k1.ll" ] || [ "$SRC/crowdnav_device.h" -nt "$W/k1.ll" ]; then
  $HIPCC $FLAGS $INJ -fprofile-instr-generate -fprofile-update=atomic -fcoverage-mapping -gline-tables-only --cuda-device-only -emit-llvm -S \
      -o "$W/k1.ll" "$SRC/crowdnav_kernel.hip" 2>&1 | grep -v "argument unused" || true
fi
python3 "$ROOT/tools/profc/rewrite_ir.py" "$W/k1.ll" "$W/k1m.ll"
# -O1 on the already optimised IR: the call sites stay where the counters were
$LL/clang -x ir -target amdgcn-amd-amdhsa -mcpu=gfx950 -O1 -fPIC -mllvm -amdgpu-internalize-symbols -c "$W/k1m.ll" -o "$W/k1_dev.o"
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$W/k1.out" "$W/k1_dev.o"
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
    -input=/dev/null -input="$W/k1.out" -output="$W/k1.hipfb"
$HIPCC $FLAGS $INJ --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$W/k1.hipfb" -c -o "$W/k1.o" "$SRC/crowdnav_kernel.hip" 2>&1 | grep -v "argument unused" || true
failed=0; for pid in "${pids[@]}"; do wait "$pid" || failed=1; done
[ "$failed" = 0 ] || { echo "profc/build.sh: a compile failed" >&2; exit 1; }
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$PKG/lib/prof/libcrowdnav_profc.so" "$W"/k1.o "$W"/k2.o "$W"/k3.o "$W"/k4.o "$W"/k5.o "$W"/abi.o "$W"/td3.o
python3 "$ROOT/tools/profc/covmap.py" "$W/k1.out" "$ROOT/tools/profc/symbols.json"
ls -la "$PKG/lib/prof/libcrowdnav_profc.so" "$ROOT/tools/profc/symbols.json"
