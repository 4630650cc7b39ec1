#!/bin/bash
# Build the product library of another git revision next to the current one, for same-box A/B runs with tools/quick_perf.py:
#   tools/ab_build.sh <git-ref> [name]   ->  <pkg>/lib/libcrowdnav_<name>.so   (default name: base)
#   CN_LIB=<pkg>/lib/libcrowdnav_base.so python tools/quick_perf.py base; python tools/quick_perf.py new
set -euo pipefail
REF="$1"; NAME="${2:-base}"; cd "$(dirname "$0")/.."; ROOT="$PWD"
PKG="drl-based-mapless-crowd-navigation-with-perceived-risk_amd"
T=$(mktemp -d)
git archive "$REF" "$PKG/csrc" include | tar -x -C "$T"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-builtin-pow -Wno-unused-function -shared \
  -o "$ROOT/$PKG/lib/libcrowdnav_$NAME.so" "$T/$PKG/csrc/crowdnav_kernel.hip" "$T/$PKG/csrc/crowdnav_abi.hip" 2>&1 | grep -v "warning\|deg2rad\|\^\|generated" || true
rm -rf "$T"; ls -la "$ROOT/$PKG/lib/libcrowdnav_$NAME.so"
