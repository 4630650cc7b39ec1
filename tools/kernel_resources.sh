#!/bin/bash
# Register / LDS / spill figures of every kernel in a built library (default: the product libcrowdnav.so).
set -euo pipefail
LIB="${1:-$(dirname "$0")/../drl-based-mapless-crowd-navigation-with-perceived-risk_amd/lib/libcrowdnav.so}"
LLVM=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
"$LLVM/llvm-objcopy" --dump-section .hip_fatbin="$TMP/fat.bin" "$LIB"
"$LLVM/clang-offload-bundler" --type=o --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$TMP/fat.bin" --output="$TMP/dev.co"
"$LLVM/llvm-readelf" --notes "$TMP/dev.co" | python3 -c '
import sys, re
txt = sys.stdin.read()
for blk in txt.split(".agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    print("%-28s vgpr %3s agpr %3s sgpr %3s sgpr_spill %4s vgpr_spill %3s scratch %5s lds_static %5s" % (
        g("name"), g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"),
        g("private_segment_fixed_size"), g("group_segment_fixed_size")))
'
rm -rf "$TMP"
