#!/bin/bash
# Register / LDS / spill figures of every kernel in a built library (default: the product libcrowdnav.so).  The library is linked
# from several translation units, each with its own offload bundle in .hip_fatbin: every bundle is unpacked.
set -euo pipefail
LIB="${1:-$(dirname "$0")/../drl-based-mapless-crowd-navigation-with-perceived-risk_amd/lib/libcrowdnav.so}"
LLVM=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
"$LLVM/llvm-objcopy" --dump-section .hip_fatbin="$TMP/fat.bin" "$LIB"
python3 - "$TMP" <<'PY'
import sys
d = sys.argv[1]
b = open(d + "/fat.bin", "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
pos = [i for i in range(len(b)) if b.startswith(magic, i)]
for n, p in enumerate(pos):
    open("%s/bundle%d.bin" % (d, n), "wb").write(b[p:pos[n + 1] if n + 1 < len(pos) else len(b)])
PY
for B in "$TMP"/bundle*.bin; do
  "$LLVM/clang-offload-bundler" --type=o --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$B" --output="$B.co"
  "$LLVM/llvm-readelf" --notes "$B.co" | python3 -c '
import sys, re
txt = sys.stdin.read()
for blk in txt.split(".agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    print("%-28s vgpr %3s agpr %3s sgpr %3s sgpr_spill %4s vgpr_spill %3s scratch %5s lds_static %5s" % (
        g("name"), g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"),
        g("private_segment_fixed_size"), g("group_segment_fixed_size")))
'
done
rm -rf "$TMP"
