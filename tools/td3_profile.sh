#!/bin/bash
# rocprofv3 kernel trace of 200 fused TD3 updates -> gpurun_out/<tag>/td3_kernel_stats.txt (per-kernel averages of the update chain)
TAG="${1:-r04}"; cd "$(dirname "$0")/.."; REPO="$PWD"; OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/td3prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/td3prof -o t -- python "$REPO/tools/td3_profile.py" > /dev/null 2>&1
f=$(find /tmp/td3prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" > "$OUT/td3_kernel_stats.txt" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "td3_" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("rocprofv3 --kernel-trace --stats of tools/td3_profile.py: 200 updates of cn_td3_update at batch 128 (100 with the actor step), per-kernel averages")
print("sum of the td3 kernels per update: %.1f us" % (tot / 200 / 1e3))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    print("%-62s calls %5s  avg %9.1f ns  %6.2f %%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]), 100 * float(r["TotalDurationNs"]) / tot))
PY
cat "$OUT/td3_kernel_stats.txt"; rm -rf /tmp/td3prof
