#!/bin/bash
# Per-kernel durations of the fused TD3 update (cn_td3_update) at batch 128: rocprofv3 --kernel-trace --stats over tools/learn_bench.py
# restricted to the fused mode.  Output: gpurun_out/<tag>/learn_kernels.txt
TAG="${1:-r05}"; cd "$(dirname "$0")/.."; OUT="$PWD/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp
D=/tmp/learn_prof; rm -rf "$D"
CN_BATCHES=128 CN_LEARN_MODES=fused rocprofv3 --kernel-trace --stats -d "$D" -o learn --output-format csv -- python tools/learn_bench.py > /dev/null 2>&1
F=$(find "$D" -name '*kernel_stats.csv' | head -1)
python - "$F" > "$OUT/learn_kernels.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].startswith(("void (anonymous namespace)::td3", "(anonymous namespace)::td3", "td3")) or "td3" in r["Name"])
print(f"{'kernel':<70} {'calls':>7} {'avg us':>8} {'share':>7}")
for r in rows:
    if "td3" not in r["Name"]: continue
    print(f"{r['Name'][:70]:<70} {int(r['Calls']):>7} {float(r['AverageNs'])/1e3:>8.2f} {100*float(r['TotalDurationNs'])/tot:>6.1f}%")
PY
cat "$OUT/learn_kernels.txt"
