#!/bin/bash
# Per-kernel durations of the fused TD3 update (cn_td3_update) at batch 128: rocprofv3 --kernel-trace --stats over tools/learn_bench.py
# restricted to the fused mode.  Output: gpurun_out/<tag>/learn_kernels.txt
TAG="${1:-r05}"; cd "$(dirname "$0")/.."; OUT="$PWD/gpurun_out/$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp
D=/tmp/learn_prof; rm -rf "$D"
CN_BATCHES=${CN_BATCHES:-128} CN_LEARN_MODES=fused rocprofv3 --kernel-trace --stats -d "$D" -o learn --output-format csv -- python tools/learn_bench.py > /dev/null 2>&1
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
T=$(find "$D" -name '*kernel_trace.csv' | head -1)
python - "$T" >> "$OUT/learn_kernels.txt" <<'PY'
# per position in the update's launch sequence (an update starts at td3_prep_kernel): mean duration, and the mean gap to the next launch
import csv, sys, re
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "td3" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seqs, cur = [], []
for r in rows:
    if "td3_prep_kernel" in r["Kernel_Name"] and cur: seqs.append(cur); cur = []
    cur.append(r)
seqs = seqs[60:]          # warm-up
for n in sorted(set(len(q) for q in seqs)):
    grp = [q for q in seqs if len(q) == n]
    if len(grp) < 20: continue
    print(f"\nupdates of {n} launches ({len(grp)} of them): span first start -> last end {sum(int(q[-1]['End_Timestamp']) - int(q[0]['Start_Timestamp']) for q in grp) / len(grp) / 1e3:.1f} us")
    for k in range(n):
        name = re.search(r"td3_\w+", grp[0][k]["Kernel_Name"]).group(0)
        dur = sum(int(q[k]["End_Timestamp"]) - int(q[k]["Start_Timestamp"]) for q in grp) / len(grp) / 1e3
        gap = sum(int(q[k + 1]["Start_Timestamp"]) - int(q[k]["End_Timestamp"]) for q in grp) / len(grp) / 1e3 if k + 1 < n else 0.0
        gx = grp[0][k].get("Grid_Size_X", "?"); 
        print(f"  {k:2d} {name:<28} grid {grp[0][k].get('Grid_Size_X','?'):>6}x{grp[0][k].get('Grid_Size_Y','?')}x{grp[0][k].get('Grid_Size_Z','?')}  {dur:6.2f} us   gap to next {gap:5.2f} us")
PY
cat "$OUT/learn_kernels.txt"
