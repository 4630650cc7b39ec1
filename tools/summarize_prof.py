#!/usr/bin/env python
"""Condense a tools/profile.sh output directory into the few numbers DESIGN.md / bench.py quote."""
import csv
import glob
import os
import sys


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def main(d):
    st = find(os.path.join(d, "trace"), "*kernel_stats.csv")
    if st:
        print("== kernel stats (%s)" % os.path.relpath(st, d))
        for row in csv.DictReader(open(st)):
            print("  %-28s calls %6s  total %12s ns  avg %12s ns  %6s%%" % (
                row.get("Name", "")[:28], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"),
                row.get("Percentage")))
    tr = find(os.path.join(d, "trace"), "*kernel_trace.csv")
    if tr:
        durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(tr))
                if "cn_env_kernel" in r["Kernel_Name"]]
        if len(durs) >= 4:
            # bench.py times the same-call-reset mode first, then the next-step-reset mode (the headline)
            h = len(durs) // 2
            print("== cn_env_kernel average duration by bench leg: same-call reset %.1f us (%d launches), "
                  "next-step reset %.1f us (%d launches)" % (sum(durs[:h]) / h / 1e3, h, sum(durs[h:]) / (len(durs) - h) / 1e3,
                                                             len(durs) - h))
    for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        f = find(os.path.join(d, tag), "*counter_collection.csv")
        if not f:
            print("== %s: no counter csv" % tag)
            continue
        acc, cnt = {}, {}
        for row in csv.DictReader(open(f)):
            if "cn_env_kernel" not in row.get("Kernel_Name", ""):
                continue
            k = row["Counter_Name"]; v = float(row["Counter_Value"])
            acc[k] = acc.get(k, 0.0) + v; cnt[k] = cnt.get(k, 0) + 1
        print("== %s (per cn_env_kernel dispatch, mean of %s)" % (tag, sorted(set(cnt.values()))))
        for k in sorted(acc):
            print("  %-24s %.6g" % (k, acc[k] / cnt[k]))
    for f in sorted(glob.glob(os.path.join(d, "bench_*.log"))):
        lines = [l for l in open(f).read().splitlines() if l.startswith("{")]
        if lines:
            import json
            j = json.loads(lines[-1])
            print("== %s: value %.4g env-steps/s, kernel_ms %.4f" % (os.path.basename(f), j["value"], j["roofline"]["kernel_ms"]))


if __name__ == "__main__":
    main(sys.argv[1])
