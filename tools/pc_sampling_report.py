#!/usr/bin/env python3
"""Histogram of a rocprofv3 PC-sampling CSV: samples per instruction (and per stall reason where the method reports one).

    tools/pc_sampling_report.py <pc_sampling csv> [kernel_trace csv]  ->  stdout: header, totals; <csv dir>/hist.tsv: the histogram

The raw CSV has one row per sampled wavefront (hundreds of MB for a second of a full device); hist.tsv keeps
(code object offset or instruction text, comment = file:line chain, every categorical column) -> count."""
import collections
import csv
import os
import sys

path = sys.argv[1]
rd = csv.reader(open(path, newline=""))
hdr = next(rd)
print("columns:", hdr)
low = [h.lower() for h in hdr]


def col(*names):
    for n in names:
        for i, h in enumerate(low):
            if n in h:
                return i
    return None


ci = col("instruction_comment", "comment")
ii = None
for i, h in enumerate(low):
    if h == "instruction" or (h.startswith("instruction") and "comment" not in h and "type" not in h):
        ii = i
        break
# categorical columns worth keeping: anything with few distinct values that is not an id / timestamp
skip = ("timestamp", "dispatch", "correlation", "exec", "wave", "chiplet", "hw_id", "workgroup", "sample")
cat = [i for i, h in enumerate(low) if i not in (ci, ii) and not any(s in h for s in skip)]
hist = collections.Counter()
first = []
n = 0
for row in rd:
    n += 1
    if len(first) < 5:
        first.append(row)
    key = tuple([row[ii] if ii is not None else "", row[ci] if ci is not None else ""] + [row[i] for i in cat])
    hist[key] += 1
print("rows:", n)
for r in first:
    print(r)
out = os.path.join(os.environ.get("PCS_OUT", os.path.dirname(path)), "hist.tsv")
with open(out, "w") as f:
    f.write("\t".join(["count", "instruction", "comment"] + [hdr[i] for i in cat]) + "\n")
    for k, c in hist.most_common():
        f.write("\t".join([str(c)] + [x.replace("\t", " ") for x in k]) + "\n")
print("distinct keys:", len(hist), "->", out)
for k, c in hist.most_common(40):
    print("%8d  %5.2f%%  %s" % (c, 100.0 * c / max(n, 1), " | ".join(k)[:260]))
