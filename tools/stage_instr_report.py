#!/usr/bin/env python
"""tools/stage_instr.sh's rocprofv3 CSVs -> the dynamic instruction ledger: wave instructions per env-step of every stage."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
order = [l.split(None, 2) for l in open(os.path.join(d, "order.txt")) if l.startswith("cut")]
hdr = [l for l in open(os.path.join(d, "order.txt")) if l.startswith("kernel")][0].split()
kname, N = hdr[1], int(hdr[3])
ncut = len(order)
cols = {}
for f in sorted(glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].split("(")[0].strip() != kname or int(r.get("Grid_Size", 0) or 0) != 64 * N:
            continue
        per.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in per.items():
        v.sort()
        cols[c] = [x[1] / N for x in v[-ncut:]]        # the last ncut full-grid dispatches = the cuts, in order
names = [c for c in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
                     "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS",
                     "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH") if c in cols]
short = {"SQ_INSTS_VALU": "VALU", "SQ_INSTS_VALU_ADD_F64": "addf64", "SQ_INSTS_VALU_MUL_F64": "mulf64", "SQ_INSTS_VALU_FMA_F64": "fmaf64",
         "SQ_INSTS_VALU_TRANS_F64": "trans", "SQ_INSTS_VALU_INT32": "int32", "SQ_INSTS_VALU_INT64": "int64", "SQ_INSTS_VALU_CVT": "cvt",
         "SQ_INSTS_SALU": "SALU", "SQ_INSTS_SMEM": "SMEM", "SQ_INSTS_LDS": "LDS", "SQ_INSTS_VMEM_RD": "VMrd", "SQ_INSTS_VMEM_WR": "VMwr",
         "SQ_INSTS_BRANCH": "branch"}
print("# dynamic ledger of %s, %d envs: wave instructions per env-step issued in each stage (rocprofv3 PMC, launches cut at the stage stamps)" % (kname, N))
print("%-52s" % "stage (ends at stamp)" + "".join("%8s" % short[c] for c in names) + "%8s" % "other")
prev = {c: 0.0 for c in names}
for i, (_, k, what) in enumerate(order):
    row = {c: cols[c][i] - prev[c] for c in names}
    f64 = sum(row.get(c, 0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    other = row.get("SQ_INSTS_VALU", 0) - f64 - row.get("SQ_INSTS_VALU_INT32", 0) - row.get("SQ_INSTS_VALU_INT64", 0) - row.get("SQ_INSTS_VALU_CVT", 0)
    label = "TOTAL (full step)" if what.strip() == "FULL" else "%s (%s)" % (what.strip(), k)
    if what.strip() == "FULL":
        row = {c: cols[c][i] for c in names}
        f64 = sum(row.get(c, 0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
        other = row.get("SQ_INSTS_VALU", 0) - f64 - row.get("SQ_INSTS_VALU_INT32", 0) - row.get("SQ_INSTS_VALU_INT64", 0) - row.get("SQ_INSTS_VALU_CVT", 0)
    print("%-52s" % label[:52] + "".join("%8.0f" % row[c] for c in names) + "%8.0f" % other)
    prev = {c: cols[c][i] for c in names}
print("other = VALU that is neither float64 arithmetic nor int32 / int64 / conversion: moves, selects, compares, lane reads and writes")
