#!/bin/bash
# LDS bank conflicts of cn_env_kernel attributed to stages by ablation (profiling build): the last 60 launches of each run carry
# the mask.   tools/lds_conflicts.sh <tag> -> gpurun_out/<tag>/lds_conflicts.txt
TAG="${1:-r03}"; cd "$(dirname "$0")/.."; REPO="$PWD"; OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
for M in 0 1 2 4 8 16 32; do
  rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN -d "$OUT/ldsc$M" -o pmc -- python $REPO/tools/lds_conflicts.py $M > "$OUT/ldsc$M.log" 2>&1
done
cd "$REPO"
python - "$OUT" <<'PY' | tee "$OUT/lds_conflicts.txt"
import csv, glob, os, sys
d = sys.argv[1]
names = {0: "nothing skipped", 1: "no pedestrians in the ray cast", 2: "type machine skipped", 4: "confirmation skipped", 8: "cone + top-K skipped",
         16: "tracker skipped (no tracks)", 32: "integer association off (float IoU test)"}
base = None
print("per env-step, last 60 launches of each run (4096 envs): LDS instructions, LDS-active cycles, bank-conflict cycles")
for m in (0, 1, 2, 4, 8, 16, 32):
    f = glob.glob(os.path.join(d, "ldsc%d" % m, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        print(m, "no csv"); continue
    rows = [r for r in csv.DictReader(open(f[0])) if r.get("Kernel_Name", "").split("(")[0].strip() == "cn_env_kernel"]
    by = {}
    for r in rows:
        by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    v = {k: sum(x[-60:]) / 60.0 / 4096.0 for k, x in by.items()}
    if m == 0:
        base = v
    print("  mask %2d %-44s insts %6.1f  active %7.1f  conflict %7.1f (%.0f %% of active)%s" % (
        m, names[m], v.get("SQ_INSTS_LDS", 0), v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0),
        100.0 * v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1e-9, v.get("SQ_LDS_IDX_ACTIVE", 1)),
        "" if m == 0 or base is None else "   -> this stage: insts %+.1f conflict %+.1f" % (base["SQ_INSTS_LDS"] - v.get("SQ_INSTS_LDS", 0), base["SQ_LDS_BANK_CONFLICT"] - v.get("SQ_LDS_BANK_CONFLICT", 0))))
PY
