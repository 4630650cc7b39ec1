#!/bin/bash
# Extra SQ / SQC counter passes of the one-launch-per-step leg (instruction cache, scalar data cache, memory latency levels):
#   tools/pmc_extra.sh <tag>   -> gpurun_out/<tag>/pmc_extra.txt
TAG="${1:-r04}"; cd "$(dirname "$0")/.."; REPO="$PWD"; OUT="$REPO/gpurun_out/$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 100 --warmup 10 --repeats 2 --groups 1 --no-cpu-baseline --no-plateau --no-other-configs"
cd /tmp
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_SALU" \
         "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_WAVES" \
         "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
         "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $C -d "$OUT/pmcx$i" -o pmc -- $BENCH > "$OUT/pmcx$i.log" 2>&1
done
cd "$REPO"
python - "$OUT" <<'PY' | tee "$OUT/pmc_extra.txt"
import csv, glob, os, sys
d = sys.argv[1]
names = set()
for f in glob.glob(os.path.join(d, "pmcx*", "**", "*counter_collection.csv"), recursive=True):
    names |= {r.get("Kernel_Name", "").split("(")[0].strip() for r in csv.DictReader(open(f)) if r.get("Kernel_Name", "").startswith("cn_env_kernel")}
for k in sorted(names):
    print("==", k, "(per env-step, full 4096-env grid)")
    for f in sorted(glob.glob(os.path.join(d, "pmcx*", "**", "*counter_collection.csv"), recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if r.get("Kernel_Name", "").split("(")[0].strip() == k]
        if not rows:
            continue
        gmax = max(int(r.get("Grid_Size", 0) or 0) for r in rows)
        acc, cnt = {}, {}
        for r in rows:
            if int(r.get("Grid_Size", 0) or 0) != gmax:
                continue
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); cnt[r["Counter_Name"]] = cnt.get(r["Counter_Name"], 0) + 1
        steps = 100.0 if "_seq" in k else 1.0
        for c in sorted(acc):
            print("  %-30s %14.1f" % (c, acc[c] / cnt[c] / (gmax / 64) / steps))
PY
rm -rf "$OUT"/pmcx[0-9]*/
