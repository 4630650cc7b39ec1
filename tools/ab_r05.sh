#!/bin/bash
# Same-box A/B of library variants under lib/ab/*.so (round 5): speed (tools/ab_perf.py, interleaved) and the dynamic instruction
# counts of one launch per step (one rocprofv3 PMC pass of tools/pc_workload.py per variant).   tools/ab_r05.sh [tag]
cd "$(dirname "$0")/.."; TAG="${1:-ab}"; OUT=gpurun_out/r05; mkdir -p $OUT; export TMPDIR=/tmp
LIBS=$(ls drl-based-mapless-crowd-navigation-with-perceived-risk_amd/lib/ab/*.so)
python tools/ab_perf.py $LIBS 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_$TAG.txt
if [ -z "${CN_AB_NO_PMC:-}" ]; then
for L in $LIBS; do
  D=/tmp/abpmc_$(basename $L .so); rm -rf $D
  (cd /tmp && CN_LIB=$OLDPWD/$L rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $D -o pmc -- python $OLDPWD/tools/pc_workload.py 4096 120 > /dev/null 2>&1)
  python - $D $(basename $L) <<'PY' | tee -a $OUT/ab_$TAG.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc, cnt = {}, {}
for r in csv.DictReader(open(f[0])) if f else []:
    if not r["Kernel_Name"].startswith("cn_env_kernel") or int(r.get("Grid_Size", 0) or 0) != 4096 * 64: continue
    k = r["Counter_Name"]; acc[k] = acc.get(k, 0) + float(r["Counter_Value"]); cnt[k] = cnt.get(k, 0) + 1
m = {k: acc[k] / cnt[k] / 4096 for k in acc}
if m:
    print("%-28s per env-step: VALU %.0f  SALU %.0f  LDS %.0f  SMEM %.0f  branch %.0f  wave quad-cycles %.0f  launch cycles %.0f  VALU busy %.3f" % (
        sys.argv[2], m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_SALU", 0), m.get("SQ_INSTS_LDS", 0), m.get("SQ_INSTS_SMEM", 0), m.get("SQ_INSTS_BRANCH", 0),
        m.get("SQ_WAVE_CYCLES", 0), m.get("SQ_BUSY_CYCLES", 0) * 4096 / 32, 4 * m.get("SQ_ACTIVE_INST_VALU", 0) * 4096 / (1024 * m.get("SQ_BUSY_CYCLES", 1) * 4096 / 32)))
PY
done
fi
