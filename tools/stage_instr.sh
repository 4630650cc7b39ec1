#!/bin/bash
# Dynamic per-stage instruction ledger (GPU box):  tools/stage_instr.sh <tag>  -> gpurun_out/<tag>/stage_instr.txt
TAG="${1:-r04}"; cd "$(dirname "$0")/.."; REPO="$PWD"; OUT="$REPO/gpurun_out/$TAG/stage_instr"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES" \
         "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  rocprofv3 --output-format csv --pmc $C -d "$OUT/pass$i" -o pmc -- python "$REPO/tools/stage_instr.py" 2>/dev/null | grep -v amdgpu > "$OUT/order.txt"
done
cd "$REPO"
python tools/stage_instr_report.py "$OUT" | tee "$REPO/gpurun_out/$TAG/stage_instr.txt"
cp "$OUT/order.txt" "$REPO/gpurun_out/$TAG/stage_instr_order.txt"; rm -rf "$OUT"
