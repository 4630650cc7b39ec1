#!/bin/bash
# rocprofv3 recipe (run on the GPU box via gpurun): kernel-trace stats + separate PMC passes.
#   tools/profile.sh <tag> [bench args...]
set -u
TAG="${1:-r05}"; shift || true
cd "$(dirname "$0")/.."
REPO="$PWD"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
BENCH="python $REPO/bench.py --steps 300 --warmup 30 --repeats 3 --no-cpu-baseline --no-plateau --no-other-configs --sustained-seconds 0 --no-sequence-traj $*"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/bench_trace.log" 2>&1
# the driver's own command (20 timed steps after 5 warm-up steps and the fixed 200-step pre-roll)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_driver" -o trace -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-plateau --no-other-configs --sustained-seconds 0 --no-sequence-traj > "$OUT/bench_trace_driver.log" 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o pmc -- $BENCH > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o pmc -- $BENCH > "$OUT/bench_pmc_write.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY \
    -d "$OUT/pmc_sq" -o pmc -- $BENCH > "$OUT/bench_pmc_sq.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    -d "$OUT/pmc_sq2" -o pmc -- $BENCH > "$OUT/bench_pmc_sq2.log" 2>&1
# cn_rollout_policy (BASELINE configs[2], the actor inside the step kernel): launches of 100 periods each
POL="python $REPO/tools/policy_perf.py 4096 --profile"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pol_trace" -o trace -- $POL > "$OUT/policy_trace.log" 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pol_fetch" -o pmc -- $POL > "$OUT/policy_pmc_fetch.log" 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pol_write" -o pmc -- $POL > "$OUT/policy_pmc_write.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES \
    -d "$OUT/pol_sq" -o pmc -- $POL > "$OUT/policy_pmc_sq.log" 2>&1
# the sequence kernel writing into trajectory buffers (bench.py's sequence_traj leg): launches of 50 steps, HBM traffic only
TRJ="python $REPO/tools/seq_traj_profile.py"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/traj_fetch" -o pmc -- $TRJ > "$OUT/traj_pmc_fetch.log" 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/traj_write" -o pmc -- $TRJ > "$OUT/traj_pmc_write.log" 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/calib_fetch" -o pmc -- python $REPO/tools/calib_pmc.py > "$OUT/calib_fetch.log" 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/calib_write" -o pmc -- python $REPO/tools/calib_pmc.py > "$OUT/calib_write.log" 2>&1
cd "$REPO"
CN_PROFILE_SEQ_STEPS=300 CN_PROFILE_POL_STEPS=100 CN_PROFILE_TRAJ_STEPS=50 python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# gpurun merges at most 64 MiB back: keep the summaries and the kernel statistics, drop the raw traces / counter dumps
mkdir -p "$OUT/keep"
for d in trace trace_driver pol_trace; do f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/keep/kernel_stats_$d.csv"; done
find "$OUT" -mindepth 1 -maxdepth 1 -type d ! -name keep -exec rm -rf {} +
