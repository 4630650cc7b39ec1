#!/bin/bash
# PC sampling of the headline step kernel (rocprofv3 --pc-sampling-beta-enabled): where a wavefront's issue slots go, per instruction.
# The sampled library is the product's code built with -gline-tables-only (lib/ab/libcrowdnav_lines.so: same instructions, the
# .loc chains give every sample its pipeline stage).  Every attempt runs under its own timeout -- the feature is a beta.
#   tools/pc_sampling.sh [method=host_trap|stochastic] [interval] [unit] [envs] [steps]
cd "$(dirname "$0")/.."; ROOT="$PWD"; OUT=$ROOT/gpurun_out/pcs; mkdir -p $OUT; export TMPDIR=/tmp
METHOD="${1:-host_trap}"; INTERVAL="${2:-100}"; UNIT="${3:-time}"; ENVS="${4:-4096}"; STEPS="${5:-600}"
LIB=$ROOT/drl-based-mapless-crowd-navigation-with-perceived-risk_amd/lib/ab/libcrowdnav_lines.so
[ -f "$LIB" ] || LIB=$ROOT/drl-based-mapless-crowd-navigation-with-perceived-risk_amd/lib/libcrowdnav.so
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
(rocprofv3-avail info --pc-sampling 2>&1 || rocprofv3-avail list --pc-sampling 2>&1) | grep -v amdgpu.ids | tee $OUT/avail.txt | tail -30
D=/tmp/pcs_$METHOD; rm -rf $D
cd /tmp
CN_LIB=$LIB CN_ARB="${CN_ARB:-auto}" timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD --pc-sampling-unit $UNIT \
    --pc-sampling-interval $INTERVAL --kernel-trace --output-format csv -d $D -o pcs -- python $ROOT/tools/pc_workload.py $ENVS $STEPS > $OUT/run_$METHOD.log 2>&1
echo "rocprofv3 exit code $?" | tee -a $OUT/run_$METHOD.log
tail -5 $OUT/run_$METHOD.log
find $D -type f | head -20
F=$(find $D -name "*pc_sampling*csv" | head -1)
if [ -n "$F" ]; then
  ls -la $F; head -3 $F
  PCS_OUT=$OUT python3 $ROOT/tools/pc_sampling_report.py $F $(find $D -name "*kernel_trace.csv" | head -1) > $OUT/report_$METHOD.txt 2>&1
  head -60 $OUT/report_$METHOD.txt
  mv -f $OUT/hist.tsv $OUT/hist_$METHOD.tsv; gzip -f $OUT/hist_$METHOD.tsv; ls -la $OUT
  # keep a compact per-instruction histogram (the raw CSV is hundreds of MB)
fi
