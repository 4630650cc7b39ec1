#!/bin/bash
# round 5, GPU session 3: the compact 720-ray layout (parity + speed), refined issue-cost table
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "one_launch_paths or config5 or 720 or dense or snapshot" > $O/gputests3.txt 2>&1; echo "pytest rc $?" >> $O/gputests3.txt
tail -15 $O/gputests3.txt
timeout 300 python tools/quick_perf.py compact720 > $O/quick_perf_compact720.txt 2>&1; cat $O/quick_perf_compact720.txt
CN_NO_SHAPE_KERNELS=1 timeout 300 python tools/quick_perf.py generic_kernels > $O/quick_perf_generic.txt 2>&1; cat $O/quick_perf_generic.txt
timeout 300 tools/micro/bin/issue_cost > $O/issue_cost.txt 2>&1; echo "micro rc $?"
grep -E "cndmask|mix_|fma64 indep|s_add32 indep|addu32 indep" $O/issue_cost.txt
