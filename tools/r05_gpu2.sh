#!/bin/bash
# round 5, GPU session 2: the new one-launch paths + bench-shape sequence test + bench launcher test, the refined issue-cost table
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "one_launch_paths or bench_shape or bench_under or policy_rollout_kernel or step_sequence_kernel or collect_policy or fused_td3 or replay" > $O/gputests2.txt 2>&1; echo "pytest rc $?" >> $O/gputests2.txt
tail -15 $O/gputests2.txt
timeout 300 tools/micro/bin/issue_cost > $O/issue_cost.txt 2>&1; echo "micro rc $?"
cat $O/issue_cost.txt
