#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
timeout 300 tools/micro/bin/issue_cost > $O/issue_cost.txt 2>&1; tail -6 $O/issue_cost.txt
timeout 2400 tools/regen_profiles.sh r05 > $O/regen.log 2>&1; tail -12 $O/regen.log
