#!/bin/bash
# round 5, GPU session 5: rocprofv3 passes (profile.sh), parity sweep
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
tools/profile.sh r05 > gpurun_out/profile_r05.log 2>&1; tail -40 gpurun_out/profile_r05.log
tools/parity_sweep.sh r05 > /dev/null 2>&1; tail -45 gpurun_out/r05/parity_sweep.txt
