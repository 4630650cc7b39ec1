#!/bin/bash
# round 6: refresh everything under profiles/r06 that depends on the kernel sources (rocprofv3 passes, bench JSONs, quick perf)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r06; export TMPDIR=/tmp
python tools/csrc_hash.py
tools/profile.sh r06 > gpurun_out/profile_r06.log 2>&1; grep -E "^== (counters|traffic)" gpurun_out/prof_r06/summary.txt
tools/regen_profiles.sh r06 > gpurun_out/r06/regen.log 2>&1; tail -8 gpurun_out/r06/regen.log
python tools/learn_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/learn_bench.txt; cat gpurun_out/r06/learn_bench.txt
tools/learn_profile.sh r06 > /dev/null 2>&1
