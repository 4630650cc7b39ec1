#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp; F=amdgpu.ids
timeout 600 python -m pytest tests -m gpu -x -q -k "bench_under" 2>&1 | tail -3
python tools/stage_timing.py 256 2>&1 | grep -v $F > $O/stage_timing.txt; head -24 $O/stage_timing.txt
{ CN_ARB=oldest_first python tools/wave_tail.py; CN_ARB=fair python tools/wave_tail.py; } 2>&1 | grep -v $F > "$O/wave_tail.txt"
python tools/tail_corr.py 2>&1 | grep -v $F > "$O/tail_corr.txt"
python tools/actor_timing.py 2>&1 | grep -v $F > "$O/actor_timing.txt"
CN_ARB=oldest_first python tools/wave_fairness.py 2>&1 | grep -v $F > "$O/wave_fairness_oldest_first.txt"
CN_ARB=fair python tools/wave_fairness.py 2>&1 | grep -v $F > "$O/wave_fairness_fair.txt"
python bench.py 2>/dev/null | tail -1 > "$O/bench_n1.json"
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > "$O/bench_n1_driver_command.json"
python tools/show_bench.py $O/bench_n1.json $O/bench_n1_driver_command.json | grep -E "value|sustained"
