#!/bin/bash
# round 5, GPU session 1: parity suite, the issue-cost table, the driver's bench command
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "pytest rc $?" >> $O/gputests.txt
tail -5 $O/gputests.txt
timeout 300 tools/micro/bin/issue_cost > $O/issue_cost.txt 2>&1; echo "micro rc $?"
cat $O/issue_cost.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc $?"
python tools/show_bench.py $O/bench_driver_cmd.json 2>/dev/null | head -40 || head -c 3000 $O/bench_driver_cmd.json
tail -5 $O/bench_driver_cmd.err
