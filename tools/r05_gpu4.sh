#!/bin/bash
# round 5, GPU session 4: full parity suite on the current tree, quick perf (s720 with 3 waves per EU + fences), driver's bench command
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/quick_perf.py s720_w3_fence > $O/quick_perf_s720_w3.txt 2>&1; cat $O/quick_perf_s720_w3.txt
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1; echo "pytest rc $?" >> $O/gputests.txt
tail -6 $O/gputests.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc $?"
python tools/show_bench.py $O/bench_driver_cmd.json | head -40
tail -3 $O/bench_driver_cmd.err
