#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "social or sf or one_launch_paths or dense" > $O/gputests_sf.txt 2>&1; tail -5 $O/gputests_sf.txt
CN_PED_MODE=2 timeout 600 python tools/quick_perf.py social-force-once-per-pair 2>&1 | grep -v amdgpu.ids | tee $O/quick_perf_sf.txt
python tools/parity_report.py --verbose 2 --envs 256 --steps 100 --ped-mode 2 --peds 100 --rays 720 --room 2.4 --reset-mode next 2>&1 | tail -3
python tools/parity_report.py --verbose 2 --envs 512 --steps 150 --ped-mode 2 --peds 60 --risk-mode 1 --min-scan 0.0 2>&1 | tail -3
python tools/parity_report.py --verbose 2 --envs 512 --steps 100 --ped-mode 2 --peds 64 --room 1.0 --reset-mode next 2>&1 | tail -3
