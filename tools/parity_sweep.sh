#!/bin/bash
# The GPU-vs-oracle sweeps DESIGN.md quotes (every observation / reward / done flag / index / counter compared, zero
# differences expected).  Writes gpurun_out/<tag>/parity_sweep.txt; copy it to profiles/<tag>/.
#   tools/parity_sweep.sh r02 [scale]      scale multiplies every --steps (10 -> 36 M env-steps, ~10 min; output parity_sweep_x10.txt)
TAG="${1:-r06}"; S="${2:-1}"; cd "$(dirname "$0")/.."; OUT="gpurun_out/$TAG"; mkdir -p "$OUT"
R="python tools/parity_report.py --verbose 2"
NAME=parity_sweep; [ "$S" != 1 ] && NAME="parity_sweep_x$S"
{
$R --envs 4096 --steps $((100 * S)) --max-steps 60 --reset-mode next
$R --envs 2048 --steps $((400 * S)) --max-steps 120
$R --envs 1024 --steps $((300 * S)) --peds 60 --reset-mode next
$R --envs 512 --steps $((200 * S)) --peds 100 --max-steps 80
$R --envs 256 --steps $((150 * S)) --peds 100 --rays 720 --room 2.4
$R --envs 512 --steps $((300 * S)) --min-scan 0.0 --max-steps 500 --reset-mode next
$R --envs 1024 --steps $((300 * S)) --peds 14 --k 4
$R --envs 512 --steps $((200 * S)) --peds 200 --rays 1025 --room 2.4 --max-steps 80
$R --envs 512 --steps $((200 * S)) --peds 60 --geos 1
$R --envs 1024 --steps $((300 * S)) --risk-mode 1 --reset-mode next
$R --envs 512 --steps $((200 * S)) --peds 100 --risk-mode 1 --k 4
$R --envs 512 --steps $((200 * S)) --peds 60 --contact 1 --min-scan 0.0
$R --envs 512 --steps $((200 * S)) --peds 40 --contact 1 --risk-mode 1 --vmax 0.5
$R --envs 1024 --steps $((300 * S)) --layout 1
$R --envs 1024 --steps $((300 * S)) --layout 2 --dt-ms 50 --reset-mode next
$R --envs 256 --steps $((200 * S)) --layout 2 --dt-ms 50 --peds 100 --min-scan 0.0
# round 3: the reference's own platform (Python-2.7 round() + GEOS <= 3.8), alone and together; social-force pedestrians
$R --envs 2048 --steps $((200 * S)) --py2 1 --reset-mode next
$R --envs 1024 --steps $((200 * S)) --peds 60 --py2 1 --geos 1
$R --envs 512 --steps $((200 * S)) --layout 2 --dt-ms 50 --py2 1
$R --envs 1024 --steps $((200 * S)) --ped-mode 2 --reset-mode next
$R --envs 512 --steps $((150 * S)) --ped-mode 2 --peds 60 --risk-mode 1 --min-scan 0.0
$R --envs 1024 --steps $((200 * S)) --ped-mode 2 --sf-tick 50 --reset-mode next
# round 4: "as Gazebo delivers it" (float32 scans, the diff-drive plugin's wheel ramp), the published log's reward, the dense social-force kernels
$R --envs 2048 --steps $((200 * S)) --scan-f32 1 --wheel-accel 1.0 --waypoint-reward 0 --reset-mode next
$R --envs 1024 --steps $((200 * S)) --peds 60 --scan-f32 1 --wheel-accel 1.0 --risk-mode 1
$R --envs 1024 --steps $((200 * S)) --peds 6 --waypoint-reward 0 --max-steps 300 --reset-mode next
$R --envs 256 --steps $((100 * S)) --ped-mode 2 --peds 100 --rays 720 --room 2.4 --reset-mode next
# round 4: the policy inside the step kernel (cn_rollout_policy, closed loop; the oracle replays the recorded actions): headline shape, generic shape, gt
$R --envs 4096 --steps $((100 * S)) --max-steps 60 --reset-mode next --policy 20
$R --envs 1000 --steps $((200 * S)) --peds 14 --k 4 --rays 300 --reset-mode next --policy 25
$R --envs 1024 --steps $((200 * S)) --risk-mode 1 --reset-mode next --policy 40
# round 5: the compact 720-ray layout (3 waves per SIMD) under the next-step reset; cn_rollout_policy for the 720-ray shape (8 environments
# per workgroup), for social-force pedestrians and for the "as Gazebo delivers it" world (float32 scans + wheel ramp)
$R --envs 1024 --steps $((120 * S)) --peds 100 --rays 720 --room 2.4 --reset-mode next --max-steps 60
$R --envs 520 --steps $((100 * S)) --peds 100 --rays 720 --room 2.4 --reset-mode next --max-steps 60 --policy 10
$R --envs 1024 --steps $((200 * S)) --ped-mode 2 --reset-mode next --policy 20
$R --envs 1000 --steps $((200 * S)) --scan-f32 1 --wheel-accel 1.0 --waypoint-reward 0 --reset-mode next --policy 25
# round 6: cn_rollout_policy for the worlds round 5 still refused -- the contact ticks (both risk modes) and the two older observation layouts
$R --envs 520 --steps $((200 * S)) --peds 40 --contact 1 --reset-mode next --policy 20
$R --envs 520 --steps $((200 * S)) --peds 40 --contact 1 --risk-mode 1 --vmax 0.5 --reset-mode next --policy 25
$R --envs 1000 --steps $((200 * S)) --layout 1 --reset-mode next --policy 20
$R --envs 1000 --steps $((200 * S)) --layout 2 --dt-ms 50 --reset-mode next --policy 40
} 2>&1 | grep -v amdgpu.ids | tee "$OUT/$NAME.txt"
