#!/bin/bash
# Round-4 training runs on the GPU box: TD3 with the reference's hyper-parameters (TRAIN:62-72) in
# presets.training(drop_cospawned=True) -- tools/train_r04.sh <set> <seconds per run>  -> gpurun_out/train_<set>/<run>.txt (+ CSV)
SET="${1:-probe}"; LIM="${2:-420}"
cd "$(dirname "$0")/.."; OUT="gpurun_out/train_$SET"; mkdir -p "$OUT"
export PYTHONPATH="$PWD/drl-based-mapless-crowd-navigation-with-perceived-risk_amd:${PYTHONPATH:-}"
run() { name="$1"; shift; CSV="--csv"; [ "${NOCSV:-0}" = 1 ] && CSV=""
  python -m crowdnav.train --scenario training_as_logged $CSV --log-every 250 --launches 100000000 --time-limit "$LIM" --out "$OUT/$name" "$@" 2>&1 | grep -v amdgpu.ids > "$OUT/$name.txt"
  echo "== $name: $*"; tail -3 "$OUT/$name.txt"; rm -f "$OUT/$name"/*.pt; }
case "$SET" in
  probe)   # the published log's reward (no way-point bonus) at the reference's update-to-data ratio of 1, and the committed reward beside it
    run e16_u16_wp0   --envs 16 --updates 16 --waypoint-reward 0
    run e16_u16_wp200 --envs 16 --updates 16 --waypoint-reward 200
    run e64_u16_b512_wp0 --envs 64 --updates 16 --batch 512 --waypoint-reward 0 ;;
  same)    # the reset convention isolated -- same seed, scenario, env count, both conventions
    run next_e16 --envs 16 --updates 16 --waypoint-reward 0 --reset-mode next
    run same_e16 --envs 16 --updates 16 --waypoint-reward 0 --reset-mode same ;;
  long)
    run e16_u16_wp0 --envs 16 --updates 16 --waypoint-reward 0 ;;
  seeds)   # robustness: the 16-env recipe under other seeds, and the env count in between (fused learner)
    for S in 1 2 3; do run fused_e16_u16_wp0_seed$S --envs 16 --updates 16 --waypoint-reward 0 --learner fused --seed $S; done
    run fused_e32_u32_wp0 --envs 32 --updates 32 --waypoint-reward 0 --learner fused
    for S in 1 2; do run fused_e64_u64_wp0_seed$S --envs 64 --updates 64 --waypoint-reward 0 --learner fused --seed $S; done ;;
  resetab) # the reset convention isolated -- same seed, scenario, env count, learner; only --reset-mode differs
    run fused_next_e16 --envs 16 --updates 16 --waypoint-reward 0 --learner fused --reset-mode next
    run fused_same_e16 --envs 16 --updates 16 --waypoint-reward 0 --learner fused --reset-mode same ;;
  init)    # NOT the reference: the actor's output layer initialised U(+-0.003); does the seed sensitivity go away?
    for S in 0 1 2 3 4 5 6 7; do run init003_fused_e16_seed$S --envs 16 --updates 16 --waypoint-reward 0 --learner fused --seed $S --actor-final-init 0.003; done ;;
  r05)     # round 5: the 16-env recipe on cn_td3_update at 0.068 ms (other summation order than round 4: other trajectories)
    for S in 0 1; do run fused_e16_u16_wp0_seed$S --envs 16 --updates 16 --waypoint-reward 0 --learner fused --seed $S; done ;;
  r05long) # round 5: the same recipe, seed 0, for as long as the second argument says (900 s: where does it level off?)
    run fused_e16_u16_wp0_seed0_long --envs 16 --updates 16 --waypoint-reward 0 --learner fused --seed 0 ;;
  final)   # the 16-env recipe on the final tree (cn_td3_update at 0.127 ms)
    run final_fused_e16_u16_wp0 --envs 16 --updates 16 --waypoint-reward 0 --learner fused ;;
  fused)   # the same runs on cn_td3_update (csrc/crowdnav_td3.hip): the reference's ratio of one update per env-step, and 4x the envs
    run fused_e16_u16_wp0 --envs 16 --updates 16 --waypoint-reward 0 --learner fused
    run fused_e64_u64_wp0 --envs 64 --updates 64 --waypoint-reward 0 --learner fused ;;
esac
