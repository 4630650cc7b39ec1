#!/usr/bin/env python
"""Profiling workload of bench.py's `sequence_traj` leg alone: cn_step_sequence launches of 50 steps at 4096 envs whose every step
writes its observation / reward / done flag to its own slot of the caller's trajectory buffers (tools/profile.sh traj_* passes:
HBM bytes per env-step of the sequence kernel when its outputs really leave the chip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import Config
from crowdnav.env import VecEnv
N, T, CALLS = 4096, int(os.environ.get("CN_PROFILE_TRAJ_STEPS", "50")), 8
env = VecEnv(Config(n_envs=N, ped_cycle_ms=1400, seed=1234, max_steps=1000)); env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.stack([torch.rand((T, N), generator=g, device="cuda") * 0.22, torch.rand((T, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
traj = dict(obs=torch.zeros((T, N, env.D), device="cuda"), reward=torch.zeros((T, N), device="cuda"), done=torch.zeros((T, N), dtype=torch.uint8, device="cuda"))
call = env.bind_step_sequence(acts, traj=traj)
for _ in range(CALLS):
    call()
torch.cuda.synchronize()
print("sequence_traj profile workload: %d launches of %s, %d steps x %d envs each" % (CALLS, env.kernel_name("sequence"), T, N))
