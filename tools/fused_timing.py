#!/usr/bin/env python
"""Where a step of the fused rollout kernel (cn_rollout) goes: s_memtime stamps of the profiling build around the actor phase, the
barriers and the stages of the environment phase, last step of a K-step launch.  Usage: python tools/fused_timing.py [N] [K]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import numpy as np
import torch
from crowdnav import _abi
_abi.LIB_PATH = _abi.LIB_PATH.replace("libcrowdnav.so", "libcrowdnav_timing.so")
_abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv
from crowdnav.td3 import Agent

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
env = VecEnv(Config(n_envs=N, ped_cycle_ms=1400, max_steps=100000)); env.reset()
agent = Agent(obs_dim=env.D, device="cuda", seed=0, memory_size=16)
tb = torch.zeros((N, 32), dtype=torch.int64, device="cuda")
env.L.cn_debug_set_timing(env.h, C.c_void_p(tb.data_ptr()))
rows = []
for rep in range(8):
    env.rollout_fused(agent, K); torch.cuda.synchronize()
    t = tb.cpu().numpy().astype(np.float64)
    if rep >= 2:
        rows.append(np.stack([t[:, 26] - t[:, 25], t[:, 27] - t[:, 26], t[:, 0] - t[:, 27], t[:, 19] - t[:, 0], t[:, 28] - t[:, 19],
                              t[:, 28] - t[:, 25]], 1))
r = np.concatenate(rows)
names = ["actor tile (16 waves)", "barrier after the actor", "env entry (LDS carve, tables)", "env step (stamps 0..19)", "barrier after the env step (waiting for the tile's slowest env)", "whole step"]
print("fused rollout, N = %d, K = %d: mean / p50 / p95 s_memtime ticks per wave" % (N, K))
for k, nm in enumerate(names):
    print("  %-68s %9.0f %9.0f %9.0f" % (nm, r[:, k].mean(), np.median(r[:, k]), np.percentile(r[:, k], 95)))
