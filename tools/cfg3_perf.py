#!/usr/bin/env python
"""BASELINE configs[2] (TD3 actor in the loop) decompositions side by side: one act -> step chain and rollout_groups with 2 / 4
stream groups.  Resets subtracted (next-step reset)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import _abi
if os.environ.get("CN_LIB"):
    _abi.LIB_PATH = os.path.abspath(os.environ["CN_LIB"]); _abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnvGroups
from crowdnav.rollout import rollout_groups
from crowdnav.td3 import Agent
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
cfg = Config(n_envs=N, ped_cycle_ms=1400)

def marker(envs):
    tot = 0
    for e in envs:
        c = e.counters(); torch.cuda.synchronize()
        tot += int((c[:, 8] - c[:, 9]).sum().item())
    return tot

for G in (1, 2, 4):
    g = VecEnvGroups(cfg, groups=G, arbitration=os.environ.get("CN_GROUP_ARB") or None); g.reset()
    agent = Agent(obs_dim=g.D, device="cuda", seed=0, memory_size=16)
    rollout_groups(g, agent, 200); torch.cuda.synchronize()
    for rep in range(2):
        m0 = marker(g.envs); torch.cuda.synchronize(); t0 = time.perf_counter()
        rollout_groups(g, agent, K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        m1 = marker(g.envs)
    print("rollout_groups G=%d: %.4f ms/step %.2f M env-steps/s" % (G, dt / K * 1e3, (N * K - (m1 - m0)) / dt / 1e6))
    g.close()
