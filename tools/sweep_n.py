#!/usr/bin/env python
"""ms/launch of cn_env_kernel as a function of the number of envs (occupancy / tail diagnosis)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import Config
from crowdnav.env import VecEnv
MODE = os.environ.get("CN_RESET_MODE", "same")
for N in [int(x) for x in (sys.argv[1:] or "256 512 1024 2048 3072 4096 6144 8192 16384".split())]:
    env = VecEnv(Config(n_envs=N, ped_cycle_ms=1400)); env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
    for i in range(50): env.step(acts[i % 16], auto_reset=MODE)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200): env.step(acts[i % 16], auto_reset=MODE)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    print("N %6d  %.4f ms/launch  %.2f M env-steps/s" % (N, ms, N / ms / 1e3))
    env.close()
