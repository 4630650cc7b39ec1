#!/usr/bin/env python
"""The workload tools/pc_sampling.sh profiles: N envs, one cn_step launch per control period (CN_ARB selects the kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import _abi
if os.environ.get("CN_LIB"):
    _abi.LIB_PATH = os.path.abspath(os.environ["CN_LIB"]); _abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
env = VecEnv(Config(n_envs=N, ped_cycle_ms=1400), arbitration=os.environ.get("CN_ARB", "auto")); env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
for i in range(steps):
    env.step(acts[i % 16], auto_reset="next")
torch.cuda.synchronize()
print("done", N, steps, env.arbitration)
