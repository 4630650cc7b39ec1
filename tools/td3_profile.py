#!/usr/bin/env python
"""200 fused TD3 updates (cn_td3_update, batch 128) for a rocprofv3 --kernel-trace --stats run: per-kernel time of the update chain."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav.td3 import Agent
B = int(os.environ.get("CN_BATCH", 128))
ag = Agent(obs_dim=398, device="cuda", seed=0, batch_size=B, memory_size=200000)
n = 100000
ag.memory.add(torch.randn((n, 398), device="cuda"), torch.rand((n, 2), device="cuda"), torch.randn(n, device="cuda"),
              torch.randn((n, 398), device="cuda"), torch.rand(n, device="cuda") < 0.05)
ag.enable_fused_update()
for i in range(200):
    ag.learn(i)
torch.cuda.synchronize()
print("done")
