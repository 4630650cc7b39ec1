#!/usr/bin/env python
"""Time the one-kernel f32-MFMA actor (cn_actor_forward) against the PyTorch actor at N = 4096."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav.td3 import Agent
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
agent = Agent(obs_dim=398, device="cuda", seed=0, memory_size=16)
obs = torch.randn((N, 398), device="cuda")
out = torch.empty((N, 2), device="cuda")
def t(fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    print("N=%d  torch actor + noise + clip: %.1f us   fused tail: %.1f us   one-kernel MFMA actor: %.1f us" % (
        N, t(lambda: agent.act(obs)), t(lambda: agent.act_fused(obs, out)), t(lambda: agent.act_mfma(obs, out))))
