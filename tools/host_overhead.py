#!/usr/bin/env python
"""Host-side cost of one VecEnv.step call (enqueue only, no synchronise inside the timed loop)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import Config
from crowdnav.env import VecEnv
env = VecEnv(Config(n_envs=64)); env.reset()
a = torch.zeros((64, 2), device="cuda"); a[:, 0] = 0.1
for _ in range(50): env.step(a, auto_reset="next")
torch.cuda.synchronize()
n = 300   # stays within the HIP queue depth so the host never blocks on the GPU
t0 = time.perf_counter()
for _ in range(n): env.step(a, auto_reset="next")
t1 = time.perf_counter()
torch.cuda.synchronize()
print("VecEnv.step host enqueue time: %.1f us per call" % ((t1 - t0) / n * 1e6))
f = env.bind_step(a, auto_reset="next")
for _ in range(50): f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): f()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("VecEnv.bind_step callable:     %.1f us per call" % ((t1 - t0) / n * 1e6))
