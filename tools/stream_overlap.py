#!/usr/bin/env python
"""Do kernels on two HIP streams of one process overlap on this box?  (torch.cuda._sleep spins one thread.)"""
import os, sys, time
import torch
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
cyc = 20_000_000
torch.cuda._sleep(1000); torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(cyc); torch.cuda.synchronize(); one = time.perf_counter() - t0
t0 = time.perf_counter()
with torch.cuda.stream(s1): torch.cuda._sleep(cyc)
with torch.cuda.stream(s2): torch.cuda._sleep(cyc)
torch.cuda.synchronize(); two = time.perf_counter() - t0
print("one sleep kernel %.2f ms; two on two streams %.2f ms  -> %s" % (one * 1e3, two * 1e3, "overlap" if two < 1.5 * one else "SERIALISED"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
from crowdnav import Config
from crowdnav.env import VecEnv
envs = [VecEnv(Config(n_envs=2048, env_index_base=i * 2048, ped_cycle_ms=1400)) for i in (0, 1)]
acts = [torch.rand((2048, 2), device="cuda") * 0.2 for _ in (0, 1)]
for e in envs: e.reset()
torch.cuda.synchronize()
def run(streams, n=200):
    for _ in range(20):
        for e, a, s in zip(envs, acts, streams):
            with torch.cuda.stream(s): e.step(a, auto_reset="next")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for e, a, s in zip(envs, acts, streams):
            with torch.cuda.stream(s): e.step(a, auto_reset="next")
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
d = torch.cuda.current_stream()
print("2 x 2048 envs, same stream: %.4f ms/step; two streams: %.4f ms/step" % (run([d, d]), run([s1, s2])))
