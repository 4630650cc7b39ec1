#!/usr/bin/env python
"""Which torch pool streams run concurrently with each other on this box?  (HIP maps streams onto a few
hardware queues; two streams on one queue serialise.)  Prints the pairwise overlap matrix."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
from crowdnav import Config
from crowdnav.env import VecEnv
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12
streams = [torch.cuda.Stream() for _ in range(M)]
print("stream handles:", [hex(s.cuda_stream) for s in streams])
n = 256
envs = [VecEnv(Config(n_envs=n, env_index_base=i * n, ped_cycle_ms=1400), stream=streams[i]) for i in range(M)]
act = torch.rand((n, 2), device="cuda") * 0.2
for e in envs: e.reset()
torch.cuda.synchronize()
def chain(idx, k=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        for i in idx: envs[i].step(act, auto_reset="next")
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6
chain(range(M))
one = [min(chain([i]) for _ in range(3)) for i in range(M)]
print("one chain: %s us/step" % " ".join("%.0f" % x for x in one))
print("pair overlap (1 = concurrent, . = serialised):")
for i in range(M):
    row = ""
    for j in range(M):
        if i == j: row += " -"; continue
        t = min(chain([i, j]) for _ in range(2))
        row += " 1" if t < 1.5 * max(one[i], one[j]) else " ."
    print("%2d %s" % (i, row))
