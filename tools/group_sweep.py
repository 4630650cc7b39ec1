#!/usr/bin/env python
"""Throughput of N envs split into G independent groups, one HIP stream each (no join between groups)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
from crowdnav import Config
from crowdnav.env import VecEnv, concurrent_streams

def run(N, G, steps=300, mode=os.environ.get("CN_MODE", "next")):
    n = N // G
    envs = [VecEnv(Config(n_envs=n, env_index_base=i * n, ped_cycle_ms=1400)) for i in range(G)]
    acts = [torch.rand((n, 2), device="cuda") * 0.2 for _ in range(G)]
    found = 1
    if G > 1:
        streams, found = concurrent_streams(G)
    else:
        streams = [torch.cuda.current_stream()]
    for e in envs: e.reset()
    torch.cuda.synchronize()
    def loop(k):
        for _ in range(k):
            for e, a, s in zip(envs, acts, streams):
                with torch.cuda.stream(s): e.step(a, auto_reset=mode)
    loop(30); torch.cuda.synchronize(); t0 = time.perf_counter(); loop(steps); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for e in envs: e.close() if hasattr(e, "close") else None
    return N * steps / dt / 1e6, dt / steps * 1e3, found

print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
Ns = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else (4096, 8192, 16384)
Gs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 4, 8)
for N in Ns:
    for G in Gs:
        if N % G: continue
        r, ms, found = run(N, G)
        print("N=%5d G=%d  %.1f M env-steps/s  (%.4f ms per step of all groups; %d concurrent streams)" % (N, G, r, ms, found), flush=True)
