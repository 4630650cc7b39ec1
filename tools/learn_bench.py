#!/usr/bin/env python
"""TD3 updates per second: the eager torch path, Agent.enable_graphs (one hipGraph launch per update) and cn_td3_update
(Agent.enable_fused_update: 7 + 5 hand-written launches, csrc/crowdnav_td3.hip; also captured into hipGraphs here)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav.td3 import Agent


def fill(ag, n=100000):
    ag.memory.add(torch.randn((n, 398), device="cuda"), torch.rand((n, 2), device="cuda"), torch.randn(n, device="cuda"),
                  torch.randn((n, 398), device="cuda"), torch.rand(n, device="cuda") < 0.05)


def timed(fn, K=400, warm=50):
    for i in range(warm): fn(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3


ONLY_FUSED = os.environ.get("CN_LEARN_MODES", "") == "fused"      # (tools/learn_profile.sh: the fused chain alone under rocprofv3)
for B in [int(x) for x in os.environ.get("CN_BATCHES", "128,1024").split(",")]:
    row = []
    if ONLY_FUSED:
        ag = Agent(obs_dim=398, device="cuda", seed=0, batch_size=B, memory_size=200000)
        fill(ag); ag.enable_fused_update()
        print("batch %5d: fused %.3f ms per update" % (B, timed(ag.learn)), flush=True)
        continue
    for mode in ("eager", "graphs", "fused"):
        ag = Agent(obs_dim=398, device="cuda", seed=0, batch_size=B, memory_size=200000)
        fill(ag)
        if mode == "graphs": ag.enable_graphs()
        if mode == "fused": ag.enable_fused_update()
        row.append(timed(ag.learn))
    # the fused update replayed as hipGraphs (the chain is enqueue-only, so it captures like any other kernel sequence)
    ag = Agent(obs_dim=398, device="cuda", seed=0, batch_size=B, memory_size=200000)
    fill(ag); ag.enable_fused_update()
    for i in range(4): ag.learn(i)
    torch.cuda.synchronize()
    gs = {}
    side = torch.cuda.Stream()
    for do_actor in (0, 1):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            ag.learn(do_actor)            # step 0: actor too; step 1: critics only
        gs[do_actor] = g
    row.append(timed(lambda i: gs[i & 1].replay()))
    print("batch %5d: eager %.3f ms per update, graphed %.3f ms, fused %.3f ms (%.0f updates/s), fused + hipGraph %.3f ms (%.0f updates/s)" % (
        B, row[0], row[1], row[2], 1e3 / row[2], row[3], 1e3 / row[3]), flush=True)
