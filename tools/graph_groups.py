#!/usr/bin/env python
"""Does one HIP graph holding K steps of every stream group (G independent chains, forked from and joined to the capturing
stream) beat eager enqueueing?  Prints env-steps/s for eager and for graphs of K = 1, 5, 20 steps, over a 20-step sample and
over 400 steps.  Usage: python tools/graph_groups.py [N] [G]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import Config
from crowdnav.env import VecEnvGroups
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_act = 20
g_ = torch.Generator(device="cuda").manual_seed(1)
acts = torch.stack([torch.rand((n_act, N), generator=g_, device="cuda") * 0.22, torch.rand((n_act, N), generator=g_, device="cuda") * 4 - 2], 2).contiguous()


def make():
    grp = VecEnvGroups(Config(n_envs=N, ped_cycle_ms=1400), groups=G); grp.reset()
    rows = [grp.rows(g) for g in range(G)]
    calls = [[grp.envs[g].bind_step(acts[i][rows[g]], auto_reset="next") for g in range(G)] for i in range(n_act)]
    for i in range(200):
        for c in calls[i % n_act]: c()
    torch.cuda.synchronize()
    return grp, calls


def eager(steps):
    grp, calls = make()
    t0 = time.perf_counter()
    for i in range(steps):
        for c in calls[i % n_act]: c()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    grp.close(); return N * steps / dt / 1e6


def graphed(steps, K):
    grp, calls = make()
    cap = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        graph.capture_begin()
        for s in grp.streams: s.wait_stream(cap)
        for i in range(K):                       # steps i = 0..K-1 of the action table, chain per group
            for c in calls[i % n_act]: c()
        for s in grp.streams: cap.wait_stream(s)
        graph.capture_end()
    torch.cuda.synchronize()
    graph.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps // K): graph.replay()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    grp.close(); return N * (steps // K) * K / dt / 1e6


for steps in (20, 400):
    print("steps %4d: eager %.1f M" % (steps, eager(steps)), end="")
    for K in (1, 5, 20):
        try:
            print(" | graph K=%d %.1f M" % (K, graphed(steps, K)), end="")
        except Exception as ex:
            print(" | graph K=%d failed: %s" % (K, str(ex)[:80]), end="")
    print(flush=True)
