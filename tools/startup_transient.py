#!/usr/bin/env python
"""How the stream-group leg gets up to speed after a device-wide synchronize (the driver's 20-step sample starts there):
completion time of every step of every group (HIP events on the group streams), 4096 envs as 4 groups of 1024.
    python tools/startup_transient.py [steps] [repeats]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import Config
from crowdnav.env import VecEnvGroups
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 24
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 5
G, N = 4, 4096
dev = torch.device("cuda", 0)
cfg = Config(n_envs=N, max_steps=1000, seed=1234, ped_cycle_ms=1400)
grp = VecEnvGroups(cfg, groups=G, device=0); grp.reset()
g_ = torch.Generator(device=dev).manual_seed(1234)
acts = torch.stack([torch.rand((64, N), generator=g_, device=dev) * 0.22, torch.rand((64, N), generator=g_, device=dev) * 4 - 2], 2).contiguous()
rows = [grp.rows(g) for g in range(G)]
calls = [[grp.envs[g].bind_step(acts[i][rows[g]], auto_reset="next") for g in range(G)] for i in range(64)]
for i in range(200):
    for c in calls[i % 64]: c()
for rep in range(REP):
    torch.cuda.synchronize(dev)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(STEPS + 1)] for _ in range(G)]
    t0 = time.perf_counter()
    for g in range(G): ev[g][0].record(grp.streams[g])
    for i in range(STEPS):
        for g, c in enumerate(calls[i % 64]):
            c(); ev[g][i + 1].record(grp.streams[g])
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    # per-step duration per group, and the completion time of step i = max over groups relative to the earliest start
    dur = [[ev[g][i].elapsed_time(ev[g][i + 1]) * 1e3 for i in range(STEPS)] for g in range(G)]
    done = [max(ev[0][0].elapsed_time(ev[g][i + 1]) for g in range(G)) * 1e3 for i in range(STEPS)]
    print("rep %d: host enqueue %.0f us, wall %.0f us -> %.1f M env-steps/s (events add launches of their own)" % (rep, t_host * 1e6, wall * 1e6, N * STEPS / wall / 1e6))
    print("   step completion (us): " + " ".join("%.0f" % d for d in done))
    print("   step-to-step (us):    " + " ".join("%.0f" % (done[i] - (done[i - 1] if i else 0)) for i in range(STEPS)))
    print("   group 0 durations:    " + " ".join("%.0f" % d for d in dur[0]))
