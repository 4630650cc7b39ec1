#!/usr/bin/env python
"""What a K-step sample of bench.py pays beside K steps: wall(K) = a + b K for the stream-group legs at 4096 envs, K = 5 .. 160,
with bench.py's own bracket (poll the streams, then torch.cuda.synchronize) and with variants of how the end is detected:
    poll      hipStreamQuery on every group stream until done, then the device synchronize          (bench.py)
    sync      the device synchronize alone (sleeps on an interrupt)
    evq       hipEventQuery on an event recorded behind the last launch of every stream, then the device synchronize
and where the intercept goes: the time from the first enqueue to the first kernel's start and from the last kernel's end to the
host's clock reading, both from the device's own time stamps (cn_device_clock is not used: HIP event times on the streams).
    python tools/sample_fixed_cost.py [groups ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import numpy as np
import torch
from crowdnav import Config
from crowdnav.env import VecEnvGroups, concurrent_streams

GS = [int(x) for x in sys.argv[1:]] or [1, 2, 4]
KS = [5, 10, 20, 40, 80, 160]
REP = 9
N = 4096
dev = torch.device("cuda", 0)
streams, conc = concurrent_streams(4, 0)
g_ = torch.Generator(device=dev).manual_seed(1234)
acts = torch.stack([torch.rand((16, N), generator=g_, device=dev) * 0.22, torch.rand((16, N), generator=g_, device=dev) * 4 - 2], 2).contiguous()


def end_poll(strs, evs):
    for s_ in strs:
        while not s_.query():
            pass
    torch.cuda.synchronize(dev)


def end_sync(strs, evs):
    torch.cuda.synchronize(dev)


def end_evq(strs, evs):
    for e_ in evs:
        while not e_.query():
            pass
    torch.cuda.synchronize(dev)


for G in GS:
    cfg = Config(n_envs=N, max_steps=1000, seed=1234, ped_cycle_ms=1400)
    grp = VecEnvGroups(cfg, groups=G, device=0, streams=streams[:G]); grp.reset()
    warm = grp.bind_step_sequence([acts[i % 16] for i in range(100)]); warm(); torch.cuda.synchronize(dev)
    tail3 = grp.bind_step_sequence([acts[i % 16] for i in range(3)])
    calls = {K: grp.bind_step_sequence([acts[i % 16] for i in range(K)]) for K in KS}
    print("== %d group(s): kernel %s" % (G, grp.envs[0].kernel_name("multi" if G > 1 else "step")))
    for name, end in (("poll", end_poll), ("sync", end_sync), ("evq", end_evq)):
        med = []
        for K in KS:
            w = []
            for r in range(REP):
                tail3()
                for s_ in grp.streams:
                    while not s_.query():
                        pass
                torch.cuda.synchronize(dev)
                evs = [torch.cuda.Event() for _ in range(G)]
                t0 = time.perf_counter()
                calls[K]()
                if name == "evq":
                    for g in range(G):
                        evs[g].record(grp.streams[g])
                end(grp.streams, evs)
                w.append(time.perf_counter() - t0)
            med.append(float(np.median(w)) * 1e6)
        b, a = np.polyfit(np.array(KS[2:], dtype=float), np.array(med[2:]), 1)
        print("  %-5s wall us at K = %s: %s | fit over K >= 20: a = %.1f us, b = %.2f us/step (%.1f M steady); K = 20 -> %.1f M"
              % (name, KS, " ".join("%.0f" % m for m in med), a, b, N / b, N * 20 / med[2]))
    # where the intercept goes (K = 20, timing events: their own packets add a little)
    K = 20
    for r in range(3):
        tail3()
        for s_ in grp.streams:
            while not s_.query():
                pass
        torch.cuda.synchronize(dev)
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(G)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(G)]
        for g in range(G):
            e0[g].record(grp.streams[g])
        t0 = time.perf_counter()
        calls[K]()
        t_enq = time.perf_counter() - t0
        for g in range(G):
            e1[g].record(grp.streams[g])
        end_poll(grp.streams, None)
        wall = time.perf_counter() - t0
        span = max(e0[0].elapsed_time(e1[g]) for g in range(G)) * 1e3
        print("  rep %d: enqueue %.0f us, wall %.0f us, first event -> last event on the device %.0f us" % (r, t_enq * 1e6, wall * 1e6, span))
    grp.close()
