#!/usr/bin/env python
"""Per-stage cycle breakdown of one wavefront of cn_env_kernel (profiling build libcrowdnav_timing.so,
s_memtime stamps at stage boundaries).  Usage: python tools/stage_timing.py [N]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import numpy as np
import torch
import crowdnav
from crowdnav import _abi
_abi.LIB_PATH = _abi.LIB_PATH.replace("libcrowdnav.so", "libcrowdnav_timing.so")
_abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv

NAMES = ["", "load state", "physics, deque, waypoint/heading/dist", "near-ped list", "ray loop (cast, end points, obs)",
         "bbox (reset only)", "gradients", "flag words", "type machine", "aliasing", "association (IoU)", "order/split words",
         "word bases", "confirmation + counters", "tracker", "speeds/defaults", "collision cone + top-K", "counters/done/tail",
         "reward + outputs", "state write-back"]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = VecEnv(Config(n_envs=N, ped_cycle_ms=1400, max_steps=100000)); env.reset()
tb = torch.zeros((N, 32), dtype=torch.int64, device="cuda")
env.L.cn_debug_set_timing(env.h, C.c_void_p(tb.data_ptr()))
if os.environ.get("CN_ABLATE"):      # e.g. CN_ABLATE=1: no pedestrians in the ray cast (results invalid, timing only)
    env.L.cn_debug_set_ablate(env.h, int(os.environ["CN_ABLATE"]))
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
acc = np.zeros(20); cnt = 0; tot = []
for i in range(60):
    tb.zero_()
    env.step(acts[i % 16], auto_reset="next")
    torch.cuda.synchronize()
    t = tb.cpu().numpy().astype(np.float64)
    ok = (t[:, 19] > 0) & (t[:, 2] > 0) & (env.done.cpu().numpy() == 0)
    if i < 10 or not ok.any():
        continue
    d = np.diff(t[ok, :20], axis=1)
    acc[1:] += d.mean(0); cnt += 1
    tot.append((t[ok, 19] - t[ok, 0]).mean())
acc /= max(cnt, 1)
print("N = %d, mean s_memtime ticks per stage over %d launches (one wavefront = one env); total %.0f" % (N, cnt, np.mean(tot)))
for k in range(1, 20):
    print("  %-36s %9.0f  %5.1f%%" % (NAMES[k], acc[k], 100 * acc[k] / acc[1:].sum()))
