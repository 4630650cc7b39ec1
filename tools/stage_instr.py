#!/usr/bin/env python
"""Dynamic instruction ledger of the step kernel, stage by stage (profiling build libcrowdnav_timing.so).

Run under rocprofv3 with PMC counters (tools/stage_instr.sh): from one snapshot of 4096 mid-episode environments the same step is
launched once per stage stamp with every wavefront ENDING at that stamp (cn_debug_set_ablate bits 8..15 -> s_endpgm at CN_T(k)),
then once in full.  The counters of the launch cut at stamp k are the instructions issued up to k; consecutive differences are the
stages' own dynamic counts.  This script only issues the launches and prints their order; tools/stage_instr_report.py reads the
rocprofv3 CSVs.  (Results of a cut launch are meaningless and discarded: every launch starts from the restored snapshot.)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import _abi
_abi.LIB_PATH = _abi.LIB_PATH.replace("libcrowdnav.so", "libcrowdnav_timing.so")
_abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv

# stamps in execution order (env_kernel_body / observe): (stamp, what ends there)
ORDER = [(0, "kernel entry, parameter loads, LDS map"), (1, "load state"), (20, "physics: trig, pedestrians, robot (dt)"),
         (21, "deque, robot (scan latency)"), (22, "sync"), (23, "distance, heading (atan2), way-point refresh at step 1"),
         (24, "way-point refresh"), (2, "twist features"), (3, "lidar set-up + near-pedestrian list"), (4, "ray loop (cast, end points, obs)"),
         (5, "scan min, bbox / deque at step 0"), (6, "gradients"), (7, "flag words"), (9, "type machine + aliasing"),
         (10, "association (IoU)"), (11, "order / split words"), (12, "word bases"), (13, "confirmation + counters"),
         (14, "tracker"), (15, "speeds, defaults"), (16, "collision cone + top-K"), (17, "counters, done, tail"),
         (18, "tracker write-back, reward, outputs"), (19, "state write-back")]
N = int(os.environ.get("CN_ENVS", 4096))
env = VecEnv(Config(n_envs=N, ped_cycle_ms=1400, max_steps=100000)); env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
for i in range(300):
    env.step(acts[i % 16], auto_reset="next")
torch.cuda.synchronize()
snap = env.snapshot()
print("kernel", env.kernel_name("step"), "envs", N)
for k, what in ORDER + [(254, "FULL")]:
    env.restore(snap)
    env.L.cn_debug_set_ablate(env.h, (k + 1) << 8)
    env.step(acts[5], auto_reset="next")
    torch.cuda.synchronize()
    print("cut %3d %s" % (k, what))
