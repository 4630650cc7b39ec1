#!/usr/bin/env python
"""Phases of one control period of cn_policy_kernel* (profiling build: s_memtime stamps of every workgroup's wave 0 in the LAST period
of a launch): lock wait, actor tile, Env.step of wave 0, wait for the workgroup's slowest wave; and how far apart the workgroups that
share a CU run.  Usage: [CN_POL_ENVS=8 CN_POL_LOCK=1] python tools/policy_phase_timing.py [N] [T]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import numpy as np
import torch
from crowdnav import _abi
_abi.LIB_PATH = _abi.LIB_PATH.replace("libcrowdnav.so", "libcrowdnav_timing.so")
_abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv
from crowdnav.td3 import Agent

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
env = VecEnv(Config(n_envs=N, ped_cycle_ms=1400, max_steps=100000)); env.reset()
agent = Agent(obs_dim=env.cfg.obs_dim, device="cuda:0", seed=0, memory_size=16)
agent.sync_fused_weights()
env.rollout_policy(agent, 50); torch.cuda.synchronize()
tb = torch.zeros((N, 32), dtype=torch.int64, device="cuda")
env.L.cn_debug_set_timing(env.h, C.c_void_p(tb.data_ptr()))
rows = []
for rep in range(8):
    tb.zero_()
    env.rollout_policy(agent, T + rep)          # T + rep: the last period lands on different phases of the pedestrians' cycle
    torch.cuda.synchronize()
    t = tb.cpu().numpy()
    wg = t[:, 27] > 0          # rows of the workgroups' first environments; [25] / [26] = that wave's HW_ID / XCC_ID (env_kernel_body)
    cu = ((t[wg][:, 26] & 15) << 8) | ((t[wg][:, 25] >> 8) & 255)
    rows.append(np.concatenate([t[wg][:, 27:32], cu[:, None]], 1).astype(np.float64))
print("kernel:", env.kernel_name("policy"), " workgroups:", rows[0].shape[0], " CN_POL_ENVS=%s CN_POL_LOCK=%s" % (os.environ.get("CN_POL_ENVS"), os.environ.get("CN_POL_LOCK")))
r = np.concatenate(rows)
names = ["lock wait + barrier", "actor tile (+ barrier, release)", "Env.step of wave 0", "wait for the slowest wave"]
for k, nm in enumerate(names):
    d = r[:, k + 1] - r[:, k]
    print("  %-34s mean %8.0f  p10 %8.0f  p90 %8.0f ticks" % (nm, d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
print("  %-34s mean %8.0f ticks" % ("period (last one)", (r[:, 4] - r[:, 0]).mean()))
# workgroups that share a CU (same XCD: one s_memtime counter): offset between their period starts
offs = []
for a in rows:
    by = {}
    for row in a:
        by.setdefault(int(row[5]), []).append(row[0])
    for v in by.values():
        if len(v) == 2:
            offs.append(abs(v[0] - v[1]))
if offs:
    offs = np.array(offs)
    print("  CUs with two workgroups: %d; |offset of their period starts| mean %.0f  p10 %.0f  median %.0f  p90 %.0f ticks" % (
        len(offs) // len(rows), offs.mean(), np.percentile(offs, 10), np.median(offs), np.percentile(offs, 90)))
