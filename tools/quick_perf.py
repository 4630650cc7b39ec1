#!/usr/bin/env python
"""A/B throughput of cn_env_kernel builds: env-steps/s for the shapes that matter (4096 envs one launch / same-call reset /
4 stream groups, 16384 envs one launch / 4 groups, config 5).  CN_LIB=<path to a libcrowdnav.so variant> selects the
library (default: the product build); run several variants back to back in one gpurun call.
    python tools/quick_perf.py [label]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import _abi
if os.environ.get("CN_LIB"):
    _abi.LIB_PATH = os.path.abspath(os.environ["CN_LIB"])
    _abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv, VecEnvGroups

label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(_abi.LIB_PATH)
STEPS, PRE = int(os.environ.get("CN_STEPS", 400)), 200


def acts_for(N):
    g = torch.Generator(device="cuda").manual_seed(1)
    return torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()


def one(cfg, mode):
    env = VecEnv(cfg, arbitration=os.environ.get("CN_ARB", "auto")); env.reset(); N = env.N; acts = acts_for(N)
    for i in range(PRE): env.step(acts[i % 16], auto_reset=mode)
    ep0 = env.counters()[:, 8].sum().item(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(STEPS): env.step(acts[i % 16], auto_reset=mode)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    resets = env.counters()[:, 8].sum().item() - ep0 if mode == "next" else 0
    env.close()
    return (N * STEPS - resets) / dt / 1e6


def groups(cfg, G):
    envs = VecEnvGroups(cfg, groups=G, arbitration=os.environ.get("CN_GROUP_ARB") or None); envs.reset(); N = envs.N; acts = acts_for(N)
    rows = [envs.rows(g) for g in range(G)]
    def loop(k):
        for i in range(k):
            for gi in range(G):
                envs.step_group(gi, acts[i % 16][rows[gi]], auto_reset="next")
    loop(PRE); ep0 = envs.episodes(); torch.cuda.synchronize(); t0 = time.perf_counter()
    loop(STEPS); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    resets = envs.episodes() - ep0
    envs.close()
    return (N * STEPS - resets) / dt / 1e6


X = dict(risk_mode=int(os.environ.get("CN_RISK", 0)), obs_layout=int(os.environ.get("CN_LAYOUT", 0)),      # CN_RISK=1: gt mode
         py2_round=int(os.environ.get("CN_PY2", 0)), geos_untyped_empty=int(os.environ.get("CN_GEOS", 0)),  # the reference's own platform
         ped_mode=int(os.environ.get("CN_PED_MODE", 0)), sf_tick_ms=int(os.environ.get("CN_SF_TICK", 0)))                                                    # 2: social-force pedestrians
c2 = Config(n_envs=4096, ped_cycle_ms=1400, **X)
c16 = Config(n_envs=16384, ped_cycle_ms=1400, **X)
c5 = Config(n_envs=4096, n_peds=100, n_rays=720, room_half=2.4, ped_cycle_ms=1400, **X)
r = [one(c2, "next"), one(c2, "same"), groups(c2, 4), one(c16, "next"), groups(c16, 4), one(c5, "next"), groups(c5, 4)]
print("%-28s 4096: 1-launch %6.2f  same-call %6.2f  4-groups %6.2f | 16384: 1-launch %6.2f  4-groups %6.2f | cfg5: %6.2f  4-groups %6.2f  (M env-steps/s)"
      % ((label,) + tuple(r)), flush=True)
