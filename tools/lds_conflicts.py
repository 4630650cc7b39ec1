#!/usr/bin/env python
"""One ablation mask of the profiling build for N launches (PMC attribution of LDS bank conflicts by stage; results invalid while
a mask is set).  Usage under rocprofv3:  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -- python tools/lds_conflicts.py <mask>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch  # noqa: E402
from crowdnav import _abi  # noqa: E402
_abi.LIB_PATH = _abi.LIB_PATH.replace("libcrowdnav.so", "libcrowdnav_timing.so")
_abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config  # noqa: E402
from crowdnav.env import VecEnv  # noqa: E402

mask = int(sys.argv[1]) if len(sys.argv) > 1 else 0
env = VecEnv(Config(n_envs=4096, ped_cycle_ms=1400))
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.stack([torch.rand((16, 4096), generator=g, device="cuda") * 0.22, torch.rand((16, 4096), generator=g, device="cuda") * 4 - 2], 2).contiguous()
for i in range(150):                       # an honest state first (tracks, pending resets), then the mask
    env.step(acts[i % 16], auto_reset="next")
env.L.cn_debug_set_ablate(env.h, mask)
for i in range(60):
    env.step(acts[i % 16], auto_reset="next")
torch.cuda.synchronize()
