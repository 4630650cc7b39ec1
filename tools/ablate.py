#!/usr/bin/env python
"""Stage-time attribution of cn_env_kernel by skipping stages (PROFILING ONLY; results are invalid
while a mask is set).  Prints ms/launch for each mask."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch  # noqa: E402
from crowdnav import _abi  # noqa: E402
_abi.LIB_PATH = _abi.LIB_PATH.replace("libcrowdnav.so", "libcrowdnav_timing.so")   # the mask only exists in the profiling build
_abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config  # noqa: E402
from crowdnav.env import VecEnv  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 20
R = int(sys.argv[3]) if len(sys.argv) > 3 else 360
env = VecEnv(Config(n_envs=N, n_peds=P, n_rays=R, ped_cycle_ms=1400, room_half=2.4 if P > 50 else 1.4))
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
names = {0: "full", 1: "-ped raycast", 2: "-type machine", 4: "-confirm loop", 8: "-CP/topK", 16: "-tracker",
         31: "all skipped"}
snap = env.snapshot()
for mask in (0, 1, 2, 4, 8, 16, 6, 24, 31, 0):
    env.restore(snap)
    env.L.cn_debug_set_ablate(env.h, mask)
    for i in range(20):
        env.step(acts[i % 16])
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(100):
        env.step(acts[i % 16])
    e1.record(); torch.cuda.synchronize()
    print("mask %2d %-16s %.4f ms/launch" % (mask, names.get(mask, ""), e0.elapsed_time(e1) / 100))
