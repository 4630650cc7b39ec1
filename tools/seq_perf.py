#!/usr/bin/env python
"""cn_step_sequence (T open-loop steps per launch) against one launch per step and stream groups, same envs, resets subtracted."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import _abi
if os.environ.get("CN_LIB"):
    _abi.LIB_PATH = os.path.abspath(os.environ["CN_LIB"]); _abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = Config(n_envs=N, ped_cycle_ms=1400, n_peds=int(os.environ.get("CN_PEDS", 20)), n_rays=int(os.environ.get("CN_RAYS", 360)),
             room_half=2.4 if int(os.environ.get("CN_PEDS", 20)) > 50 else 1.4)
env = VecEnv(cfg); env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
def acts(T):
    return torch.stack([torch.rand((T, N), generator=g, device="cuda") * 0.22, torch.rand((T, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
def marker():
    c = env.counters(); torch.cuda.synchronize(); return int((c[:, 8] - c[:, 9]).sum().item())
env.step_sequence(acts(300)); torch.cuda.synchronize()
for T in (20, 100, 1000):
    a = acts(T); call = env.bind_step_sequence(a)
    for rep in range(3):
        m0 = marker(); torch.cuda.synchronize(); t0 = time.perf_counter(); call(); torch.cuda.synchronize(); dt = time.perf_counter() - t0; m1 = marker()
    print("sequence N=%d T=%4d: %.4f ms/step %.2f M env-steps/s" % (N, T, dt / T * 1e3, (N * T - (m1 - m0)) / dt / 1e6))
a = acts(64)
for rep in range(2):
    m0 = marker(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(400): env.step(a[i % 64], auto_reset="next")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0; m1 = marker()
print("one launch per step N=%d: %.4f ms/step %.2f M env-steps/s" % (N, dt / 400 * 1e3, (N * 400 - (m1 - m0)) / dt / 1e6))
