import ctypes as C, os, sys
sys.path.insert(0, "/root/repo/drl-based-mapless-crowd-navigation-with-perceived-risk_amd")
import numpy as np, torch
from crowdnav import _abi
_abi.LIB_PATH = _abi.LIB_PATH.replace("libcrowdnav.so", "libcrowdnav_timing.so"); _abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = VecEnv(Config(n_envs=N, ped_cycle_ms=1400, max_steps=100000)); env.reset()
tb = torch.zeros((N, 32), dtype=torch.int64, device="cuda")
env.L.cn_debug_set_timing(env.h, C.c_void_p(tb.data_ptr()))
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
seq = [1, 27, 28, 20, 21, 22, 23, 24, 2]
names = ["trig of the step (sincos x 4, packed)", "pedestrians: one pass over 160 ms", "robot advance 150 ms", "deque + advance 10 ms", "-> observe entry", "waypoint (step 1)", "dist + heading (atan2) + round", "waypoint refresh", "sincos(w), sincos(yaw), origin"]
acc = np.zeros(len(seq) - 1); cnt = 0
for i in range(60):
    tb.zero_(); env.step(acts[i % 16], auto_reset="next"); torch.cuda.synchronize()
    t = tb.cpu().numpy().astype(np.float64)
    ok = (t[:, 19] > 0) & (t[:, 20] > 0) & (env.done.cpu().numpy() == 0)
    if i < 10 or not ok.any(): continue
    acc += np.array([(t[ok, seq[k + 1]] - t[ok, seq[k]]).mean() for k in range(len(seq) - 1)]); cnt += 1
for n_, a in zip(names, acc / cnt): print("  %-36s %8.0f" % (n_, a))
