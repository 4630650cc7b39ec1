#!/usr/bin/env python
"""What makes a wavefront's ray loop / tracker / cone slow?  Correlates per-env stage ticks (profiling build) with env
features read from a snapshot: wall proximity, tracks, confirmed objects.  Usage: python tools/tail_corr.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import numpy as np, torch
from crowdnav import _abi
_abi.LIB_PATH = _abi.LIB_PATH.replace("libcrowdnav.so", "libcrowdnav_timing.so"); _abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv
N = 4096
env = VecEnv(Config(n_envs=N, ped_cycle_ms=1400)); env.reset()
tb = torch.zeros((N, 32), dtype=torch.int64, device="cuda")
env.L.cn_debug_set_timing(env.h, C.c_void_p(tb.data_ptr()))
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
for i in range(230):
    env.step(acts[i % 16], auto_reset="next")
rows = []
for i in range(20):
    tb.zero_(); env.step(acts[i % 16], auto_reset="next"); torch.cuda.synchronize()
    t = tb.cpu().numpy().astype(np.float64)
    snap = env.snapshot()
    sd = np.frombuffer(snap[:N * 24 * 8].tobytes(), dtype=np.float64).reshape(N, 24)
    si = np.frombuffer(snap[N * 24 * 8:N * 24 * 8 + N * 16 * 4].tobytes(), dtype=np.int32).reshape(N, 16)
    P = 20
    off = N * 24 * 8 + N * 16 * 4
    pp = np.frombuffer(snap[off:off + N * P * 16].tobytes(), dtype=np.float64).reshape(N, P, 2)
    d = np.hypot(pp[:, :, 0] - sd[:, None, 0], pp[:, :, 1] - sd[:, None, 1])
    nnear = (d < 0.66).sum(1)
    wx = (np.abs(sd[:, 0]) > 0.78).astype(int); wy = (np.abs(sd[:, 1]) > 0.78).astype(int)
    ok = (t[:, 19] > 0) & (t[:, 4] > 0) & (t[:, 3] > 0)
    rows.append(np.stack([t[:, 4] - t[:, 3], t[:, 14] - t[:, 13], t[:, 16] - t[:, 15], nnear, wx + wy, si[:, 2], si[:, 10], t[:, 19] - t[:, 0]], 1)[ok])
r = np.concatenate(rows)
names = ["ray", "tracker", "cone", "nnear", "walls", "ntracks", "nconf", "life"]
for k, nm in ((3, "near pedestrians"), (4, "walls in reach"), (5, "tracks"), (6, "confirmed objects")):
    print("by %s:" % nm)
    for v in sorted(set(r[:, k].astype(int)))[:12]:
        m = r[:, k].astype(int) == v
        print("   %2d: n %7d  ray %7.0f  tracker %7.0f  cone %7.0f  life %7.0f" % (v, m.sum(), r[m, 0].mean(), r[m, 1].mean(), r[m, 2].mean(), r[m, 7].mean()))
