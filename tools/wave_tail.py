#!/usr/bin/env python
"""How long does each wavefront of a cn_env_kernel launch live?  (profiling build, s_memtime at entry and exit).
Prints the distribution of per-env wave lifetimes for stepping envs and for envs that spend the launch on Env.reset
(next-step reset), which is what bounds a launch from below.  Usage: python tools/wave_tail.py [N]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import numpy as np, torch
from crowdnav import _abi
_abi.LIB_PATH = _abi.LIB_PATH.replace("libcrowdnav.so", "libcrowdnav_timing.so"); _abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = VecEnv(Config(n_envs=N, ped_cycle_ms=1400), arbitration=os.environ.get("CN_ARB", "auto")); env.reset()
print("N = %d, arbitration %s (CN_ARB=auto|oldest_first|fair)" % (N, env.arbitration))
tb = torch.zeros((N, 32), dtype=torch.int64, device="cuda")
env.L.cn_debug_set_timing(env.h, C.c_void_p(tb.data_ptr()))
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
step, reset, span = [], [], []
ORDER = [0, 1, 20, 21, 22, 23, 24] + list(range(2, 20))      # program order of the stamps
NAMES = {1: "load state", 20: "peds + robot 150 ms", 21: "deque, robot 10 ms", 22: "(pre-observe)", 23: "wp(step 1), dist, heading",
         24: "wp refresh", 2: "sincos(w)", 3: "near-ped list", 4: "ray loop", 5: "bbox (reset)", 6: "gradients", 7: "flag words",
         8: "type machine", 9: "aliasing", 10: "association", 11: "order/split", 12: "word bases", 13: "confirmation",
         14: "tracker", 15: "speeds/defaults", 16: "cone + top-K", 17: "counters/tail", 18: "reward + outputs", 19: "write-back"}
stg = {"median": [], "slowest": [], "reset": []}
for i in range(300):
    pend = env.counters()[:, 9].clone()
    tb.zero_()
    env.step(acts[i % 16], auto_reset="next"); torch.cuda.synchronize()
    if i < 200: continue
    t = tb.cpu().numpy().astype(np.float64); pr = pend.cpu().numpy() != 0
    life = t[:, 19] - t[:, 0]
    step.append(life[~pr]); reset.append(life[pr])
    tt = t[:, ORDER]
    d = np.diff(tt, axis=1)
    ok = (~pr) & (tt > 0).all(1)
    if ok.sum() > 100:
        lo, hi = np.percentile(life[ok], [40, 60]); cut = np.percentile(life[ok], 99.5)
        stg["median"].append(d[ok & (life >= lo) & (life <= hi)].mean(0)); stg["slowest"].append(d[ok & (life >= cut)].mean(0))
    okr = pr & (t[:, ORDER[2:]] >= 0).all(1)
    span.append((t[:, 19].max() - t[:, 0].min(), np.median(life[~pr]), life.max(), int(pr.sum())))
s = np.concatenate(step); r = np.concatenate(reset)
print("stepping envs : n %8d  ticks mean %7.0f  p50 %7.0f  p99 %7.0f  max %7.0f" % (len(s), s.mean(), np.median(s), np.percentile(s, 99), s.max()))
print("resetting envs: n %8d  ticks mean %7.0f  p50 %7.0f  p99 %7.0f  max %7.0f" % (len(r), r.mean(), np.median(r), np.percentile(r, 99), r.max()))
sp = np.array(span)
print("per launch: first entry -> last exit %7.0f ticks; median stepping wave %7.0f; slowest wave %7.0f; resetting envs per launch %.1f"
      % tuple(sp.mean(0)))
if stg["median"]:
    m, w = np.mean(stg["median"], 0), np.mean(stg["slowest"], 0)
    print("ticks per stage: median stepping waves (40-60 %%) vs the slowest 0.5 %% of each launch")
    for k in range(len(m)):
        print("  %-28s %8.0f %8.0f  %+8.0f" % (NAMES.get(ORDER[k + 1], str(ORDER[k + 1])), m[k], w[k], w[k] - m[k]))
    print("  %-28s %8.0f %8.0f" % ("total", m.sum(), w.sum()))
