#!/usr/bin/env python
"""Do the wavefronts that share a SIMD share it fairly?  (profiling build: s_memtime at entry / exit + HW_ID / XCC_ID per wave.)
For one-launch-per-step at N envs: groups the waves of each launch by (XCC, SE, CU, SIMD), ranks them by entry time and prints
lifetime and exit time by rank -- an arbiter that favours the oldest wave shows up as lifetimes growing with rank and the SIMD's
last exit far behind its first.  Usage: [CN_ARB=fair] python tools/wave_fairness.py [N]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import numpy as np, torch
from crowdnav import _abi
_abi.LIB_PATH = os.path.abspath(os.environ["CN_LIB"]) if os.environ.get("CN_LIB") else _abi.LIB_PATH.replace("libcrowdnav.so", "libcrowdnav_timing.so"); _abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = VecEnv(Config(n_envs=N, ped_cycle_ms=1400), arbitration=os.environ.get("CN_ARB", "oldest_first")); env.reset()
print("N = %d, arbitration %s (CN_ARB=oldest_first|fair)" % (N, env.arbitration))
tb = torch.zeros((N, 32), dtype=torch.int64, device="cuda")
env.L.cn_debug_set_timing(env.h, C.c_void_p(tb.data_ptr()))
g = torch.Generator(device="cuda").manual_seed(1)
acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
by_rank = {}; per_simd = []; occ = []; slot_life = {}
for i in range(260):
    tb.zero_()
    env.step(acts[i % 16], auto_reset="next"); torch.cuda.synchronize()
    if i < 200: continue
    t = tb.cpu().numpy()
    t0, t1, hw, xcc = t[:, 0], t[:, 19], t[:, 25], t[:, 26] & 0xf
    wave, simd, cu, se = hw & 0xf, (hw >> 4) & 3, (hw >> 8) & 0xf, (hw >> 13) & 7
    key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
    for k in np.unique(key):
        m = np.nonzero(key == k)[0]
        o = m[np.argsort(t0[m], kind="stable")]
        base = t0[xcc == xcc[o[0]]].min()                 # s_memtime is per XCD
        occ.append(len(o))
        per_simd.append((len(o), t0[o].max() - t0[o].min(), t1[o].min() - base, t1[o].max() - base))
        for r, e in enumerate(o):
            by_rank.setdefault((len(o), r), []).append((t0[e] - base, t1[e] - t0[e], t1[e] - base))
            slot_life.setdefault(int(wave[e]), []).append(t1[e] - t0[e])
    if i == 259:
        print("launch span per XCD (first entry -> last exit): %s ticks; SIMDs in use %d; CUs in use %d" % (
            " ".join(str(int(t1[xcc == x].max() - t0[xcc == x].min())) for x in np.unique(xcc)), len(np.unique(key)), len(np.unique(key // 4))))
occ = np.array(occ)
print("waves per SIMD: " + "  ".join("%d: %.1f %%" % (k, 100.0 * (occ == k).mean()) for k in np.unique(occ)))
ps = np.array(per_simd, dtype=np.float64)
for k in np.unique(occ):
    m = ps[:, 0] == k
    print("SIMDs with %d waves: entry spread %6.0f   first exit %7.0f   last exit %7.0f" % (k, ps[m, 1].mean(), ps[m, 2].mean(), ps[m, 3].mean()))
print("by entry rank within the SIMD (entry after launch start, lifetime, exit after launch start):")
for (k, r) in sorted(by_rank):
    a = np.array(by_rank[(k, r)], dtype=np.float64)
    print("  %d waves, rank %d: n %6d  entry %6.0f  life %7.0f  exit %7.0f (p99 %7.0f)" % (k, r, len(a), a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), np.percentile(a[:, 2], 99)))
print("by hardware wave slot: " + "  ".join("%d: %.0f" % (k, np.mean(v)) for k, v in sorted(slot_life.items())))
