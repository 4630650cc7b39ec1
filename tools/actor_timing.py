#!/usr/bin/env python
"""Where does a cn_actor_kernel tile spend its time?  (profiling build: s_memtime stamps of wave 0 of every workgroup.)
Usage: python tools/actor_timing.py [N]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import numpy as np, torch
from crowdnav import _abi
_abi.LIB_PATH = os.path.abspath(os.environ["CN_LIB"]) if os.environ.get("CN_LIB") else _abi.LIB_PATH.replace("libcrowdnav.so", "libcrowdnav_timing.so"); _abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav.td3 import Agent
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
agent = Agent(obs_dim=398, device="cuda", seed=0, memory_size=16)
obs = torch.randn((N, 398), device="cuda"); out = torch.empty((N, 2), device="cuda")
L = _abi.lib()
nb = (N + 15) // 16
tb = torch.zeros((nb, 8), dtype=torch.int64, device="cuda")
L.cn_debug_set_actor_timing.argtypes = [C.c_void_p]
assert L.cn_debug_set_actor_timing(C.c_void_p(tb.data_ptr())) == 0
NAMES = ["stage obs (loads + LDS stores)", "barrier", "layer 1 (+ first block of W2)", "barrier", "layer 2", "barrier", "layer 3 + heads"]
acc = []
with torch.no_grad():
    for i in range(60):
        tb.zero_(); agent.act_mfma(obs, out); torch.cuda.synchronize()
        if i >= 20:
            t = tb.cpu().numpy().astype(np.float64); acc.append(np.diff(t, axis=1).mean(0)); tot = (t[:, 7] - t[:, 0])
m = np.mean(acc, 0)
print("N = %d, %d workgroups; mean s_memtime ticks per phase of wave 0 (total %.0f; last sample: tile min %.0f max %.0f)" % (N, nb, m.sum(), tot.min(), tot.max()))
for n_, v in zip(NAMES, m):
    print("  %-34s %8.0f  %5.1f %%" % (n_, v, 100 * v / m.sum()))
