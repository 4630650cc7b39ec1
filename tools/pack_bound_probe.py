#!/usr/bin/env python
"""Upper bound for "several environments per wavefront in the narrow stages" (DESIGN section 11): in the profiling build three of every
four wavefronts END at a stage stamp (their environments' results are then invalid -- this is a timing probe only) while the fourth
runs the whole step, i.e. the stages after the stamp cost one wavefront per four environments, with no hand-off and no packing
overhead at all.  Compared with the same library uncut.   python tools/pack_bound_probe.py [stamp ...]"""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import _abi
_abi.LIB_PATH = _abi.LIB_PATH.replace("libcrowdnav.so", "libcrowdnav_timing.so")
_abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv, VecEnvGroups

STEPS, PRE = 400, 100
stamps = [int(x) for x in sys.argv[1:]] or [10, 13, 15]


def acts_for(N):
    g = torch.Generator(device="cuda").manual_seed(1)
    return torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()


def run(G, mask):
    cfg = Config(n_envs=4096, ped_cycle_ms=1400, max_steps=1000000)
    envs = VecEnvGroups(cfg, groups=G); envs.reset(); acts = acts_for(4096)
    rows = [envs.rows(g) for g in range(G)]
    def loop(k):
        for i in range(k):
            for gi in range(G):
                envs.step_group(gi, acts[i % 16][rows[gi]], auto_reset="next")
    loop(PRE); torch.cuda.synchronize()          # a normal pre-roll: the tracker tables fill up as in any run
    for e in envs.envs:
        e.L.cn_debug_set_ablate(e.h, mask)
    loop(20); torch.cuda.synchronize(); t0 = time.perf_counter()
    loop(STEPS); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    name = envs.envs[0].kernel_name("multi" if G > 1 else "step")
    envs.close()
    return 4096 * STEPS / dt / 1e6, name


for G in (1, 4):
    base, name = run(G, 0)
    print("%d stream group(s), %s (profiling build): uncut %.1f M env-steps/s" % (G, name, base))
    for k in stamps:
        v, _ = run(G, ((k + 1) << 8) | (1 << 16))
        print("   three of four wavefronts end at stamp %2d: %.1f M  (x %.2f)" % (k, v, v / base))
