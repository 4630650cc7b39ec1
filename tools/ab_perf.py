#!/usr/bin/env python
"""Quick A/B of libcrowdnav.so variants on the headline shape (CN_LIB=<path> selects the library): env-steps/s for one launch per
step, 2 and 4 stream groups at 4096 envs and 4 groups at 16384 envs; interleaved repeats so that box drift hits every variant alike.
    python tools/ab_perf.py lib_a.so lib_b.so ...      (each variant runs in its own child process, `rounds` times in turn)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CN_AB_CHILD"):
    sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
    import torch
    from crowdnav import _abi
    _abi.LIB_PATH = os.path.abspath(os.environ["CN_LIB"]); _abi.build = lambda force=False: _abi.LIB_PATH
    from crowdnav import Config
    from crowdnav.env import VecEnvGroups
    STEPS, PRE = 300, 150
    def run(N, G, P=20, R=360, room=1.4):
        envs = VecEnvGroups(Config(n_envs=N, ped_cycle_ms=1400, n_peds=P, n_rays=R, room_half=room), groups=G); envs.reset()
        g = torch.Generator(device="cuda").manual_seed(1)
        acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
        pre = envs.bind_step_sequence([acts[i % 16] for i in range(PRE)]); call = envs.bind_step_sequence([acts[i % 16] for i in range(STEPS)])
        pre(); ep0 = envs.episodes(); torch.cuda.synchronize(); t0 = time.perf_counter()
        call(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        r = (N * STEPS - (envs.episodes() - ep0)) / dt / 1e6
        envs.close(); return r
    if os.environ.get("CN_AB_CFG5"):
        out = [run(4096, 1, 100, 720, 2.4), run(4096, 2, 100, 720, 2.4), run(4096, 4, 100, 720, 2.4), run(2048, 1, 100, 720, 2.4)]
    else:
        out = [run(4096, 1), run(4096, 2), run(4096, 4), run(16384, 4)]
    print(" ".join("%.2f" % x for x in out)); sys.exit(0)
libs = sys.argv[1:]
rounds = int(os.environ.get("CN_AB_ROUNDS", "3"))
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        lib_, _, kv = l.partition("@")            # "lib.so@VAR=VALUE": the same library with an environment switch
        extra = dict([kv.split("=", 1)]) if kv else {}
        o = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, CN_AB_CHILD="1", CN_LIB=lib_, **extra), capture_output=True, text=True)
        line = [x for x in o.stdout.splitlines() if x and x[0].isdigit()]
        if not line:
            print(l, "FAILED", o.stderr[-500:]); continue
        res[l].append([float(x) for x in line[-1].split()])
print("%-40s %s" % ("library (median of %d)" % rounds, ("cfg5: 4096x1   4096x2   4096x4   2048x1" if os.environ.get("CN_AB_CFG5") else "4096x1   4096x2   4096x4  16384x4")))
for l in libs:
    if res[l]:
        cols = list(zip(*res[l]))
        print("%-40s %s" % (os.path.basename(l), "  ".join("%7.2f" % sorted(c)[len(c) // 2] for c in cols)))
