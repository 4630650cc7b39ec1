#!/usr/bin/env python
"""Two wavefronts per environment (cn_env_kernel_s360_x2) against the one-wave kernels: bit-identical outputs / state, and speed at
small grids.  CN_X2=0 / 1 force the choice; default: x2 up to 8 x CUs environments."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
if os.environ.get("CN_X2_CHILD"):
    import numpy as np, torch
    from crowdnav import Config
    from crowdnav.env import VecEnv
    mode, N = os.environ["CN_X2_CHILD"], int(sys.argv[1])
    cfg = Config(n_envs=N, ped_cycle_ms=1400, seed=1234, max_steps=60 if mode == "check" else 1000)
    env = VecEnv(cfg); env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
    if mode == "check":
        import hashlib
        h = hashlib.sha256()
        for i in range(150):
            env.step(acts[i % 16], auto_reset="next")
            h.update(env.obs.cpu().numpy().tobytes()); h.update(env.reward.cpu().numpy().tobytes()); h.update(env.done.cpu().numpy().tobytes()); h.update(env.topk_idx.cpu().numpy().tobytes())
        h.update(env.snapshot().tobytes())
        print(env.kernel_name("step"), h.hexdigest()[:16], int(env.counters()[:, 8].sum()))
    else:
        for i in range(100): env.step(acts[i % 16], auto_reset="next")
        ep0 = int(env.counters()[:, 8].sum()); torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 600
        for i in range(K): env.step(acts[i % 16], auto_reset="next")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%s N=%5d: %.2f us per step, %.2f M env-steps/s" % (env.kernel_name("step"), N, dt / K * 1e6, (N * K - (int(env.counters()[:, 8].sum()) - ep0)) / dt / 1e6))
    sys.exit(0)
def child(mode, N, x2):
    e = dict(os.environ, CN_X2_CHILD=mode)
    if x2 is not None: e["CN_X2"] = str(x2)
    o = subprocess.run([sys.executable, os.path.abspath(__file__), str(N)], env=e, capture_output=True, text=True, timeout=120)
    return (o.stdout.strip().splitlines() or ["FAILED: " + o.stderr[-300:]])[-1]
for N in (4, 130, 1024):
    a, b = child("check", N, 0), child("check", N, 1)
    print("check N=%4d: one wave [%s]  two waves [%s]  %s" % (N, a, b, "IDENTICAL" if a.split()[1:] == b.split()[1:] and "FAILED" not in a else "DIFFERENT"))
if "--speed" in sys.argv:
    for N in (1, 256, 1024, 2048, 4096):
        for x2 in (0, 1):
            print(child("speed", N, x2))
        if os.environ.get("CN_X2_PRIO_AB"):
            os.environ["CN_X2_PRIO"] = "1"; print("prio: " + child("speed", N, 1)); del os.environ["CN_X2_PRIO"]
