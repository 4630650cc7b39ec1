import os, sys, time
sys.path.insert(0, "/root/repo/drl-based-mapless-crowd-navigation-with-perceived-risk_amd")
import torch
from crowdnav import Config
from crowdnav.env import VecEnv
def run(N, arb, steps=600):
    env = VecEnv(Config(n_envs=N, ped_cycle_ms=1400), arbitration=arb); env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
    for i in range(200): env.step(acts[i % 16], auto_reset="next")
    c = env.counters(); m0 = int((c[:, 8] - c[:, 9]).sum().item()); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): env.step(acts[i % 16], auto_reset="next")
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c = env.counters(); m1 = int((c[:, 8] - c[:, 9]).sum().item())
    env.close()
    return (N * steps - (m1 - m0)) / dt / 1e6
for N in (1024, 1536, 2048, 3072, 4096, 6144, 8192):
    a = [run(N, "oldest_first"), run(N, "fair"), run(N, "oldest_first"), run(N, "fair")]
    print("N %5d: oldest-first %.2f %.2f   fair %.2f %.2f M env-steps/s" % (N, a[0], a[2], a[1], a[3]), flush=True)
