import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import Config
from crowdnav.env import VecEnv
from crowdnav.td3 import Agent
for N in (4096, 8192):
    env = VecEnv(Config(n_envs=N, ped_cycle_ms=1400)); env.reset()
    agent = Agent(obs_dim=env.D, device="cuda", seed=0, memory_size=16)
    env.rollout_fused(agent, 300); torch.cuda.synchronize()
    for K in (20, 200, 1000):
        c0 = env.counters().clone(); torch.cuda.synchronize()
        t0 = time.perf_counter(); env.rollout_fused(agent, K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        c1 = env.counters()
        resets = int(((c1[:, 8] - c1[:, 9]).sum() - (c0[:, 8] - c0[:, 9]).sum()).item())
        print("fused rollout N=%d K=%4d: %.4f ms/step %.2f M env-steps/s (resets %d)" % (N, K, dt / K * 1e3, (N * K - resets) / dt / 1e6, resets))
    # chain for comparison
    for K in (200,):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(K):
            a = agent.act_mfma(env.obs); env.step(a, auto_reset="next")
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("chain N=%d K=%d: %.4f ms/step %.2f M/s (resets not subtracted)" % (N, K, dt / K * 1e3, N * K / dt / 1e6))
    env.close()
