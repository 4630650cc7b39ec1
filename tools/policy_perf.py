#!/usr/bin/env python
"""cn_rollout_policy (T periods with the TD3 actor inside the step kernel, one launch) against the cn_actor_forward -> cn_step
chain (two launches per period), same envs and actor, resets subtracted.  BASELINE configs[2]."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import _abi
if os.environ.get("CN_LIB"):
    _abi.LIB_PATH = os.path.abspath(os.environ["CN_LIB"]); _abi.build = lambda force=False: _abi.LIB_PATH
from crowdnav import Config
from crowdnav.env import VecEnv
from crowdnav.td3 import Agent
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = Config(n_envs=N, ped_cycle_ms=1400)
env = VecEnv(cfg); env.reset()
agent = Agent(obs_dim=cfg.obs_dim, device="cuda:0", seed=0, memory_size=16)
agent.sync_fused_weights()
def marker():
    c = env.counters(); torch.cuda.synchronize(); return int((c[:, 8] - c[:, 9]).sum().item())
if "--profile" in sys.argv:       # tools/profile.sh: every launch the same (100 periods), so per-dispatch counters / 100 = per period
    for _ in range(13):
        env.rollout_policy(agent, 100)
    torch.cuda.synchronize()
    sys.exit(0)
env.rollout_policy(agent, 300); torch.cuda.synchronize()
print("kernel:", env.kernel_name("policy"))
for T in (1, 20, 100, 1000):
    call = env.bind_rollout_policy(agent, T)
    reps = 200 if T == 1 else 1
    for rep in range(3):
        m0 = marker(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): call()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0; m1 = marker()
    print("policy rollout N=%d T=%4d: %.4f ms/period %.2f M env-steps/s" % (N, T, dt / (T * reps) * 1e3, (N * T * reps - (m1 - m0)) / dt / 1e6))
act = torch.zeros((N, 2), device="cuda")
ca = agent.bind_act_mfma(env.obs, act, add_noise=True); cs = env.bind_step(act, auto_reset="next")
for rep in range(3):
    m0 = marker(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(400): ca(); cs()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0; m1 = marker()
print("act -> step chain N=%d: %.4f ms/period %.2f M env-steps/s" % (N, dt / 400 * 1e3, (N * 400 - (m1 - m0)) / dt / 1e6))
