#!/usr/bin/env python
"""BASELINE.json configs 2, 3 and 5 on one MI355X (config 4 = 8 GPUs is the driver's `bench.py --gpus 8`)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import Config
from crowdnav.env import VecEnv
from crowdnav.td3 import Agent

def run(name, cfg, actor=False, steps=400, mode="next", fused=False, mfma=False):
    env = VecEnv(cfg); env.reset(); N = env.N
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
    agent = Agent(obs_dim=env.D, device="cuda", seed=0, memory_size=16) if actor else None
    obs = env.obs
    pol = (lambda o: agent.act_mfma(o)) if mfma else ((lambda o: agent.act_fused(o)) if fused else (lambda o: agent.act(o)))
    for i in range(40):
        obs, _, _ = env.step(pol(obs) if actor else acts[i % 16], auto_reset=mode)
    ep0 = env.counters()[:, 8].sum().item(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        obs, _, _ = env.step(pol(obs) if actor else acts[i % 16], auto_reset=mode)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    resets = env.counters()[:, 8].sum().item() - ep0 if mode == "next" else 0
    print("%-52s %8.4f ms/step  %8.2f M env-steps/s" % (name, dt / steps * 1e3, (N * steps - resets) / dt / 1e6))
    env.close()

run("config 2: 4096 x 20 peds x 360 rays (open loop)", Config(n_envs=4096, ped_cycle_ms=1400))
run("config 3: 4096 x 20 peds, TD3 actor in the loop", Config(n_envs=4096, ped_cycle_ms=1400), actor=True)
run("config 3 with the fused policy tail (cn_policy_tail)", Config(n_envs=4096, ped_cycle_ms=1400), actor=True, fused=True)
run("config 3 with the one-kernel f32-MFMA actor (cn_actor_forward)", Config(n_envs=4096, ped_cycle_ms=1400), actor=True, mfma=True)

def run_graphed(name, cfg, steps=400):
    from crowdnav.rollout import GraphedRollout
    env = VecEnv(cfg); N = env.N
    agent = Agent(obs_dim=env.D, device="cuda", seed=0, memory_size=16)
    with torch.no_grad():
        g = GraphedRollout(env, agent)
        for i in range(40): g.step()
        ep0 = env.counters()[:, 8].sum().item(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps): g.step()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    resets = env.counters()[:, 8].sum().item() - ep0
    print("%-52s %8.4f ms/step  %8.2f M env-steps/s" % (name, dt / steps * 1e3, (N * steps - resets) / dt / 1e6))
    env.close()

def run_groups(name, cfg, G, actor=False, steps=400):
    from crowdnav.env import VecEnvGroups
    from crowdnav.rollout import rollout_groups
    envs = VecEnvGroups(cfg, groups=G); envs.reset(); N = envs.N
    if actor:
        agent = Agent(obs_dim=envs.D, device="cuda", seed=0, memory_size=16)
        rollout_groups(envs, agent, 40)
        ep0 = envs.episodes(); torch.cuda.synchronize(); t0 = time.perf_counter()
        rollout_groups(envs, agent, steps)
    else:
        g = torch.Generator(device="cuda").manual_seed(1)
        acts = torch.stack([torch.rand((16, N), generator=g, device="cuda") * 0.22, torch.rand((16, N), generator=g, device="cuda") * 4 - 2], 2).contiguous()
        def loop(k):
            for i in range(k):
                for gi in range(G):
                    envs.step_group(gi, acts[i % 16][envs.rows(gi)], auto_reset="next")
        loop(40)
        ep0 = envs.episodes(); torch.cuda.synchronize(); t0 = time.perf_counter()
        loop(steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    resets = envs.episodes() - ep0
    print("%-52s %8.4f ms/step  %8.2f M env-steps/s" % (name, dt / steps * 1e3, (N * steps - resets) / dt / 1e6))
    envs.close()

run_groups("config 2 as 4 stream groups (VecEnvGroups)", Config(n_envs=4096, ped_cycle_ms=1400), 4)
run_groups("config 3, MFMA actor, 2 stream groups (rollout_groups)", Config(n_envs=4096, ped_cycle_ms=1400), 2, actor=True)
run_groups("config 3, MFMA actor, 4 stream groups (rollout_groups)", Config(n_envs=4096, ped_cycle_ms=1400), 4, actor=True)
run_graphed("config 3 as one HIP graph per step", Config(n_envs=4096, ped_cycle_ms=1400))
run("config 4 shard: 2048 x 20 peds x 360 rays (1 of 8 GPUs)", Config(n_envs=2048, ped_cycle_ms=1400))
run("config 5: 4096 x 100 peds x 720 rays", Config(n_envs=4096, n_peds=100, n_rays=720, room_half=2.4, ped_cycle_ms=1400))
run_groups("config 4 shard as 2 stream groups", Config(n_envs=2048, ped_cycle_ms=1400), 2)
run_groups("config 5 as 4 stream groups", Config(n_envs=4096, n_peds=100, n_rays=720, room_half=2.4, ped_cycle_ms=1400), 4)
run("16384 x 20 peds x 360 rays on one GPU", Config(n_envs=16384, ped_cycle_ms=1400))
run_groups("16384 x 20 peds x 360 rays as 4 stream groups", Config(n_envs=16384, ped_cycle_ms=1400), 4)
run_groups("16384 envs, MFMA actor in the loop, 4 stream groups", Config(n_envs=16384, ped_cycle_ms=1400), 4, actor=True)
run("obs_layout 1 (environment_stage_1_original, 363 inputs): 4096 x 20 peds x 360 rays", Config(n_envs=4096, ped_cycle_ms=1400, obs_layout=1))
run_groups("obs_layout 1 as 4 stream groups", Config(n_envs=4096, ped_cycle_ms=1400, obs_layout=1), 4)
run("obs_layout 1: 16384 envs", Config(n_envs=16384, ped_cycle_ms=1400, obs_layout=1))
