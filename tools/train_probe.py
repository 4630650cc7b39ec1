#!/usr/bin/env python
"""Diagnostic: the crowdnav.train loop for a few thousand launches with the policy's greedy actions and critic values printed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
import torch
from crowdnav import train as T
from crowdnav.td3 import Agent
N = int(sys.argv[1]); L = int(sys.argv[2]); U = int(sys.argv[3]) if len(sys.argv) > 3 else 16
env = T.make_env("training_as_logged", N, 500, 0, 0)
agent = Agent(obs_dim=env.D, device="cuda:0", seed=0, batch_size=128, memory_size=1_000_000)
obs = env.reset()
resetting = torch.zeros(env.N, dtype=torch.bool, device=obs.device)
p0 = [p.detach().clone() for p in agent.actor.parameters()]
ret = 0.0; nd = 0
for it in range(1, L + 1):
    act = agent.act_fused(obs, add_noise=True)
    prev = obs.clone()
    obs, reward, done = env.step(act, auto_reset="next")
    agent.memory.add_masked(prev, act, reward, obs, done, ~resetting)
    resetting = done.bool().clone()
    if len(agent.memory) > 128:
        for u in range(U):
            loss = agent.learn(it * U + u)
    if done.any():
        r = env.returns()[0]; ret += float(r[done.bool()].sum()); nd += int(done.sum())
    if it % 1000 == 0:
        with torch.no_grad():
            g = agent.actor(obs)
            q = agent.q1(obs, g)
            m = agent.memory
            drift = sum(float((p - q_).abs().sum()) for p, q_ in zip(agent.actor.parameters(), p0))
        print("N %d launch %5d: greedy v mean %.3f w mean %+.3f |w| %.3f  q1 %.1f  critic loss %.2f  actor drift %.1f  r mean %.3f d frac %.4f  mean return %.1f" % (
            N, it, float(g[:, 0].mean()), float(g[:, 1].mean()), float(g[:, 1].abs().mean()), float(q.mean()), float(loss), drift,
            float(m.r[:m.size].mean()), float(m.d[:m.size].mean()), ret / max(1, nd)), flush=True)
        ret = 0.0; nd = 0
