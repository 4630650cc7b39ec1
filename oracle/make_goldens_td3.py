#!/usr/bin/env python
"""TEST INFRASTRUCTURE.  Golden vectors for the caller of the hot path: the reference's
own TD3 classes (turtlebot3_rl_sim/src/td3.py -- imports only torch/numpy, so it runs here unmodified) are
imported from /root/reference and driven on seeded inputs:

  actor_*    Actor(398, 2, 256).forward on 32 observations, weights from torch.manual_seed(7)
  upd_*      four consecutive Agent.learn(step) calls (step = 0..3, policy_update = 2 -> both branches) of a small
             agent (46 -> 32 -> 32), batch 16, with the replay sample order and the target-policy noise pinned:
             initial parameters of the six networks, the batch, the noise, and every parameter after each call.

Writes tests/golden/td3.npz (data only).  Usage: python oracle/make_goldens_td3.py"""
import importlib.util
import os
import random
import sys

import numpy as np
import torch

REF = "/root/reference/turtlebot3_rl_sim/src/td3.py"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_td3", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.device = torch.device("cpu")
    return m


def flat(prefix, module, out):
    for k, v in module.state_dict().items():
        out["%s.%s" % (prefix, k)] = v.detach().cpu().numpy().copy()


def main():
    ref = load_ref()
    out = {}
    # ---- Actor.forward ----
    torch.manual_seed(7)
    actor = ref.Actor(398, 2, 256, 0.22, 2.0)
    rng = np.random.RandomState(3)
    obs = np.concatenate([rng.uniform(0.08, 0.6, (32, 359)), rng.uniform(-3.2, 3.2, (32, 39))], 1).astype(np.float32)
    with torch.no_grad():
        act = actor(torch.from_numpy(obs)).numpy()
    out["actor_seed"] = np.int64(7); out["actor_obs"] = obs; out["actor_out"] = act
    # ---- Agent.act clip bounds (noise off) ----
    ag = ref.Agent(398, 2, 256, 3e-4, 3e-4, 128, 1000, 0.99, 0.005, 0.22, 2.0, 0.2, 0.5, 2)
    ag.actor_local.load_state_dict(actor.state_dict())
    out["act_single"] = np.stack([ag.act(obs[i].astype(np.float64), 0, add_noise=False)[0] for i in range(8)])
    # ---- four Agent.learn calls ----
    torch.manual_seed(11)
    H, B = 32, 16
    a = ref.Agent(46, 2, H, 3e-4, 3e-4, B, 1000, 0.99, 0.005, 0.22, 2.0, 0.2, 0.5, 2)
    nets = dict(actor=a.actor_local, actor_t=a.actor_target, q1=a.critic_local1, q1_t=a.critic_target1,
                q2=a.critic_local2, q2_t=a.critic_target2)
    for k, m in nets.items():
        flat("init." + k, m, out)
    s = rng.uniform(-1, 1, (B, 46)).astype(np.float32); s2 = rng.uniform(-1, 1, (B, 46)).astype(np.float32)
    ac = np.stack([rng.uniform(0, 0.22, B), rng.uniform(-2, 2, B)], 1).astype(np.float32)
    r = rng.uniform(-5, 5, B).astype(np.float32); d = (rng.uniform(0, 1, B) < 0.25)
    for i in range(B):
        a.step(s[i], ac[i][None, :], float(r[i]), s2[i], bool(d[i]))      # the trainer stores action as (1, 2) (TRAIN:128)
    out["upd_s"], out["upd_a"], out["upd_r"], out["upd_s2"], out["upd_d"] = s, ac, r, s2, d.astype(np.float32)
    random.sample = lambda pop, k: list(pop)[:k]                            # pinned replay order
    ref.random.sample = random.sample
    noises = rng.standard_normal((4, B, 2)).astype(np.float32)
    out["upd_noise"] = noises
    it = iter(noises)
    orig = torch.randn_like
    torch.randn_like = lambda t, *aa, **kw: torch.from_numpy(next(it)).clone()   # pinned target-policy noise
    for step in range(4):
        a.learn(step)
        for k, m in nets.items():
            flat("step%d.%s" % (step, k), m, out)
    torch.randn_like = orig
    path = os.path.join(ROOT, "tests", "golden", "td3.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
