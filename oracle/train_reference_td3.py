#!/usr/bin/env python
"""TEST INFRASTRUCTURE, container-only (imports /root/reference at run time; nothing on the GPU box uses it).

The REFERENCE's own TD3 (turtlebot3_rl_sim/src/td3.py: Agent, ReplayBuffer, GaussianExploration, loaded unmodified) trained by the
reference's own loop (start_td3_training.py:104-168, restated below line for line) on this build's environment through the CPU
oracle -- presets.training(drop_cospawned=True) with the reward the published log shows (waypoint_reward 0).  One env, one update of
128 per env-step, sigma = 1 exploration: the published recipe, on CPU.  The question it answers: is the seed sensitivity of the
batched trainer (profiles/r04/train/: about half of the runs end with saturated actor heads) a property of the build, or of the
algorithm as published?

  python oracle/train_reference_td3.py --seed 0 --episodes 1500 --out /tmp/ref_td3_seed0.csv
"""
import argparse
import csv
import importlib.util
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
from crowdnav import presets            # noqa: E402
from oracle import oracle               # noqa: E402


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_td3", "/root/reference/turtlebot3_rl_sim/src/td3.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.device = torch.device("cpu")
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--episodes", type=int, default=1500)
    ap.add_argument("--nsteps", type=int, default=1000)
    ap.add_argument("--waypoint-reward", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(1)
    torch.manual_seed(a.seed); np.random.seed(a.seed); random.seed(a.seed)
    ref = load_ref()
    cfg, init = presets.training(n_envs=1, max_steps=a.nsteps, seed=1000 + a.seed, drop_cospawned=True, waypoint_reward=a.waypoint_reward)
    env = oracle.Oracle(cfg.as_dict()); env.set_ped_init(init)
    oracle.set_num_threads(1)
    # TRAIN:82-99 (the resume branch's values = the ones of the published runs)
    agent = ref.Agent(366 + 4 * 8, 2, 256, 3e-4, 3e-4, 128, 1000000, 0.99, 0.005, 0.22, 2.0, 0.2, 0.5, 2)
    rows = []
    t0 = time.time()
    step_counter = 0
    for ep in range(a.episodes):
        cumulated_reward = 0.0
        state = env.reset()[0]                                            # TRAIN:113-117
        for step in range(a.nsteps):
            step_counter += 1
            state = np.float32(state)
            action = agent.act(state, step, add_noise=True)               # TRAIN:123
            obs, reward, done, _ = env.step(np.asarray(action, dtype=np.float64).reshape(1, 2), step_counter=[step + 1])   # TRAIN:125
            c = env.counters()[0]
            reward, done = float(reward[0]), bool(done[0])
            cumulated_reward += reward
            next_state = np.float32(obs[0])
            agent.memory.add(state, action, reward, next_state, done)     # TRAIN:132-136
            if len(agent.memory) > 128:
                agent.learn(step)
            if not done:
                state = next_state
            else:
                seen = int(c[2])
                rows.append([ep + 1, bool(c[4]), bool(c[5]), cumulated_reward, step + 1,
                             1.0 - c[0] / seen if seen else float("nan"), 1.0 - c[1] / seen if seen else float("nan"), (step + 1) * 0.16])
                break
        if (ep + 1) % 100 == 0:
            last = rows[-100:]
            print("seed %d  episode %5d  env-steps %7d  success(last 100) %.2f  mean return %7.1f  mean steps %5.1f  %.0f s" % (
                a.seed, ep + 1, step_counter, sum(r[1] for r in last) / len(last), sum(r[3] for r in last) / len(last),
                sum(r[4] for r in last) / len(last), time.time() - t0), flush=True)
    last = rows[-500:]
    print("seed %d  last %d episodes: success %.3f  mean return %.1f  mean steps %.1f  (%d env-steps, %.0f s)" % (
        a.seed, len(last), sum(r[1] for r in last) / len(last), sum(r[3] for r in last) / len(last), sum(r[4] for r in last) / len(last),
        step_counter, time.time() - t0), flush=True)
    if a.out:
        with open(a.out, "w", newline="") as fp:
            w = csv.writer(fp)
            w.writerow(['episode_number', 'success_episode', 'failure_episode', 'episode_reward', 'episode_step', 'ego_safety_score',
                        'social_safety_score', 'timelapse'])
            w.writerows(rows)


if __name__ == "__main__":
    main()
