#!/usr/bin/env python
"""TEST INFRASTRUCTURE, container-only (reads /root/reference at run time; nothing on the GPU box imports it).

End-to-end sanity check of the observation / action conventions: the reference's own TRAINED TD3 actors
(models/td3/**/td3_actor_model_ep*.pt, trained in Gazebo) drive the build's simulator through the CPU oracle,
greedy (no exploration noise), in the training world and in the README's scripted evaluation scenarios.  A policy
that was trained against the reference Env only reaches goals here if heading/distance signs, scan ordering, top-K
feature layout and the (v, w) action semantics all agree with what it was trained on.

The published training logs (results/td3/**/td3_training.csv) hold training-time episodes, i.e. with the exploration
noise sigma = 1.0 of TD3:67-78, so the comparable rows here are the sigma = 1 ones.

  python oracle/eval_reference_policy.py [--envs 128] [--episodes 1]      -> table on stdout
"""
import argparse
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
from crowdnav import presets            # noqa: E402  (data + Config only; no GPU needed)
from crowdnav.td3 import Actor          # noqa: E402
from oracle import oracle               # noqa: E402

MODELS = "/root/reference/turtlebot3_rl_sim/src/models/td3"


def run(actor, cfg, init, vel, episodes, policy="actor", sigma=0.0):
    """Greedy (sigma = 0) or training-time (sigma = 1.0, TD3:67-78 + clip TD3:214-215) roll-outs; returns the five columns of the
    reference's training CSV (TRAIN:139-160): success rate, mean episode return (sum of rewards, TRAIN:127), mean steps, mean ego /
    social safety score (ENV:1269-1283) -- plus how often ENV:1116's way-point bonus was paid."""
    o = oracle.Oracle(cfg.as_dict() if hasattr(cfg, "as_dict") else cfg)
    if init is not None:
        o.set_ped_init(init)
    if vel is not None:
        o.set_ped_preset_vel(vel)
    obs = o.reset()
    N = o.N
    bonus = int(o.cfg.waypoint_reward)
    succ = fail = 0; steps = []; ego = []; soc = []; rets = []; nb = []
    ep_done = np.zeros(N, dtype=np.int64)
    ret = np.zeros(N); nbonus = np.zeros(N)
    rng = np.random.RandomState(0)
    while ep_done.min() < episodes:
        if policy == "actor":
            with torch.no_grad():
                a = actor(torch.from_numpy(obs.astype(np.float32))).numpy().astype(np.float64)
            if sigma:
                a = a + rng.normal(0.0, sigma, a.shape)
            a[:, 0] = np.clip(a[:, 0], 0.0, 0.22); a[:, 1] = np.clip(a[:, 1], -2.0, 2.0)
        else:
            a = np.stack([rng.uniform(0, 0.22, N), rng.uniform(-2, 2, N)], 1)
        pre = o.counters().copy()
        obs, rew, done, _ = o.step(a, auto_reset=True)
        ret += rew
        if bonus:      # a step's reward is -2 + dtg + htg (+ bonus) (+- 200 at the end): the bonus is what lifts it above that range
            nbonus += (rew - np.where(done != 0, np.where(rew > 0, 200.0, -200.0), 0.0)) > 2.5
        if done.any():
            c = o.counters()
            for e in np.nonzero(done)[0]:
                if ep_done[e] < episodes:
                    ep_done[e] += 1
                    succ += int(c[e, 4]); fail += int(c[e, 5])
                    steps.append(int(pre[e, 3]) + 1); rets.append(ret[e]); nb.append(nbonus[e])
                    if pre[e, 2] > 0:
                        ego.append(1.0 - pre[e, 0] / pre[e, 2]); soc.append(1.0 - pre[e, 1] / pre[e, 2])
                ret[e] = 0.0; nbonus[e] = 0.0
    n = int(ep_done.sum())
    return dict(n=n, success=succ / n, failure=fail / n, steps=float(np.mean(steps)), ret=float(np.mean(rets)),
                ret_max=float(np.max(rets)), bonuses=float(np.mean(nb)),
                ego=float(np.mean(ego)) if ego else float("nan"), social=float(np.mean(soc)) if soc else float("nan"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=128)
    ap.add_argument("--episodes", type=int, default=1)
    ap.add_argument("--max-steps", type=int, default=1000)
    ap.add_argument("--switches", action="store_true", help="only the five-column table of the round-4 switches")
    a = ap.parse_args()
    oracle.set_num_threads(min(16, os.cpu_count() or 1))
    actors = {}
    for k in (1, 4, 8, 12, 16):        # the K-ablation checkpoints of the 366 + 4K layout
        f = os.path.join(MODELS, "turtlebot3_top_%d_obstacle" % k, "td3_actor_model_ep2500.pt")
        sd = torch.load(f, map_location="cpu")
        assert sd["linear1.weight"].shape[1] == 366 + 4 * k
        m = Actor(366 + 4 * k, 2, 256); m.load_state_dict(sd); m.eval()
        actors[k] = m
    print("failure = collision or step limit (the reference counts both as failure_episode, ENV:1131-1158)")
    print("%-98s %5s | %7s %8s %6s %6s %6s | %7s %7s" % ("policy / world", "eps", "success", "return", "steps", "ego", "social", "max ret", "bonuses"))
    print("%-98s %5d | %7.3f %8.1f %6.1f %6.3f %6.3f | %7.0f %7s" % (
        "PUBLISHED LOG results/td3/revamped/new_tracking_cp_gcp_nobonus_corrected_3/td3_training.csv, last 500 episodes", 500,
        0.608, -9.2, 94.8, 0.991, 0.904, 173, "0"))

    def show(name, r):
        print("%-98s %5d | %7.3f %8.1f %6.1f %6.3f %6.3f | %7.0f %7.2f" % (name, r["n"], r["success"], r["ret"], r["steps"], r["ego"], r["social"],
                                                                       r["ret_max"], r["bonuses"]), flush=True)

    # the five columns of the published log against this simulator, switch by switch (cn_config.waypoint_reward / scan_f32 / wheel_accel)
    if a.switches:
        for drop, label in ((True, "obstacles 1-6"), (False, "all 14 obstacles")):
            for sw, sname in ((dict(), "as committed (bonus 200, float64 scan, kinematic)"),
                              (dict(waypoint_reward=0), "waypoint_reward 0"),
                              (dict(waypoint_reward=0, scan_f32=1), "waypoint_reward 0 + scan_f32"),
                              (dict(waypoint_reward=0, wheel_accel=1.0), "waypoint_reward 0 + wheel_accel 1"),
                              (dict(waypoint_reward=0, scan_f32=1, wheel_accel=1.0), "waypoint_reward 0 + scan_f32 + wheel_accel 1")):
                cfg, init = presets.training(n_envs=a.envs, max_steps=a.max_steps, seed=77, k_obstacles=8, drop_cospawned=drop, **sw)
                show("top_8 ep2500, sigma 1 / training world, %s: %s" % (label, sname), run(actors[8], cfg, init, None, a.episodes, sigma=1.0))
        return
    def world(k, vmax=None):
        cfg, init = presets.training(n_envs=a.envs, max_steps=a.max_steps, seed=77, k_obstacles=k)
        if vmax is not None:
            cfg.ped_vmax = vmax
        return cfg, init

    cfg, init = world(8)
    show("uniform-random actions / training world, 14 walkers U(-0.2, 0.2) m/s", run(None, cfg, init, None, a.episodes, policy="random"))
    for k, m in actors.items():
        for sigma in (1.0, 0.0):
            cfg, init = world(k)
            show("top_%d_obstacle ep2500, sigma %.0f / training world, walkers U(-0.2, 0.2) m/s" % (k, sigma), run(m, cfg, init, None, a.episodes, sigma=sigma))
    # how many of the 14 walkers are really in the room?  (obstacles 7-14 are created at one point: presets.training's docstring)
    for npeds in (10, 8, 7, 6, 4):
        cfg, init = world(8)
        cfg.n_peds = npeds
        show("top_8_obstacle ep2500, sigma 1 / training world, only obstacles 1-%d in the room (round still 1.4 s)" % npeds,
             run(actors[8], cfg, init[:, :npeds].copy(), None, a.episodes, sigma=1.0))
    for vmax in (0.1, 0.03, 0.0):
        for k in (1, 8):
            cfg, init = world(k, vmax)
            show("top_%d_obstacle ep2500, sigma 1 / training world, walkers U(-%.2f, %.2f) m/s" % (k, vmax, vmax), run(actors[k], cfg, init, None, a.episodes, sigma=1.0))
    for kind in ("crossing", "towards", "ahead", "random"):
        for n in (4, 8, 12, 20):
            cfg, init, vel = presets.evaluation(kind, n, n_envs=a.envs, max_steps=a.max_steps, seed=77, k_obstacles=8)
            show("top_8_obstacle ep2500, sigma 1 / test room, %s x %d" % (kind, n), run(actors[8], cfg, init, vel, a.episodes, sigma=1.0))


if __name__ == "__main__":
    main()
