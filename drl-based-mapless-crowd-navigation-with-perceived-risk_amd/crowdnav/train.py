"""Batched counterpart of start_td3_training.py (TRAIN:40-168): TD3 on N environments of one MI355X.

    python -m crowdnav.train --envs 1024 --launches 3000 --out runs/td3
    python -m crowdnav.train --evaluate --load runs/td3 --load-episode 3000 --scenario crossing_8

What it keeps from the reference loop: Agent hyper-parameters (TRAIN:62-72), exploration noise sigma = 1.0 with the
clip to v in [0, 0.22], w in [-2, 2], 1-based per-env step counters, `learn()` only once the replay holds more than a
batch, target-network checkpoints named td3_{actor,critic1,critic2}_model_ep<N>.pt, one CSV row per finished episode
(utils.record_data schema).  What is batched: N envs step per launch with the NEXT-STEP reset convention (the fast kernel,
one observation per wavefront: a finished env spends its next launch on Env.reset, and that launch is not a transition --
it is masked out of the replay; the observation a finished env returns is the terminal one, so it is the transition's
s' as it stands), and `--updates` TD3 updates of `--batch` samples follow each launch (the reference does one update of
128 per single env step)."""
import argparse
import os
import time

import torch

from . import presets
from .config import Config
from .env import VecEnv
from .rollout import EpisodeStats, evaluate
from .td3 import Agent


def make_env(scenario, n_envs, max_steps, seed, device, ped_vmax=None):
    if scenario in ("training", "training_as_logged"):
        # training_as_logged: without obstacles 7-14, which the world file creates at one point (presets.training's docstring)
        cfg, init = presets.training(n_envs=n_envs, max_steps=max_steps, seed=seed, drop_cospawned=scenario == "training_as_logged")
        if ped_vmax is not None:
            cfg.ped_vmax = ped_vmax
        vel = None
    elif scenario == "bench":
        cfg, init, vel = Config(n_envs=n_envs, max_steps=max_steps, seed=seed, ped_cycle_ms=1400), None, None
    else:
        kind, n = scenario.rsplit("_", 1)
        cfg, init, vel = presets.evaluation(kind, int(n), n_envs=n_envs, max_steps=max_steps, seed=seed)
    env = VecEnv(cfg, device=device)
    if init is not None:
        env.set_ped_init(init)
    if vel is not None:
        env.set_ped_preset_vel(vel)
    return env


def train(a):
    dev = a.device
    torch.cuda.set_device(dev)        # policy kernels and torch ops of this process all target the env's GPU
    env = make_env(a.scenario, a.envs, a.max_steps, a.seed, dev, a.ped_vmax)
    agent = Agent(obs_dim=env.D, device="cuda:%d" % dev, seed=a.seed, batch_size=a.batch, memory_size=a.memory)
    if a.load:
        agent.load_models(*[os.path.join(a.load, "td3_%s_model_ep%d.pt" % (n, a.load_episode)) for n in ("actor", "critic1", "critic2")])
        ns = os.path.join(a.load, "noise_state_ep%d.txt" % a.load_episode)
        if os.path.exists(ns):           # continue the exploration-noise stream instead of replaying it
            agent.set_noise_state(*[int(x) for x in open(ns).read().split()])
    if a.graphs:
        agent.enable_graphs()         # a TD3 update as one hipGraph launch (the eager update is launch-bound at batch 128)
    stats = EpisodeStats()
    os.makedirs(a.out, exist_ok=True)
    obs = env.reset()
    t0 = time.time()
    episodes = succ_w = done_w = 0
    ret_w = 0.0
    next_ckpt = a.checkpoint_every
    log = open(os.path.join(a.out, "progress.txt"), "a")
    resetting = torch.zeros(env.N, dtype=torch.bool, device=obs.device)   # envs whose NEXT launch is their Env.reset
    env_steps = 0
    for it in range(1, a.launches + 1):
        act = agent.act_fused(obs, add_noise=True)                     # TD3:196-223, sigma = 1.0, clipped
        prev = obs.clone()
        obs, reward, done = env.step(act, auto_reset="next")
        agent.memory.add_masked(prev, act, reward, obs, done, ~resetting)   # TRAIN:129-131; s' of a finished env = its terminal obs
        env_steps += env.N - int(resetting.sum().item())
        resetting = done.bool().clone()
        if len(agent.memory) > a.batch:
            for u in range(a.updates):
                agent.learn(it * a.updates + u)                          # TRAIN:132-136
        nd = int(done.sum().item())
        if nd:
            c = env.counters(); ret = env.returns()[0]
            idx = torch.nonzero(done).flatten()
            s = c[idx, 4].sum().item()
            episodes += nd; done_w += nd; succ_w += s; ret_w += ret[idx].sum().item()
            if a.csv:
                cc, rr = c.cpu(), ret.cpu()
                for e in idx.cpu().tolist():
                    stats.add_from_counters(cc[e], rr[e].item(), time.time() - t0)
            if episodes >= next_ckpt:                                    # TRAIN:150-154 (every 100 episodes there)
                agent.save(a.out, next_ckpt)
                open(os.path.join(a.out, "noise_state_ep%d.txt" % next_ckpt), "w").write("%d %d\n" % agent.noise_state())
                next_ckpt += a.checkpoint_every
        if it % a.log_every == 0 and done_w:
            line = "launch %6d  env-steps %10d  episodes %8d  success %.3f  mean return %8.1f  replay %8d  %.0f s" % (
                it, env_steps, episodes, succ_w / done_w, ret_w / done_w, len(agent.memory), time.time() - t0)
            print(line, flush=True); log.write(line + "\n"); log.flush()
            succ_w = done_w = 0; ret_w = 0.0
    agent.save(a.out, episodes)
    open(os.path.join(a.out, "noise_state_ep%d.txt" % episodes), "w").write("%d %d\n" % agent.noise_state())
    if a.csv:
        stats.write_csv(a.out, "td3_training")
    return agent, episodes


def run_evaluation(a):
    torch.cuda.set_device(a.device)
    env = make_env(a.scenario, a.envs, a.max_steps, a.seed, a.device, a.ped_vmax)
    agent = Agent(obs_dim=env.D, device="cuda:%d" % a.device, seed=a.seed, memory_size=16)
    agent.load_models(*[os.path.join(a.load, "td3_%s_model_ep%d.pt" % (n, a.load_episode)) for n in ("actor", "critic1", "critic2")])
    st = evaluate(env, agent, episodes_per_env=a.episodes_per_env)
    n = len(st.rows)
    print("%s: %d episodes, success %.3f, failure %.3f, mean steps %.1f, ego %.3f, social %.3f" % (
        a.scenario, n, sum(r[1] for r in st.rows) / n, sum(r[2] for r in st.rows) / n, sum(r[4] for r in st.rows) / n,
        sum(r[5] for r in st.rows if r[5] == r[5]) / max(1, sum(1 for r in st.rows if r[5] == r[5])),
        sum(r[6] for r in st.rows if r[6] == r[6]) / max(1, sum(1 for r in st.rows if r[6] == r[6]))))
    if a.out:
        print("wrote", st.write_csv(a.out, "td3_training_test_" + a.scenario))
    return st


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--scenario", default="training", help="training | training_as_logged | bench | {crossing,towards,ahead,random}_{4,8,12,20}")
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--launches", type=int, default=3000)
    ap.add_argument("--max-steps", type=int, default=1000, help="nsteps (configs/td3.yaml)")
    ap.add_argument("--updates", type=int, default=4, help="TD3 updates per launch")
    ap.add_argument("--batch", type=int, default=128, help="TRAIN:62")
    ap.add_argument("--memory", type=int, default=1_000_000, help="TRAIN:63")
    ap.add_argument("--checkpoint-every", type=int, default=100000, help="episodes between checkpoints (TRAIN:150: 100)")
    ap.add_argument("--log-every", type=int, default=100)
    ap.add_argument("--ped-vmax", type=float, default=None, help="training world only: walker speed bound (CROWD:101 -> 0.2)")
    ap.add_argument("--graphs", type=int, default=1, help="1: capture the TD3 update into hipGraphs (Agent.enable_graphs)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default="runs/td3")
    ap.add_argument("--csv", action="store_true", help="one CSV row per finished episode (costs a host sync per launch)")
    ap.add_argument("--load", default=None)
    ap.add_argument("--load-episode", type=int, default=0)
    ap.add_argument("--evaluate", action="store_true")
    ap.add_argument("--episodes-per-env", type=int, default=1)
    a = ap.parse_args(argv)
    if a.evaluate:
        return run_evaluation(a)
    return train(a)


if __name__ == "__main__":
    main()
