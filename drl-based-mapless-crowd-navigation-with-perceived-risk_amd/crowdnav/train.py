"""Batched counterpart of start_td3_training.py (TRAIN:40-168): TD3 on N environments of one MI355X.

    python -m crowdnav.train --scenario training_as_logged --waypoint-reward 0 --envs 16 --updates 16 --launches 25000 --csv --out runs/td3
    python -m crowdnav.train --evaluate --load runs/td3 --load-episode latest --scenario crossing_8

Checkpoints are labelled with the episode count the weights really have behind them (N envs finish episodes in batches, so the
count at a log interval is rarely a round number); every save also rewrites `latest_checkpoint.txt` with that count, and
`--load-episode latest` (the default) reads it -- resume / evaluate commands can be written before the run exists.

What it keeps from the reference loop: Agent hyper-parameters (TRAIN:62-72), exploration noise sigma = 1.0 with the
clip to v in [0, 0.22], w in [-2, 2], 1-based per-env step counters, `learn()` only once the replay holds more than a
batch, target-network checkpoints named td3_{actor,critic1,critic2}_model_ep<N>.pt, one CSV row per finished episode
(utils.record_data schema).  What is batched: N envs step per launch and `--updates` TD3 updates of `--batch` samples
follow each launch (the reference does one update of 128 per single env step: --envs E --updates E keeps its ratio).

The loop enqueues only -- no host synchronisation per launch: the policy is the whole actor as ONE kernel (cn_actor_forward on
the weights packed after the launch's updates), the env step is the fast kernel (next-step reset: a finished env spends its
next launch on Env.reset, and that launch is not a transition -- it is masked out of the replay on the device; the
observation a finished env returns is the terminal one, so it is the transition's s' as it stands), episode statistics and the
CSV rows accumulate in device tensors and are read once per `--log-every` launches.  `--reset-mode same` keeps the older
path (same-call reset + final_obs) for A/B runs."""
import argparse
import os
import time

import torch

from . import presets
from .config import Config
from .env import VecEnv
from .rollout import EpisodeStats, evaluate
from .td3 import Agent


def make_env(scenario, n_envs, max_steps, seed, device, ped_vmax=None, **switches):
    """switches: cn_config fields applied on top of the scenario (waypoint_reward, scan_f32, wheel_accel, ...)."""
    sw = {k: v for k, v in switches.items() if v is not None}
    if scenario in ("training", "training_as_logged"):
        # training_as_logged: without obstacles 7-14, which the world file creates at one point (presets.training's docstring)
        cfg, init = presets.training(n_envs=n_envs, max_steps=max_steps, seed=seed, drop_cospawned=scenario == "training_as_logged", **sw)
        if ped_vmax is not None:
            cfg.ped_vmax = ped_vmax
        vel = None
    elif scenario == "bench":
        cfg, init, vel = Config(n_envs=n_envs, max_steps=max_steps, seed=seed, ped_cycle_ms=1400, **sw), None, None
    else:
        kind, n = scenario.rsplit("_", 1)
        cfg, init, vel = presets.evaluation(kind, int(n), n_envs=n_envs, max_steps=max_steps, seed=seed, **sw)
    env = VecEnv(cfg, device=device)
    if init is not None:
        env.set_ped_init(init)
    if vel is not None:
        env.set_ped_preset_vel(vel)
    return env


class DeviceEpisodeLog:
    """Finished episodes, recorded on the device: running totals for the progress line and one row per episode for the CSV
    (success, failure, return, steps, ego / social violations, obstacle-present steps, launch index) -- appended with a
    cumulative-sum scatter, rows of envs that did not finish go to a spare row.  One host read per flush().  On a HIP device add()
    is libcrowdnav's cn_episode_log_add (one launch instead of ~20 PyTorch kernels); `fused=False` keeps the PyTorch formulation."""

    def __init__(self, device, max_rows, fused=True):
        self.max_rows = int(max_rows)
        self.fused = bool(fused) and torch.device(device).type == "cuda"
        self._log = None
        self.rows = torch.zeros((self.max_rows + 1, 8), dtype=torch.float32, device=device)
        self.n = torch.zeros((), dtype=torch.int64, device=device)
        self.tot = torch.zeros(5, dtype=torch.float64, device=device)     # episodes, successes, return sum, step sum, env-steps
        self._flushed = 0

    def _add_fused(self, done, counters, last_return, launch, transitions):
        import ctypes as C
        from . import _abi
        L = _abi.lib()
        dev = self.rows.device
        if self._log is None:
            self._log = _abi.CnEpisodeLog(rows=self.rows.data_ptr(), max_rows=self.max_rows, n_dev=self.n.data_ptr(), tot_dev=self.tot.data_ptr())
        n = done.shape[0]
        d8 = done.contiguous() if done.dtype in (torch.uint8, torch.bool) else (done != 0)
        t8 = transitions.contiguous() if transitions.dtype in (torch.uint8, torch.bool) else (transitions != 0)
        cnt = counters if counters.dtype == torch.int32 and counters.is_contiguous() else counters.to(torch.int32).contiguous()
        ret = last_return if last_return.dtype == torch.float32 and last_return.is_contiguous() else last_return.float().contiguous()
        rc = L.cn_episode_log_add(C.byref(self._log), C.c_void_p(d8.data_ptr()), C.c_void_p(cnt.data_ptr()), cnt.shape[1], C.c_void_p(ret.data_ptr()),
                                  C.c_void_p(t8.data_ptr()), float(launch), n, dev.index if dev.index is not None else torch.cuda.current_device(),
                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise _abi.CrowdNavError("cn_episode_log_add: %s" % L.cn_td3_last_error().decode())

    def add(self, done, counters, last_return, launch, transitions):
        if self.fused:
            return self._add_fused(done, counters, last_return, launch, transitions)
        d = done.bool()
        k = d.to(torch.int64)
        c = torch.cumsum(k, 0)
        idx = torch.where(d, torch.clamp(self.n + c - 1, max=self.max_rows), torch.full_like(c, self.max_rows))
        cf = counters.to(torch.float32)
        row = torch.stack([cf[:, 4], cf[:, 5], last_return, cf[:, 13], cf[:, 10], cf[:, 11], cf[:, 12],
                           torch.full_like(last_return, float(launch))], 1)
        self.rows.index_copy_(0, idx, row)
        self.n.add_(c[-1])
        df = d.to(torch.float64)
        self.tot.add_(torch.stack([df.sum(), (cf[:, 4].double() * df).sum(), (last_return.double() * df).sum(),
                                   (cf[:, 13].double() * df).sum(), transitions.sum().double()]))

    def flush(self):
        """-> (new rows as a CPU tensor, totals since the previous flush as a list); one synchronisation."""
        n = min(int(self.n.item()), self.max_rows)
        new = self.rows[self._flushed:n].cpu()
        self._flushed = n
        tot = self.tot.cpu().tolist()
        self.tot.zero_()
        return new, tot


def resolve_load_episode(load_dir, episode):
    """--load-episode: an integer, or "latest" = the count in <load_dir>/latest_checkpoint.txt (save_checkpoint writes it)."""
    if isinstance(episode, str) and episode.strip().lower() == "latest":
        path = os.path.join(load_dir, "latest_checkpoint.txt")
        if not os.path.exists(path):
            raise FileNotFoundError("--load-episode latest: %s does not exist (no checkpoint was saved into %s)" % (path, load_dir))
        return int(open(path).read().split()[0])
    return int(episode)


def save_checkpoint(agent, outdir, episodes):
    """TRAIN:150-154's checkpoint + the exploration-noise stream's position + the `latest` pointer."""
    agent.save(outdir, episodes)
    open(os.path.join(outdir, "noise_state_ep%d.txt" % episodes), "w").write("%d %d\n" % agent.noise_state())
    tmp = os.path.join(outdir, ".latest_checkpoint.txt.%d" % os.getpid())
    open(tmp, "w").write("%d\n" % episodes)
    os.replace(tmp, os.path.join(outdir, "latest_checkpoint.txt"))


def train(a):
    dev = a.device
    torch.cuda.set_device(dev)        # policy kernels and torch ops of this process all target the env's GPU
    env = make_env(a.scenario, a.envs, a.max_steps, a.seed, dev, a.ped_vmax, waypoint_reward=a.waypoint_reward,
                   scan_f32=a.scan_f32, wheel_accel=a.wheel_accel)
    agent = Agent(obs_dim=env.D, device="cuda:%d" % dev, seed=a.seed, batch_size=a.batch, memory_size=a.memory,
                  actor_final_init=getattr(a, "actor_final_init", None))
    if a.load:
        a.load_episode = resolve_load_episode(a.load, a.load_episode)
        agent.load_models(*[os.path.join(a.load, "td3_%s_model_ep%d.pt" % (n, a.load_episode)) for n in ("actor", "critic1", "critic2")])
        ns = os.path.join(a.load, "noise_state_ep%d.txt" % a.load_episode)
        if os.path.exists(ns):           # continue the exploration-noise stream instead of replaying it
            agent.set_noise_state(*[int(x) for x in open(ns).read().split()])
    if a.learner == "fused":
        agent.enable_fused_update()   # cn_td3_update: the update as 7 (+ 5) hand-written launches
    elif a.graphs:
        agent.enable_graphs()         # a TD3 update as one hipGraph launch (the eager update is launch-bound at batch 128)
    stats = EpisodeStats()
    os.makedirs(a.out, exist_ok=True)
    # a run continued into the directory it was loaded from appends to that run's CSV (as progress.txt always did)
    resumed = bool(a.load) and os.path.abspath(a.load) == os.path.abspath(a.out)
    obs = env.reset()
    t0 = time.time()
    episodes = 0
    env_steps = 0
    next_ckpt = a.checkpoint_every
    log = open(os.path.join(a.out, "progress.txt"), "a")
    N = env.N
    same = a.reset_mode == "same"
    resetting = torch.zeros(N, dtype=torch.bool, device=obs.device)   # envs whose NEXT launch is their Env.reset
    all_rows = torch.ones(N, dtype=torch.bool, device=obs.device)
    prev = torch.empty_like(obs)
    elog = DeviceEpisodeLog(obs.device, a.max_csv_rows)
    learning = False
    updates_done = 0
    agent.sync_fused_weights()
    step_s = (env.cfg.dt_ms + env.cfg.scan_latency_ms) / 1000.0
    win = []                                                           # (successes, episodes) of the recent log windows
    warned_overflow = False
    for it in range(1, a.launches + 1):
        act = agent.act_mfma(obs, add_noise=True)                      # TD3:196-223 as one kernel, sigma = 1.0, clipped
        prev.copy_(obs)
        if same:
            obs, reward, done = env.step(act, auto_reset="same", want_final=True)
            agent.memory.add_masked(prev, act, reward, env.final_obs, done, all_rows)
            keep = all_rows
        else:
            obs, reward, done = env.step(act, auto_reset="next")
            keep = ~resetting
            agent.memory.add_masked(prev, act, reward, obs, done, keep)    # TRAIN:129-131; s' of a finished env = its terminal obs
            resetting = done.bool()
        elog.add(done, env.counters(), env.returns()[0], it, keep)
        if not learning:                                               # TRAIN:132: only once the replay holds more than a batch
            learning = agent.memory.ready(a.batch)                     # (a host read only while the bounds straddle it; none afterwards)
        if learning:
            for u in range(a.updates):
                updates_done += 1
                agent.learn(updates_done)                              # TRAIN:133-136
            agent.sync_fused_weights()                                 # the actor the next launch acts with
        last_launch = it == a.launches or (a.time_limit and it % a.log_every == 0 and time.time() - t0 > a.time_limit)
        if it % a.log_every == 0 or last_launch:
            rows, tot = elog.flush()
            ne = int(tot[0])
            episodes += ne; env_steps += int(tot[4])
            for r in rows.tolist():
                seen = int(r[6])
                stats.add(int(r[0]), int(r[1]), r[2], int(r[3]), 1.0 - r[4] / seen if seen else float("nan"),
                          1.0 - r[5] / seen if seen else float("nan"), int(r[3]) * step_s)
            if ne:
                win.append((tot[1], ne))
                line = "launch %6d  env-steps %10d  updates %9d  episodes %8d  success %.3f  mean return %8.1f  mean steps %6.1f  %.0f s" % (
                    it, env_steps, updates_done, episodes, tot[1] / ne, tot[2] / ne, tot[3] / ne, time.time() - t0)
                print(line, flush=True); log.write(line + "\n"); log.flush()
            if not warned_overflow:
                sc_ = env.status_counts()
                if sc_["track_overflow"] or sc_["conf_overflow"]:
                    warned_overflow = True
                    line = ("WARNING: %d env(s) outgrew the track table and %d the confirmed-object table (status bits CN_ST_TRACK_OVERFLOW / "
                            "CN_ST_CONF_OVERFLOW): their risk features use the tracks that fit and differ from the reference's unbounded "
                            "lists from there on; Config(track_capacity=64) doubles the table" % (sc_["track_overflow"], sc_["conf_overflow"]))
                    print(line, flush=True); log.write(line + "\n"); log.flush()
            if a.csv:
                stats.append_csv(a.out, "td3_training", resume=resumed)   # incremental: a killed run keeps its rows up to here
            if episodes >= next_ckpt:                                    # TRAIN:150-154 (every 100 episodes there)
                # labelled with the episode count the weights really have behind them (checked at log time, so it can be past
                # the threshold that triggered it)
                save_checkpoint(agent, a.out, episodes)
                while next_ckpt <= episodes:
                    next_ckpt += a.checkpoint_every
            if last_launch:
                break
    agent.memory.sync_len()
    save_checkpoint(agent, a.out, episodes)
    last = stats.rows[-500:]
    if last:
        line = "last %d episodes: success %.3f  mean return %.1f  mean steps %.1f  ego %.3f  social %.3f  | %d updates, %.0f updates/s, %.0f env-steps/s overall" % (
            len(last), sum(r[1] for r in last) / len(last), sum(r[3] for r in last) / len(last), sum(r[4] for r in last) / len(last),
            sum(r[5] for r in last if r[5] == r[5]) / max(1, sum(1 for r in last if r[5] == r[5])),
            sum(r[6] for r in last if r[6] == r[6]) / max(1, sum(1 for r in last if r[6] == r[6])),
            updates_done, updates_done / max(1e-9, time.time() - t0), env_steps / max(1e-9, time.time() - t0))
        print(line, flush=True); log.write(line + "\n"); log.flush()
    if a.csv:
        stats.append_csv(a.out, "td3_training", resume=resumed)
    return agent, episodes


def run_evaluation(a):
    torch.cuda.set_device(a.device)
    env = make_env(a.scenario, a.envs, a.max_steps, a.seed, a.device, a.ped_vmax, waypoint_reward=a.waypoint_reward,
                   scan_f32=a.scan_f32, wheel_accel=a.wheel_accel)
    agent = Agent(obs_dim=env.D, device="cuda:%d" % a.device, seed=a.seed, memory_size=16)
    a.load_episode = resolve_load_episode(a.load, a.load_episode)
    agent.load_models(*[os.path.join(a.load, "td3_%s_model_ep%d.pt" % (n, a.load_episode)) for n in ("actor", "critic1", "critic2")])
    st = evaluate(env, agent, episodes_per_env=a.episodes_per_env)
    n = len(st.rows)
    print("%s: %d episodes, success %.3f, failure %.3f, mean return %.1f, mean steps %.1f, ego %.3f, social %.3f" % (
        a.scenario, n, sum(r[1] for r in st.rows) / n, sum(r[2] for r in st.rows) / n, sum(r[3] for r in st.rows) / n, sum(r[4] for r in st.rows) / n,
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
    ap.add_argument("--time-limit", type=float, default=0.0, help="stop after this many seconds (checked at log time); 0 = run all launches")
    ap.add_argument("--max-steps", type=int, default=1000, help="nsteps (configs/td3.yaml)")
    ap.add_argument("--updates", type=int, default=4, help="TD3 updates per launch")
    ap.add_argument("--batch", type=int, default=128, help="TRAIN:62")
    ap.add_argument("--memory", type=int, default=1_000_000, help="TRAIN:63")
    ap.add_argument("--checkpoint-every", type=int, default=100000, help="episodes between checkpoints (TRAIN:150: 100); checked at log time")
    ap.add_argument("--log-every", type=int, default=100)
    ap.add_argument("--ped-vmax", type=float, default=None, help="training world only: walker speed bound (CROWD:101 -> 0.2)")
    ap.add_argument("--waypoint-reward", type=int, default=None, help="cn_config.waypoint_reward: ENV:1116's 200 (default) or 0 = the published log's reward")
    ap.add_argument("--scan-f32", type=int, default=None, help="cn_config.scan_f32")
    ap.add_argument("--wheel-accel", type=float, default=None, help="cn_config.wheel_accel (XACRO:70: 1.0)")
    ap.add_argument("--reset-mode", default="next", choices=["next", "same"], help="next: the fast kernel, reset launches masked out of the replay; same: same-call reset + final_obs")
    ap.add_argument("--graphs", type=int, default=1, help="1: capture the TD3 update into hipGraphs (Agent.enable_graphs)")
    ap.add_argument("--learner", default="torch", choices=["torch", "fused"], help="torch: the PyTorch update (eager / hipGraph); fused: cn_td3_update (csrc/crowdnav_td3.hip)")
    ap.add_argument("--actor-final-init", type=float, default=None, help="NOT the reference: U(+-x) initialisation of the actor's output layer (e.g. 0.003)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default="runs/td3")
    ap.add_argument("--csv", action="store_true", help="one CSV row per finished episode in the reference's 8-column schema (recorded on the device, appended to the file at "
                    "every log interval); `timelapse` = the episode's own virtual duration, steps x (0.15 s + scan wait) -- TRAIN:141 "
                    "measures wall time since the episode's start, which the reference's time.sleep(0.15) makes the same quantity")
    ap.add_argument("--max-csv-rows", type=int, default=2_000_000)
    ap.add_argument("--load", default=None)
    ap.add_argument("--load-episode", default="latest", help="the <N> of td3_*_model_ep<N>.pt, or `latest` = the count in <load>/latest_checkpoint.txt")
    ap.add_argument("--evaluate", action="store_true")
    ap.add_argument("--episodes-per-env", type=int, default=1)
    a = ap.parse_args(argv)
    if a.evaluate:
        return run_evaluation(a)
    return train(a)


if __name__ == "__main__":
    main()
