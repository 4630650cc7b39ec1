"""Batched counterpart of the reference's training loop (start_td3_training.py:104-168).

The reference drives ONE env: reset() -> done=False -> for step: act -> env.step(a, step+1, "continuous")
-> accumulate reward -> on done: scores, CSV row.  Here N envs advance per launch; each env keeps its own
1-based step counter and return on the device, a finished env spends its next launch on Env.reset
(cn_step auto_reset 2, the fast kernel), and episode statistics come back as tensors.  Across GPUs the envs shard by global
index with no data-path collective; the one exchange is the all-gather of per-env episode returns."""
import csv
import os

import torch


def shard_range(n_total, rank, world):
    """Contiguous shard of global env indices owned by `rank`."""
    per = n_total // world
    assert per * world == n_total, "envs must divide evenly across ranks"
    return rank * per, per


def gather_returns(local_returns):
    """All-gather of per-env episode returns across ranks (RCCL over xGMI on GPUs, gloo on CPU).
    Identity when torch.distributed is not initialised."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_returns
    world = dist.get_world_size()
    out = torch.empty(world * local_returns.numel(), dtype=local_returns.dtype, device=local_returns.device)
    dist.all_gather_into_tensor(out, local_returns.contiguous())
    return out


class EpisodeStats:
    """Per-episode rows in the reference's CSV schema (utils.py:53-64)."""
    HEADERS = ['episode_number', 'success_episode', 'failure_episode', 'episode_reward', 'episode_step',
               'ego_safety_score', 'social_safety_score', 'timelapse']

    def __init__(self):
        self.rows = []

    def add(self, success, failure, reward, steps, ego, social, timelapse=0.0):
        self.rows.append([len(self.rows) + 1, bool(success), bool(failure), float(reward), int(steps), float(ego),
                          float(social), float(timelapse)])

    def add_from_counters(self, c, episode_return, timelapse=None, step_seconds=0.16):
        """One row from an env's cn_get_counters record taken after the launch in which it finished.  timelapse: TRAIN:141's
        `time.time() - start_time`; by default the episode's own (virtual) duration, steps x (time.sleep(0.15) + the /scan
        wait) -- what the reference measures when it is not slowed down by its host."""
        if timelapse is None:
            timelapse = int(c[13]) * step_seconds
        seen = int(c[12])
        ego = 1.0 - int(c[10]) * 1.0 / seen if seen else float("nan")      # ENV:1277-1283 (ZeroDivisionError there)
        soc = 1.0 - int(c[11]) * 1.0 / seen if seen else float("nan")      # ENV:1269-1275
        self.add(int(c[4]), int(c[5]), episode_return, int(c[13]), ego, soc, timelapse)

    def write_csv(self, outdir, filename):
        os.makedirs(outdir, exist_ok=True)
        path = os.path.join(outdir, filename + ".csv")
        with open(path, "w", newline="") as fp:
            w = csv.writer(fp, dialect="excel")
            w.writerow(self.HEADERS)
            w.writerows(self.rows)
        self._flushed = (path, len(self.rows))
        return path

    def append_csv(self, outdir, filename, resume=False):
        """Incremental write_csv: the header the first time, then only the rows added since the last call -- so a run that is
        killed keeps every episode up to its last log interval (the reference appends one row per episode, UTL:53-64).
        resume=True (a run continued with --load into the same --out): an existing file is kept and appended to, and this
        run's episode numbers continue after its last row -- utils.record_data never truncates either.  A fresh start
        (resume=False) truncates."""
        path = os.path.join(outdir, filename + ".csv")
        done = getattr(self, "_flushed", (None, 0))
        if done[0] != path:
            os.makedirs(outdir, exist_ok=True)
            self._episode_base = 0
            if resume and os.path.exists(path) and os.path.getsize(path) > 0:
                with open(path, newline="") as fp:
                    prior = [r for r in csv.reader(fp, dialect="excel") if r]
                if prior and prior[0] == self.HEADERS:
                    prior = prior[1:]
                try:
                    self._episode_base = int(prior[-1][0]) if prior else 0
                except ValueError:
                    self._episode_base = len(prior)
            else:
                with open(path, "w", newline="") as fp:
                    csv.writer(fp, dialect="excel").writerow(self.HEADERS)
            done = (path, 0)
        if len(self.rows) > done[1]:
            base = getattr(self, "_episode_base", 0)
            with open(path, "a", newline="") as fp:
                csv.writer(fp, dialect="excel").writerows([[r[0] + base] + r[1:] for r in self.rows[done[1]:]])
        self._flushed = (path, len(self.rows))
        return path


def rollout(env, agent, n_steps, learn=False, add_noise=True, stats=None, policy=None, auto_reset="next"):
    """Actor-in-the-loop rollout (BASELINE config 3).  Returns total env-steps taken.
    `env` is a crowdnav.env.VecEnv; `agent` a crowdnav.td3.Agent on the same device.
    policy: "mfma" = the whole actor as one libcrowdnav kernel (cn_actor_forward; default when the weights are
    static), "tail" = PyTorch GEMMs + fused output stage (cn_policy_tail; default while learning, since the
    weights change every update), "torch" = plain PyTorch; or a callable `policy(obs, t) -> [N, 2] float32 device
    tensor` (scripted / replayed actions; `agent` may then be None).
    auto_reset: "next" (default; the fast kernel: a finished env's next launch is its Env.reset, its action is ignored and the
    launch is neither a transition nor an episode row) or "same" (the reset inside the finishing launch, s' from final_obs)."""
    if policy is None:
        policy = "tail" if learn else "mfma"
    if callable(policy):
        act_fn = None
    else:
        act_fn = {"mfma": agent.act_mfma, "tail": agent.act_fused, "torch": agent.act}[policy]
    if policy == "mfma":
        agent.sync_fused_weights()
    obs = env.obs if getattr(env, "_started", False) else env.reset()
    env._started = True
    # next-step reset convention (the fast kernel): a finished env's next launch is its Env.reset -- not a transition, not an
    # episode row; the observation it returned with done = 1 is the terminal one
    resetting = getattr(env, "_resetting", None)
    if resetting is None:
        resetting = torch.zeros(env.N, dtype=torch.bool, device=obs.device)
    for t in range(n_steps):
        act = policy(obs, t) if act_fn is None else act_fn(obs, add_noise=add_noise)
        if learn:
            prev = obs.clone()
        if auto_reset == "next":
            obs, reward, done = env.step(act, auto_reset="next")
            if learn:
                agent.memory.add_masked(prev, act, reward, obs, done, ~resetting)
            resetting = done.bool()
        else:
            obs, reward, done = env.step(act, auto_reset="same", want_final=learn)
            if learn:
                agent.memory.add_masked(prev, act, reward, env.final_obs, done, torch.ones_like(resetting))
        if learn:
            agent.learn(t)            # gates itself on DeviceReplay.ready(batch): no host read once the ring holds a batch
        if stats is not None and bool(done.any()):
            # columns 10..13 keep the finished episode's counters as they stood when Env.step returned done (what TRAIN:142-147
            # reads), terminal step included
            c = env.counters().cpu()
            ret, _ = env.returns()
            ret = ret.cpu()
            for e in torch.nonzero(done.cpu()).flatten().tolist():
                stats.add_from_counters(c[e], ret[e].item())
    env._resetting = resetting
    return n_steps * env.N


def collect_policy(env, agent, n_steps, periods=16, add_noise=True, store=True):
    """Data collection with the policy INSIDE the step kernel (cn_rollout_policy): `periods` control periods per launch, the
    actor frozen at the weights of `agent.sync_fused_weights()` for the whole call (TRAIN:104-168's act -> step -> memory.add
    with a policy lag of at most `periods` periods).  store: the transitions go to agent.memory in the order the per-step loop
    `rollout(env, agent, ..., policy="mfma")` would add them -- period-major, env order, an env's reset launch skipped -- by one
    masked add per launch (no host read).  Enqueues only; returns env-steps issued (N per period)."""
    agent.sync_fused_weights()
    if not getattr(env, "_started", False):
        env.reset()
        env._started = True
    N, D = env.N, env.D
    resetting = getattr(env, "_resetting", None)
    if resetting is None:
        resetting = torch.zeros(N, dtype=torch.bool, device=env.device)
    left = int(n_steps)
    while left > 0:
        T = min(int(periods), left)
        if not store:
            env.rollout_policy(agent, T, add_noise=add_noise)
            resetting = env.done.bool()
        else:
            assert T * N <= agent.memory.cap, "one launch's transitions must fit the replay ring"
            bufs = getattr(env, "_pol_traj", None)
            if bufs is None or bufs["obs"].shape[0] != T + 1:
                bufs = dict(obs=torch.zeros((T + 1, N, D), dtype=torch.float32, device=env.device),
                            action=torch.zeros((T, N, 2), dtype=torch.float32, device=env.device),
                            reward=torch.zeros((T, N), dtype=torch.float32, device=env.device),
                            done=torch.zeros((T + 1, N), dtype=torch.uint8, device=env.device))
                env._pol_traj = bufs
            with env._on_stream():
                bufs["obs"][0].copy_(env.obs)                       # s of period 0; slots 1..T receive what the steps return
                bufs["done"][0].copy_(resetting)
            traj = dict(obs=bufs["obs"][1:], action=bufs["action"], reward=bufs["reward"], done=bufs["done"][1:])
            env.rollout_policy(agent, T, traj=traj, add_noise=add_noise, obs0=bufs["obs"][0])
            with env._on_stream():
                keep = ~bufs["done"][:T].bool().reshape(T * N)      # period t is a transition unless the env finished in period t - 1
                agent.memory.add_masked(bufs["obs"][:T].reshape(T * N, D), bufs["action"].reshape(T * N, 2), bufs["reward"].reshape(T * N),
                                        bufs["obs"][1:].reshape(T * N, D), bufs["done"][1:].reshape(T * N), keep)
                resetting = bufs["done"][T].bool()
        left -= T
    env._resetting = resetting
    return int(n_steps) * N


def rollout_groups(envs, agent, n_steps, add_noise=True, auto_reset="next"):
    """Actor-in-the-loop rollout over a crowdnav.env.VecEnvGroups: each group runs its own act -> step chain
    (cn_actor_forward, then cn_step) on its own HIP stream, so the actor of one group overlaps the env step of
    another and the env launches of different groups fill each other's idle phases.
    No join between groups until the end.  Static weights (evaluation / data collection between updates);
    call agent.sync_fused_weights() after an update.  Returns env-steps issued (N per step)."""
    agent.sync_fused_weights()
    if not getattr(envs, "_started", False):
        envs.reset()
        envs._started = True
    if getattr(envs, "_act", None) is None:
        envs._act = torch.zeros((envs.N, 2), dtype=torch.float32, device=envs.device)
    envs.fork()
    rows = [envs.rows(g) for g in range(envs.G)]
    # pre-marshalled launches: the loop is host-bound otherwise (2 G ctypes calls per step)
    calls = []
    for g in range(envs.G):
        calls.append(agent.bind_act_mfma(envs.obs[rows[g]], envs._act[rows[g]], add_noise=add_noise,
                                         stream=envs.streams[g], noise_seed=agent.group_noise_seed(g)))
        calls.append(envs.envs[g].bind_step(envs._act[rows[g]], auto_reset=auto_reset))
    for _ in range(n_steps):
        for c in calls:
            c()
    envs.join()
    return n_steps * envs.N


def evaluate(env, agent, episodes_per_env=1, max_launches=100000):
    """The reference's evaluation run (README "Start testing"; TRAIN:142-161 with learning = False): greedy
    actor, one CSV row per finished episode (success, failure, return, steps, ego/social safety scores).
    Every env contributes exactly its FIRST `episodes_per_env` episodes and the loop runs until every env has
    reached that quota: with auto-reset, short episodes would otherwise be over-represented (an env that fails early
    starts a second episode while long ones are still running)."""
    stats = EpisodeStats()
    env.reset()
    obs = env.obs
    step_s = (env.cfg.dt_ms + env.cfg.scan_latency_ms) / 1000.0           # one Env.step in the env's own clock
    finished = torch.zeros(env.N, dtype=torch.int64, device=obs.device)   # per-env count of recorded episodes
    launches = 0
    while int(finished.min().item()) < episodes_per_env and launches < max_launches:
        act = agent.act(obs, add_noise=False)
        obs, reward, done = env.step(act, auto_reset="next")        # (a finished env's next launch is its reset: done = 0 there)
        launches += 1
        take = done.bool() & (finished < episodes_per_env)
        if bool(take.any()):
            c = env.counters().cpu(); ret = env.returns()[0].cpu()
            for e in torch.nonzero(take.cpu()).flatten().tolist():
                stats.add_from_counters(c[e], ret[e].item(), step_seconds=step_s)      # timelapse = the episode's duration (TRAIN:141)
            finished += take.to(finished.dtype)
    return stats


class GraphedRollout:
    """Actor-in-the-loop stepping captured once into a HIP graph (torch.cuda.CUDAGraph): the actor's three
    GEMMs, the exploration noise, the clip and the fused env-step kernel replay as ONE graph launch per
    step, which removes the per-kernel launch gaps that dominate BASELINE config 3 at this step time.
    cn_step only enqueues on the capturing stream, so it is captured like any other kernel."""

    def __init__(self, env, agent, add_noise=True, auto_reset="next"):
        self.env, self.agent = env, agent
        if not getattr(env, "_started", False):
            env.reset()
            env._started = True
        self.obs = env.obs
        self.act = torch.zeros((env.N, 2), dtype=torch.float32, device=env.device)
        s = torch.cuda.Stream(device=env.device)
        s.wait_stream(torch.cuda.current_stream(env.device))
        with torch.cuda.stream(s):           # warm-up outside capture (allocator, lazy init)
            for _ in range(3):
                self.act.copy_(self._policy(add_noise))
                env.step(self.act, auto_reset=auto_reset)
        torch.cuda.current_stream(env.device).wait_stream(s)
        torch.cuda.synchronize(env.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.act.copy_(self._policy(add_noise))
            env.step(self.act, auto_reset=auto_reset)

    def _policy(self, add_noise):
        # three library GEMMs + the fused output stage (cn_policy_tail).  The noise counter is frozen into the
        # captured launch, so a replayed graph would repeat its noise: the counter is folded with the per-env
        # step count that the env kernel keeps in the observation-independent state... simpler and graph-safe:
        # draw the Gaussian with torch's default generator (philox offsets advance under replay).
        a = self.agent.actor(self.obs)
        if add_noise:
            a = a + torch.randn_like(a) * self.agent.explore_sigma
        return torch.max(torch.min(a, self.agent._hi), self.agent._lo)

    def step(self):
        self.graph.replay()
        return self.env.obs, self.env.reward, self.env.done
