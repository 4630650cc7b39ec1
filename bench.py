#!/usr/bin/env python
"""bench.py -- env-steps/sec of the fused HIP environment step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the driver's
  way: RANK / LOCAL_RANK / WORLD_SIZE come from the environment), or run bare -- `python bench.py --gpus N` then re-executes
  itself under torch.distributed.run with N ranks on 127.0.0.1, one rank per GPU over RCCL.
  --envs-total T: strong scaling, T environments split over the ranks (BASELINE config 4: 16384 over 8 GPUs = 2048 per
  GPU); default is weak scaling, --envs (4096) per GPU.

A "step" is every env of this rank's shard stepped once through cn_step (4096 envs x 20 pedestrians x 360 rays, K = 8,
BASELINE.json configs[1]) -- as `--groups` launches on independent streams (default 4 x 1024 envs), as 2 launches, or as
one; every decomposition is timed over the same K steps with the same bracket and the best one is the headline
(`config.legs_env_steps_s` carries all three).  `value` = env-steps/s summed over all ranks, inputs resident in HBM,
auto-reset included; a launch an env spends on its reset is not counted as an env-step.  Envs shard across
ranks with no data-path collective; the one collective is the RCCL all-gather of per-env episode returns after the timed
region (8e E1).

Adds to the JSON line:
  roofline      achieved algorithmic HBM bytes/s of cn_env_kernel (HIP events on the launch stream)
  cpu_baseline  the CPU oracle (plain-C port of the reference path, oracle/cn_oracle.c) timed on this
                box's host cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes(P, R, K):
    """Algorithmic HBM bytes per env-step of the float64-state layout:
    pedestrian pos+vel read+write 2*32P, obs f32 write 4(R-1+7+4K), top-K idx 4K, scalar records
    (24 f64 + 16 i32) read+write, action 8, reward 4, done 1."""
    return 64 * P + 4 * (R - 1 + 7 + 4 * K) + 4 * K + 2 * (24 * 8 + 16 * 4) + 8 + 4 + 1


D4_BYTES_PER_ENV_STEP = 2400.0   # SURVEY 8(d) D4's own per-env-step figure (float32 state), for round-to-round comparison


def profiled_counters():
    """Issue-side figures of cn_env_kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN/counters.json, written by tools/summarize_prof.py): VALU busy fraction and wave instructions per
    env-step.  None when no profile is committed."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "counters.json")))
    if not c:
        return None
    try:
        return json.load(open(c[-1]))
    except Exception:
        return None


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: become the launcher (one rank per GPU)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def profiled_traffic():
    """HBM bytes per cn_env_kernel launch from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN/traffic.json: FETCH_SIZE and WRITE_SIZE from separate passes, calibrated on this box's
    known-byte streams).  None when no profile is committed."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if not c:
        return None
    try:
        return float(json.load(open(c[-1]))["bytes_per_launch"])
    except Exception:
        return None


def baseline_metric():
    """The metric string exactly as BASELINE.json spells it."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "env-steps/sec @4096 envs\u00d720 peds\u00d7360 rays; HBM GB/s vs roofline"


def usable_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 CPU quota
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU (weak scaling)")
    ap.add_argument("--envs-total", type=int, default=0,
                    help="strong scaling: this many environments split evenly over the ranks (overrides --envs)")
    ap.add_argument("--preroll", type=int, default=200,
                    help="untimed steps after every reset so that the timed sample sees de-phased envs and real resets")
    ap.add_argument("--no-plateau", action="store_true", help="skip the 16384-env issue-bound measurement")
    ap.add_argument("--peds", type=int, default=20)
    ap.add_argument("--rays", type=int, default=360)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--groups", type=int, default=4,
                    help="headline leg: the rank's envs as this many independent stream groups (1 = one launch per step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))

    import torch
    import torch.distributed as dist
    from crowdnav import Config
    from crowdnav.env import VecEnv, VecEnvGroups

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # CN_BENCH_DRYRUN_GLOO=1: exercise the N > 1 code path on a single-GPU box (every rank on cuda:0,
    # gloo collectives on host copies).  Never set by the driver; numbers from it are meaningless.
    dry = os.environ.get("CN_BENCH_DRYRUN_GLOO") == "1"
    if a.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not dry and torch.cuda.device_count() < (local_rank + 1):
        raise SystemExit("bench.py: rank %d has no GPU (%d visible); CN_BENCH_DRYRUN_GLOO=1 shares cuda:0 for a dry run"
                         % (rank, torch.cuda.device_count()))
    dev_index = 0 if dry else local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = torch.device("cpu") if dry else dev   # where collective buffers live

    if a.envs_total:
        if a.envs_total % world:
            raise SystemExit("bench.py: --envs-total must divide evenly over the ranks")
        N = a.envs_total // world
    else:
        N = a.envs
    cfg = Config(n_envs=N, n_peds=a.peds, n_rays=a.rays, k_obstacles=a.k, max_steps=1000, seed=1234,
                 env_index_base=rank * N, ped_cycle_ms=1400,
                 room_half=2.40 if a.peds > 50 else 1.40)
    env = VecEnv(cfg, device=dev_index)
    env.reset()
    # open-loop actions v ~ U(0, 0.22), w ~ U(-2, 2): counter-based, seed 1234 + global env index
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    n_act = 64
    acts = torch.stack([torch.rand((n_act, N), generator=g, device=dev) * 0.22,
                        torch.rand((n_act, N), generator=g, device=dev) * 4.0 - 2.0], 2).contiguous()

    # The opening bracket.  A GPU that sits idle for a few ms clocks down and needs ~10 launches to come back
    # (tools/startup_transient.py: the same 20 steps take 800 us right after work, 870 us after 5 ms of idling, 960 us
    # after 100 ms), and a counters read-back or a rank barrier is such a pause.  So the episode counters are snapshotted
    # on the device, the rank barrier comes first and the last `warm_tail` of the W warm-up steps run between it and the
    # torch.cuda.synchronize() that starts the clock: barrier + synchronize still bracket the K timed steps, all W + the
    # pre-roll steps are still untimed, and the timed region starts on a GPU in the state a long run keeps it in.
    warm_tail = min(a.warmup, 3)

    def rank_barrier():
        if world > 1:
            dist.barrier()

    def barrier(streams=()):
        """The contract's bracket: barrier over the ranks + torch.cuda.synchronize().  hipDeviceSynchronize blocks on an
        interrupt and wakes up 60-75 us after the last kernel has finished (tools/startup_transient.py) -- 7 % of a
        20-step sample of 0.85 ms that is host notification latency, not device work.  So the streams that carry the
        timed launches are polled to completion first (hipStreamQuery, ~1 us a call); the synchronize that follows
        then returns at once and still is the bracket."""
        for s_ in streams:
            while not s_.query():
                pass
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(mode):
        """K launches of cn_step; returns (wall s, kernel ms/launch, env-steps actually taken by this rank)."""
        n_pre = a.preroll + a.warmup
        for i in range(n_pre - warm_tail):
            env.step(acts[i % n_act], auto_reset=mode)
        ep0 = env.counters()[:, 8].sum()              # device scalar, read after the timed region (no host sync here)
        rank_barrier()
        for i in range(n_pre - warm_tail, n_pre):
            env.step(acts[i % n_act], auto_reset=mode)
        stream = torch.cuda.current_stream(dev)
        while not stream.query():                      # spin instead of sleeping in the synchronize (see timed_groups)
            pass
        torch.cuda.synchronize(dev)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for i in range(a.steps):
            env.step(acts[i % n_act], auto_reset=mode)
        ev1.record(stream)
        barrier((stream,))
        wall_ = time.perf_counter() - t0
        k_ms = ev0.elapsed_time(ev1) / a.steps       # cn_env_kernel is the only kernel in the timed region
        ep1 = env.counters()[:, 8].sum().item()
        torch.cuda.synchronize(dev)
        taken = N * a.steps
        if mode == "next":                            # a finished env spends one launch on its reset: not an env-step
            taken -= int(ep1 - ep0.item())
        return wall_, k_ms, taken

    conc = 1
    enq_ms = {}
    def timed_groups(G, mode="next", gcfg=None, gacts=None, steps=None):
        """The same N envs as G independent groups (crowdnav.env.VecEnvGroups): one step = every group stepped
        once, each on its own HIP stream, no join between groups inside the timed region.  Returns (wall s,
        mean per-stream ms/launch from HIP events on each group's stream, env-steps taken)."""
        gcfg = gcfg or cfg
        acts_ = gacts if gacts is not None else acts
        steps_ = steps or a.steps
        grp = VecEnvGroups(gcfg, groups=G, device=dev_index)
        nonlocal conc
        conc = grp.concurrent
        grp.reset()
        rows = [grp.rows(g) for g in range(G)]
        # pre-marshalled launches: ~1.5 us of host time per foreign call instead of ~7 us for VecEnv.step, so a short timed
        # sample (the driver uses 20 steps = 80 launches of ~40 us) is not paced by the Python enqueue loop
        # one cn_step_multi per step: all groups' launches behind one foreign call
        calls = [[grp.bind_step_all(acts_[i], auto_reset=mode)] for i in range(n_act)]
        def episodes_dev():                            # finished episodes per group as device scalars, no host sync
            out = []
            for e_ in grp.envs:
                with torch.cuda.stream(e_.stream):
                    out.append(e_.counters()[:, 8].sum())
            return out
        n_pre = a.preroll + a.warmup
        # the timed steps as ONE pre-marshalled cn_step_multi (steps_ x G entries, step-major): the host side of the timed
        # region is then a C loop over hipLaunchKernel, as it would be in a C++ trainer, not a Python loop
        timed_call = grp.bind_step_sequence([acts_[i % n_act] for i in range(steps_)], auto_reset=mode)
        for i in range(n_pre - warm_tail):
            for c in calls[i % n_act]:
                c()
        ep0 = episodes_dev()
        rank_barrier()
        for i in range(n_pre - warm_tail, n_pre):
            for c in calls[i % n_act]:
                c()
        for s_ in grp.streams:                         # spin, do not sleep: a host core that blocked in hipDeviceSynchronize
            while not s_.query():                      # comes back clocked down and its first launches cost 2-3x
                pass
        torch.cuda.synchronize(dev)
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(G)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(G)]
        t0 = time.perf_counter()
        for g in range(G):
            ev0[g].record(grp.streams[g])
        timed_call()                                   # all steps_ x G launches: one cn_step_multi, a C loop
        enq_ms.setdefault(G, (time.perf_counter() - t0) * 1e3 / steps_)   # host time to enqueue one step of all groups (first leg with G)
        for g in range(G):
            ev1[g].record(grp.streams[g])
        barrier(grp.streams)
        wall_ = time.perf_counter() - t0
        k_ms = sum(ev0[g].elapsed_time(ev1[g]) for g in range(G)) / G / steps_
        taken = gcfg.n_envs * steps_ - (grp.episodes() - sum(int(x.item()) for x in ep0) if mode == "next" else 0)
        grp.close()
        return wall_, k_ms, taken

    # legs: reset inside the same call (auto_reset = 1); next-step reset (auto_reset = 2) as one launch per step;
    # next-step reset with the envs as `--groups` independent stream groups (the headline when groups > 1)
    wall_same, kms_same, taken_same = timed("same")
    env.reset()
    wall_1, kernel_ms_1, taken_1 = timed("next")
    G = max(1, a.groups)
    if G > 1:
        wall, kernel_ms, taken = timed_groups(G)
    else:
        wall, kernel_ms, taken = wall_1, kernel_ms_1, taken_1
    # A third decomposition of the same step: 2 groups.  4 groups need a launch every ~10 us from the host thread; on a slow or
    # busy host a short sample is then paced by the enqueue loop (20-step samples: 84 M with 4 groups, 92 M with 2 on one
    # box; 91-93 M with 4 on others), while 2048-env launches need half the launch rate and give 94 % of the 4-group rate.
    G2 = 2 if G > 2 and N % 2 == 0 else 0
    conc_requested = conc                              # concurrent streams found for the requested group count
    wall_2, kernel_ms_2, taken_2 = timed_groups(G2) if G2 else (wall_1, kernel_ms_1, taken_1)
    conc = conc_requested
    # Issue-bound ceiling of this kernel on this GPU, measured in the same run: 16384 resident envs in 4 stream groups
    # (every SIMD has work in every phase).  Rank 0 of a single-GPU run only.
    plateau = None
    if world == 1 and not a.no_plateau and (a.peds, a.rays) == (20, 360):
        import dataclasses
        pcfg = dataclasses.replace(cfg, n_envs=16384)
        gp = torch.Generator(device=dev).manual_seed(99)
        pacts = torch.stack([torch.rand((n_act, 16384), generator=gp, device=dev) * 0.22,
                             torch.rand((n_act, 16384), generator=gp, device=dev) * 4.0 - 2.0], 2).contiguous()
        pw, pk, pt = timed_groups(max(1, a.groups), gcfg=pcfg, gacts=pacts, steps=max(50, min(a.steps, 300)))
        plateau = pt / pw
        del pacts
    if world > 1:
        t = torch.tensor([wall, float(taken), wall_same, float(taken_same), wall_1, float(taken_1), wall_2, float(taken_2)],
                         dtype=torch.float64, device=cdev)
        tm = t.clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ts = t.clone(); dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        wall, wall_same, wall_1, wall_2 = float(tm[0].item()), float(tm[2].item()), float(tm[4].item()), float(tm[6].item())
        taken_all, taken_same_all, taken_1_all, taken_2_all = (float(ts[1].item()), float(ts[3].item()), float(ts[5].item()),
                                                               float(ts[7].item()))
        # the path's one exchange: all-gather of per-env episode returns over xGMI
        ret, _ = env.returns()
        ret = ret.to(cdev)
        gathered = torch.empty(world * N, dtype=torch.float32, device=cdev)
        dist.all_gather_into_tensor(gathered, ret)     # first call builds the communicator rings: not timed
        torch.cuda.synchronize(dev)
        dist.barrier()
        tg0 = time.perf_counter()
        for _ in range(10):
            dist.all_gather_into_tensor(gathered, ret)
        torch.cuda.synchronize(dev)
        gather_ms = (time.perf_counter() - tg0) * 1e3 / 10
        # every rank holds every env's return; rank r's slice must be what rank r computed
        assert torch.equal(gathered[rank * N:(rank + 1) * N], ret)
    else:
        taken_all, taken_same_all, taken_1_all, taken_2_all, gather_ms = (float(taken), float(taken_same), float(taken_1),
                                                                          float(taken_2), None)

    # Headline = the best of the next-step-reset legs (the same K steps of the same envs, decomposed into `--groups`, 2 or 1
    # launches per step); the JSON says which one and carries the others.  A runtime that maps every stream onto one hardware
    # queue makes the one-launch leg the headline, a slow host the 2-group leg.  Decided on the rank-aggregated numbers, so
    # every rank agrees.
    groups_requested = G
    legs = {"%d_groups" % G: taken_all / wall, "1_launch": taken_1_all / wall_1}
    if G2:
        legs["2_groups"] = taken_2_all / wall_2
        if taken_2_all / wall_2 > taken_all / wall:
            G, wall, kernel_ms, taken_all = 2, wall_2, kernel_ms_2, taken_2_all
    if G > 1 and taken_1_all / wall_1 > taken_all / wall:
        G, wall, kernel_ms, taken_all = 1, wall_1, kernel_ms_1, taken_1_all
    value = taken_all / wall
    B = algorithmic_bytes(a.peds, a.rays, a.k)
    n_launch = N // G                                  # envs per cn_env_kernel launch in the headline leg
    # every launch moves its envs' state, reset or step; G launches are in flight at once, one per stream
    achieved = G * B * n_launch / (kernel_ms * 1e-3) / 1e9
    achieved_1 = B * N / (kernel_ms_1 * 1e-3) / 1e9
    counters = profiled_counters() if (a.peds, a.rays) == (20, 360) else None
    traffic = profiled_traffic() if (a.envs, a.peds, a.rays) == (4096, 20, 360) else None
    valu_n = ((counters or {}).get("wave_instr_per_env_step") or {}).get("valu")
    valu_peak = (1024 * 2.4e9 / (4.0 * valu_n)) if valu_n else None
    out = {
        "metric": baseline_metric(),
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall / a.steps * 1e3, "higher_is_better": True, "scaling": "strong" if a.envs_total else "weak",
        "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d envs/GPU x %d pedestrians x %d rays, K=%d, lidar-tracker risk features, "
                               "next-step auto-reset (reset launches not counted as env-steps), open-loop "
                               "U(0,0.22)xU(-2,2) actions, %d untimed pre-roll steps; one step = every env stepped once, "
                               "the envs running as %d independent stream group(s) of %d" % (
                                   ("BASELINE configs[3] shape, %d envs total over %d GPU(s) (strong scaling)" % (a.envs_total, world))
                                   if a.envs_total else ("BASELINE configs[1] per GPU (weak scaling over %d GPU(s))" % world),
                                   N, a.peds, a.rays, a.k, a.preroll, G, n_launch),
                   "envs_per_gpu": N, "envs_total": N * world, "stream_groups": G, "stream_groups_requested": groups_requested,
                   "legs_env_steps_s": legs,
                   # host time the enqueue loop needs per step of a group leg (a leg is host-paced when this approaches ms_per_step)
                   "host_enqueue_ms_per_step": {"%d_groups" % k: v for k, v in sorted(enq_ms.items()) if k in (groups_requested, 2)},
                   "concurrent_hw_queues_found": conc, "parallelism": "env-sharded x%d" % world,
                   "one_launch_per_step_value": taken_1_all / wall_1, "one_launch_per_step_ms": wall_1 / a.steps * 1e3,
                   "same_call_reset_value": taken_same_all / wall_same, "same_call_reset_ms_per_step": wall_same / a.steps * 1e3,
                   "returns_allgather_ms": gather_ms},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic * n_launch / 4096.0 if traffic is not None else None,
                     "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE+WRITE_SIZE, calibrated; profiles/)",
                     "algorithmic_bytes_per_launch": B * n_launch, "envs_per_launch": n_launch,
                     "concurrent_launches": G,
                     "kernel": "cn_env_kernel", "kernel_ms": kernel_ms, "bytes_per_env_step": B,
                     # the same speed priced with SURVEY 8(d) D4's 2400 B per env-step (float32 state), so that the HBM
                     # fraction stays comparable across rounds whatever the state dtype is
                     "frac_d4": G * D4_BYTES_PER_ENV_STEP * n_launch / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     # what actually binds: instruction issue (float64 VALU) and one wavefront's critical path, not HBM
                     "binding": "instruction issue (f64 VALU) above ~8k resident envs; one wavefront's critical path at 4096",
                     "issue_bound_env_steps_s": plateau,
                     "frac_of_issue_bound": (value / world / plateau) if plateau else None,
                     # the hardware ceiling of the binding unit: every wave64 VALU instruction occupies its SIMD's vector pipe
                     # for 4 cycles; 1024 SIMDs x 2.4 GHz (MI355X_MICROARCH.md) / (4 x measured VALU instructions per env-step)
                     "valu_peak_env_steps_s": valu_peak,
                     "frac_of_valu_peak": (value / world / valu_peak) if valu_peak else None,
                     "valu_busy": (counters or {}).get("valu_busy"),
                     "wave_instr_per_env_step": (counters or {}).get("wave_instr_per_env_step"),
                     "counters_source": (counters or {}).get("source"),
                     "note": "achieved = concurrent_launches x algorithmic bytes per launch / mean launch duration on "
                             "its own stream (HIP events per group stream)",
                     "one_launch_per_step": {"envs_per_launch": N, "kernel_ms": kernel_ms_1, "achieved": achieved_1,
                                             "frac": achieved_1 / HBM_PEAK_GBS}},
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import oracle
        ncpu = usable_cpus()

        def time_oracle(threads, n_s, seconds):
            oracle.set_num_threads(threads)
            c2 = Config(n_envs=n_s, n_peds=a.peds, n_rays=a.rays, k_obstacles=a.k, max_steps=1000, seed=1234,
                        ped_cycle_ms=1400, room_half=cfg.room_half)
            orc = oracle.Oracle(c2.as_dict())
            orc.reset()
            acts_c = acts[:, :n_s].double().cpu().numpy()
            orc.step(acts_c[0], auto_reset=True)
            tc0 = time.perf_counter(); k = 0
            while time.perf_counter() - tc0 < seconds:
                orc.step(acts_c[k % n_act], auto_reset=True); k += 1
            tc = time.perf_counter() - tc0
            return n_s * k / tc, k, tc

        v1, k1, t1c = time_oracle(1, 256, a.cpu_seconds * 0.4)
        n_all = min(N, 16 * ncpu)
        vall, kall, tallc = time_oracle(ncpu, n_all, a.cpu_seconds * 0.6)
        out["cpu_baseline"] = {"value": max(v1, vall), "unit": "env-steps/s", "cores": ncpu if vall >= v1 else 1,
                               "kind": "port", "one_core_value": v1,
                               "sample": "oracle/cn_oracle.c (plain-C port of the reference path) on the same "
                                         "workload: %d envs x %d steps on 1 thread (%.1f s) and %d envs x %d steps "
                                         "with OpenMP over envs on %d threads (%.1f s)" % (256, k1, t1c, n_all, kall,
                                                                                           ncpu, tallc)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
