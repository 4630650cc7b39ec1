#!/usr/bin/env python
"""bench.py -- env-steps/sec of the fused HIP environment step (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the driver's
  way: RANK / LOCAL_RANK / WORLD_SIZE come from the environment), or run bare -- `python bench.py --gpus N` then re-executes
  itself under torch.distributed.run with N ranks on 127.0.0.1, one rank per GPU over RCCL.
  --envs-total T: strong scaling, T environments split over the ranks (BASELINE config 4: 16384 over 8 GPUs = 2048 per
  GPU); default is weak scaling, --envs (4096) per GPU.

A "step" is every env of this rank's shard stepped once through cn_step (4096 envs x 20 pedestrians x 360 rays, K = 8,
BASELINE.json configs[1]).  The shard runs as `--groups` independent stream groups (default 4 x 1024 envs), as 2, or as one
launch per step.  Protocol (no choice is made on the reported sample):
  1. after the W warm-up + pre-roll steps every decomposition runs ONE untimed-for-the-report probe of K steps; the best
     probe AMONG THE DECOMPOSITIONS THAT CAN SERVE A POLICY IN THE LOOP -- one cn_step launch per step, alone or as stream
     groups: SURVEY 8(d) D1's "one env-step = one cn_step for one env" -- decides which one is the headline
     (`config.headline_choice`).  The open-loop single-launch forms (cn_step_sequence in place / into trajectory buffers) are
     timed as comparables (`config.leg_sequence*`), never as `value` (round 5);
  2. every decomposition is then timed `--repeats` (5) times: each sample = exactly K steps bracketed by barrier +
     torch.cuda.synchronize() on both sides (max over ranks); `value` / `ms_per_step` are the MEDIAN sample of the headline
     decomposition, `config.legs_env_steps_s` the medians of all of them, `config.samples_env_steps_s` every sample.
`value` = env-steps/s summed over all ranks, inputs resident in HBM, auto-reset included; a launch an env
spends on its reset is not counted as an env-step (counted exactly: finished-episode and reset-pending counters on the
device before and after the K steps).  Envs shard across ranks with no data-path collective; the one collective is the
RCCL all-gather of per-env episode returns after the timed region (8e E1).

Adds to the JSON line:
  roofline       achieved algorithmic HBM bytes/s of cn_env_kernel (HIP events on the launch streams); `frac` is priced with
                 SURVEY 8(d) D4's 2400 B per env-step, `frac_f64_layout` with this build's float64 state layout
  cpu_baseline   the CPU oracle (plain-C port of the reference path, oracle/cn_oracle.c) timed on this box's host cores on a
                 bounded sample of the same workload (rank 0, N = 1 only), the reference's own Python rate beside it
  config.other_configs   BASELINE configs[2] (TD3 actor in the loop) and configs[4] (100 pedestrians x 720 rays) under
                 the same bracket, each with its own ms_per_step and D4-priced roofline fraction (N = 1 only); their
                 headline numbers are repeated as flat scalars (config.configs2_* / configs4_* / leg_*) for record keepers
                 that drop nested objects
  config.sustained_*     the headline decomposition stepped back to back for >= --sustained-seconds (6 s): env-steps/s, the
                 shader clock the chip held (s_memtime cycles / 100 MHz s_memrealtime ticks around the run, cn_device_clock) and
                 the burst / sustained ratio.  Runs right after the headline legs, long before the CPU baseline.
"""
import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
VECTOR_PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md "Peak FP32 (vector)" (packed v_pk_fma_f32: 2 x 64 lanes x 2 flop per 4-cycle issue)
# float64 vector peak: the guide has no row for it.  AMD's MI355X data sheet: 78.6 TFLOP/s (half the packed-f32 rate) -- and it is what
# tools/micro/issue_cost.hip measures: one v_fma_f64 per 4 cycles per SIMD = 1024 SIMDs x 64 lanes x 2 flop / 4 cycles x 2.4 GHz.
VECTOR_PEAK_F64_TFLOPS = 78.6
N_ACT = 64             # distinct open-loop action tensors cycled through


def algorithmic_bytes(P, R, K):
    """Algorithmic HBM bytes per env-step of the float64-state layout:
    pedestrian pos+vel read+write 2*32P, obs f32 write 4(R-1+7+4K), top-K idx 4K, scalar records
    (24 f64 + 16 i32) read+write, action 8, reward 4, done 1."""
    return 64 * P + 4 * (R - 1 + 7 + 4 * K) + 4 * K + 2 * (24 * 8 + 16 * 4) + 8 + 4 + 1


def d4_bytes(P, R, K):
    """SURVEY 8(d) D4's own per-env-step figure (float32 state): 2400 B at 20 pedestrians x 360 rays, 6400 B at 100 x 720
    (16 P of pedestrian state read + written, the observation and indices written once, ~0.4 KB of scalars)."""
    if (P, R) == (20, 360):
        return 2400.0
    if (P, R) == (100, 720):
        return 6400.0
    return float(32 * P + 4 * (R - 1 + 7 + 4 * K) + 4 * K + 256 + 13)


def profiled_json(name):
    """profiles/rNN/<name> of the latest round IF it was taken with the kernel sources of this tree (csrc_hash stamp written by
    tools/summarize_prof.py); (None, reason) otherwise -- a profile of another kernel says nothing about this one."""
    import glob
    from csrc_hash import csrc_hash
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)))
    if not c:
        return None, "no profile committed"
    try:
        d = json.load(open(c[-1]))
    except Exception as ex:
        return None, "unreadable %s: %s" % (c[-1], ex)
    if d.get("csrc_hash") != csrc_hash():
        return None, "%s was taken with csrc %s, this tree is %s" % (os.path.relpath(c[-1], ROOT), d.get("csrc_hash"), csrc_hash())
    return d, os.path.relpath(c[-1], ROOT)


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


def baseline_metric():
    """The metric string exactly as BASELINE.json spells it."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "env-steps/sec @4096 envs×20 peds×360 rays; HBM GB/s vs roofline"


def usable_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 CPU quota
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, n)


def median_index(vals):
    """Index of the median element (the lower middle one for an even count)."""
    order = sorted(range(len(vals)), key=lambda i: vals[i])
    return order[(len(vals) - 1) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--repeats", type=int, default=5, help="timed samples of K steps per decomposition; the median is reported")
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU (weak scaling)")
    ap.add_argument("--envs-total", type=int, default=0,
                    help="strong scaling: this many environments split evenly over the ranks (overrides --envs)")
    ap.add_argument("--preroll", type=int, default=200,
                    help="untimed steps after every reset so that the timed samples see de-phased envs and real resets")
    ap.add_argument("--no-plateau", action="store_true", help="skip the 16384-env issue-bound measurement")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the BASELINE configs[2] / configs[4] legs")
    ap.add_argument("--peds", type=int, default=20)
    ap.add_argument("--rays", type=int, default=360)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--groups", type=int, default=4,
                    help="largest decomposition tried: the rank's envs as this many independent stream groups")
    ap.add_argument("--sustained-seconds", type=float, default=6.0,
                    help="length of the sustained leg (headline decomposition back to back); 0 = skip")
    ap.add_argument("--no-sequence-traj", action="store_true", help="skip the sequence_traj leg (profiling passes: one kind of launch per kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))

    import torch
    import torch.distributed as dist
    from crowdnav import Config
    from crowdnav.env import VecEnvGroups, concurrent_streams

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
    # Under a launcher (torch.distributed.run sets RANK / WORLD_SIZE) the process group is created and every collective of the
    # N > 1 path runs even at world size 1 -- communicator init, the barrier, the sample all-reduces, the RCCL all-gather of
    # returns -- so that branch has executed on real hardware before an 8-GPU run meets it.  The driver's N = 1 run (plain
    # `python bench.py`, no launcher environment) takes no collective at all.
    use_dist = world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ and os.environ.get("CN_BENCH_NO_DIST") != "1")
    if use_dist:
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
    K, R = max(1, a.steps), max(1, a.repeats)
    cfg = Config(n_envs=N, n_peds=a.peds, n_rays=a.rays, k_obstacles=a.k, max_steps=1000, seed=1234,
                 env_index_base=rank * N, ped_cycle_ms=1400,
                 room_half=2.40 if a.peds > 50 else 1.40)

    def open_loop_actions(n, seed):
        """v ~ U(0, 0.22), w ~ U(-2, 2): N_ACT tensors [n, 2] resident in HBM"""
        g = torch.Generator(device=dev).manual_seed(seed)
        return torch.stack([torch.rand((N_ACT, n), generator=g, device=dev) * 0.22,
                            torch.rand((N_ACT, n), generator=g, device=dev) * 4.0 - 2.0], 2).contiguous()

    acts = open_loop_actions(N, 1234 + rank)
    # streams that really run concurrently (HIP maps streams onto a few hardware queues), probed once for every leg
    Gmax = max(1, a.groups)
    streams, conc = concurrent_streams(Gmax, dev_index)
    warm_tail = min(a.warmup, 3)

    def rank_barrier():
        if use_dist:
            dist.barrier()

    def bracket(strs):
        """The contract's bracket: barrier over the ranks + torch.cuda.synchronize().  The streams that carry the launches are
        polled to completion first (hipStreamQuery), so that the synchronize returns at once instead of sleeping on an
        interrupt and waking up 60-75 us late (tools/startup_transient.py)."""
        for s_ in strs:
            while not s_.query():
                pass
        rank_barrier()
        torch.cuda.synchronize(dev)

    class Leg:
        """One decomposition of a config: its envs as G stream groups (G = 1: one launch per step), open loop or with the TD3
        actor in the loop (each group's act -> step chain on its own stream, crowdnav.rollout.rollout_groups)."""

        def __init__(self, lcfg, G, mode="next", lacts=None, agent=None, sequence=False, arbitration=None, python_loop=False,
                     policy_sequence=False, traj=False):
            self.cfg, self.G, self.mode, self.agent, self.sequence = lcfg, G, mode, agent, sequence
            self.lacts = lacts
            self.timed_call = None
            self.python_loop = python_loop             # the timed K steps enqueued one foreign call per step from Python (as a trainer would)
            # arbitration None: the library's defaults (cn_set_arbitration) -- one launch per step picks the fair kernel when it
            # fills the device on its own, overlapping stream groups stay on the hardware's oldest-first order
            self.grp = VecEnvGroups(lcfg, groups=G, device=dev_index, streams=streams[:G] if G <= len(streams) else None,
                                    arbitration=arbitration)
            self.arbitration = "rotating per step (sequence kernel)" if sequence else self.grp.envs[0].arbitration
            # the device kernel this leg's launches run (cn_kernel_name): the key of profiles/rNN/{counters,traffic}.json
            e0 = self.grp.envs[0]
            self.kernel = e0.kernel_name("policy" if policy_sequence else "sequence" if sequence else "same" if mode == "same" else
                                         "multi" if G > 1 else "step")
            self.grp.reset()
            self.enq_ms = None
            if sequence:
                # cn_step_sequence: the K timed steps as ONE launch of persistent wavefronts (open-loop actions [K, N, 2] in HBM)
                self.calls = [self.grp.bind_step_all(lacts[i], auto_reset=mode) for i in range(N_ACT)]
                self.seq_actions = lacts[torch.arange(K, device=dev) % N_ACT].contiguous()
                if not traj:
                    self.timed_call = self.grp.envs[0].bind_step_sequence(self.seq_actions)
                else:
                    # the same launch(es) with TRAJECTORY buffers: step t's observation / reward / done land in their own HBM slot
                    # (what an offline consumer of an open-loop phase reads); launches of at most 50 steps so that the buffers stay
                    # small whatever K is (50 x 4096 x 398 x 4 B = 326 MB)
                    Tc = min(K, 50)
                    D_ = e0.D
                    self.traj = dict(obs=torch.zeros((Tc, lcfg.n_envs, D_), dtype=torch.float32, device=dev),
                                     reward=torch.zeros((Tc, lcfg.n_envs), dtype=torch.float32, device=dev),
                                     done=torch.zeros((Tc, lcfg.n_envs), dtype=torch.uint8, device=dev))
                    parts = []
                    for k0 in range(0, K, Tc):
                        n_ = min(Tc, K - k0)
                        parts.append(e0.bind_step_sequence(self.seq_actions[k0:k0 + n_],
                                                           traj={k_: v_[:n_] for k_, v_ in self.traj.items()}))
                    self.timed_call = (lambda ps=tuple(parts): [c_() for c_ in ps]) if len(parts) > 1 else parts[0]
            elif agent is None:
                self.calls = [self.grp.bind_step_all(lacts[i], auto_reset=mode) for i in range(N_ACT)]
                # the K timed steps as ONE pre-marshalled cn_step_multi (K x G entries): the host side of a sample is a C loop
                self.timed_call = self.grp.bind_step_sequence([lacts[i % N_ACT] for i in range(K)], auto_reset=mode)
            elif policy_sequence:
                # cn_rollout_policy: the K timed periods (actor -> Env.step, closed loop) as ONE launch; warm-up = one-period launches
                agent.sync_fused_weights()
                self.chain = [e0.bind_rollout_policy(agent, 1, add_noise=True)]
                self.timed_call = e0.bind_rollout_policy(agent, K, add_noise=True)
            else:
                agent.sync_fused_weights()
                self.act = torch.zeros((lcfg.n_envs, 2), dtype=torch.float32, device=dev)
                chain = []
                for g in range(G):
                    rows = self.grp.rows(g)
                    chain.append(agent.bind_act_mfma(self.grp.obs[rows], self.act[rows], add_noise=True,
                                                     stream=self.grp.streams[g], noise_seed=agent.group_noise_seed(g)))
                    chain.append(self.grp.envs[g].bind_step(self.act[rows], auto_reset=mode))
                self.chain = chain

        def run(self, k, i0=0):
            if self.agent is None:
                for i in range(k):
                    self.calls[(i0 + i) % N_ACT]()
            else:
                for _ in range(k):
                    for c in self.chain:
                        c()

        def _reset_launch_marker(self):
            """finished episodes minus pending resets, summed over this leg's envs, as device scalars (no host sync): the
            difference of two markers is exactly the number of launches envs spent on a reset in between (auto_reset 2)"""
            out = []
            for e_ in self.grp.envs:
                with torch.cuda.stream(e_.stream):
                    c = e_.counters()
                    out.append((c[:, 8].sum() - c[:, 9].sum()).clone())
            return out

        def sample(self, steps=None):
            """exactly K steps between two brackets -> (wall s, mean ms per launch on its stream, env-steps taken)"""
            steps = K if steps is None else steps
            grp, G = self.grp, self.G
            rank_barrier()
            self.run(warm_tail)                        # the last warm-up steps run after the rank barrier (no idle gap)
            m0 = self._reset_launch_marker()
            for s_ in grp.streams:
                while not s_.query():
                    pass
            torch.cuda.synchronize(dev)
            ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(G)]
            ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(G)]
            for g in range(G):
                ev0[g].record(grp.streams[g])
            t0 = time.perf_counter()                   # the K steps start here: the next host action is their first enqueue
            if self.timed_call is not None and steps == K and not self.python_loop:
                self.timed_call()
            else:
                self.run(steps)
            t_enq = time.perf_counter() - t0
            for g in range(G):
                ev1[g].record(grp.streams[g])
            bracket(grp.streams)
            wall = time.perf_counter() - t0
            if self.enq_ms is None:
                self.enq_ms = t_enq * 1e3 / steps
            k_ms = sum(ev0[g].elapsed_time(ev1[g]) for g in range(G)) / G / steps
            taken = self.cfg.n_envs * steps
            if self.mode == "next":
                m1 = self._reset_launch_marker()
                torch.cuda.synchronize(dev)
                taken -= int(sum(x.item() for x in m1) - sum(x.item() for x in m0))
            return wall, k_ms, taken

        def sustained(self, seconds):
            """The decomposition stepped back to back for ~`seconds`: pre-marshalled calls of S steps each, the host at most two
            calls ahead of the device.  -> dict(env_steps_s, seconds, env_steps, clock_mhz, clock_mhz_idle)."""
            grp, e0 = self.grp, self.grp.envs[0]
            S = 500
            per_call = self.cfg.n_envs * S
            if self.sequence:
                acts_ = self.lacts[torch.arange(S, device=dev) % N_ACT].contiguous()
                call = e0.bind_step_sequence(acts_)
            else:
                call = grp.bind_step_sequence([self.lacts[i % N_ACT] for i in range(S)], auto_reset=self.mode)
            # the shader clock of an otherwise idle chip, for comparison (the probe wave alone, 2 ms)
            for s_ in grp.streams:
                s_.synchronize()
            side = torch.cuda.Stream(device=dev)
            time.sleep(0.25)
            ci = e0.device_clock(2000, stream=side); torch.cuda.synchronize(dev)
            call(); self.run(warm_tail)
            m0 = self._reset_launch_marker()
            for s_ in grp.streams:
                s_.synchronize()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            evs, probes = [], []
            n_probes = min(8, max(1, int(seconds / 1.5)))
            n_calls = 0
            while True:                                    # the host stays at most two calls (~40 ms) ahead of the device, so its clock
                call()                                     # follows the device's: stop once `seconds` have passed
                ev = torch.cuda.Event(); ev.record(grp.streams[0]); evs.append(ev)
                n_calls += 1
                if n_calls >= 3:
                    evs[n_calls - 3].synchronize()
                    el = time.perf_counter() - t0
                    # clock probes on a side stream while the env launches keep coming: one every ~1.5 s, 5 ms each
                    if len(probes) < n_probes and el >= seconds * (len(probes) + 0.5) / n_probes:
                        probes.append(e0.device_clock(5000, stream=side))
                    if el >= seconds:
                        break
            bracket(grp.streams)
            wall = time.perf_counter() - t0
            taken = n_calls * per_call
            if self.mode == "next":
                m1 = self._reset_launch_marker()
                torch.cuda.synchronize(dev)
                taken -= int(sum(x.item() for x in m1) - sum(x.item() for x in m0))
            torch.cuda.synchronize(dev)

            def mhz(x_):          # shader-clock cycles per microsecond of the 100 MHz counter, inside one probe wave
                x_ = x_.cpu().tolist()
                return x_[0] / max(1.0, x_[1] / 100.0)
            pm = [mhz(x_) for x_ in probes]
            return {"env_steps_s": taken / wall, "seconds": wall, "env_steps": taken, "launches": n_calls * S * self.G,
                    "clock_mhz": (sum(pm) / len(pm)) if pm else None, "clock_mhz_probes": pm, "clock_mhz_idle": mhz(ci)}

        def close(self):
            self.grp.close()

    def reduce_samples(samples):
        """per-sample (wall, kernel ms, taken) over the ranks: max wall, mean kernel ms, summed env-steps"""
        if not use_dist:
            return samples, None
        t = torch.tensor(samples, dtype=torch.float64, device=cdev)          # [n, 3]
        mx = t.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = t.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        allr = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allr, t)
        out = [(float(mx[i, 0]), float(sm[i, 1]) / world, float(sm[i, 2])) for i in range(len(samples))]
        return out, [x.cpu().tolist() for x in allr]

    def measure(lcfg, candidates, lacts=None, agent=None, mode="next", repeats=R, probe=True, sequence_leg=False):
        """Every candidate decomposition of one config: pre-roll + warm-up, one probe of K steps each (the best probe among the
        ELIGIBLE ones decides the headline), then `repeats` timed samples each.  Eligible = can serve a policy in the loop: stream
        groups / one launch per step (integers) and cn_rollout_policy; the open-loop cn_step_sequence forms and the A/B legs are
        comparables only.  Returns a dict with the chosen decomposition's median sample and all legs."""
        legs = {}
        if sequence_leg:
            # + the A/B of cn_set_arbitration on one launch per step (only where the library's default is the fair kernel)
            # + one launch per step enqueued from a Python loop, one foreign call per step (what a Python trainer pays; the other
            #   legs pre-marshal their K steps into ONE cn_step_multi / cn_step_sequence call)
            # + cn_step_sequence: in place, and into trajectory buffers (every step's outputs in their own HBM slot)
            candidates = (list(candidates) + (["1_groups_oldest_first"] if 1 in candidates and lcfg.n_envs >= 2048 else [])
                          + (["1_groups_python_enqueue"] if 1 in candidates else []) + ["sequence"] + ([] if a.no_sequence_traj else ["sequence_traj"]))
        eligible = [G for G in candidates if isinstance(G, int) or G == "policy_sequence"]
        for G in candidates:
            lg = (Leg(lcfg, 1, lacts=lacts, sequence=True) if G == "sequence" else
                  Leg(lcfg, 1, lacts=lacts, sequence=True, traj=True) if G == "sequence_traj" else
                  Leg(lcfg, 1, mode=mode, lacts=lacts, agent=agent, arbitration="oldest_first") if G == "1_groups_oldest_first" else
                  Leg(lcfg, 1, mode=mode, lacts=lacts, agent=agent, python_loop=True) if G == "1_groups_python_enqueue" else
                  Leg(lcfg, 1, agent=agent, policy_sequence=True) if G == "policy_sequence" else
                  Leg(lcfg, G, mode=mode, lacts=lacts, agent=agent))
            lg.run(a.preroll + a.warmup - warm_tail)
            legs[G] = lg
        probes = {}
        if probe and len(candidates) > 1:
            legs[candidates[-1]].sample()        # unreported: the first timed burst after the set-up pauses runs on ramping clocks
            raw = [legs[G].sample() for G in candidates]
            red, _ = reduce_samples(raw)
            probes = {G: red[i][2] / red[i][0] for i, G in enumerate(candidates)}
            chosen = max(eligible, key=lambda G: probes[G])
        else:
            chosen = eligible[0]
        out = {"chosen": chosen, "eligible": [str(G) for G in eligible],
               "probe_env_steps_s": {str(G): v for G, v in probes.items()}, "legs": {}}
        for G in candidates:
            raw = [legs[G].sample() for _ in range(repeats)]
            red, per_rank = reduce_samples(raw)
            vals = [s[2] / s[0] for s in red]
            mi = median_index(vals)
            out["legs"][G] = {"median": vals[mi], "samples": vals, "wall": red[mi][0], "kernel_ms": red[mi][1],
                              "taken": red[mi][2], "enq_ms": legs[G].enq_ms, "arbitration": legs[G].arbitration, "kernel": legs[G].kernel,
                              "per_rank": [pr[mi] for pr in per_rank] if per_rank else None}
            legs[G].close()
        return out

    cands = [Gmax] + [g for g in (2, 1) if g < Gmax and N % g == 0]
    main_m = measure(cfg, cands, lacts=acts, sequence_leg=(a.peds, a.rays) == (20, 360))
    same_m = measure(cfg, [1], lacts=acts, mode="same", repeats=min(R, 3), probe=False)
    def leg_name(g):
        return g if isinstance(g, str) else "%d_groups" % g

    Gc = main_m["chosen"]                              # an int: the envs as that many stream groups (1 = one launch per step)
    G = 1 if isinstance(Gc, str) else Gc
    hl = main_m["legs"][Gc]
    value, wall, kernel_ms, taken_all = hl["median"], hl["wall"], hl["kernel_ms"], hl["taken"]
    one = main_m["legs"].get(1)

    # Sustained: the headline decomposition back to back for >= 6 s, right after its burst samples (the CPU baseline, which takes
    # most of this script's wall time, comes last).  A 20-step sample lasts 0.7 ms; this one shows what the chip holds.
    sustained = None
    if a.sustained_seconds > 0 and (a.peds, a.rays) == (20, 360):
        lg = Leg(cfg, G, lacts=acts)
        lg.run(a.preroll)
        sustained = lg.sustained(a.sustained_seconds)
        lg.close()
        if use_dist:
            t_ = torch.tensor([sustained["env_steps_s"], sustained["seconds"], sustained["clock_mhz"] or 0.0], dtype=torch.float64, device=cdev)
            sm_ = t_.clone(); dist.all_reduce(sm_, op=dist.ReduceOp.SUM)
            mx_ = t_.clone(); dist.all_reduce(mx_, op=dist.ReduceOp.MAX)
            sustained.update({"env_steps_s": float(sm_[0]), "seconds": float(mx_[1]), "clock_mhz": float(sm_[2]) / world})

    # Issue-bound ceiling of this kernel on this GPU, measured in the same run: 16384 resident envs in stream groups
    # (every SIMD has work in every phase).  Single-GPU run only.
    plateau = None
    single_default = world == 1 and (a.peds, a.rays) == (20, 360)
    if single_default and not a.no_plateau:
        import dataclasses
        pcfg = dataclasses.replace(cfg, n_envs=16384)
        pm = measure(pcfg, [Gmax], lacts=open_loop_actions(16384, 99), repeats=min(R, 3), probe=False)
        plateau = pm["legs"][Gmax]["median"]

    # BASELINE configs[2] (TD3 actor in the loop: random-init 398-256-256-2 actor, sigma = 1 exploration, cn_actor_forward ->
    # cn_step chains) and configs[4] (100 pedestrians x 720 rays in the 4.8 m room) under the same bracket and protocol
    other = None
    if single_default and N == 4096 and not a.no_other_configs:
        from crowdnav.td3 import Agent
        other = {}
        agent = Agent(obs_dim=cfg.obs_dim, device="cuda:%d" % dev_index, seed=0, memory_size=16)
        m3 = measure(cfg, list(cands) + ["policy_sequence"], agent=agent, repeats=min(R, 3))
        c5 = Config(n_envs=N, n_peds=100, n_rays=720, k_obstacles=a.k, max_steps=1000, seed=1234, ped_cycle_ms=1400, room_half=2.40)
        m5 = measure(c5, ([Gmax, 1] if Gmax > 1 else [1]) + ["sequence"], lacts=acts, repeats=min(R, 3))
        # BASELINE configs[3]'s per-GPU shard: 16384 envs over 8 GPUs = 2048 envs on this one (the other seven would run the same)
        import dataclasses as _dc
        c3s = _dc.replace(cfg, n_envs=2048)
        m3s = measure(c3s, [g for g in (4, 2, 1) if g <= Gmax], lacts=open_loop_actions(2048, 77), repeats=min(R, 3))
        for key, m, P_, R_, what in (("configs[3]_shard", m3s, 20, 360, "2048 envs x 20 pedestrians x 360 rays, K=8: ONE GPU's shard of BASELINE configs[3] "
                                      "(16384 envs over 8 GPUs), open loop; two wavefronts per environment (cn_env_kernel_s360_x2)"),
                                     ("configs[2]", m3, 20, 360, "4096 envs x 20 pedestrians x 360 rays, K=8, TD3 actor in the loop "
                                      "(f32-MFMA actor + exploration noise -> Env.step, closed loop: policy_sequence = the K periods as ONE "
                                      "cn_rollout_policy launch, the actor inside the step kernel; N_groups = a cn_actor_forward -> cn_step chain per stream group)"),
                                     ("configs[4]", m5, 100, 720, "4096 envs x 100 pedestrians x 720 rays, K=8, room 4.8 m, open loop (sequence = the K steps as "
                                      "ONE cn_step_sequence launch, open-loop only; N_groups = one launch per step and stream group)")):
            l_ = m["legs"][m["chosen"]]
            d4 = d4_bytes(P_, R_, a.k)
            other[key] = {"workload": what, "value": l_["median"], "unit": "env-steps/s", "ms_per_step": l_["wall"] / K * 1e3,
                          "decomposition": ("%d stream group(s)" % m["chosen"]) if isinstance(m["chosen"], int) else m["chosen"],
                          "kernel": l_["kernel"],
                          "samples_env_steps_s": l_["samples"],
                          "legs_env_steps_s": {leg_name(g): v["median"] for g, v in m["legs"].items()},
                          "probe_env_steps_s": m["probe_env_steps_s"],
                          "roofline": {"bound": "hbm", "bytes_per_env_step_d4": d4, "achieved": l_["median"] * d4 / 1e9,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": l_["median"] * d4 / 1e9 / HBM_PEAK_GBS}}

    # The caller of the path: the TD3 update the collected transitions feed -- cn_td3_update (csrc/crowdnav_td3.hip) at the
    # reference's batch of 128, sampled from a 100 000-row device replay of synthetic transitions, 50 untimed + 400 timed updates
    # (tools/learn_bench.py is the same measurement with the PyTorch update beside it).  Not an env-steps figure: reported as flat keys.
    learner = None
    if single_default and N == 4096 and not a.no_other_configs:
        from crowdnav.td3 import Agent as _Agent
        la = _Agent(obs_dim=cfg.obs_dim, device="cuda:%d" % dev_index, seed=0, batch_size=128, memory_size=200000)
        nrep = 100000
        la.memory.add(torch.randn((nrep, cfg.obs_dim), device=la.device), torch.rand((nrep, 2), device=la.device),
                      torch.randn(nrep, device=la.device), torch.randn((nrep, cfg.obs_dim), device=la.device),
                      torch.rand(nrep, device=la.device) < 0.05)
        la.enable_fused_update()
        for i in range(50):
            la.learn(i)
        torch.cuda.synchronize(dev_index)
        t0 = time.perf_counter()
        for i in range(400):
            la.learn(i)
        torch.cuda.synchronize(dev_index)
        dt = (time.perf_counter() - t0) / 400
        learner = {"what": "cn_td3_update, batch 128, obs_dim %d, hidden 256, policy_delay 2 (every other update steps the actor and the targets), "
                           "replay of %d synthetic rows on the device" % (cfg.obs_dim, nrep),
                   "update_ms": dt * 1e3, "updates_s": 1.0 / dt, "launches_per_update": "7 (critic step) / 12 (with the actor step)"}
        del la

    gather = None
    if use_dist:
        # the path's one exchange: all-gather of per-env episode returns over xGMI.  A fresh handle stepped a few
        # hundred launches so that every env has finished episodes to report.
        lg = Leg(cfg, 1, lacts=acts)
        lg.run(a.preroll)
        ret, _ = lg.grp.envs[0].returns()
        torch.cuda.synchronize(dev)
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
        # every rank holds every env's return; rank r's slice must be what rank r computed, and all ranks must agree
        assert torch.equal(gathered[rank * N:(rank + 1) * N], ret)
        gh = gathered.cpu().numpy()
        crc = zlib.crc32(gh.tobytes())
        crcs = torch.tensor([crc], dtype=torch.int64, device=cdev)
        allc = [torch.empty_like(crcs) for _ in range(world)]
        dist.all_gather(allc, crcs)
        seen = torch.tensor([1.0], device=cdev); dist.all_reduce(seen)
        try:
            ver = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            ver = None
        gather = {"ms": gather_ms, "bytes_per_rank": 4 * N, "ranks_seen": int(seen.item()),
                  "crc32_of_gathered_returns": crc, "all_ranks_agree": all(int(c.item()) == crc for c in allc),
                  "sum_of_gathered_returns": float(gh.astype("float64").sum()), "nonzero_returns": int((gh != 0).sum()),
                  "collective": "gloo (dry run)" if dry else "rccl", "rccl_version": ver}
        lg.close()

    B = algorithmic_bytes(a.peds, a.rays, a.k)
    D4 = d4_bytes(a.peds, a.rays, a.k)
    n_launch = N // G                                  # envs per cn_env_kernel launch in the headline leg
    # every launch moves its envs' state, reset or step; G launches are in flight at once, one per stream
    achieved = G * B * n_launch / (kernel_ms * 1e-3) / 1e9
    achieved_d4 = G * D4 * n_launch / (kernel_ms * 1e-3) / 1e9
    # PMC figures of the HEADLINE kernel (and of no other): profiles/rNN/{counters,traffic}.json are keyed by kernel name
    hk = hl["kernel"]
    counters_all, counters_src = profiled_json("counters.json") if (a.peds, a.rays) == (20, 360) else (None, "other workload")
    traffic_all, traffic_src = profiled_json("traffic.json") if (N, a.peds, a.rays) == (4096, 20, 360) else (None, "other workload")
    counters = (counters_all or {}).get("kernels", {}).get(hk)
    traffic = (traffic_all or {}).get("kernels", {}).get(hk)
    if counters_all and not counters:
        counters_src = "%s has no entry for %s" % (counters_src, hk)
    if traffic_all and not traffic:
        traffic_src = "%s has no entry for %s" % (traffic_src, hk)
    # the other configs' kernels, where the committed profile has them (cn_policy_kernel_s360: tools/profile.sh's pol_* passes)
    for oc in (other or {}).values():
        ck = (counters_all or {}).get("kernels", {}).get(oc.get("kernel"))
        tk = (traffic_all or {}).get("kernels", {}).get(oc.get("kernel"))
        if ck:
            oc["roofline"].update({"valu_busy": ck.get("valu_busy"), "wave_instr_per_env_step": ck.get("wave_instr_per_env_step"),
                                   "counters_source": counters_src})
        if tk:
            oc["roofline"].update({"traffic_bytes_per_env_step": tk["bytes_per_env_step"],
                                   "wasted_traffic_ratio": tk["bytes_per_env_step"] / oc["roofline"]["bytes_per_env_step_d4"]})
        if str(oc.get("kernel", "")).startswith("cn_policy_kernel") or "actor" in oc.get("workload", ""):
            # the actor's share priced on the f32 matrix cores (MI355X_MICROARCH.md: 157.3 TF, v_mfma_f32_16x16x4_f32)
            mfma_flops = 2.0 * (((cfg.obs_dim + 31) // 32 * 32) * 256 + 256 * 256)
            oc["roofline"].update({"actor_mfma_flops_per_env_step": mfma_flops,
                                       "frac_mfma_f32": oc["value"] * mfma_flops / (VECTOR_PEAK_F32_TFLOPS * 1e12)})
    steps_per_launch = 1
    # HBM bytes the counters saw per launch of the headline kernel, scaled to this run's launch (envs per launch x steps per launch)
    traffic_b = float(traffic["bytes_per_env_step"]) * n_launch * steps_per_launch if traffic else None
    flops_d5 = 79e3 if (a.peds, a.rays) == (20, 360) else 623e3 if (a.peds, a.rays) == (100, 720) else None
    out = {
        "metric": baseline_metric(),
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": wall / K * 1e3, "higher_is_better": True, "scaling": "strong" if a.envs_total else "weak",
        "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d envs/GPU x %d pedestrians x %d rays, K=%d, lidar-tracker risk features, "
                               "next-step auto-reset (reset launches not counted as env-steps), open-loop "
                               "U(0,0.22)xU(-2,2) actions, %d untimed pre-roll steps; one step = every env stepped once by ONE cn_step "
                               "launch (SURVEY 8d D1: the decomposition that can serve a policy in the loop), "
                               "%s; median of %d samples of %d steps; config.sustained_env_steps_s = the same decomposition for >= 6 s; the "
                               "open-loop single-launch forms are comparables only: config.leg_sequence_env_steps_s (in place), "
                               "config.leg_sequence_traj_env_steps_s (into trajectory buffers)" % (
                                   ("BASELINE configs[3] shape, %d envs total over %d GPU(s) (strong scaling)" % (a.envs_total, world))
                                   if a.envs_total else ("BASELINE configs[1] per GPU (weak scaling over %d GPU(s))" % world),
                                   N, a.peds, a.rays, a.k, a.preroll,
                                   "the envs running as %d independent stream group(s) of %d" % (G, n_launch), R, K),
                   "envs_per_gpu": N, "envs_total": N * world, "stream_groups": G, "stream_groups_requested": Gmax,
                   "decomposition": leg_name(Gc),
                   "repeats": R, "samples_env_steps_s": hl["samples"],
                   "headline_choice": {"rule": "best untimed probe of K steps among the decompositions that can serve a policy in the loop "
                                               "(one cn_step launch per step, alone or as stream groups), taken before the timed samples; "
                                               "cn_step_sequence legs are never eligible",
                                       "eligible": main_m["eligible"],
                                       "probe_env_steps_s": main_m["probe_env_steps_s"], "chosen": leg_name(Gc)},
                   "legs_env_steps_s": {leg_name(g): v["median"] for g, v in main_m["legs"].items()},
                   "legs_samples_env_steps_s": {leg_name(g): v["samples"] for g, v in main_m["legs"].items()},
                   # include/crowdnav.h cn_set_arbitration: which issue order each leg's kernel ran with
                   "legs_arbitration": {leg_name(g): v["arbitration"] for g, v in main_m["legs"].items()},
                   # host time the enqueue loop needs per step of a leg (a leg is host-paced when this approaches ms_per_step)
                   "host_enqueue_ms_per_step": {leg_name(g): v["enq_ms"] for g, v in main_m["legs"].items()},
                   "concurrent_hw_queues_found": conc, "parallelism": "env-sharded x%d" % world,
                   "one_launch_per_step_value": one["median"] if one else None,
                   "one_launch_per_step_ms": one["wall"] / K * 1e3 if one else None,
                   "same_call_reset_value": same_m["legs"][1]["median"],
                   "same_call_reset_ms_per_step": same_m["legs"][1]["wall"] / K * 1e3,
                   "per_rank": ([{"rank": r_, "wall_ms": pr[0] * 1e3, "env_steps": pr[2], "value": pr[2] / pr[0]}
                                 for r_, pr in enumerate(hl["per_rank"])] if hl["per_rank"] else None),
                   "returns_allgather": gather,
                   "other_configs": other,
                   "sustained": sustained},
        "roofline": {"bound": "hbm", "achieved": achieved_d4, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     # priced with SURVEY 8(d) D4's per-env-step figure (float32 state), the contract's definition
                     "frac": achieved_d4 / HBM_PEAK_GBS, "bytes_per_env_step": D4,
                     # the same speed priced with the bytes this build's float64 state layout really has to move
                     "achieved_f64_layout": achieved, "frac_f64_layout": achieved / HBM_PEAK_GBS, "bytes_per_env_step_f64_layout": B,
                     "traffic": traffic_b,
                     "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE+WRITE_SIZE, calibrated; profiles/)",
                     "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": D4 * n_launch * steps_per_launch, "envs_per_launch": n_launch,
                     "concurrent_launches": G,
                     # the kernel every figure of this block belongs to (cn_kernel_name of the headline leg's handle); for the sequence
                     # kernel kernel_ms = launch duration / K and the counters are per control period
                     "kernel": hk, "kernel_ms": kernel_ms, "steps_per_launch": steps_per_launch,
                     "wasted_traffic_ratio": (traffic_b / (D4 * n_launch * steps_per_launch)) if traffic_b else None,
                     "wasted_traffic_ratio_f64_layout": (traffic_b / (B * n_launch * steps_per_launch)) if traffic_b else None,
                     # the same speed priced in flops: SURVEY 8(d) D5 x env-steps/s over the FLOAT64 vector peak (the path computes in
                     # f64; rounds 3-4 divided by the packed-f32 peak, twice too generous) -- the f32 figure stays beside it
                     "flops_per_env_step_d5": flops_d5,
                     "frac_valu_f64": (value / world * flops_d5 / (VECTOR_PEAK_F64_TFLOPS * 1e12)) if flops_d5 else None,
                     "vector_peak_f64_tflops": VECTOR_PEAK_F64_TFLOPS,
                     "frac_valu_f32_peak": (value / world * flops_d5 / (VECTOR_PEAK_F32_TFLOPS * 1e12)) if flops_d5 else None,
                     "vector_peak_f32_tflops": VECTOR_PEAK_F32_TFLOPS,
                     "legs_kernels": {leg_name(g): v["kernel"] for g, v in main_m["legs"].items()},
                     # what actually binds: instruction issue (float64 VALU) and one wavefront's critical path, not HBM
                     "binding": "instruction issue (f64 VALU) above ~8k resident envs; one wavefront's critical path at 4096",
                     "issue_bound_env_steps_s": plateau,
                     "frac_of_issue_bound": (value / world / plateau) if plateau else None,
                     "valu_busy": (counters or {}).get("valu_busy"),
                     "wave_instr_per_env_step": (counters or {}).get("wave_instr_per_env_step"),
                     "counters_source": counters_src,
                     "note": "achieved = concurrent_launches x algorithmic bytes per launch / mean launch duration on "
                             "its own stream (HIP events per group stream, the headline's median sample)",
                     "one_launch_per_step": ({"envs_per_launch": N, "kernel": one["kernel"], "kernel_ms": one["kernel_ms"],
                                              "traffic": (float(traffic_all["kernels"][one["kernel"]]["bytes_per_launch"])
                                                          if traffic_all and one["kernel"] in traffic_all.get("kernels", {}) else None),
                                              "valu_busy": ((counters_all or {}).get("kernels", {}).get(one["kernel"]) or {}).get("valu_busy"),
                                              "wave_instr_per_env_step": ((counters_all or {}).get("kernels", {}).get(one["kernel"]) or {}).get("wave_instr_per_env_step"),
                                              "achieved": D4 * N / (one["kernel_ms"] * 1e-3) / 1e9,
                                              "frac": D4 * N / (one["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS} if one else None)},
    }
    # Flat scalar copies (record keepers that drop nested objects keep these): every leg, the other configs, the sustained leg
    fc = out["config"]
    for g, v in main_m["legs"].items():
        fc["leg_%s_env_steps_s" % leg_name(g)] = v["median"]
        fc["leg_%s_kernel" % leg_name(g)] = v["kernel"]
    fc["plateau_16384_envs_env_steps_s"] = plateau
    for leg_, key_ in (("sequence", ""), ("sequence_traj", "+traj")):         # HBM bytes the counters saw per env-step of the open-loop legs
        lk_ = main_m["legs"].get(leg_)
        tk_ = (traffic_all or {}).get("kernels", {}).get(lk_["kernel"] + key_) if lk_ else None
        if tk_:
            fc["leg_%s_traffic_bytes_per_env_step" % leg_] = tk_["bytes_per_env_step"]
    for key, oc in (other or {}).items():
        k_ = key.replace("[", "").replace("]", "")
        fc["%s_env_steps_s" % k_] = oc["value"]
        fc["%s_ms_per_step" % k_] = oc["ms_per_step"]
        fc["%s_decomposition" % k_] = oc["decomposition"]
        fc["%s_kernel" % k_] = oc["kernel"]
        fc["%s_roofline_frac" % k_] = oc["roofline"]["frac"]
        for g, v in oc["legs_env_steps_s"].items():
            fc["%s_leg_%s_env_steps_s" % (k_, g)] = v
    if learner:
        fc["learner"] = learner["what"]
        fc["learner_td3_update_ms"] = learner["update_ms"]
        fc["learner_td3_updates_s"] = learner["updates_s"]
        fc["learner_launches_per_update"] = learner["launches_per_update"]
    if sustained:
        fc["sustained_env_steps_s"] = sustained["env_steps_s"]
        fc["sustained_seconds"] = sustained["seconds"]
        fc["sustained_clock_mhz"] = sustained["clock_mhz"]
        fc["sustained_clock_mhz_idle"] = sustained.get("clock_mhz_idle")
        fc["burst_over_sustained"] = value / sustained["env_steps_s"]
        out["roofline"]["sustained_achieved"] = sustained["env_steps_s"] / world * D4 / 1e9
        out["roofline"]["sustained_frac"] = sustained["env_steps_s"] / world * D4 / 1e9 / HBM_PEAK_GBS
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
                orc.step(acts_c[k % N_ACT], auto_reset=True); k += 1
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
                                                                                           ncpu, tallc),
                               # the reference's own Python (environment_stage_1_nobonus.py under oracle/harness, sleeps
                               # virtualised) cannot travel to the GPU box; BASELINE.md's figure, measured in the container
                               "reference_python": {"value": 102.0, "unit": "env-steps/s", "cores": 1, "where": "container",
                                                    "source": "BASELINE.md"}}
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
