#!/usr/bin/env python
"""Randomised GPU-vs-oracle soak over the CROSS PRODUCT of the configuration switches (diagnostic; the assertions live in tests/).

tools/parity_sweep.sh walks 37 hand-picked configurations and tests/test_gpu_parity.py::test_randomised_configurations a dozen random
geometries of the default world; this draws worlds from every switch cn_create accepts at once -- observation layout x risk mode x
pedestrian model x contact ticks x Python-2 rounding x GEOS <= 3.8 semantics x float32 scans x wheel ramp x reward variant x K x
pedestrians x rays x room x clocks x reset convention -- and drives each through one of the three launch forms (cn_step per step,
cn_step_sequence with trajectory buffers, cn_rollout_policy closed loop), small grids (two wavefronts per environment) and larger
ones, until the time budget is spent.  Every observation / reward / done flag / top-K index / counter is compared; a configuration
cn_create refuses is counted and named, not an error.
    python tools/fuzz_parity.py [--seconds 300] [--seed 1] [--verbose 3]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))


HEADLINE_FRAC = 0.0


def draw(rng):
    layout = int(rng.choice([0, 0, 0, 1, 2]))
    risk = int(rng.choice([0, 0, 1])) if layout == 0 else 0
    ped_mode = int(rng.choice([0, 0, 2]))
    R = int(rng.choice([int(rng.integers(8, 130)), 181, 360, 360, 361, int(rng.integers(130, 800)), 720, 1025]))
    P = int(rng.choice([0, 1, int(rng.integers(2, 30)), 20, int(rng.integers(30, 70)), int(rng.integers(70, 140))]))
    room = float(rng.uniform(1.0, 3.0)) if P < 60 else float(rng.uniform(2.0, 3.2))
    kw = dict(n_envs=int(rng.choice([1, 3, 8, 17, 64, 130, 300])), n_peds=P, n_rays=R,
              k_obstacles=int(rng.integers(1, 17)), max_steps=int(rng.integers(6, 60)), room_half=room,
              obs_layout=layout, risk_mode=risk, ped_mode=ped_mode, ped_contact=int(rng.choice([0, 0, 1])) if ped_mode == 0 else 0,
              py2_round=int(rng.choice([0, 0, 1])), geos_untyped_empty=int(rng.choice([0, 0, 1])),
              scan_f32=int(rng.choice([0, 0, 1])), wheel_accel=float(rng.choice([0.0, 0.0, 1.0, 2.5])),
              waypoint_reward=int(rng.choice([200, 200, 0])), sf_tick_ms=int(rng.choice([0, 0, 50])) if ped_mode == 2 else 0,
              goal_x=float(rng.uniform(-0.9, 0.9)), goal_y=float(rng.uniform(-0.9, 0.9)),
              spawn_x=float(rng.uniform(-0.7, 0.7)), spawn_y=float(rng.uniform(-0.7, 0.7)), spawn_yaw=float(rng.uniform(-3.1, 3.1)),
              dt_ms=int(rng.choice([50, 100, 150, 150, 200])), scan_latency_ms=int(rng.choice([5, 10, 10, 20])),
              settle_ms=int(rng.choice([0, 50, 100, 100])), ped_cycle_ms=int(rng.choice([0, 300, 700, 1400, 2000])),
              ped_vmax=float(rng.uniform(0.05, 0.5)), min_scan_range=float(rng.choice([0.0, 0.12, 0.12])),
              seed=int(rng.integers(1, 1 << 30)) if rng.random() < 0.8 else int(rng.integers(1 << 40, 1 << 62)),
              env_index_base=int(rng.integers(0, 1 << 20)) if rng.random() < 0.8 else int(rng.integers(1 << 33, 1 << 45)))
    if rng.random() < 0.5:          # the sensor, the bodies and the goal geometry too (every second world keeps the reference's)
        kw.update(lidar_max=float(rng.choice([0.6, 0.6, 1.0, 3.5])), max_scan_range=float(rng.choice([0.6, 0.6, 0.5, 1.0])),
                  lidar_min=float(rng.choice([0.08, 0.08, 0.0, 0.12])), lidar_span=float(rng.choice([6.28, 6.28, 3.14, 4.0])),
                  lidar_offset_x=float(rng.choice([-0.032, -0.032, 0.0, 0.05])), ped_radius=float(rng.choice([0.0505, 0.0505, 0.1, 0.03])),
                  robot_clearance=float(rng.choice([0.09, 0.09, 0.15])), waypoint_radius=float(rng.choice([0.3, 0.3, 0.2, 0.5])),
                  goal_eps=float(rng.choice([0.2, 0.2, 0.1, 0.4])), start_x=float(rng.uniform(-1, 1)), start_y=float(rng.uniform(-1, 1)),
                  track_capacity=int(rng.choice([0, 0, 64])), ped_stagger_ms=int(rng.choice([100, 100, 50, 200])))
    if rng.random() < HEADLINE_FRAC:
        # the two BENCHMARKED shapes, whose kernels are compiled for exactly these sizes (cn_env_kernel[_fair]_s360[_w4|_x2], _seq_s360,
        # cn_policy_kernel_s360 and the _s720 family): everything that keeps a world on them stays random, and the grid sizes walk
        # through the selection rules of choose_kernel (two wavefronts per env up to 2048 envs, 4 envs per workgroup up to 4096, ...)
        dense = rng.random() < 0.3
        kw.update(n_peds=100 if dense else 20, n_rays=720 if dense else 360, k_obstacles=8, obs_layout=0, risk_mode=0, ped_mode=0, ped_contact=0,
                  wheel_accel=0.0, sf_tick_ms=0, track_capacity=0, room_half=float(rng.uniform(2.0, 3.2)) if dense else float(rng.uniform(1.0, 3.0)),
                  n_envs=int(rng.choice([1, 17, 64, 300, 1000, 2100, 4100] if not dense else [1, 17, 64, 300, 1000])))
    form = str(rng.choice(["step", "step", "sequence", "policy", "manual"]))
    mode = "next" if form != "step" else str(rng.choice(["next", "same"]))
    return kw, form, mode


def diff(tag, g, c):
    """where two arrays differ: first env, its columns, both values"""
    badm = np.argwhere(g != c)
    if not len(badm):
        return
    e = int(badm[0][0]); cols = np.nonzero(np.atleast_1d(g[e] != c[e]))[0]
    print("   %s: %d entries in %d env(s) differ; env %d cols %s\n      gpu %s\n      cpu %s"
          % (tag, len(badm), len(set(badm[:, 0])), e, cols[:16], np.atleast_1d(g[e])[cols[:8]], np.atleast_1d(c[e])[cols[:8]]))


def run_world(kw, form, mode, T, detail=False, as_steps=False):
    """one world through one launch form -> (list of differences (empty = equal), kernel name, 'refused' / None).  Actions and the
    launch length come from a generator seeded with the world's own seed, so (kw, form, mode) reproduces the run."""
    import torch
    from crowdnav import Config
    from crowdnav.env import VecEnv
    from crowdnav.td3 import Agent
    from oracle import oracle
    cfg = Config(**kw)
    env = VecEnv(cfg)                               # cn_create's own validation raises here
    rng = np.random.default_rng(int(kw["seed"]) ^ 0x5eed)
    N = cfg.n_envs
    orc = oracle.Oracle(cfg.as_dict())
    env.enable_f64_obs()
    what = "policy" if form == "policy" else "sequence" if form == "sequence" else "same" if (mode == "same" and form == "step") else "step"
    kn = env.kernel_name(what)
    env.reset(); torch.cuda.synchronize()
    bad, skipped, bad_envs = [], None, set()

    def check(tag, g, c):
        if not np.array_equal(g, c):
            bad.append(tag)
            bad_envs.update(int(x) for x in np.nonzero((g != c).reshape(g.shape[0], -1).any(1))[0])
            if detail:
                diff(tag, g, c)
    check("reset observation", env.obs_f64.cpu().numpy(), orc.reset())
    if form == "step":
        t_snap = int(rng.integers(1, T)) if rng.random() < 0.4 else -1      # 40 %: snapshot mid-run, continue on a FRESH handle restored from it
        for t in range(T):
            if t == t_snap:
                blob = env.snapshot()
                env2 = VecEnv(cfg); env2.enable_f64_obs(); env2.restore(blob)
                env.close(); env = env2
            act = np.stack([rng.uniform(0, 0.22, N), rng.uniform(-2, 2, N)], 1).astype(np.float32)
            env.step(torch.from_numpy(act).cuda(), auto_reset=mode); torch.cuda.synchronize()
            oc, rc, dc, ic = orc.step(act.astype(np.float64), auto_reset=mode)
            check("obs @%d" % t, env.obs_f64.cpu().numpy(), oc)
            check("reward @%d" % t, env.reward.cpu().numpy(), rc.astype(np.float32))
            check("done @%d" % t, env.done.cpu().numpy(), dc)
            check("idx @%d" % t, env.topk_idx.cpu().numpy(), ic)
            if bad:
                break
    elif form == "manual":
        # the reference's own loop (TRAIN:106-166): no auto-reset, an explicit step counter per environment, Env.reset() for the
        # environments that finished (cn_reset with a mask) -- plus a few that did not (a trainer may cut an episode short)
        sc = np.zeros(N, dtype=np.int32)
        for t in range(T):
            act = np.stack([rng.uniform(0, 0.22, N), rng.uniform(-2, 2, N)], 1).astype(np.float32)
            env.step(torch.from_numpy(act).cuda(), step_counter=sc, auto_reset=False); torch.cuda.synchronize()
            oc, rc, dc, ic = orc.step(act.astype(np.float64), step_counter=sc, auto_reset=False)
            check("obs @%d" % t, env.obs_f64.cpu().numpy(), oc)
            check("reward @%d" % t, env.reward.cpu().numpy(), rc.astype(np.float32))
            check("done @%d" % t, env.done.cpu().numpy(), dc)
            check("idx @%d" % t, env.topk_idx.cpu().numpy(), ic)
            if bad:
                break
            mask = (dc != 0) | (rng.random(N) < 0.02)
            sc += 1
            if mask.any():
                env.reset(mask=mask); torch.cuda.synchronize()
                orst = orc.reset(mask=mask)
                full_g, full_c = env.obs_f64.cpu().numpy(), orst
                rows_bad = np.nonzero(mask & (full_g != full_c).any(1))[0]
                if len(rows_bad):
                    bad.append("masked reset @%d" % t); bad_envs.update(int(x) for x in rows_bad)
                    if detail:
                        e_ = int(rows_bad[0]); cols_ = np.nonzero(full_g[e_] != full_c[e_])[0]
                        print("   masked reset @%d: env %d (done before the reset: %d, step counter %d) cols %s\n      gpu %s\n      cpu %s" % (t, e_, int(dc[e_]), int(sc[e_]), cols_[:16], full_g[e_][cols_[:8]], full_c[e_][cols_[:8]]))
                        dg, dc__ = env.debug_env(e_), orc.debug(e_)
                        for k_ in ("n_tracks", "n_confirmed", "n_entries", "collision_prob", "ego_score", "status"):
                            print("      %-14s gpu %r cpu %r" % (k_, dg[k_], dc__[k_]))
                        print("      track_pose gpu", np.asarray(dg["track_pose"]).tolist(), "\n      track_pose cpu", np.asarray(dc__["track_pose"]).tolist())
                        print("      track_vel gpu", np.asarray(dg["track_vel"]).tolist(), "\n      track_vel cpu", np.asarray(dc__["track_vel"]).tolist())
                        print("      track_t gpu", np.asarray(dg["track_t"]).tolist(), "\n      track_t cpu", np.asarray(dc__["track_t"]).tolist())
                        print("      entry_cp cpu", np.asarray(dc__["entry_cp"]).tolist())
                sc[mask] = 0
                if bad:
                    break
    else:
        Tc = int(rng.choice([1, 5, 12, 20]))
        traj = dict(action=torch.zeros((Tc, N, 2), device="cuda"), obs=torch.zeros((Tc, N, env.D), device="cuda"),
                    reward=torch.zeros((Tc, N), device="cuda"), done=torch.zeros((Tc, N), dtype=torch.uint8, device="cuda"))
        agent = Agent(obs_dim=cfg.obs_dim, device="cuda:0", seed=int(kw["seed"]) & 0xffff, memory_size=16) if form == "policy" else None
        for t0 in range(0, T, Tc):
            if form == "policy":
                try:
                    env.rollout_policy(agent, Tc, traj=traj)
                except Exception as ex:             # a shape of which not even 8 environments fit one CU's LDS has no policy kernel
                    if t0 or "LDS" not in str(ex):
                        raise
                    skipped = "refused"
                    break
            else:
                A_ = np.stack([rng.uniform(0, 0.22, (Tc, N)), rng.uniform(-2, 2, (Tc, N))], 2).astype(np.float32)
                traj["action"].copy_(torch.from_numpy(A_))
                if as_steps:                        # the same actions through cn_step, one launch per step (is it the sequence kernel or the world?)
                    for t in range(Tc):
                        env.step(traj["action"][t], auto_reset="next"); torch.cuda.synchronize()
                        traj["obs"][t].copy_(env.obs); traj["reward"][t].copy_(env.reward); traj["done"][t].copy_(env.done)
                        if detail and os.environ.get("CN_FUZZ_DUMP") and (t0 + t) == int(os.environ["CN_FUZZ_DUMP"].split(",")[0]):
                            e_ = int(os.environ["CN_FUZZ_DUMP"].split(",")[1])
                            dump_gpu = env.debug_env(e_)
                            print("   GPU debug env %d step %d:" % (e_, t0 + t), {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in dump_gpu.items()})
                else:
                    env.bind_step_sequence(traj["action"], traj={k: traj[k] for k in ("obs", "reward", "done")})()
            torch.cuda.synchronize()
            A = traj["action"].cpu().numpy(); O = traj["obs"].cpu().numpy(); Rw = traj["reward"].cpu().numpy(); Dn = traj["done"].cpu().numpy()
            for t in range(Tc):
                oc, rc, dc, ic = orc.step(A[t].astype(np.float64), auto_reset="next")
                check("obs @%d" % (t0 + t), O[t], oc.astype(np.float32))
                check("reward @%d" % (t0 + t), Rw[t], rc.astype(np.float32))
                check("done @%d" % (t0 + t), Dn[t], dc)
                if bad:
                    break
            if bad:
                break
    if not bad and not skipped:
        check("counters", env.counters().cpu().numpy()[:, :6], orc.counters())
    if bad and detail and form == "step":
        try:
            e = 0
            dg, dc_ = env.debug_env(e), orc.debug(e)
            print("   env 0: tracks %d/%d nconf %d/%d bb %r/%r status %d/%d" % (dg["n_tracks"], dc_["n_tracks"], dg["n_confirmed"], dc_["n_confirmed"], dg["bb"], dc_["bb"], dg["status"], dc_["status"]))
        except Exception as ex:
            print("   (debug_env: %s)" % ex)
    if bad:
        # The reference's track list and confirmed-object list grow without bound (Python lists); the kernel's tables hold
        # track_capacity (32 / 64) tracks and max_conf objects and RAISE A STATUS BIT when a world outgrows them (CN_ST_TRACK_OVERFLOW 1,
        # CN_ST_CONF_OVERFLOW 8, include/crowdnav.h).  A difference in an environment whose status carries one of the two is that
        # documented limit, not a disagreement about the arithmetic.
        st = env.counters().cpu().numpy()[:, 6].astype(np.int64)
        flagged = [e for e in sorted(bad_envs) if st[e] & 9]
        if bad_envs and len(flagged) == len(bad_envs):
            skipped = "overflow"
        if detail:
            print("   differing envs %s; status words %s (1 = track table full, 8 = confirmed-object table full); tracks %s"
                  % (sorted(bad_envs)[:8], [int(st[e]) for e in sorted(bad_envs)[:8]], [int(x) for x in env.counters().cpu().numpy()[sorted(bad_envs)[:8], 7]]))
    env.close()
    return bad, kn, skipped


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--verbose", type=int, default=3)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--headline-frac", type=float, default=0.0, help="fraction of the worlds drawn on the two benchmarked shapes (their specialised kernels)")
    ap.add_argument("--repro", default=None, help="a world as printed by a MISMATCH line: \"{'n_envs': ...}\"; with --form / --mode")
    ap.add_argument("--form", default="step")
    ap.add_argument("--mode", default="next")
    ap.add_argument("--as-steps", action="store_true", help="--repro of a sequence world: the same actions through cn_step, one launch per step")
    a = ap.parse_args()
    from oracle import oracle
    oracle.set_num_threads()
    global HEADLINE_FRAC
    HEADLINE_FRAC = a.headline_frac
    if a.repro:
        import ast
        kw = ast.literal_eval(a.repro)
        bad, kn, skipped = run_world(kw, a.form, a.mode, a.steps, detail=True, as_steps=a.as_steps)
        print("repro (%s, %s, kernel %s): %s%s" % (a.form, a.mode, kn, bad or "no difference in %d steps" % a.steps, " [table overflow, flagged]" if skipped == "overflow" else ""))
        sys.exit(1 if bad else 0)
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    n_trials = n_refused = n_bad = n_form_refused = n_overflow = 0
    env_steps = 0
    kernels = {}
    forms = {}
    while time.time() < t_end:
        kw, form, mode = draw(rng)
        steps = a.steps if rng.random() < 0.8 else 4 * a.steps       # every fifth world four times as long (tracks persist across resets)
        try:
            bad, kn, skipped = run_world(kw, form, mode, steps)
        except Exception as ex:                     # cn_create's own validation (the message names the field)
            if "cn_create" not in str(ex):
                raise
            n_refused += 1
            if n_refused <= 6:
                print("refused: %s | %s" % (str(ex).splitlines()[-1][:170], {k: kw[k] for k in ("n_peds", "n_rays", "k_obstacles", "obs_layout", "risk_mode", "ped_mode")}))
            continue
        n_trials += 1
        kernels[kn] = kernels.get(kn, 0) + 1
        forms[form] = forms.get(form, 0) + 1
        if skipped == "refused":
            n_form_refused += 1
            continue
        if skipped == "overflow":
            n_overflow += 1
            env_steps += kw["n_envs"] * steps
            if n_overflow <= 3:
                print("  table overflow (flagged in the status word) %d: %s %s | %s" % (n_overflow, kn, bad[:2], {k: kw[k] for k in ("n_envs", "n_peds", "n_rays", "k_obstacles", "max_steps")}))
                if a.verbose > 3:
                    print("     --form %s --mode %s --steps %d --repro \"%r\"" % (form, mode, steps, kw))
            continue
        env_steps += kw["n_envs"] * steps
        if bad:
            n_bad += 1
            print("  bad world %d: %s %s %s %s | --form %s --mode %s --steps %d --repro \"%r\"" % (n_bad, form, mode, kn, bad[:3], form, mode, steps, kw))
    print("fuzz_parity: seed %d, %.0f s: %d worlds (%d more refused by cn_create), %d env-steps compared, %d world(s) with a difference; %d more outgrew the track / confirmed-object tables (status bit raised); cn_rollout_policy refused %d shape(s) for LDS"
          % (a.seed, a.seconds, n_trials, n_refused, env_steps, n_bad, n_overflow, n_form_refused))
    print("  launch forms:", dict(sorted(forms.items())))
    print("  kernels exercised (%d):" % len(kernels), dict(sorted(kernels.items(), key=lambda kv: -kv[1])))
    sys.exit(1 if n_bad else 0)


if __name__ == "__main__":
    main()
