#!/usr/bin/env python
"""Randomised soak of the EXTERNAL-sensor path (cn_observe_external: Env.get_state + compute_reward on /scan and /odom messages that
Gazebo or a physical robot delivers) against the oracle's restatement fed with the same messages, call by call (diagnostic; the
assertions live in tests/: test_golden_replay_through_the_kernel replays the 3 272 recorded calls of the reference's own runs,
test_external_scans_with_nan_zero_inf_and_out_of_range_values 80 synthetic ones).

Each world: N robots in their own little scenes -- a few discs drifting on straight lines, a wall or two -- ray-cast by a few lines of
numpy (any scan is a valid message; the scenes only make the segmentation / tracker / collision cone see coherent objects), then
degraded the way sensors do: NaN / 0.0 / +inf drop-outs, returns beyond max_scan_range and below lidar_min, float32 rounding, whole
scans of one value; odometry with jitter and jumps, clocks that repeat or leap, episodes that end by the step counter.  The
configuration is drawn too: observation layout 0 / 1 / 2, K, rays, Python-2 rounding, GEOS <= 3.8 semantics, min_scan_range, goal.
Compared per call: observation (float64), reward, done flag, top-K indices (layout 0), safety counters, status bits.
    python tools/fuzz_external.py [--seconds 120] [--seed 1] [--envs 32] [--calls 150]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "drl-based-mapless-crowd-navigation-with-perceived-risk_amd"))


def scene_scan(rng, R, span, pose, discs, walls, maxr):
    """ranges [R] of one robot: rays k * span / (R - 1) + yaw against discs [(x, y, r)] and axis-aligned walls [(axis, coordinate)]"""
    ang = pose[2] + np.arange(R) * (span / (R - 1))
    dx, dy = np.cos(ang), np.sin(ang)
    t = np.full(R, np.inf)
    for (cx, cy, rad) in discs:
        ox, oy = cx - pose[0], cy - pose[1]
        b = ox * dx + oy * dy
        disc = b * b - (ox * ox + oy * oy - rad * rad)
        hit = (disc >= 0) & (b > 0)
        th = b - np.sqrt(np.where(hit, disc, 0.0))
        t = np.where(hit & (th > 0), np.minimum(t, th), t)
    for (axis, c) in walls:
        d = dx if axis == 0 else dy
        o = pose[0] if axis == 0 else pose[1]
        with np.errstate(divide="ignore", invalid="ignore"):
            tw = (c - o) / d
        t = np.where((d != 0) & (tw > 0), np.minimum(t, tw), t)
    return np.where(t > maxr * rng.uniform(1.0, 1.6), np.inf, t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--envs", type=int, default=32)
    ap.add_argument("--calls", type=int, default=150)
    ap.add_argument("--verbose", type=int, default=4)
    a = ap.parse_args()
    import torch
    from crowdnav import Config
    from crowdnav.env import VecEnv
    from oracle import oracle
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    worlds = calls = bad_worlds = soft = n_overflow = n_dtzero = 0
    shown = 0
    kernels = {}
    N = a.envs
    while time.time() < t_end:
        layout = int(rng.choice([0, 0, 1, 2]))
        R = int(rng.choice([181, 360, 360, 500, 720]))
        kw = dict(n_envs=N, n_peds=0, n_rays=R, k_obstacles=int(rng.integers(1, 17)), max_steps=int(rng.integers(10, 80)), obs_layout=layout,
                  dt_ms=50 if layout == 2 else int(rng.choice([100, 150, 200])), py2_round=int(rng.choice([0, 0, 1])),
                  geos_untyped_empty=int(rng.choice([0, 0, 1])), min_scan_range=float(rng.choice([0.0, 0.12, 0.12])),
                  goal_x=float(rng.uniform(-1.5, 1.5)), goal_y=float(rng.uniform(-1.5, 1.5)), seed=int(rng.integers(1, 1 << 30)),
                  waypoint_reward=int(rng.choice([200, 0])), track_capacity=int(rng.choice([64, 64, 0])))
        try:
            cfg = Config(**kw); env = VecEnv(cfg)
        except Exception as ex:
            print("refused:", str(ex)[-150:]); continue
        env.enable_f64_obs()
        orc = oracle.Oracle(cfg.as_dict())
        kn = env.kernel_name("external"); kernels[kn] = kernels.get(kn, 0) + 1
        worlds += 1
        maxr, span = cfg.max_scan_range, cfg.lidar_span
        pose = np.stack([rng.uniform(-1, 1, N), rng.uniform(-1, 1, N), rng.uniform(-3.1, 3.1, N)], 1)
        nd = rng.integers(0, 7, N)
        dpos = [np.stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.5, 1.5, n)], 1) for n in nd]
        dvel = [rng.uniform(-0.4, 0.4, (n, 2)) * (rng.random((n, 1)) < 0.7) for n in nd]
        drad = [rng.choice([0.0505, 0.1, 0.178, 0.03], n) for n in nd]
        walls = [[(int(rng.integers(0, 2)), float(rng.choice([-1.4, 1.4, -0.8, 2.0]))) for _ in range(int(rng.integers(0, 3)))] for _ in range(N)]
        now = np.full(N, 10.0) + rng.uniform(0, 5, N)
        sc = np.zeros(N, dtype=np.int64)
        bad = None
        for i in range(a.calls):
            is_reset = (i == 0) or (rng.random() < 0.03)
            if is_reset:
                sc[:] = 0
            dt = cfg.dt_ms / 1000.0
            v, w = rng.uniform(0, 0.22, N), rng.uniform(-2, 2, N)
            # the robots drive (roughly) what they were told; odometry adds jitter, now and then a jump
            pose[:, 2] += w * dt
            pose[:, 0] += v * np.cos(pose[:, 2]) * dt + rng.normal(0, 0.002, N); pose[:, 1] += v * np.sin(pose[:, 2]) * dt + rng.normal(0, 0.002, N)
            jump = rng.random(N) < 0.01
            pose[jump, :2] += rng.uniform(-0.5, 0.5, (int(jump.sum()), 2))
            pose[:, 2] = (pose[:, 2] + np.pi) % (2 * np.pi) - np.pi
            step_t = np.where(rng.random(N) < 0.04, 0.0, dt + rng.normal(0, 0.004, N))      # 4 %: the clock repeats
            step_t = np.where(rng.random(N) < 0.01, 3.0, step_t)                              # 1 %: it leaps
            now += np.maximum(step_t, 0.0)
            ranges = np.empty((N, R))
            for e in range(N):
                dpos[e] += dvel[e] * dt
                r = scene_scan(rng, R, span, pose[e], [(dpos[e][j, 0], dpos[e][j, 1], drad[e][j]) for j in range(nd[e])], walls[e], maxr)
                kind = rng.random()
                if kind < 0.02: r[:] = np.inf
                elif kind < 0.04: r[:] = 0.0
                elif kind < 0.05: r[:] = np.nan
                elif kind < 0.07: r[:] = rng.uniform(0.1, 0.7)
                else:
                    m = rng.random(R)
                    p = rng.choice([0.0, 0.0, 0.005, 0.02, 0.1])
                    r[m < p] = np.nan; r[(m >= p) & (m < 2 * p)] = 0.0; r[(m >= 2 * p) & (m < 3 * p)] = np.inf
                    r = r + np.where(np.isfinite(r), rng.normal(0, rng.choice([0.0, 0.0, 0.001, 0.01]), R), 0.0)
                    if rng.random() < 0.5: r = r.astype(np.float32).astype(np.float64)
                    if rng.random() < 0.3: r = np.where(np.isfinite(r), np.round(r, 3), r)    # exact thousandths: rounding ties
                ranges[e] = np.where(r < 0, 0.0, r)      # (a negative range is not a LaserScan value; drivers report 0 / inf / NaN for no return)
            dq = pose[:, :2] + rng.normal(0, 0.001, (N, 2))
            end_ts = np.where(rng.random(N) < 0.02, 0.0, dt + rng.normal(0, 0.002, N))
            oc = np.zeros((N, cfg.obs_dim)); rc = np.zeros(N); dc = np.zeros(N, dtype=bool); ic = np.zeros((N, cfg.k_obstacles), dtype=np.int32)
            for e in range(N):
                inp = dict(deque_x=float(dq[e, 0]), deque_y=float(dq[e, 1]), end_timestep=float(end_ts[e]), px=float(pose[e, 0]), py=float(pose[e, 1]),
                           yaw=float(pose[e, 2]), v=float(v[e]), w=float(w[e]), now=float(now[e]), step_counter=int(sc[e]), is_reset=int(is_reset))
                oc[e], rc[e], dc[e], ic[e] = orc.ext_call(e, ranges[e], **inp)
                if is_reset:
                    orc.ext_set_done(e, False)
            odom = np.stack([pose[:, 0], pose[:, 1], pose[:, 2], v, w, now, dq[:, 0], dq[:, 1], end_ts, np.zeros(N)], 1)
            env.observe_external(ranges, odom, step_counter=sc.astype(np.int32), is_reset=is_reset)
            torch.cuda.synchronize()
            calls += N
            og = env.obs_f64.cpu().numpy()
            why = []
            neq = ~((og == oc) | (np.isnan(og) & np.isnan(oc)))          # NaN = NaN here (a repeated clock gives 0 / 0 on both sides)
            if neq.any():
                with np.errstate(invalid="ignore"):
                    d = np.abs(og - oc)[neq]
                if not np.isfinite(d).all() or d.max() > 1e-12:
                    why.append("obs")
                else:
                    soft += 1; neq[:] = False
            if not is_reset:
                if not np.array_equal(env.reward.cpu().numpy().astype(np.float64), rc.astype(np.float32).astype(np.float64)): why.append("reward")
                if not np.array_equal(env.done.cpu().numpy().astype(bool), dc): why.append("done")
                if layout == 0 and not np.array_equal(env.topk_idx.cpu().numpy(), ic): why.append("idx")
            else:
                env.done.zero_()                          # TRAIN:116
            cg = env.counters().cpu().numpy()
            if not np.array_equal(cg[:, :3], np.asarray(orc.counters())[:, :3]): why.append("counters")
            if why:
                st = cg[:, 6].astype(np.int64)
                rowm = neq.any(1)
                if not is_reset:
                    rowm |= (env.reward.cpu().numpy() != rc.astype(np.float32)) | (env.done.cpu().numpy().astype(bool) != dc)
                    if layout == 0:
                        rowm |= (env.topk_idx.cpu().numpy() != ic).any(1)
                rowm |= (cg[:, :3] != np.asarray(orc.counters())[:, :3]).any(1)
                rows = np.nonzero(rowm)[0]
                # flagged by the kernel itself in the env's status word: a full track / confirmed-object table (1, 8) -- the documented
                # capacity limit -- or a repeated time stamp (4: CN_ST_DT_ZERO; the reference divides by zero there and then sorts NaNs)
                if len(rows) > 0 and all(st[e] & 9 for e in rows):
                    bad = "overflow"; n_overflow += 1
                elif len(rows) > 0 and all(st[e] & 4 for e in rows):
                    bad = "overflow"; n_dtzero += 1
                else:
                    bad = "call %d: %s" % (i, why)
                    if shown < a.verbose:
                        shown += 1
                        e = int(rows[0]) if len(rows) else 0
                        cols = np.nonzero(neq[e])[0]
                        print("DIFFERENCE world %d (%s) %s env %d cols %s\n   gpu %s\n   cpu %s\n   status %d  Config(**%r)" % (
                            worlds, kn, bad, e, cols[:12], og[e][cols[:8]], oc[e][cols[:8]], int(st[e]), kw))
                        dg, dc_ = env.debug_env(e), orc.debug(e)
                        for k_ in ("n_tracks", "n_confirmed", "n_entries", "bb", "status", "collision_prob", "ego_score"):
                            print("   %-14s gpu %r  cpu %r" % (k_, dg[k_], dc_[k_]))
                        print("   track_pose gpu", np.asarray(dg["track_pose"]).round(4).tolist()[:10], "\n   track_pose cpu", np.asarray(dc_["track_pose"]).round(4).tolist()[:10])
                        print("   track_dist gpu", np.asarray(dg["track_dist"]).tolist()[:10], "\n   track_dist cpu", np.asarray(dc_["track_dist"]).tolist()[:10])
                        fin = np.isfinite(ranges[e]); print("   scan: %d finite, %d nan, %d zero, %d inf; min finite %.4f; is_reset %s step %d" % (int(fin.sum()), int(np.isnan(ranges[e]).sum()), int((ranges[e] == 0).sum()), int(np.isposinf(ranges[e]).sum()), float(ranges[e][fin & (ranges[e] > 0)].min()) if (fin & (ranges[e] > 0)).any() else -1, is_reset, int(sc[e])))
                break
            sc += 1
        if bad and bad != "overflow":
            bad_worlds += 1
        env.close()
    print("fuzz_external: seed %d, %.0f s: %d worlds x %d robots, %d calls compared, %d world(s) with a difference (%d calls equal to 1e-12 but not bit for bit); apart: %d world(s) outgrew a table (status 1 / 8), %d differ only in envs whose clock repeated (status 4: the reference divides by zero and sorts NaNs)"
          % (a.seed, a.seconds, worlds, N, calls, bad_worlds, soft, n_overflow, n_dtzero))
    print("  kernels:", kernels)
    sys.exit(1 if bad_worlds else 0)


if __name__ == "__main__":
    main()
