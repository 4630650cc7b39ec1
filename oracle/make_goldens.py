#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE's own Python under oracle/harness.

Container-only (needs /root/reference).  The committed .npz files are data: inputs and the
outputs the reference produced for them.  Re-run with:  python oracle/make_goldens.py

  seq_<name>.npz   sequence level: every Env.reset()/Env.step() of a few seeded
                   episodes -- what Gazebo/ROS handed to get_state (ranges[360], pose, yaw, twist,
                   clock, step counter, deque pose, timestep) and what the reference returned
                   (obs[366+4K], reward, done, safety counters, track table, CP scalars, waypoint)
  func.npz         function level: utils.py helpers and Env helpers on seeded random inputs
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from oracle.harness.refenv import Harness, HarnessOriginal, HarnessRealworld  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
MAXT = 24

SEQ_CONFIGS = {
    # name: (config overrides, episodes, action ranges)
    "train20": (dict(n_peds=20, max_steps=200, seed=11), 5, (0.0, 0.22, -2.0, 2.0)),
    "dense100": (dict(n_peds=100, max_steps=120, seed=12), 4, (0.02, 0.12, -1.0, 1.0)),
    "eval60": (dict(n_peds=60, max_steps=80, seed=13, min_scan_range=0.0, goal_x=-1.0, goal_y=1.0), 2,
               (0.05, 0.22, -1.0, 1.0)),
    "k4": (dict(n_peds=40, max_steps=100, seed=14, k_obstacles=4), 3, (0.0, 0.22, -2.0, 2.0)),
    # the reference under GEOS <= 3.8 empty-result semantics (cn_config.geos_untyped_empty; shapely_shim.UNTYPED_EMPTY):
    # get_collision_point gives up at the first candidate segment that misses (UTL:279-289)
    "geos38": (dict(n_peds=60, max_steps=120, seed=15, geos_untyped_empty=1), 3, (0.0, 0.22, -2.0, 2.0)),
    # the reference fed by the simulator's "as Gazebo delivers it" switches: float32 LaserScan.ranges (cn_config.scan_f32) and the
    # diff-drive plugin's wheel-speed ramp (cn_config.wheel_accel; /odom then reports the wheels' twist, not the command)
    "gazebo20": (dict(n_peds=20, max_steps=150, seed=17, scan_f32=1, wheel_accel=1.0), 4, (0.0, 0.22, -2.0, 2.0)),
}
# the reference under Python-2.7 round() (cn_config.py2_round; harness refenv.py2_round), fed by TieSim: sensor data on exact ties
TIE_CONFIGS = {
    "py2tie": (dict(n_peds=40, max_steps=60, seed=16, py2_round=1), 3, (0.0, 0.22, -2.0, 2.0)),
}


class TieSim(object):
    """Plays Gazebo for the harness like oracle.Oracle does, but puts a share of what it hands to the reference EXACTLY on decimal
    ties -- lidar ranges and robot coordinates on odd multiples of 1/16 (ties of round(x, 3): ENV:324-327, 1025, 1208, UTL:122-123),
    and on the first calls a pose 0.625 m from the goal (a tie of ENV:255's round(np.float64, 2)) -- so that a Python-2.7 round()
    (cn_config.py2_round) and a Python-3 one give different runs.  The simulator underneath keeps its own (unsnapped) state: the
    reference only ever sees what this object returns, and that is what the golden records."""

    def __init__(self, sim, seed):
        self.sim, self.cfg = sim, sim.cfg
        self.rng = np.random.default_rng(seed)
        self.calls = 0
        self._key, self._snapped = None, None      # one snapped pose per simulator state: the harness reads the pose several times

    def hsim_reset(self, env=0):
        self.sim.hsim_reset(env)

    def hsim_advance(self, ms, v, w, env=0):
        self.sim.hsim_advance(ms, v, w, env)

    def hsim_scan(self, env=0):
        r = self.sim.hsim_scan(env)
        ties = np.array([0.1875, 0.3125, 0.4375, 0.5625])
        fin = np.isfinite(r) & (self.rng.uniform(size=r.shape) < 0.35)
        idx = np.abs(r[fin, None] - ties[None, :]).argmin(1)
        r[fin] = ties[idx]
        return r

    def sim_state(self, env=0):
        robot, pp, pv, rg = self.sim.sim_state(env)
        key = robot.tobytes()
        if key != self._key:
            self._key = key
            robot = robot.copy()
            self.calls += 1
            if self.calls <= 2:
                robot[0], robot[1] = -0.375, 1.0                     # 0.625 m from the goal (-1, 1): round(0.625, 2) is a tie
            else:
                for k in (0, 1):
                    if self.rng.uniform() < 0.5:
                        robot[k] = (2.0 * np.floor(robot[k] * 8.0) + 1.0) / 16.0     # an odd multiple of 1/16 nearby
            self._snapped = robot
        return self._snapped.copy(), pp, pv, rg

    def get_ped_init(self):
        return self.sim.get_ped_init()


def gen_seq(name, kw, episodes, arange, tie_sim=False):
    cfgkw = dict(n_envs=1, **kw)
    sim = oracle.Oracle(**cfgkw)
    if tie_sim:
        sim = TieSim(sim, kw["seed"])
    h = Harness(sim)
    rng = np.random.default_rng(kw["seed"])
    rows = []
    cols = {k: [] for k in ("ranges", "px", "py", "yaw", "v", "w", "now", "step_counter", "is_reset", "deque_x",
                            "deque_y", "end_timestep", "action", "obs", "reward", "done", "counters", "n_tracks",
                            "track_pose", "track_dist", "track_speed", "track_vel", "collision_prob", "ego_score",
                            "wp", "bb", "status")}

    def push(rec, action, obs, reward, done):
        snap = h.snapshot()
        for k in ("ranges", "px", "py", "yaw", "v", "w", "now", "step_counter", "is_reset", "deque_x", "deque_y",
                  "end_timestep"):
            cols[k].append(rec[k])
        cols["action"].append(action)
        cols["obs"].append(obs); cols["reward"].append(reward); cols["done"].append(done)
        cols["counters"].append(snap["counters"]); cols["n_tracks"].append(snap["n_tracks"])
        n = snap["n_tracks"]
        assert n <= MAXT, n

        def pad(a, shape):
            out = np.zeros(shape)
            out[:n] = a
            return out

        cols["track_pose"].append(pad(snap["track_pose"], (MAXT, 2)))
        cols["track_dist"].append(pad(snap["track_dist"], (MAXT,)))
        cols["track_speed"].append(pad(snap["track_speed"], (MAXT,)))
        cols["track_vel"].append(pad(snap["track_vel"], (MAXT, 2)))
        cols["collision_prob"].append(snap["collision_prob"]); cols["ego_score"].append(snap["ego_score"])
        cols["wp"].append(snap["wp"]); cols["bb"].append(snap["bb"]); cols["status"].append(snap["status"])

    max_steps = kw["max_steps"]
    for ep in range(episodes):
        obs = h.reset()
        push(h.trace[-1], (0.0, 0.0), obs, 0.0, False)
        for st in range(max_steps):
            # float32-representable: the product's ABI takes float32 actions (the TD3 actor's dtype)
            a = (float(np.float32(rng.uniform(arange[0], arange[1]))), float(np.float32(rng.uniform(arange[2], arange[3]))))
            obs, r, d = h.step(a, st + 1)
            push(h.trace[-1], a, obs, r, d)
            if d:
                break
    arrs = {k: np.asarray(v) for k, v in cols.items()}
    arrs["ped_init"] = sim.get_ped_init()
    arrs["config_keys"] = np.array(sorted(kw.keys()))
    arrs["config_vals"] = np.array([float(kw[k]) for k in sorted(kw.keys())])
    path = os.path.join(OUT, "seq_%s.npz" % name)
    np.savez_compressed(path, **arrs)
    nt = arrs["n_tracks"]
    print("%-10s calls=%4d  tracks mean %.2f max %d  (>K on %d calls)  done=%d  %.0f KB" % (
        name, len(nt), nt.mean(), nt.max(), int((nt > kw.get("k_obstacles", 8)).sum()), int(arrs["done"].sum()),
        os.path.getsize(path) / 1024))


ORIG_CONFIGS = {
    # environment_stage_1_original.py (obs_layout 1, 363 inputs): name -> (config overrides, episodes, action ranges)
    "orig20": (dict(n_peds=20, max_steps=150, seed=21, obs_layout=1), 6, (0.0, 0.22, -2.0, 2.0)),
    "orig60": (dict(n_peds=60, max_steps=80, seed=22, obs_layout=1, room_half=1.8), 4, (0.05, 0.22, -1.0, 1.0)),
}


def gen_seq_original(name, kw, episodes, arange):
    """Sequence goldens of the 363-input environment: inputs (ranges, pose, yaw, step counter) and what the reference
    returned (obs[363], reward, done, success/failure flags, previous_distance / previous_heading)."""
    sim = oracle.Oracle(n_envs=1, **kw)
    h = HarnessOriginal(sim)
    rng = np.random.default_rng(kw["seed"])
    cols = {k: [] for k in ("ranges", "px", "py", "yaw", "v", "w", "now", "step_counter", "is_reset", "action", "obs",
                            "reward", "done", "status", "prev")}

    def push(rec, action, obs, reward, done):
        snap = h.snapshot()
        for k in ("ranges", "px", "py", "yaw", "v", "w", "now", "step_counter", "is_reset"):
            cols[k].append(rec[k])
        cols["action"].append(action); cols["obs"].append(obs); cols["reward"].append(reward); cols["done"].append(done)
        cols["status"].append(snap["status"]); cols["prev"].append(snap["prev"])

    for ep in range(episodes):
        obs = h.reset()
        push(h.trace[-1], (0.0, 0.0), obs, 0.0, False)
        for st in range(kw["max_steps"]):
            a = (float(np.float32(rng.uniform(arange[0], arange[1]))), float(np.float32(rng.uniform(arange[2], arange[3]))))
            obs, r, d = h.step(a, st + 1)
            push(h.trace[-1], a, obs, r, d)
            if d:
                break
    arrs = {k: np.asarray(v) for k, v in cols.items()}
    arrs["ped_init"] = sim.get_ped_init()
    arrs["config_keys"] = np.array(sorted(kw.keys()))
    arrs["config_vals"] = np.array([float(kw[k]) for k in sorted(kw.keys())])
    path = os.path.join(OUT, "seq_%s.npz" % name)
    np.savez_compressed(path, **arrs)
    print("%-10s calls=%4d  done=%d (success %d)  %.0f KB" % (name, len(arrs["done"]), int(arrs["done"].sum()),
                                                              int(sum(1 for i in range(len(arrs["done"])) if arrs["done"][i] and arrs["status"][i][0])),
                                                              os.path.getsize(path) / 1024))


RW_CONFIGS = {
    # environment_stage_1_nobonus_realworld.py (obs_layout 2, 370 inputs): Env.step holds a command for 0.05 s (RW:880-883)
    "rw20": (dict(n_peds=20, max_steps=250, seed=31, obs_layout=2, dt_ms=50), 5, (0.0, 0.22, -2.0, 2.0)),
    "rw60": (dict(n_peds=60, max_steps=200, seed=32, obs_layout=2, dt_ms=50, min_scan_range=0.0, room_half=1.8), 3,
             (0.05, 0.22, -1.0, 1.0)),
}


def gen_seq_realworld(name, kw, episodes, arange):
    """Sequence goldens of the 370-input environment: what Gazebo/ROS handed get_state and what the reference returned
    (obs[370], reward, done, counters, track table, collision probability, bbox size, previous distance / heading)."""
    sim = oracle.Oracle(n_envs=1, **kw)
    h = HarnessRealworld(sim)
    rng = np.random.default_rng(kw["seed"])
    cols = {k: [] for k in ("ranges", "px", "py", "yaw", "v", "w", "now", "step_counter", "is_reset", "deque_x", "deque_y",
                            "end_timestep", "action", "obs", "reward", "done", "counters", "n_tracks", "track_pose",
                            "track_dist", "track_speed", "track_vel", "collision_prob", "bb", "status", "prev")}

    def push(rec, action, obs, reward, done):
        snap = h.snapshot()
        for k in ("ranges", "px", "py", "yaw", "v", "w", "now", "step_counter", "is_reset", "deque_x", "deque_y", "end_timestep"):
            cols[k].append(rec[k])
        cols["action"].append(action); cols["obs"].append(obs); cols["reward"].append(reward); cols["done"].append(done)
        n = snap["n_tracks"]
        assert n <= MAXT, n

        def pad(a, shape):
            out = np.zeros(shape); out[:n] = a
            return out
        cols["counters"].append(snap["counters"]); cols["n_tracks"].append(n)
        cols["track_pose"].append(pad(snap["track_pose"], (MAXT, 2))); cols["track_dist"].append(pad(snap["track_dist"], (MAXT,)))
        cols["track_speed"].append(pad(snap["track_speed"], (MAXT,))); cols["track_vel"].append(pad(snap["track_vel"], (MAXT, 2)))
        cols["collision_prob"].append(snap["collision_prob"]); cols["bb"].append(snap["bb"])
        cols["status"].append(snap["status"]); cols["prev"].append(snap["prev"])

    for ep in range(episodes):
        obs = h.reset()
        push(h.trace[-1], (0.0, 0.0), obs, 0.0, False)
        for st in range(kw["max_steps"]):
            a = (float(np.float32(rng.uniform(arange[0], arange[1]))), float(np.float32(rng.uniform(arange[2], arange[3]))))
            obs, r, d = h.step(a, st + 1)
            push(h.trace[-1], a, obs, r, d)
            if d:
                break
    arrs = {k: np.asarray(v) for k, v in cols.items()}
    arrs["ped_init"] = sim.get_ped_init()
    arrs["config_keys"] = np.array(sorted(kw.keys()))
    arrs["config_vals"] = np.array([float(kw[k]) for k in sorted(kw.keys())])
    path = os.path.join(OUT, "seq_%s.npz" % name)
    np.savez_compressed(path, **arrs)
    nt = arrs["n_tracks"]
    print("%-10s calls=%4d  tracks mean %.2f max %d  done=%d (success %d)  %.0f KB" % (
        name, len(nt), nt.mean(), nt.max(), int(arrs["done"].sum()),
        int(sum(1 for i in range(len(nt)) if arrs["done"][i] and arrs["status"][i][0])), os.path.getsize(path) / 1024))


def gen_func():
    sim = oracle.Oracle(n_envs=1, n_peds=4)
    h = Harness(sim)
    U, E = h.utils, h.env
    rng = np.random.default_rng(2024)
    g = {}
    # A5 get_scan_ranges (UTL:375-392)
    from oracle.harness.refenv import LaserScan
    rg = rng.uniform(0.05, 0.9, (64, 360))
    rg[rng.uniform(size=rg.shape) < 0.5] = np.inf
    rg[rng.uniform(size=rg.shape) < 0.02] = 0.0
    rg[rng.uniform(size=rg.shape) < 0.02] = np.nan
    g["scan_in"] = rg
    g["scan_out"] = np.array([U.get_scan_ranges(LaserScan(list(map(float, r))), 360, 0.6) for r in rg], dtype=np.float64)
    # A6 convert_laserscan_to_coordinate (UTL:110-126)
    from oracle.harness.refenv import Point
    sc = rng.uniform(0.08, 0.6, (64, 359)); pose = rng.uniform(-1.3, 1.3, (64, 2)); yaw = rng.uniform(-np.pi, np.pi, 64)
    g["pts_scan"], g["pts_pose"], g["pts_yaw"] = sc, pose, yaw
    g["pts_out"] = np.array([U.convert_laserscan_to_coordinate(list(map(float, sc[i])), 360,
                                                               Point(float(pose[i, 0]), float(pose[i, 1])),
                                                               float(yaw[i]), 360) for i in range(64)])
    # A10 compute_average_bounding_box_size (UTL:405-419)
    g["bb_out"] = np.array([U.compute_average_bounding_box_size(p.tolist()) for p in g["pts_out"]])
    # A9 get_local_goal_waypoints (UTL:296-314)
    a = rng.uniform(-1.3, 1.3, (512, 2)); gl = rng.uniform(-1.3, 1.3, (512, 2))
    gl[:64] = a[:64] + rng.uniform(-0.25, 0.25, (64, 2))               # goal inside the polygon
    ang = np.arange(64) * np.pi / 32.0                                   # exactly through vertex directions
    gl[64:128] = a[64:128] + 0.8 * np.stack([np.cos(ang), -np.sin(ang)], 1)
    gl[128:192] = a[128:192] + 0.8 * np.stack([np.cos(ang + np.pi / 64), -np.sin(ang + np.pi / 64)], 1)
    g["wp_agent"], g["wp_goal"] = a, gl
    g["wp_out"] = np.array([U.get_local_goal_waypoints([float(a[i, 0]), float(a[i, 1])],
                                                       [float(gl[i, 0]), float(gl[i, 1])], 0.3) for i in range(512)])
    # A22 get_collision_point (UTL:251-293)
    n = 1024
    a0 = np.round(rng.uniform(-1.3, 1.3, (n, 2)), 3); a1 = np.round(a0 + rng.uniform(-0.05, 0.05, (n, 2)), 3)
    ob = np.round(a0 + rng.uniform(-0.7, 0.7, (n, 2)), 3)
    a1[:32, 1] = 0.0                                                     # ZeroDivisionError branch
    ob[32:96] = np.round(a0[32:96] + rng.uniform(-0.15, 0.15, (64, 2)), 3)  # agent inside the ring -> one point
    out = []
    for i in range(n):
        d = U.get_collision_point([[float(a0[i, 0]), float(a0[i, 1])], [float(a1[i, 0]), float(a1[i, 1])]],
                                  [float(ob[i, 0]), float(ob[i, 1])], 0.178)
        out.append(np.nan if d is None else d)
    g["cp_a0"], g["cp_a1"], g["cp_ob"], g["cp_out"] = a0, a1, ob, np.array(out)
    # ... and under GEOS <= 3.8 empty-result semantics (geos_untyped_empty): None at the first candidate that misses
    from oracle.harness import shapely_shim
    shapely_shim.UNTYPED_EMPTY = True
    out = []
    for i in range(n):
        d = U.get_collision_point([[float(a0[i, 0]), float(a0[i, 1])], [float(a1[i, 0]), float(a1[i, 1])]],
                                  [float(ob[i, 0]), float(ob[i, 1])], 0.178)
        out.append(np.nan if d is None else d)
    g["cp_out_geos38"] = np.array(out)
    g["wp_out_geos38"] = np.array([U.get_local_goal_waypoints([float(g["wp_agent"][i, 0]), float(g["wp_agent"][i, 1])],
                                                              [float(g["wp_goal"][i, 0]), float(g["wp_goal"][i, 1])], 0.3)
                                   for i in range(512)])
    shapely_shim.UNTYPED_EMPTY = False
    # the survey's spot value
    g["cp_spot"] = np.array([U.get_collision_point([[0, 0.5], [-0.03, 0.5]], [-0.5, 0.5], 0.178)])
    # A16/A20 is_associated / get_iou (UTL:435-460)
    p1 = np.round(rng.uniform(-1, 1, (2048, 2)), 3); p2 = np.round(p1 + rng.uniform(-0.12, 0.12, (2048, 2)), 3)
    s = np.where(rng.uniform(size=2048) < 0.5, 0.0505, rng.uniform(0.009, 0.025, 2048))
    g["iou_p1"], g["iou_p2"], g["iou_s"] = p1, p2, s
    g["iou_out"] = np.array([U.get_iou(p1[i].tolist(), p2[i].tolist(), float(s[i])) for i in range(2048)])
    g["assoc_out"] = np.array([U.is_associated(p1[i].tolist(), p2[i].tolist(), float(s[i])) for i in range(2048)])
    # A18 estimate_num_obs_scans (UTL:395-402)
    d = np.concatenate([np.round(rng.uniform(0.08, 0.6, 500), 3), [0.6, 0.36, 0.12]])
    g["est_d"] = d
    g["est_out"] = np.array([U.estimate_num_obs_scans(float(x), 0.6, 0.12) for x in d])
    g["est_out0"] = np.array([U.estimate_num_obs_scans(float(x), 0.6, 0.0) for x in d])
    # A23 collision probabilities (UTL:317-345)
    ttc = np.concatenate([rng.uniform(-3, 3, 500), [0.15, 0.1, 1e-9, -1e-9]])
    g["cpttc_in"] = ttc
    g["cpttc_out"] = np.array([U.compute_collision_prob(float(x)) for x in ttc])
    gd = np.concatenate([np.round(rng.uniform(0.0, 0.7, 500), 3), [0.6, 0.12, 0.05]])
    g["gcp_in"] = gd
    g["gcp_out"] = np.array([U.compute_general_collision_prob(float(x), 0.6, 0.12) for x in gd])
    # A24 top-K rule, ENV:882-883 verbatim on random CP lists (ties included)
    K = 8
    lists, keep = [], []
    for t in range(256):
        nn = int(rng.integers(1, 20))
        cp = np.round(rng.uniform(-0.2, 1.0, nn), int(rng.integers(1, 4)))
        entries = [[float(cp[i]), i] for i in range(nn)]
        kept = sorted(entries, key=lambda x: x[0], reverse=True)[-K:]
        row = np.full(20, np.nan); row[:nn] = cp
        kk = np.full(K, -1); kk[:len(kept)] = [e[1] for e in kept]
        lists.append(row); keep.append(kk)
    g["topk_cp"], g["topk_idx"] = np.array(lists), np.array(keep)
    # A7/A8 heading and distance (ENV:191-237)
    from oracle.harness.refenv import Quaternion
    import math
    pos = rng.uniform(-1.3, 1.3, (256, 2)); wpt = rng.uniform(-1.3, 1.3, (256, 2)); yw = rng.uniform(-3.14, 3.14, 256)
    hd, dist = [], []
    for i in range(256):
        E.waypoint_desired_point.x, E.waypoint_desired_point.y = float(wpt[i, 0]), float(wpt[i, 1])
        P = Point(float(pos[i, 0]), float(pos[i, 1]), 0.0)
        q = Quaternion(0.0, 0.0, math.sin(yw[i] / 2.0), math.cos(yw[i] / 2.0))
        hd.append(E.get_heading_to_goal(P, q)); dist.append(float(E.get_distance_to_goal(P)))
        yw[i] = E.get_angle_from_point(q)  # the yaw the reference itself derives
    g["hd_pos"], g["hd_wp"], g["hd_yaw"], g["hd_out"], g["dist_out"] = pos, wpt, yw, np.array(hd), np.array(dist)
    # A28 half-open goal box (ENV:1303-1319)
    bx = np.concatenate([rng.uniform(-1.4, -0.6, (200, 2)) * [1, -1], [[-0.8, 1.2], [-1.2, 0.8], [-0.8, 0.8], [-1.2, 1.2]]])
    g["box_in"] = bx
    g["box_out"] = np.array([E.is_in_true_desired_position(Point(float(b[0]), float(b[1]))) for b in bx])
    # A30 compute_reward sign table (ENV:1046-1162)
    rows = []
    for ch in (-0.5, 0.0, 0.5):
        for ph in (-0.5, 0.0, 0.5):
            for cd in (1.0, 1.1):
                for pd in (1.0, 1.1, 1.2):
                    for done in (False, True):
                        st = [0.6] * 359 + [ch, cd] + [0.0] * 37
                        E.previous_heading, E.previous_distance = ph, pd
                        E.waypoint_desired_point.x, E.waypoint_desired_point.y = 5.0, 5.0
                        h._push_odom()
                        import contextlib, io
                        with contextlib.redirect_stdout(io.StringIO()):
                            r, _ = E.compute_reward(st, 3, done)
                        rows.append([ch, ph, cd, pd, float(done), float(r)])
    g["reward_table"] = np.array(rows)
    path = os.path.join(OUT, "func.npz")
    np.savez_compressed(path, **g)
    print("func.npz %.0f KB, %d arrays" % (os.path.getsize(path) / 1024, len(g)))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]            # e.g. `python oracle/make_goldens.py orig20 orig60` regenerates just those
    for name, (kw, eps, ar) in SEQ_CONFIGS.items():
        if not only or name in only:
            gen_seq(name, kw, eps, ar)
    for name, (kw, eps, ar) in TIE_CONFIGS.items():
        if not only or name in only:
            gen_seq(name, kw, eps, ar, tie_sim=True)
    for name, (kw, eps, ar) in ORIG_CONFIGS.items():
        if not only or name in only:
            gen_seq_original(name, kw, eps, ar)
    for name, (kw, eps, ar) in RW_CONFIGS.items():
        if not only or name in only:
            gen_seq_realworld(name, kw, eps, ar)
    if not only or "func" in only:
        gen_func()
