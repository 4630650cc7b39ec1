"""ctypes binding of oracle/_build/libcn_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package never does (tests/test_layout.py greps for that).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CN_ORACLE_LIB: load another build of the same source instead (oracle/Makefile `make ubsan`; never set by the product or the bench)
_LIB_PATH = os.environ.get("CN_ORACLE_LIB") or os.path.join(_HERE, "_build", "libcn_oracle.so")
MAX_TRACKS = 64


class CnoConfig(C.Structure):
    _fields_ = [
        ("n_envs", C.c_int32), ("n_peds", C.c_int32), ("n_rays", C.c_int32), ("k_obstacles", C.c_int32),
        ("max_steps", C.c_int32), ("ped_mode", C.c_int32), ("dt_ms", C.c_int32), ("scan_latency_ms", C.c_int32),
        ("settle_ms", C.c_int32), ("ped_cycle_ms", C.c_int32), ("ped_stagger_ms", C.c_int32), ("reserved0", C.c_int32),
        ("obs_layout", C.c_int32), ("geos_untyped_empty", C.c_int32), ("ped_contact", C.c_int32), ("risk_mode", C.c_int32),
        ("py2_round", C.c_int32), ("sf_tick_ms", C.c_int32), ("scan_f32", C.c_int32), ("waypoint_reward", C.c_int32),
        ("env_index_base", C.c_int64), ("seed", C.c_uint64),
        ("room_half", C.c_double), ("ped_radius", C.c_double), ("ped_vmax", C.c_double),
        ("robot_clearance", C.c_double), ("lidar_min", C.c_double), ("lidar_max", C.c_double),
        ("lidar_span", C.c_double), ("lidar_offset_x", C.c_double), ("max_scan_range", C.c_double),
        ("min_scan_range", C.c_double), ("goal_x", C.c_double), ("goal_y", C.c_double),
        ("start_x", C.c_double), ("start_y", C.c_double), ("spawn_x", C.c_double), ("spawn_y", C.c_double),
        ("spawn_yaw", C.c_double), ("waypoint_radius", C.c_double), ("goal_eps", C.c_double),
        ("sf_tau", C.c_double), ("sf_A", C.c_double), ("sf_B", C.c_double), ("sf_wall_A", C.c_double),
        ("sf_wall_B", C.c_double), ("sf_goal_eps", C.c_double), ("wheel_accel", C.c_double), ("wheel_separation", C.c_double),
    ]


class CnoExtIn(C.Structure):
    _fields_ = [
        ("deque_x", C.c_double), ("deque_y", C.c_double), ("end_timestep", C.c_double),
        ("px", C.c_double), ("py", C.c_double), ("yaw", C.c_double), ("v", C.c_double), ("w", C.c_double),
        ("now", C.c_double), ("step_counter", C.c_int32), ("is_reset", C.c_int32),
    ]


class CnoDebug(C.Structure):
    _fields_ = [
        ("n_confirmed", C.c_int32), ("n_tracks", C.c_int32), ("n_entries", C.c_int32), ("status", C.c_int32),
        ("bb", C.c_double), ("collision_prob", C.c_double), ("ego_score", C.c_double),
        ("wpx", C.c_double), ("wpy", C.c_double),
        ("track_pose", C.c_double * 2 * MAX_TRACKS), ("track_dist", C.c_double * MAX_TRACKS),
        ("track_speed", C.c_double * MAX_TRACKS), ("track_vel", C.c_double * 2 * MAX_TRACKS),
        ("track_t", C.c_double * MAX_TRACKS), ("track_dqlen", C.c_int32 * MAX_TRACKS),
        ("entry_cp", C.c_double * MAX_TRACKS), ("entry_ego", C.c_double * MAX_TRACKS),
    ]


# Reference defaults (SURVEY.md appendix B cites every value).
DEFAULTS = dict(
    n_envs=1, n_peds=20, n_rays=360, k_obstacles=8, max_steps=1000, ped_mode=0, dt_ms=150, scan_latency_ms=10,
    settle_ms=100, ped_cycle_ms=0, ped_stagger_ms=100, reserved0=0, obs_layout=0, geos_untyped_empty=0, ped_contact=0, risk_mode=0, py2_round=0, sf_tick_ms=0, scan_f32=0, waypoint_reward=200, env_index_base=0, seed=1234,
    room_half=1.40, ped_radius=0.0505, ped_vmax=0.2, robot_clearance=0.09, lidar_min=0.08, lidar_max=0.60,
    lidar_span=6.28, lidar_offset_x=-0.032, max_scan_range=0.6, min_scan_range=0.12, goal_x=-1.0, goal_y=1.0,
    start_x=0.75, start_y=-0.75, spawn_x=1.0, spawn_y=-1.0, spawn_yaw=3.14, waypoint_radius=0.3, goal_eps=0.20,
    sf_tau=0.5, sf_A=0.8, sf_B=0.10, sf_wall_A=1.0, sf_wall_B=0.05, sf_goal_eps=0.10, wheel_accel=0.0, wheel_separation=0.160,
)


def make_config(**kw):
    d = dict(DEFAULTS)
    kw = dict(kw)
    kw.pop("track_capacity", None)   # product-only field (the oracle always has 64 slots)
    for k, v in kw.items():
        if k not in d:
            raise KeyError(k)
        d[k] = v
    if not d["ped_cycle_ms"]:
        d["ped_cycle_ms"] = max(100, 100 * d["n_peds"])  # CROWD:128-144: 0.1 s per obstacle per cycle
    return CnoConfig(**d)


def build(force=False):
    if os.environ.get("CN_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f))
                                              for f in ("cn_oracle.c", "cn_oracle.h", "Makefile"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.cno_create.argtypes = [C.POINTER(CnoConfig), C.POINTER(C.c_void_p)]
        L.cno_destroy.argtypes = [C.c_void_p]
        L.cno_obs_dim.argtypes = [C.c_void_p]
        L.cno_set_ped_init.argtypes = [C.c_void_p, dp]
        L.cno_get_ped_init.argtypes = [C.c_void_p, dp]
        L.cno_set_ped_preset_vel.argtypes = [C.c_void_p, dp]
        L.cno_reset.argtypes = [C.c_void_p, C.c_void_p, dp]
        L.cno_step.argtypes = [C.c_void_p, dp, C.c_void_p, C.c_int, dp, C.c_void_p, dp, C.c_void_p, C.c_void_p]
        L.cno_get_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.cno_get_returns.argtypes = [C.c_void_p, dp]
        L.cno_get_sim_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cno_get_debug.argtypes = [C.c_void_p, C.c_int, C.POINTER(CnoDebug)]
        L.cno_set_num_threads.argtypes = [C.c_int]
        L.cno_ext_call.argtypes = [C.c_void_p, C.c_int, C.POINTER(CnoExtIn), dp, dp, dp, C.c_void_p, C.c_void_p]
        L.cno_ext_set_done.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.cno_py_round.argtypes = [C.c_double, C.c_int]; L.cno_py_round.restype = C.c_double
        L.cno_np_around.argtypes = [C.c_double, C.c_int]; L.cno_np_around.restype = C.c_double
        L.cno_det_sincos.argtypes = [C.c_double, dp, dp]
        L.cno_scan_sanitize.argtypes = [dp, C.c_int, C.c_double, dp]
        L.cno_scan_to_points.argtypes = [dp, C.c_int, C.c_double, C.c_double, C.c_double, dp]
        L.cno_waypoint.argtypes = [C.c_double] * 5 + [dp]
        L.cno_collision_point.argtypes = [C.c_double] * 7 + [dp]
        L.cno_collision_prob.argtypes = [C.c_double]; L.cno_collision_prob.restype = C.c_double
        L.cno_general_collision_prob.argtypes = [C.c_double] * 3; L.cno_general_collision_prob.restype = C.c_double
        L.cno_topk.argtypes = [dp, C.c_int, C.c_int, C.c_void_p]
        L.cno_heading_to_goal.argtypes = [C.c_void_p, C.c_int] + [C.c_double] * 5; L.cno_heading_to_goal.restype = C.c_double
        L.cno_distance_to_goal.argtypes = [C.c_double] * 4; L.cno_distance_to_goal.restype = C.c_double
        L.cno_in_box.argtypes = [C.c_double] * 5
        L.cno_compute_reward.argtypes = [C.c_void_p, C.c_int] + [C.c_double] * 8 + [C.c_int]; L.cno_compute_reward.restype = C.c_double
        L.cno_collision_point_geos.argtypes = [C.c_double] * 7 + [C.c_int, dp]
        L.cno_iou.argtypes = [C.c_double] * 5; L.cno_iou.restype = C.c_double
        L.cno_bbox_size.argtypes = [dp, C.c_int]; L.cno_bbox_size.restype = C.c_double
        L.cno_estimate_num_obs_scans.argtypes = [C.c_double] * 3
        L.cno_raycast.argtypes = [C.POINTER(CnoConfig), C.c_double, C.c_double, C.c_double, dp, C.c_int, dp]
        L.cno_rng_u01.argtypes = [C.c_uint64, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.cno_rng_u01.restype = C.c_double
        L.cno_set_state.argtypes = [C.c_void_p, C.c_int, dp, C.c_void_p, dp, dp, dp, C.c_int, dp, dp, dp]
        L.cno_get_state.argtypes = [C.c_void_p, C.c_int, dp, C.c_void_p, dp, dp, dp, C.c_int, dp]
        L.cno_det_exp.argtypes = [C.c_double]; L.cno_det_exp.restype = C.c_double
        L.cno_set_py2_round.argtypes = [C.c_int]; L.cno_set_py2_round.restype = None
        L.cno_hsim_reset.argtypes = [C.c_void_p, C.c_int]
        L.cno_hsim_advance.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_double, C.c_double]
        L.cno_hsim_scan.argtypes = [C.c_void_p, C.c_int, dp]
        L.cno_set_robot.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
        _lib = L
    return _lib


def _ar(mode):
    return {False: 0, True: 1, None: 0, "next": 2, "same": 1}.get(mode, mode)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    """Batched CPU oracle: same call shapes as the product's VecEnv, float64 throughout."""

    def __init__(self, cfg=None, **kw):
        self.cfg = cfg if isinstance(cfg, CnoConfig) else make_config(**(cfg or {}), **kw)
        self.L = lib()
        self.h = C.c_void_p()
        rc = self.L.cno_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise RuntimeError("cno_create failed: %d" % rc)
        self.N, self.P, self.R, self.K = (self.cfg.n_envs, self.cfg.n_peds, self.cfg.n_rays, self.cfg.k_obstacles)
        self.D = self.L.cno_obs_dim(self.h)

    def __del__(self):
        try:
            if self.h:
                self.L.cno_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_ped_init(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(self.N, self.P, 2)
        self.L.cno_set_ped_init(self.h, _dp(xy))

    def get_ped_init(self):
        xy = np.zeros((self.N, self.P, 2))
        self.L.cno_get_ped_init(self.h, _dp(xy))
        return xy

    def set_ped_preset_vel(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(self.N, self.P, 2)
        self.L.cno_set_ped_preset_vel(self.h, _dp(v))

    def reset(self, mask=None):
        obs = np.zeros((self.N, self.D))
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
        self.L.cno_reset(self.h, m.ctypes.data if m is not None else None, _dp(obs))
        return obs

    def step(self, action, step_counter=None, auto_reset=False, want_final=False):
        a = np.ascontiguousarray(action, dtype=np.float64).reshape(self.N, 2)
        obs = np.zeros((self.N, self.D))
        fin = np.zeros((self.N, self.D)) if want_final else None
        rew = np.zeros(self.N)
        done = np.zeros(self.N, dtype=np.uint8)
        idx = np.zeros((self.N, self.K), dtype=np.int32)
        sc = None
        if step_counter is not None:
            sc = np.ascontiguousarray(step_counter, dtype=np.int32).reshape(self.N)
        self.L.cno_step(self.h, _dp(a), sc.ctypes.data if sc is not None else None, _ar(auto_reset), _dp(obs),
                        fin.ctypes.data if fin is not None else None, _dp(rew), done.ctypes.data, idx.ctypes.data)
        if want_final:
            return obs, rew, done, idx, fin
        return obs, rew, done, idx

    def counters(self):
        out = np.zeros((self.N, 6), dtype=np.int32)
        self.L.cno_get_counters(self.h, out.ctypes.data)
        return out

    def returns(self):
        out = np.zeros(self.N)
        self.L.cno_get_returns(self.h, _dp(out))
        return out

    def sim_state(self, env=0):
        robot = np.zeros(5); pp = np.zeros((self.P, 2)); pv = np.zeros((self.P, 2)); rg = np.zeros(self.R)
        self.L.cno_get_sim_state(self.h, env, robot.ctypes.data, pp.ctypes.data, pv.ctypes.data, rg.ctypes.data)
        return robot, pp, pv, rg

    def debug(self, env=0):
        d = CnoDebug()
        self.L.cno_get_debug(self.h, env, C.byref(d))
        n = d.n_tracks
        return dict(n_confirmed=d.n_confirmed, n_tracks=n, n_entries=d.n_entries, status=d.status, bb=d.bb,
                    collision_prob=d.collision_prob, ego_score=d.ego_score, wp=(d.wpx, d.wpy),
                    track_pose=np.array(d.track_pose)[:n].copy(), track_dist=np.array(d.track_dist)[:n].copy(),
                    track_speed=np.array(d.track_speed)[:n].copy(), track_vel=np.array(d.track_vel)[:n].copy(),
                    track_t=np.array(d.track_t)[:n].copy(), track_dqlen=np.array(d.track_dqlen)[:n].copy(),
                    entry_cp=np.array(d.entry_cp)[:d.n_entries].copy(), entry_ego=np.array(d.entry_ego)[:d.n_entries].copy())

    # ---- state exchange in the product's snapshot layout ------------------
    def set_state(self, env, sd, si, ped_p, ped_v, trk, ped_init=None, ped_preset=None, ped_aux=None):
        c = lambda a, dt: None if a is None else np.ascontiguousarray(a, dtype=dt)
        sd, si, ped_p, ped_v, trk = c(sd, np.float64), c(si, np.int32), c(ped_p, np.float64), c(ped_v, np.float64), c(trk, np.float64)
        ped_init, ped_preset, ped_aux = c(ped_init, np.float64), c(ped_preset, np.float64), c(ped_aux, np.float64)
        q = lambda a: None if a is None else _dp(a)
        rc = self.L.cno_set_state(self.h, int(env), q(sd), si.ctypes.data if si is not None else None, q(ped_p), q(ped_v), q(trk),
                                  int(trk.shape[0]) if trk is not None else 0, q(ped_init), q(ped_preset), q(ped_aux))
        if rc != 0:
            raise RuntimeError("cno_set_state failed: %d" % rc)

    def get_state(self, env, trk_cap=MAX_TRACKS):
        sd = np.zeros(24); si = np.zeros(16, dtype=np.int32); pp = np.zeros((self.P, 2)); pv = np.zeros((self.P, 2))
        trk = np.zeros((trk_cap, 12)); aux = np.zeros((self.P, 3))
        self.L.cno_get_state(self.h, int(env), _dp(sd), si.ctypes.data, _dp(pp), _dp(pv), _dp(trk), int(trk_cap), _dp(aux))
        return dict(sd=sd, si=si, ped_p=pp, ped_v=pv, trk=trk, ped_aux=aux)

    # ---- harness-facing simulator access -------------------------------------------------
    def hsim_reset(self, env=0):
        self.L.cno_hsim_reset(self.h, env)

    def hsim_advance(self, ms, v, w, env=0):
        self.L.cno_hsim_advance(self.h, env, int(ms), float(v), float(w))

    def hsim_scan(self, env=0):
        rg = np.zeros(self.R)
        self.L.cno_hsim_scan(self.h, env, _dp(rg))
        return rg

    # ---- golden replay --------------------------------------------------------------------
    def ext_call(self, env, ranges, **kw):
        inp = CnoExtIn(**kw)
        rg = np.ascontiguousarray(ranges, dtype=np.float64)
        obs = np.zeros(self.D); rew = C.c_double(0.0); done = C.c_uint8(0)
        idx = np.zeros(self.K, dtype=np.int32)
        self.L.cno_ext_call(self.h, env, C.byref(inp), _dp(rg), _dp(obs), C.byref(rew), C.byref(done),
                            idx.ctypes.data)
        return obs, rew.value, bool(done.value), idx

    def ext_set_done(self, env, done):
        self.L.cno_ext_set_done(self.h, env, int(done))


def load_snapshot(path):
    """An Oracle seeded from a GPU snapshot file (crowdnav.env.VecEnv.save_snapshot): the configuration comes
    from the file's header, every env's state from its SoA arrays.  Stepping this oracle and the restored GPU handle with the
    same actions must then agree step for step -- tools/bisect_divergence.py reports the first env / field that does not."""
    z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz")
    if str(z["format"]) != "crowdnav-snapshot-1":
        raise ValueError("%s is not a crowdnav snapshot file" % path)
    kw = {str(k): float(v) for k, v in zip(z["config_keys"], z["config_vals"])}
    ints = [n for n, t in CnoConfig._fields_ if t in (C.c_int32, C.c_int64, C.c_uint64)]
    kw.pop("track_capacity", None)
    cfg = {k: (int(v) if k in ints else v) for k, v in kw.items()}
    cfg["seed"] = int(z["config_seed"]); cfg["env_index_base"] = int(z["config_env_index_base"])     # exact 64-bit values
    o = Oracle(cfg)
    for e in range(o.N):
        o.set_state(e, z["sd"][e], z["si"][e], z["ped_p"][e], z["ped_v"][e], z["trk"][e], z["ped_init"][e], z["ped_preset"][e],
                    z["ped_aux"][e])
    return o


def usable_cpus():
    """CPUs this process may really use: the affinity mask cut down by the cgroup quota (a GPU box reports its 100+ hardware
    threads in os.cpu_count() while the container owns 16; OpenMP teams larger than that spin against each other)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, n)


def set_num_threads(n=None):
    """OpenMP threads over envs for the batched calls; None = every usable CPU."""
    return lib().cno_set_num_threads(int(n if n is not None else usable_cpus()))
