"""The simulator half (rows A1-A4) has no reference source (Gazebo owns it), so the oracle's simulator is
pinned by analytic known answers: ray vs a single disc / the walls in closed form, straight-line /
pure-rotation / arc robot motion, constant-velocity pedestrians with wall clamping, the crowd schedule, and
the deterministic sin/cos against libm.  CPU only.  (The GPU simulator is then required to match the oracle's
bit for bit by tests/test_gpu_parity.py.)"""
import ctypes as C
import math

import numpy as np


def _raycast(oracle_mod, cfg, rx, ry, yaw, peds):
    L = oracle_mod.lib()
    peds = np.ascontiguousarray(peds, dtype=np.float64).reshape(-1, 2)
    out = np.zeros(cfg.n_rays)
    L.cno_raycast(C.byref(cfg), rx, ry, yaw, peds.ctypes.data_as(C.POINTER(C.c_double)), len(peds),
                  out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def test_det_sincos_matches_libm(oracle_mod):
    L = oracle_mod.lib()
    s, c = C.c_double(), C.c_double()
    worst = 0.0
    for x in np.concatenate([np.linspace(-12, 12, 20001), np.arange(360) * 6.28 / 359, [0.0, math.pi / 2, math.pi, 3.14]]):
        L.cno_det_sincos(float(x), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - math.sin(x)), abs(c.value - math.cos(x)))
    assert worst < 3e-16


def test_hypot_restatement_is_the_c_librarys(oracle_mod):
    """cno_hypot (glibc 2.35's algorithm spelled out: the reference's math.hypot under Python 2.7 is the C library's) against the
    hypot of the C library this test runs on, BIT FOR BIT, on the arguments the path feeds it (differences of coordinates in
    thousandths) and on general ones -- ENV:826 compares two speeds for exact equality, so the last bit is part of the semantics;
    within half an ulp of the exact value on Pythagorean-free samples is implied by glibc's own bound and not asserted here."""
    L = oracle_mod.lib()
    L.cno_hypot_array.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]; L.cno_hypot_array.restype = None
    libm = C.CDLL("libm.so.6"); libm.hypot.restype = C.c_double; libm.hypot.argtypes = [C.c_double, C.c_double]
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.integers(-3000, 3000, 1000000) / 1000.0 - rng.integers(-3000, 3000, 1000000) / 1000.0, rng.uniform(-5, 5, 1000000), [0.0, 3.0, 0.0, -2.5, 1.0]])
    y = np.concatenate([rng.integers(-3000, 3000, 1000000) / 1000.0 - rng.integers(-3000, 3000, 1000000) / 1000.0, rng.uniform(-5, 5, 1000000), [0.0, 0.0, -1.25, 0.0, 1e-17]])
    got = np.empty_like(x)
    L.cno_hypot_array(x.size, x.ctypes.data, y.ctypes.data, got.ctypes.data)
    assert np.array_equal(got[-5:], [0.0, 3.0, 1.25, 2.5, 1.0])
    ref = np.hypot(x, y)                                    # numpy's loop calls the C library's hypot
    assert all(libm.hypot(float(x[i]), float(y[i])) == ref[i] for i in range(0, x.size, 997))
    assert np.array_equal(got, ref), "this C library's hypot is not glibc 2.35's algorithm: %d of %d differ" % (int((got != ref).sum()), x.size)
    # and the one the kernel used through round 6, sqrt(fma(a, a, b b)), is NOT that function (why it was replaced)
    a, b = np.maximum(np.abs(x), np.abs(y)), np.minimum(np.abs(x), np.abs(y))
    old = np.array([math.sqrt(math.fma(float(a[i]), float(a[i]), float(b[i] * b[i]))) for i in range(0, 20000)]) if hasattr(math, "fma") else None
    if old is not None:
        assert (old != ref[:20000]).mean() > 0.02


def test_lidar_single_disc_closed_form(oracle_mod):
    cfg = oracle_mod.make_config(n_peds=1, room_half=5.0, lidar_offset_x=0.0)
    d, r = 0.40, cfg.ped_radius
    rg = _raycast(oracle_mod, cfg, 0.0, 0.0, 0.0, [[d, 0.0]])
    step = cfg.lidar_span / 359
    for k in range(360):
        a = k * step
        perp = abs(d * math.sin(a))
        if math.cos(a) > 0 and perp < r:
            exp = d * math.cos(a) - math.sqrt(r * r - perp * perp)
            assert abs(rg[k] - exp) < 1e-12, k
        else:
            assert math.isinf(rg[k]), k
    assert abs(rg[0] - (d - r)) < 1e-15            # forward ray: centre distance minus radius
    # yaw rotates the scan: the disc appears at robot-frame angle -yaw
    rg2 = _raycast(oracle_mod, cfg, 0.0, 0.0, 10 * step, [[d, 0.0]])
    assert np.isfinite(rg2[359 - 10 + 1]) or np.isfinite(rg2[350])


def test_lidar_walls_closed_form_and_range_limits(oracle_mod):
    cfg = oracle_mod.make_config(n_peds=0, room_half=1.40, lidar_offset_x=0.0)
    rg = _raycast(oracle_mod, cfg, 1.0, -1.0, 0.0, np.zeros((0, 2)))     # 0.4 m from the +x and -y walls
    step = cfg.lidar_span / 359
    for k in range(360):
        a = k * step
        cands = []
        if math.cos(a) > 1e-12: cands.append(0.4 / math.cos(a))
        if math.sin(a) < -1e-12: cands.append(0.4 / -math.sin(a))
        if math.cos(a) < -1e-12: cands.append(2.4 / -math.cos(a))
        if math.sin(a) > 1e-12: cands.append(2.4 / math.sin(a))
        t = min(cands)
        if t > cfg.lidar_max:
            assert math.isinf(rg[k]), k
        else:
            assert abs(rg[k] - max(t, cfg.lidar_min)) < 1e-12, k
    # closer than range_min reads as range_min; the mount offset shifts the origin backwards along the heading
    cfg2 = oracle_mod.make_config(n_peds=0, room_half=1.40)
    rg = _raycast(oracle_mod, cfg2, 1.35, 0.0, 0.0, np.zeros((0, 2)))
    assert abs(rg[0] - (1.40 - (1.35 - 0.032))) < 1e-12


def test_robot_kinematics_known_answers(oracle_mod):
    o = oracle_mod.Oracle(n_envs=1, n_peds=0, spawn_x=0.0, spawn_y=0.0, spawn_yaw=0.0, room_half=5.0)
    o.hsim_reset()
    o.hsim_advance(150, 0.2, 0.0)                   # straight line: x += v * dt
    r = o.sim_state()[0]
    assert abs(r[0] - 0.03) < 1e-15 and r[1] == 0.0 and r[2] == 0.0
    o.hsim_advance(150, 0.0, 2.0)                   # pure rotation: yaw += w * dt
    r = o.sim_state()[0]
    assert abs(r[0] - 0.03) < 1e-15 and abs(r[2] - 0.3) < 1e-15
    # arc: many short intervals of the mid-point rule converge to the exact unicycle arc
    o.hsim_reset()
    v, w, n = 0.2, 1.0, 1000
    for _ in range(n):
        o.hsim_advance(1, v, w)
    r = o.sim_state()[0]
    T = n / 1000.0
    assert abs(r[0] - v / w * math.sin(w * T)) < 1e-7 and abs(r[1] - v / w * (1 - math.cos(w * T))) < 1e-7
    # yaw wraps into (-pi, pi]
    o.hsim_reset()
    for _ in range(30):
        o.hsim_advance(150, 0.0, 2.0)
    assert -math.pi < o.sim_state()[0][2] <= math.pi


def test_pedestrian_schedule_integration_and_walls(oracle_mod):
    P = 4
    o = oracle_mod.Oracle(n_envs=1, n_peds=P, ped_mode=1, ped_cycle_ms=400, room_half=1.40)
    init = np.array([[[0.0, 0.0], [0.5, 0.5], [1.30, 0.0], [-1.0, -1.3]]])
    vel = np.array([[[0.1, -0.05], [0.0, 0.2], [0.2, 0.0], [0.0, -0.2]]])
    o.set_ped_init(init); o.set_ped_preset_vel(vel)
    o.hsim_reset()
    o.hsim_advance(1000, 0.0, 0.0)
    _, pp, pv, _ = o.sim_state()
    # pedestrian i starts moving at t = 100*i ms (CROWD:144 stagger) and keeps its constant preset velocity
    for i in range(2):
        moved = (1000 - 100 * i) / 1000.0
        assert np.allclose(pp[i], init[0, i] + vel[0, i] * moved, atol=1e-12)
    lim = 1.40 - 0.0505
    assert abs(pp[2][0] - lim) < 1e-15 and pp[2][1] == 0.0          # clamped at the +x wall, slides
    assert abs(pp[3][1] + lim) < 1e-15
    assert np.array_equal(pv, vel[0])
    # a simulator reset zeroes the twists; a pedestrian stands still until its next scheduled update
    o.hsim_reset()
    _, pp, pv, _ = o.sim_state()
    assert np.array_equal(pp, init[0]) and not pv.any()
    # crowd clock is 1000 ms; update instants are 100*i + 400*m: ped 2 is due exactly at 1000, ped 3 at 1100, ped 0 at 1200
    o.hsim_advance(50, 0.0, 0.0)                     # [1000, 1050): only pedestrian 2 gets its velocity back
    _, pp, pv, _ = o.sim_state()
    assert np.array_equal(pp[[0, 1, 3]], init[0][[0, 1, 3]]) and not pv[[0, 1, 3]].any()
    assert np.array_equal(pv[2], vel[0, 2]) and abs(pp[2][0] - (1.30 + 0.2 * 0.05)) < 1e-15
    o.hsim_advance(200, 0.0, 0.0)                    # [1050, 1250): ped 3 at 1100, ped 0 at 1200; ped 1 not before 1300
    _, pp, pv, _ = o.sim_state()
    assert np.array_equal(pv[0], vel[0, 0]) and np.array_equal(pv[3], vel[0, 3]) and not pv[1].any()
    assert np.allclose(pp[0], init[0, 0] + vel[0, 0] * 0.05, atol=1e-15)


def test_random_walker_velocity_law(oracle_mod):
    L = oracle_mod.lib()
    u = np.array([L.cno_rng_u01(1234, e, 1, p, b) for e in range(20) for p in range(20) for b in range(10)])
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.02 and abs(u.std() - 12 ** -0.5) < 0.02
    assert L.cno_rng_u01(1234, 3, 1, 2, 7) == L.cno_rng_u01(1234, 3, 1, 2, 7) != L.cno_rng_u01(1234, 4, 1, 2, 7)
    o = oracle_mod.Oracle(n_envs=3, n_peds=20, ped_cycle_ms=1400)
    o.reset()
    for _ in range(30):
        o.step(np.zeros((3, 2)))
    for e in range(3):
        pv = o.sim_state(e)[2]
        assert np.abs(pv).max() <= 0.2 and np.abs(pv).max() > 0.05     # CROWD:101-102 U(-0.2, 0.2)
    xy = o.get_ped_init()
    assert np.abs(xy).max() <= 1.30 and (np.hypot(xy[..., 0] - 1.0, xy[..., 1] + 1.0) >= 0.4).all()


def test_rounding_restatements_match_python_and_numpy(oracle_mod):
    """A29: round(float, nd) (correctly rounded decimal, what the harness runs the reference under) and
    round(np.float64, nd) / np.around (multiply, rint, divide) -- on random values, on exact binary ties
    (k/16 + 1/32 style dyadics such as 0.0625 -> 0.062) and on decimal near-ties (x.xxx5 literals)."""
    L = oracle_mod.lib()
    rng = np.random.default_rng(9)
    xs = np.concatenate([rng.uniform(-3.5, 3.5, 20000), rng.uniform(-1e-3, 1e-3, 2000),
                         np.arange(-2000, 2000) / 16.0 + 1.0 / 32.0,            # exact ties at 3 decimals? (0.03125 ...)
                         np.arange(-4000, 4000) / 2000.0 + 0.0005,              # decimal half-way literals at 3 digits
                         np.arange(-400, 400) / 200.0 + 0.005,                  # ... and at 2 digits
                         np.array([0.0625, 0.1875, 2.5e-4, -0.0625, 0.5, 1.5, 2.5, 0.6, 0.105, 0.12, 1e-9, -1e-9])])
    for nd in (2, 3):
        for x in xs:
            x = float(x)
            assert L.cno_py_round(x, nd) == round(x, nd), (x, nd)
            assert L.cno_np_around(x, nd) == float(np.around(np.float64(x), nd)), (x, nd)
            assert L.cno_np_around(x, nd) == float(round(np.float64(x), nd)), (x, nd)


# ---- cn_config.ped_contact = 1 (row A2): frictionless rigid contact, WORLD:86-145 -------------------------------------
def _contact_world(oracle_mod, init, vel, **kw):
    init, vel = np.asarray(init, dtype=np.float64)[None], np.asarray(vel, dtype=np.float64)[None]
    o = oracle_mod.Oracle(n_envs=1, n_peds=init.shape[1], ped_mode=1, ped_contact=1, ped_cycle_ms=100 * init.shape[1],
                          room_half=2.4, spawn_x=-2.0, spawn_y=-2.0, spawn_yaw=0.0, **kw)
    o.set_ped_init(init); o.set_ped_preset_vel(vel)
    o.hsim_reset()
    return o


def test_contact_head_on_pair_stops(oracle_mod):
    """Two discs approach head-on at +-0.2 m/s: they meet when the gap closes, stop (inelastic, equal mass) and never overlap
    by more than one physics tick of closing motion; the line of centres keeps its direction (no tunnelling)."""
    o = _contact_world(oracle_mod, [[-0.3, 0.0], [0.3, 0.0]], [[0.2, 0.0], [-0.2, 0.0]], ped_stagger_ms=0)
    gaps = []
    for _ in range(30):                                   # 3 s: the gap of 0.6 - 0.101 closes at 0.4 m/s after ~1.25 s
        o.hsim_advance(100, 0.0, 0.0)
        _, pp, pv, _ = o.sim_state()
        gaps.append(pp[1, 0] - pp[0, 0])
        assert abs(pp[0, 1]) < 1e-12 and abs(pp[1, 1]) < 1e-12
    gaps = np.array(gaps)
    assert (gaps > 2 * 0.0505 - 0.4 * 0.010 - 1e-9).all()         # never deeper than one 10 ms tick of closing speed
    assert abs(gaps[-1] - 2 * 0.0505) < 5e-3                       # resting in contact
    assert np.abs(pv[:, 0]).max() < 1e-9                           # stopped
    assert gaps[0] < 0.6 and (np.diff(gaps) <= 1e-12).all()        # monotone approach, no bounce
    # the preset velocity comes back at the next crowd assignment (set_model_state re-asserts the twist, CROWD:128-144)
    # and the contact takes it away again within the same tick
    o.hsim_advance(400, 0.0, 0.0)
    _, pp, pv, _ = o.sim_state()
    assert pp[1, 0] - pp[0, 0] > 2 * 0.0505 - 0.4 * 0.010 - 1e-9


def test_contact_oblique_keeps_the_tangential_motion(oracle_mod):
    """Frictionless: only the normal component of the relative velocity is removed."""
    o = _contact_world(oracle_mod, [[-0.2, 0.0], [0.2, 0.0]], [[0.2, 0.1], [-0.2, 0.1]], ped_stagger_ms=0)
    o.hsim_advance(2000, 0.0, 0.0)
    _, pp, pv, _ = o.sim_state()
    assert np.allclose(pv[:, 1], 0.1, atol=1e-9) and np.abs(pv[:, 0]).max() < 1e-6      # both keep drifting in +y, side by side
    assert abs((pp[1, 0] - pp[0, 0]) - 2 * 0.0505) < 5e-3 and abs(pp[0, 1] - pp[1, 1]) < 1e-9


def test_contact_robot_pushes_a_disc(oracle_mod):
    """The robot is kinematic: a standing disc in its way is pushed ahead at the robot's speed and stays in front of it."""
    o = _contact_world(oracle_mod, [[-1.5, -2.0]], [[0.0, 0.0]])
    rc, r = 0.09, 0.0505                                   # robot_clearance (contact radius of the robot), ped_radius
    for k in range(40):                                    # robot drives +x at 0.2 m/s from x = -2: contact after ~(0.5 - 0.1405) / 0.2 s
        o.hsim_advance(100, 0.2, 0.0)
        robot, pp, pv, _ = o.sim_state()
        assert pp[0, 0] - robot[0] > rc + r - 0.2 * 0.010 - 1e-9          # never deeper than one tick of the robot's motion
    assert abs((pp[0, 0] - robot[0]) - (rc + r)) < 3e-3 and abs(pv[0, 0] - 0.2) < 1e-9 and abs(pp[0, 1] + 2.0) < 1e-12
    assert robot[0] > -2.0 + 0.2 * 3.9                     # the robot was not slowed down


def test_contact_mode_without_contacts_is_the_plain_simulator(oracle_mod):
    """Far-apart walkers: the 10 ms ticks change the rounding of the integration, nothing else (1e-12 over 30 s)."""
    kw = dict(n_envs=2, n_peds=6, ped_vmax=0.05, room_half=3.0, max_steps=400, seed=5)
    a, b = oracle_mod.Oracle(ped_contact=0, **kw), oracle_mod.Oracle(ped_contact=1, **kw)
    init = np.array([[[-2.5, -2.5], [-2.5, 2.5], [2.5, 2.5], [0.0, 2.5], [0.0, -2.5], [2.5, -2.5]]] * 2)
    for o in (a, b):
        o.set_ped_init(init); o.reset()
    for t in range(200):
        act = np.array([[0.05, 0.3], [0.1, -0.2]])
        a.step(act); b.step(act)
    for e in range(2):
        ra, pa, va, _ = a.sim_state(e); rb, pb, vb, _ = b.sim_state(e)
        assert np.allclose(pa, pb, atol=1e-10) and np.array_equal(va, vb)
        assert np.allclose(ra, rb, atol=1e-3)              # the robot's arc is integrated in 10 ms pieces (more accurate)


# ---- cn_config.ped_mode = 2: social-force pedestrians (BASELINE north_star; include/crowdnav.h states the model) -------------
# No reference source (CROWD:98-126 is a random-velocity walker), so the model is pinned by analytic cases: relaxation to the
# desired speed, antisymmetric pair repulsion, the wall stand-off distance, the goal sequence, and the deterministic exp.
def _sf_world(oracle_mod, pos, goals, **kw):
    pos, goals = np.asarray(pos, dtype=np.float64), np.asarray(goals, dtype=np.float64)
    P = len(pos)
    cfg = dict(n_envs=1, n_peds=P, ped_mode=2, room_half=6.0, spawn_x=5.5, spawn_y=5.5, spawn_yaw=0.0, seed=77)
    cfg.update(kw)
    o = oracle_mod.Oracle(**cfg)
    o.set_ped_init(pos[None])
    o.hsim_reset()
    aux = np.concatenate([goals, np.zeros((P, 1))], 1)
    o.set_state(0, None, None, None, None, None, ped_aux=aux)
    L = oracle_mod.lib()
    v0 = np.array([o.cfg.ped_vmax * (0.5 + 0.5 * L.cno_rng_u01(o.cfg.seed, 0, 4, i, 0)) for i in range(P)])
    return o, v0


def test_det_exp_matches_libm(oracle_mod):
    L = oracle_mod.lib()
    xs = np.concatenate([np.linspace(-12.5, 3.0, 40001), [0.0, -1.0, 1.0, -0.5 * math.log(2), 0.5 * math.log(2), -700.0, 700.0]])
    worst = max(abs(L.cno_det_exp(float(x)) / math.exp(x) - 1.0) for x in xs)
    assert worst < 4e-16
    assert L.cno_det_exp(0.0) == 1.0 and L.cno_det_exp(-701.0) == 0.0 and math.isinf(L.cno_det_exp(701.0))


def test_social_force_relaxes_to_the_desired_speed(oracle_mod):
    """One pedestrian, nothing within reach: dv/dt = (v0 e - v) / tau with explicit 10 ms ticks, v_n = v0 (1 - (1 - h / tau)^n)."""
    o, v0 = _sf_world(oracle_mod, [[-4.0, 0.0]], [[4.0, 0.0]])
    assert 0.1 <= v0[0] <= 0.2                            # ped_vmax (0.5 + 0.5 u)
    for n in (1, 10, 50, 300):
        o.hsim_reset(); o.set_state(0, None, None, None, None, None, ped_aux=[[4.0, 0.0, 0.0]])
        o.hsim_advance(10 * n, 0.0, 0.0)
        _, pp, pv, _ = o.sim_state()
        want = v0[0] * (1.0 - (1.0 - 0.01 / 0.5) ** n)
        assert abs(pv[0, 0] - want) < 1e-13 and pv[0, 1] == 0.0 and pp[0, 1] == 0.0, n
    assert abs(pv[0, 0] - v0[0]) < 3e-3 * v0[0]           # 3 s = 6 tau: within 0.3 % of the desired speed
    # position = h * sum of the velocities (semi-implicit Euler: v first, then x)
    want_x = -4.0 + 0.01 * sum(v0[0] * (1.0 - 0.98 ** k) for k in range(1, 301))
    assert abs(pp[0, 0] - want_x) < 1e-12


def test_social_force_tick_is_part_of_the_model(oracle_mod):
    """sf_tick_ms: the explicit scheme's step.  With 50 ms ticks the relaxation follows v_n = v0 (1 - (1 - 0.05 / tau)^n); an advance
    that is not a multiple of the tick ends with a shorter one (160 ms = 3 x 50 + 10)."""
    o, v0 = _sf_world(oracle_mod, [[-4.0, 0.0]], [[4.0, 0.0]], sf_tick_ms=50)
    o.hsim_advance(500, 0.0, 0.0)
    _, pp, pv, _ = o.sim_state()
    assert abs(pv[0, 0] - v0[0] * (1.0 - 0.9 ** 10)) < 1e-13
    o.hsim_reset(); o.set_state(0, None, None, None, None, None, ped_aux=[[4.0, 0.0, 0.0]])
    o.hsim_advance(160, 0.0, 0.0)
    _, pp, pv, _ = o.sim_state()
    want = v0[0] * (1.0 - 0.9 ** 3); want = want + (v0[0] - want) / 0.5 * 0.01
    assert abs(pv[0, 0] - want) < 1e-13


def test_social_force_pair_repulsion_is_antisymmetric_and_has_the_stated_strength(oracle_mod):
    """Two pedestrians 0.2 m apart, goals straight ahead in +y: after one tick from rest the x-velocities are exactly opposite and
    equal h A exp((2 r - d) / B) -- to the model's force grid of 2^-36 m/s^2 (round 5: pair components are summed exactly on
    that grid, so the sum has no order); they drift apart until the force has faded, never crossing."""
    o, v0 = _sf_world(oracle_mod, [[-0.1, 0.0], [0.1, 0.0]], [[-0.1, 5.0], [0.1, 5.0]])
    c = o.cfg
    o.hsim_advance(10, 0.0, 0.0)
    _, pp, pv, _ = o.sim_state()
    f = c.sf_A * math.exp((2 * c.ped_radius - 0.2) / c.sf_B)
    assert pv[0, 0] == -pv[1, 0] and abs(pv[1, 0] - f * 0.01) <= 0.5 * 2.0 ** -36 * 0.01 + 1e-15
    assert np.allclose(pv[:, 1], v0 * 0.01 / c.sf_tau, rtol=0, atol=1e-15)      # the goal term alone drives y
    gaps = []
    for _ in range(100):
        o.hsim_advance(100, 0.0, 0.0)
        _, pp, pv, _ = o.sim_state()
        gaps.append(pp[1, 0] - pp[0, 0])
    gaps = np.array(gaps)
    assert gaps[0] > 0.2 and gaps.max() > 0.35 and (gaps > 0.2).all()           # pushed apart, never through each other
    assert (pp[:, 1] > 0.9).all()                                              # ... while both keep walking to their goals


def test_social_force_pair_sums_are_exact_and_order_free(oracle_mod):
    """Round 5: every pedestrian-pedestrian force component enters the sum as a multiple of 2^-36 m/s^2, so the total is exact and has
    no order (the kernels evaluate each unordered pair once and scatter +- with LDS atomics).  Checked on the oracle: a ring of
    pedestrians listed in two different index orders ends the tick with the same pair accelerations, pedestrian for pedestrian, to
    within the few grid steps the (index-keyed) goal term's own rounding leaves in the difference; the configuration bound that
    keeps the sums exact is enforced at creation.  (GPU = oracle bit for bit on these worlds: tests/test_gpu_parity.py.)"""
    rng = np.random.default_rng(5)
    P = 24
    ang = rng.uniform(0, 2 * np.pi, P); rad = rng.uniform(0.2, 0.9, P)
    pos = np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1)
    goals = pos * 4.0
    perm = rng.permutation(P)
    o1, v1 = _sf_world(oracle_mod, pos, goals, sf_goal_eps=0.0)
    o2, v2 = _sf_world(oracle_mod, pos[perm], goals[perm], sf_goal_eps=0.0)
    o1.hsim_advance(10, 0.0, 0.0); o2.hsim_advance(10, 0.0, 0.0)
    _, _, pv1, _ = o1.sim_state(); _, _, pv2, _ = o2.sim_state()
    # the pair term: velocity change minus the (per-pedestrian, order-free) goal term v0 e_goal / tau * h
    c = o1.cfg
    def pair_part(pv, v0, p_, g_):
        e = (g_ - p_) / np.linalg.norm(g_ - p_, axis=1, keepdims=True)
        return pv - (v0[:, None] * e) / c.sf_tau * 0.01
    a1, a2 = pair_part(pv1, v1, pos, goals), pair_part(pv2, v2, pos[perm], goals[perm])
    assert np.abs(a1).max() > 1e-3                                   # the ring is crowded enough to repel
    # same pedestrians, other summation order: identical to the last bit once the desired-speed draw (keyed by index) is taken out
    q = 2.0 ** -36 * 0.01
    assert np.abs(a1[perm] - a2).max() <= 4 * q                      # (the goal term's own rounding differs with v0_i: a few grid steps)
    L = oracle_mod.lib()
    import ctypes as C
    bad = oracle_mod.make_config(n_envs=1, n_peds=100, ped_mode=2, sf_B=0.005)         # 100 * 0.8 * e^(0.101 / 0.005) >> 2^15
    h = C.c_void_p()
    assert L.cno_create(C.byref(bad), C.byref(h)) == -2


def test_social_force_wall_stand_off(oracle_mod):
    """A pedestrian whose goal lies behind the +x wall stops where the wall's push equals the goal's pull:
    v0 / tau = A_w exp((r - d) / B_w)  ->  d = r - B_w ln(v0 / (tau A_w))."""
    o, v0 = _sf_world(oracle_mod, [[1.0, 0.3]], [[9.0, 0.3]], room_half=2.0, spawn_x=-1.8, spawn_y=-1.8)
    c = o.cfg
    o.hsim_advance(60000, 0.0, 0.0)
    _, pp, pv, _ = o.sim_state()
    d_eq = c.ped_radius - c.sf_wall_B * math.log(v0[0] / (c.sf_tau * c.sf_wall_A))
    assert abs((2.0 - pp[0, 0]) - d_eq) < 1e-6 and abs(pv[0, 0]) < 1e-7 and abs(pp[0, 1] - 0.3) < 1e-9
    assert d_eq > c.ped_radius                                                   # it stands off, it is not clamped into the wall


def test_social_force_goal_sequence_and_robot_repulsion(oracle_mod):
    """A goal within sf_goal_eps is replaced by the next one of the pedestrian's counter-based sequence (stream 3, uniform in the
    room shrunk by 0.1 m); a robot parked next to a pedestrian pushes it away along the line of centres."""
    o, v0 = _sf_world(oracle_mod, [[0.0, 0.0], [3.0, 3.0]], [[0.05, 0.0], [3.0, -3.0]])
    L = oracle_mod.lib()
    c = o.cfg
    o.hsim_advance(10, 0.0, 0.0)
    aux = o.get_state(0)["ped_aux"]
    lo, span = -c.room_half + 0.1, 2 * c.room_half - 0.2
    assert aux[0, 2] == 1.0 and aux[1, 2] == 0.0                                  # pedestrian 0 was within 0.1 m of its goal
    assert aux[0, 0] == lo + span * L.cno_rng_u01(c.seed, 0, 3, 0, 2) or abs(aux[0, 0] - (lo + span * L.cno_rng_u01(c.seed, 0, 3, 0, 2))) < 1e-15
    assert abs(aux[0, 1] - (lo + span * L.cno_rng_u01(c.seed, 0, 3, 0, 3))) < 1e-15
    assert np.array_equal(aux[1, :2], [3.0, -3.0])
    # robot 0.2 m to the left of a pedestrian at rest whose goal is straight up: the push is +x
    o2, _ = _sf_world(oracle_mod, [[0.0, 0.0]], [[0.0, 5.0]], spawn_x=-0.2, spawn_y=0.0)
    o2.hsim_advance(10, 0.0, 0.0)
    _, pp, pv, _ = o2.sim_state()
    f = c.sf_A * math.exp((c.ped_radius + c.robot_clearance - 0.2) / c.sf_B)
    assert abs(pv[0, 0] - f * 0.01) < 1e-15 and pv[0, 1] > 0.0


def test_social_force_crowd_keeps_moving_and_mostly_apart(oracle_mod):
    """Twenty pedestrians in the training room for 60 s: nobody leaves the room, speeds stay under the 1.3 v0 cap, goals keep
    being reached, and overlaps (the model has no hard contact) are rare and shallow."""
    o = oracle_mod.Oracle(n_envs=4, n_peds=20, ped_mode=2, seed=5, max_steps=100000)
    o.reset()
    L = oracle_mod.lib()
    overlap = deep = 0
    for t in range(375):
        o.step(np.zeros((4, 2)))
        for e in range(4):
            _, pp, pv, _ = o.sim_state(e)
            assert np.abs(pp).max() <= 1.4 - 0.0505 + 1e-12
            d = np.hypot(pp[:, None, 0] - pp[None, :, 0], pp[:, None, 1] - pp[None, :, 1]) + np.eye(20)
            overlap += int((d < 0.101).sum()) // 2; deep += int((d < 0.06).sum()) // 2
    for e in range(4):
        _, pp, pv, _ = o.sim_state(e)
        v0 = np.array([0.2 * (0.5 + 0.5 * L.cno_rng_u01(5, e, 4, i, 0)) for i in range(20)])
        assert (np.hypot(pv[:, 0], pv[:, 1]) <= 1.3 * v0 + 1e-12).all()
        assert o.get_state(e)["ped_aux"][:, 2].sum() >= 20        # on average every pedestrian reached a goal
    assert overlap < 0.02 * 375 * 4 * 190 and deep == 0


# ---- cn_config.wheel_accel: the diff-drive plugin's wheel-speed ramp (XACRO:57-72; include/crowdnav.h states the model) ----------
def _wheel_sim(oracle_mod, **kw):
    o = oracle_mod.Oracle(n_envs=1, n_peds=0, wheel_accel=1.0, spawn_x=0.0, spawn_y=0.0, spawn_yaw=0.0, room_half=5.0, **kw)
    o.hsim_reset()
    return o


def test_wheel_ramp_reaches_a_forward_command_in_22_ticks(oracle_mod):
    """From rest to v = 0.22 m/s at 1 m/s^2 on 10 ms ticks: the wheel speed is 0.01 k after tick k, the tick's speed moves the robot
    over that tick, so 150 ms cover 0.01 * 0.01 * (1 + ... + 15) = 0.012 m (the kinematic robot: 0.033 m) and /odom reports 0.15 m/s;
    the command is reached on tick 22 and held."""
    o = _wheel_sim(oracle_mod)
    o.hsim_advance(150, 0.22, 0.0)
    r = o.sim_state()[0]
    assert abs(r[3] - 0.15) < 1e-12 and r[4] == 0.0
    assert abs(r[0] - 0.012) < 1e-12 and r[1] == 0.0 and r[2] == 0.0
    o.hsim_advance(70, 0.22, 0.0)            # ticks 16 .. 22
    r = o.sim_state()[0]
    assert abs(r[3] - 0.22) < 1e-12
    x22 = 0.01 * 0.01 * sum(range(1, 23))
    assert abs(r[0] - x22) < 1e-12
    o.hsim_advance(100, 0.22, 0.0)
    assert abs(o.sim_state()[0][0] - (x22 + 0.022)) < 1e-12 and o.sim_state()[0][3] == 0.22
    # and back down: a stop command takes the same 0.22 s
    o.hsim_advance(100, 0.0, 0.0)
    assert abs(o.sim_state()[0][3] - 0.12) < 1e-12
    o.hsim_advance(120, 0.0, 0.0)
    assert o.sim_state()[0][3] == 0.0
    k = oracle_mod.Oracle(n_envs=1, n_peds=0, spawn_x=0.0, spawn_y=0.0, spawn_yaw=0.0, room_half=5.0)   # wheel_accel 0: kinematic
    k.hsim_reset(); k.hsim_advance(150, 0.22, 0.0)
    assert abs(k.sim_state()[0][0] - 0.033) < 1e-12 and k.sim_state()[0][3] == 0.22


def test_wheel_ramp_either_wheel_within_tolerance_releases_both(oracle_mod):
    """gazebo_ros_diff_drive's condition is an OR: when ONE wheel is within 0.01 m/s of its target both are set to their targets.
    v = 0.1, w = 0.9375 -> targets 0.025 / 0.175 m/s: two ticks of +0.01 on both wheels (pure translation), then the left wheel is
    0.005 from its target and the right one jumps from 0.02 to 0.175."""
    o = _wheel_sim(oracle_mod)
    o.hsim_advance(20, 0.1, 0.9375)
    r = o.sim_state()[0]
    assert abs(r[3] - 0.02) < 1e-15 and abs(r[4]) < 1e-15 and abs(r[0] - 0.0003) < 1e-15 and r[2] == 0.0
    o.hsim_advance(10, 0.1, 0.9375)
    r = o.sim_state()[0]
    assert abs(r[3] - 0.1) < 1e-15 and abs(r[4] - 0.9375) < 1e-13
    assert abs(r[2] - 0.009375) < 1e-15                      # that tick already turns at the full rate


def test_wheel_ramp_turn_command_and_odom_twist_in_the_observation(oracle_mod):
    """A pure turn command w = 2 rad/s (wheel targets -0.16 / +0.16 m/s) takes 16 ticks; the observation's twist features
    (ENV:267-268: -v cos(w), v sin(w) with the ANGULAR velocity used as the angle) are built from the wheels' twist, not the command."""
    o = _wheel_sim(oracle_mod)
    o.hsim_advance(100, 0.0, 2.0)
    r = o.sim_state()[0]
    assert abs(r[4] - 0.1 * 2 / 0.16) < 1e-12 and abs(r[3]) < 1e-15     # 10 ticks: wheels at -/+0.10 -> w = 1.25 rad/s
    yaw = sum(0.01 * k * 2 / 0.16 * 0.01 for k in range(1, 11))
    assert abs(r[2] - yaw) < 1e-12
    o2 = oracle_mod.Oracle(n_envs=1, n_peds=0, wheel_accel=1.0, max_steps=50)
    o2.reset()
    obs, _, _, _ = o2.step(np.array([[0.22, 0.0]]))
    v = 0.16                                                            # 150 + 10 ms after the command: 16 ticks of +0.01
    assert abs(obs[0, 364] - (-v)) < 1e-3 and abs(obs[0, 365]) < 1e-12  # state[364] = -v cos(0), state[365] = v sin(0)


def test_scan_f32_rounds_every_range_to_float32(oracle_mod):
    cfg = oracle_mod.make_config(n_peds=1, room_half=1.40, scan_f32=1)
    cfg0 = oracle_mod.make_config(n_peds=1, room_half=1.40)
    rg, rg0 = (_raycast(oracle_mod, c, 1.0, -1.0, 0.3, [[0.7, -0.8]]) for c in (cfg, cfg0))
    fin = np.isfinite(rg0)
    assert fin.sum() > 50 and np.array_equal(np.isfinite(rg), fin)
    assert np.array_equal(rg[fin], rg0[fin].astype(np.float32).astype(np.float64)) and not np.array_equal(rg[fin], rg0[fin])


def test_waypoint_reward_switch(oracle_mod):
    """ENV:1109-1116 pays `waypoint_reward` when the robot is within goal_eps of the way-point; 0 = the published log's reward
    (its largest episode return in 3021 episodes is 173 < 200, while here the bonus is paid several times per episode)."""
    L = oracle_mod.lib()
    for bonus in (200, 0):
        o = oracle_mod.Oracle(n_envs=1, n_peds=0, waypoint_reward=bonus)
        o.reset()
        # cno_compute_reward(handle, env, cur_head, cur_dist, prev_head, prev_dist, px, py, wpx, wpy, done)
        r = L.cno_compute_reward(o.h, 0, 0.1, 0.25, 0.2, 0.3, 0.5, -0.5, 0.4, -0.4, 0)
        assert r == -2 + 1 + 1 + bonus
