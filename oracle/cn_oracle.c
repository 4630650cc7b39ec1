/*
 * cn_oracle.c -- TEST INFRASTRUCTURE ONLY (see cn_oracle.h).
 *
 * Plain-C float64 restatement of the reference environment step.  ENV = environment_stage_1_nobonus.py,
 * UTL = utils.py, CROWD = crowd_behaviors/simulate_crowd.py (all under
 * /root/reference/turtlebot3_rl_sim/src).  Every block cites the lines it follows.
 *
 * Semantics pinned here (and by oracle/harness, which runs the reference's own Python):
 *   - Python-2 integer division at UTL:113 (360/359 == 1) and ENV:577 (len/2)
 *   - Python-3 round() (correctly rounded, ties-to-even on the exact binary value); differs from
 *     Python-2 round only on exact dyadic ties such as 0.0625
 *   - round(np.float64, n) and np.around use numpy's multiply / rint / divide
 *   - dict iteration = insertion order (uuid4 hash order of Python 2 is unreproducible)
 *   - time.time() is a virtual clock: sleep(d) adds d; the /scan wait adds scan_latency
 *   - shapely Point.buffer(r).boundary is the regular 64-gon with vertices at angle -k*pi/32;
 *     ring/segment intersection is the textbook two-parameter solve below ("parity unpinned":
 *     shapely/GEOS are not part of the reference tree)
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include "cn_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TY_NONE 0
#define TY_W 1
#define TY_O 2
#define ST_TRACK_OVERFLOW 1
#define ST_TTC_ZERO 2
#define ST_DT_ZERO 4

typedef struct { double x, y; } v2;

typedef struct {
    v2 pose;
    double dist;
    v2 dq[2];
    int dq_len;
    double t;     /* time stamp, or a duration after a match (ENV:710) */
    double speed; /* -1 until matched (ENV:667) */
    v2 vel;
    int id;       /* risk_mode gt: the pedestrian this entry is */
} track_t;

typedef struct {
    /* simulator (no reference source; DESIGN.md "physics") */
    double rx, ry, ryaw, rv, rw;
    double cmd_v, cmd_w;   /* wheel_accel > 0: the twist /cmd_vel last carried ((rv, rw) is then the wheels' real twist) */
    double* ped_p;
    double* ped_v;
    double* ped_init;
    double* ped_preset;
    double* ranges;
    double* ped_aux;   /* ped_mode 2: [P][3] goal x, goal y, goal counter */
    int64_t crowd_ms;
    /* virtual clock */
    double clock;
    /* Env attributes that persist between calls */
    double wpx, wpy;
    double prev_dist, prev_head;
    int done;
    v2 agent_dq[2];
    int agent_dq_len;
    double agent_vel_timestep;
    double bb;
    track_t tracks[CNO_MAX_TRACKS];
    int ntracks;
    double ego_score_cp;
    double collision_prob;
    double entry_cp[CNO_MAX_TRACKS];   /* debug: CP of every entry before the top-K cut (ENV:818-860) */
    double entry_ego[CNO_MAX_TRACKS];
    int ego_viol, social_viol, obst_steps;
    int ep_success, ep_failure;
    int ep_step;
    double ep_return, last_return;
    int status;
    int n_confirmed, n_entries;
    int pending_reset;
    int episodes;                                      /* finished episodes since creation */
    int last_ego_viol, last_social_viol, last_obst_steps, last_ep_steps;   /* the last finished episode's counters (TRAIN:142-147) */
} env_t;

struct cno_sim {
    cno_config cfg;
    int n, D;
    env_t* envs;
    double* lidar_c;
    double* lidar_s;
    double poly_c[64], poly_s[64];
};

static int g_threads = 1;

/* ------------------------------------------------------------------------------------------
 * numeric helpers
 * ---------------------------------------------------------------------------------------- */

/* cn_config.py2_round of the handle whose call is running (set on entry of every public function that takes a handle;
 * read-only inside the OpenMP region).  0: Python-3 round(), 1: Python-2.7 round() (floatobject.c _Py_double_round: correctly
 * rounded, an EXACT tie -- 2-valuation of x equal to -(nd + 1) -- goes away from zero). */
/* Thread-local, and set from the HANDLE's configuration by whichever thread is about to run that handle's arithmetic: on entry of
 * every public function that takes a handle and again at the top of every OpenMP iteration (worker threads have their own copy).
 * Handles with different py2_round values can therefore coexist AND be stepped concurrently from different threads; the
 * handle-less helpers (cno_py_round ...) use the calling thread's last cno_set_py2_round / handle call. */
static __thread int g_py2 = 0;
void cno_set_py2_round(int on) { g_py2 = on ? 1 : 0; }

/* Python round(x, nd): correctly rounded to nd decimals -- ties-to-even on the exact value (Python 3) or away from zero
 * (Python 2.7, g_py2) -- then the nearest double of that decimal.  x*p = y + err exactly (fma), so the only case the
 * rounded product can mislead rint() is y landing exactly on a half-integer. */
double cno_py_round(double x, int nd)
{
    double p = (nd == 3) ? 1000.0 : (nd == 2 ? 100.0 : pow(10.0, nd));
    if (!isfinite(x)) return x;
    double y = x * p;
    double r = rint(y);
    double d = y - r;
    if (fabs(d) == 0.5) {
        double err = fma(x, p, -y);
        if (err > 0.0) r = y + 0.5;
        else if (err < 0.0) r = y - 0.5;
        else if (g_py2) r = y + copysign(0.5, y);     /* exact tie: Python 2.7 rounds half away from zero */
    }
    return r / p;
}

/* numpy around / round(np.float64, nd) under Python 3: multiply, rint, divide */
double cno_np_around(double x, int nd)
{
    double p = (nd == 3) ? 1000.0 : (nd == 2 ? 100.0 : pow(10.0, nd));
    return rint(x * p) / p;
}

/* round(np.float64, nd) (ENV:255, ORIG:280, RW:209): Python 3 dispatches to np.float64.__round__ (numpy's arithmetic); the
 * Python-2.7 builtin converts its argument to a C double and rounds it like any float */
static double round_np64(double x, int nd) { return g_py2 ? cno_py_round(x, nd) : cno_np_around(x, nd); }

/* Deterministic sin/cos used by the simulator (physics + lidar direction table): only + * fma
 * and rint, so every IEEE-754 implementation returns the same bits.  Cody-Waite reduction by
 * pi/2 (three-part constant) followed by the classic degree-13/14 minimax kernels on [-pi/4, pi/4]. */
/* math.hypot (ENV:754, UTL:283-284, 409-413) = the C library's hypot under the reference's Python 2.7 (oracle/harness/refenv.py
 * binds math.hypot to libm for that reason; CPython >= 3.8 has its own, correctly rounded one, which differs from this image's on
 * 0.5 % of the path's arguments).  Its last bit decides exact comparisons downstream -- ENV:826 `relative_vel == 0` picks between
 * two formulas whose results differ by a factor of two, and a robot driving straight past a static object makes the two speeds
 * equal on paper -- so the function is part of the semantics and is spelled out here instead of left to whatever libm the oracle
 * is linked against: glibc 2.35's algorithm (sysdeps/ieee754/dbl-64/e_hypot.c, the kernel without hardware fma: the square root of
 * the plain sum of squares and one correction step, < 1 ulp), the C library the goldens were recorded on; bit-equal to this
 * image's hypot() on every sample of tests/test_simulator_known_answers.py::test_hypot_restatement_is_the_c_librarys.  The device
 * code restates the same operations (crowdnav_device.h cn_hypot).  Lengths of a few metres: glibc's scaling branches for huge /
 * tiny arguments never apply; hypot(x, 0) = |x|. */
double cno_hypot(double x, double y)
{
    double ax = fabs(x), ay = fabs(y);
    if (ax < ay) { double t = ax; ax = ay; ay = t; }
    if (ay == 0.0 || ax >= ay / 0x1p-54) return ax + ay;
    double h = sqrt(ax * ax + ay * ay), t1, t2;
    if (h <= 2.0 * ay) {
        double delta = h - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    } else {
        double delta = h - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    h -= (t1 + t2) / (2.0 * h);
    return h;
}
void cno_hypot_array(int n, const double* x, const double* y, double* out) { for (int i = 0; i < n; ++i) out[i] = cno_hypot(x[i], y[i]); }

void cno_det_sincos(double x, double* sn, double* cs)
{
    const double two_over_pi = 6.36619772367581382433e-01;
    const double p1 = 1.57079632673412561417e+00;  /* first 33 bits of pi/2 */
    const double p2 = 6.07710050630396597660e-11;  /* next 33 bits */
    const double p3 = 2.02226624879595063154e-21;  /* remainder */
    double fn = rint(x * two_over_pi);
    double r = fma(-fn, p1, x);
    r = fma(-fn, p2, r);
    r = fma(-fn, p3, r);
    double z = r * r;
    /* sin kernel */
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double ps = fma(z, S6, S5);
    ps = fma(z, ps, S4);
    ps = fma(z, ps, S3);
    ps = fma(z, ps, S2);
    ps = fma(z, ps, S1);
    double s = fma(r * z, ps, r);
    /* cos kernel */
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double pc = fma(z, C6, C5);
    pc = fma(z, pc, C4);
    pc = fma(z, pc, C3);
    pc = fma(z, pc, C2);
    pc = fma(z, pc, C1);
    double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    long q = (long)fn;
    switch (q & 3) {
    case 0: *sn = s;  *cs = c;  break;
    case 1: *sn = c;  *cs = -s; break;
    case 2: *sn = -s; *cs = -c; break;
    default: *sn = -c; *cs = s; break;
    }
}

static uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* counter-based uniform [0,1): stateless, keyed by (seed, global env index, stream, a, b) */
double cno_rng_u01(uint64_t seed, int64_t env, uint32_t stream, uint32_t a, uint32_t b)
{
    uint64_t h = mix64(seed ^ mix64((uint64_t)env));
    h = mix64(h ^ (((uint64_t)stream << 32) | (uint64_t)a));
    h = mix64(h ^ (uint64_t)b);
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

/* ------------------------------------------------------------------------------------------
 * simulator: pedestrians (CROWD:98-144 velocity law), diff drive, lidar.
 * ---------------------------------------------------------------------------------------- */

static double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

static void ped_advance(const cno_sim* s, env_t* e, int64_t gid, int64_t t0, int64_t t1)
{
    const cno_config* c = &s->cfg;
    const double lo = -c->room_half + c->ped_radius, hi = c->room_half - c->ped_radius;
    const int64_t T = c->ped_cycle_ms;
    for (int i = 0; i < c->n_peds; ++i) {
        double x = e->ped_p[2 * i], y = e->ped_p[2 * i + 1];
        double vx = e->ped_v[2 * i], vy = e->ped_v[2 * i + 1];
        int64_t offs = (int64_t)i * c->ped_stagger_ms;
        int64_t m = (t0 <= offs) ? 0 : (t0 - offs + T - 1) / T;
        int64_t a = offs + m * T; /* first assignment instant >= t0 */
        int64_t tc = t0;
        while (a < t1) {
            if (a > tc) {
                double ds = (double)(a - tc) / 1000.0;
                x = clampd(fma(vx, ds, x), lo, hi);
                y = clampd(fma(vy, ds, y), lo, hi);
                tc = a;
            }
            if (c->ped_mode == 0) { /* CROWD:101-102 random.uniform(-vmax, vmax) */
                double u0 = cno_rng_u01(c->seed, gid, 1u, (uint32_t)i, (uint32_t)(2 * m));
                double u1 = cno_rng_u01(c->seed, gid, 1u, (uint32_t)i, (uint32_t)(2 * m + 1));
                vx = fma(2.0 * c->ped_vmax, u0, -c->ped_vmax);
                vy = fma(2.0 * c->ped_vmax, u1, -c->ped_vmax);
            } else {
                vx = e->ped_preset[2 * i];
                vy = e->ped_preset[2 * i + 1];
            }
            a += T;
            m += 1;
        }
        if (t1 > tc) {
            double ds = (double)(t1 - tc) / 1000.0;
            x = clampd(fma(vx, ds, x), lo, hi);
            y = clampd(fma(vy, ds, y), lo, hi);
        }
        e->ped_p[2 * i] = x; e->ped_p[2 * i + 1] = y;
        e->ped_v[2 * i] = vx; e->ped_v[2 * i + 1] = vy;
    }
}

/* Diff-drive for one interval with (v, w) held: the mid-point rule of turtlebot3_fake.cpp:156-162
 * (x += ds*cos(th + dth/2), y += ds*sin(th + dth/2), th += dth). */
static void robot_advance(const cno_sim* s, env_t* e, int64_t ms)
{
    const cno_config* c = &s->cfg;
    double dts = (double)ms / 1000.0;
    double ds = e->rv * dts, dth = e->rw * dts;
    double sn, cs;
    cno_det_sincos(fma(0.5, dth, e->ryaw), &sn, &cs);
    double lim = c->room_half - c->robot_clearance;
    e->rx = clampd(fma(ds, cs, e->rx), -lim, lim);
    e->ry = clampd(fma(ds, sn, e->ry), -lim, lim);
    double th = e->ryaw + dth;
    if (th > M_PI) th -= 2.0 * M_PI;
    else if (th <= -M_PI) th += 2.0 * M_PI;
    e->ryaw = th;
}

/* cn_config.wheel_accel > 0 (XACRO:57-72): the wheel-speed ramp of libgazebo_ros_diff_drive.so, as include/crowdnav.h states it
 * (third-party plugin, restated from its published source: UpdateChild's branch on wheel_accel).  The command (cmd_v, cmd_w) is
 * what /cmd_vel last carried; (rv, rw) is the twist the wheels really have -- what /odom reports (ENV:239-243).  Plugin ticks of
 * at most 10 ms (updateRate 100); the tick's new wheel speeds move the robot over that tick by the mid-point rule. */
static void robot_advance_wheels(const cno_sim* s, env_t* e, int64_t ms)
{
    const cno_config* c = &s->cfg;
    const double a = c->wheel_accel, half = 0.5 * c->wheel_separation;
    const double tl = e->cmd_v - e->cmd_w * half, tr = e->cmd_v + e->cmd_w * half;
    double cl = e->rv - e->rw * half, cr = e->rv + e->rw * half;
    for (int64_t tt = 0; tt < ms; ) {
        const int64_t h = (ms - tt < 10) ? (ms - tt) : 10;
        const double ah = a * ((double)h / 1000.0);
        if (fabs(tl - cl) < 0.01 || fabs(tr - cr) < 0.01) { cl = tl; cr = tr; }
        else {
            cl += (tl >= cl) ? fmin(tl - cl, ah) : fmax(tl - cl, -ah);
            cr += (tr >= cr) ? fmin(tr - cr, ah) : fmax(tr - cr, -ah);
        }
        e->rv = (cl + cr) * 0.5;
        e->rw = (cr - cl) / c->wheel_separation;
        robot_advance(s, e, h);
        tt += h;
    }
}

/* cn_config.ped_contact = 1 (row A2; WORLD:86-145: rigid mu = 0 cylinders, r = 0.0505, 1 kg): the world advances in physics
 * ticks of at most 10 ms.  Per tick, in this order:
 *   1. crowd velocity assignments whose instant falls inside the tick [t, t + h) take effect at t (CROWD:98-144; exact when
 *      every interval is a multiple of 10 ms, which the reference's 150 / 10 / 100 ms are);
 *   2. the robot moves by the mid-point rule over h (it is kinematic: driven by its wheel controller, never pushed);
 *   3. every pedestrian integrates x <- clamp(fma(v, h, x)) into the room;
 *   4. contacts, Jacobi style (every correction is computed from the positions / velocities after step 3, summed in index
 *      order, then applied): two overlapping discs each back off half the penetration along the centre line and, if they are
 *      approaching, each gives up half the closing speed (inelastic, frictionless: a head-on pair stops); a disc overlapping
 *      the robot (radius robot_clearance) backs off the whole penetration and loses its closing speed relative to the robot's
 *      linear velocity (the robot pushes it);  then the room clamp again.
 * A pedestrian's velocity persists until its next assignment, as a Gazebo body's twist does between set_model_state calls. */
static void sim_advance_contact(const cno_sim* s, env_t* e, int64_t gid, int64_t ms)
{
    const cno_config* c = &s->cfg;
    const int P = c->n_peds;
    const double lo = -c->room_half + c->ped_radius, hi = c->room_half - c->ped_radius;
    const double r = c->ped_radius, rr2 = (2.0 * r) * (2.0 * r);
    const double Rr = r + c->robot_clearance, Rr2 = Rr * Rr;
    const int64_t T = c->ped_cycle_ms;
    double* dxy = (double*)malloc(sizeof(double) * 4 * (size_t)(P > 0 ? P : 1));
    int64_t t = e->crowd_ms;
    const int64_t t1 = t + ms;
    while (t < t1) {
        const int64_t h = (t1 - t < 10) ? (t1 - t) : 10;
        const double hs = (double)h / 1000.0;
        for (int i = 0; i < P; ++i) {                       /* 1. assignments with instant in [t, t + h) */
            int64_t offs = (int64_t)i * c->ped_stagger_ms;
            int64_t m = (t <= offs) ? 0 : (t - offs + T - 1) / T;
            int64_t a = offs + m * T;
            while (a < t + h) {
                if (c->ped_mode == 0) {
                    double u0 = cno_rng_u01(c->seed, gid, 1u, (uint32_t)i, (uint32_t)(2 * m));
                    double u1 = cno_rng_u01(c->seed, gid, 1u, (uint32_t)i, (uint32_t)(2 * m + 1));
                    e->ped_v[2 * i] = fma(2.0 * c->ped_vmax, u0, -c->ped_vmax);
                    e->ped_v[2 * i + 1] = fma(2.0 * c->ped_vmax, u1, -c->ped_vmax);
                } else {
                    e->ped_v[2 * i] = e->ped_preset[2 * i]; e->ped_v[2 * i + 1] = e->ped_preset[2 * i + 1];
                }
                a += T; m += 1;
            }
        }
        robot_advance(s, e, h);                              /* 2. */
        double syaw, cyaw;
        cno_det_sincos(e->ryaw, &syaw, &cyaw);
        const double rvx = e->rv * cyaw, rvy = e->rv * syaw; /* the robot's linear velocity in the world frame */
        for (int i = 0; i < P; ++i) {                       /* 3. */
            e->ped_p[2 * i] = clampd(fma(e->ped_v[2 * i], hs, e->ped_p[2 * i]), lo, hi);
            e->ped_p[2 * i + 1] = clampd(fma(e->ped_v[2 * i + 1], hs, e->ped_p[2 * i + 1]), lo, hi);
        }
        for (int i = 0; i < P; ++i) {                       /* 4. corrections from the post-integration state */
            const double xi = e->ped_p[2 * i], yi = e->ped_p[2 * i + 1], vxi = e->ped_v[2 * i], vyi = e->ped_v[2 * i + 1];
            double ax = 0.0, ay = 0.0, bx = 0.0, by = 0.0;   /* position and velocity corrections */
            for (int j = 0; j < P; ++j) {
                if (j == i) continue;
                const double ddx = xi - e->ped_p[2 * j], ddy = yi - e->ped_p[2 * j + 1];
                const double d2 = fma(ddx, ddx, ddy * ddy);
                if (!(d2 < rr2) || !(d2 > 0.0)) continue;
                const double d = sqrt(d2), nx = ddx / d, ny = ddy / d;
                const double pen = 2.0 * r - d;
                ax = fma(0.5 * pen, nx, ax); ay = fma(0.5 * pen, ny, ay);
                const double vn = fma(vxi - e->ped_v[2 * j], nx, (vyi - e->ped_v[2 * j + 1]) * ny);
                if (vn < 0.0) { bx = fma(-0.5 * vn, nx, bx); by = fma(-0.5 * vn, ny, by); }
            }
            {
                const double ddx = xi - e->rx, ddy = yi - e->ry;
                const double d2 = fma(ddx, ddx, ddy * ddy);
                if (d2 < Rr2 && d2 > 0.0) {
                    const double d = sqrt(d2), nx = ddx / d, ny = ddy / d;
                    const double pen = Rr - d;
                    ax = fma(pen, nx, ax); ay = fma(pen, ny, ay);
                    const double vn = fma(vxi - rvx, nx, (vyi - rvy) * ny);
                    if (vn < 0.0) { bx = fma(-vn, nx, bx); by = fma(-vn, ny, by); }
                }
            }
            dxy[4 * i] = ax; dxy[4 * i + 1] = ay; dxy[4 * i + 2] = bx; dxy[4 * i + 3] = by;
        }
        for (int i = 0; i < P; ++i) {
            e->ped_p[2 * i] = clampd(e->ped_p[2 * i] + dxy[4 * i], lo, hi);
            e->ped_p[2 * i + 1] = clampd(e->ped_p[2 * i + 1] + dxy[4 * i + 1], lo, hi);
            e->ped_v[2 * i] += dxy[4 * i + 2]; e->ped_v[2 * i + 1] += dxy[4 * i + 3];
        }
        t += h;
    }
    e->crowd_ms = t1;
    free(dxy);
}

/* Deterministic exp used by the social-force model: only + * fma rint and an exponent-field scaling, so every IEEE-754
 * implementation returns the same bits (the GPU kernel evaluates the same sequence).  Cody-Waite reduction by ln 2, degree-13
 * Taylor polynomial on |r| <= ln2 / 2 (truncation 4e-18 relative), |x| <= 700. */
double cno_det_exp(double x)
{
    const double LOG2E = 1.44269504088896338700e+00, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    if (x < -700.0) return 0.0;
    if (x > 700.0) return INFINITY;
    const double kf = rint(x * LOG2E);
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    double q = 1.0 / 6227020800.0;                  /* 1/13! */
    q = fma(q, r, 1.0 / 479001600.0);               /* 1/12! */
    q = fma(q, r, 1.0 / 39916800.0);
    q = fma(q, r, 1.0 / 3628800.0);
    q = fma(q, r, 1.0 / 362880.0);
    q = fma(q, r, 1.0 / 40320.0);
    q = fma(q, r, 1.0 / 5040.0);
    q = fma(q, r, 1.0 / 720.0);
    q = fma(q, r, 1.0 / 120.0);
    q = fma(q, r, 1.0 / 24.0);
    q = fma(q, r, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    union { uint64_t u; double d; } sc;
    sc.u = (uint64_t)((int64_t)kf + 1023) << 52;     /* 2^k, k in [-1010, 1010] */
    return q * sc.d;
}

/* ped_mode 2: goal m of pedestrian i = uniform in the room shrunk by 0.1 m, counter-based (stream 3); desired speed (stream 4) */
static void sf_goal(const cno_config* c, int64_t gid, int i, uint32_t m, double* gx, double* gy)
{
    const double lo = -c->room_half + 0.1, span = 2.0 * c->room_half - 0.2;
    *gx = fma(span, cno_rng_u01(c->seed, gid, 3u, (uint32_t)i, 2u * m), lo);
    *gy = fma(span, cno_rng_u01(c->seed, gid, 3u, (uint32_t)i, 2u * m + 1u), lo);
}
static double sf_desired_speed(const cno_config* c, int64_t gid, int i)
{
    return c->ped_vmax * fma(0.5, cno_rng_u01(c->seed, gid, 4u, (uint32_t)i, 0u), 0.5);
}

/* cn_config.ped_mode = 2 (BASELINE north_star "per-env pedestrian social-force integration"; include/crowdnav.h states the
 * model): Helbing-Molnar goal attraction + exponential repulsion from the other pedestrians, the four walls and the robot, on
 * physics ticks of at most sf_tick_ms (10 ms by default).  Per tick: the (kinematic) robot moves by the mid-point rule; every pedestrian's acceleration
 * is evaluated from the tick-start pedestrian state (Jacobi) and the robot's new position, contributions summed in the order
 * goal, pedestrians by index, walls -x +x -y +y, robot; semi-implicit Euler (v first, capped at 1.3 v0, then x, clamped into
 * the room).  Contributions whose exponent is below -12 are dropped (6e-6 of the strength).  A goal within sf_goal_eps at a
 * tick start is replaced by the next one of the pedestrian's sequence.  No reference source (CROWD:98-126 is a random-velocity
 * walker): pinned by the analytic cases of tests/test_simulator_known_answers.py. */
/* A pedestrian-pedestrian force component enters the sum on a grid of 2^-36 m/s^2 (1.5e-11): every partial sum of such values is
 * exactly representable (cn_create / cno_create bound P A e^{2r/B} below 2^15), so the total does not depend on the order of the
 * additions -- which is what lets the kernels evaluate each unordered pair ONCE and scatter +- its contribution (round 5). */
static double sf_quant(double v) { return rint(v * 68719476736.0) * (1.0 / 68719476736.0); }

static void sim_advance_sf(const cno_sim* s, env_t* e, int64_t gid, int64_t ms)
{
    const cno_config* c = &s->cfg;
    const int P = c->n_peds;
    const double H = c->room_half, r = c->ped_radius, lo = -H + r, hi = H - r;
    const double A = c->sf_A, B = c->sf_B, Aw = c->sf_wall_A, Bw = c->sf_wall_B, tau = c->sf_tau;
    const double eps2 = c->sf_goal_eps * c->sf_goal_eps, Rr = r + c->robot_clearance;
    double* nxt = (double*)malloc(sizeof(double) * 4 * (size_t)(P > 0 ? P : 1));
    const int64_t tick = c->sf_tick_ms > 0 ? c->sf_tick_ms : 10;
    int64_t t = 0;
    while (t < ms) {
        const int64_t h = (ms - t < tick) ? (ms - t) : tick;
        const double hs = (double)h / 1000.0;
        robot_advance(s, e, h);
        for (int i = 0; i < P; ++i) {
            const double xi = e->ped_p[2 * i], yi = e->ped_p[2 * i + 1], vxi = e->ped_v[2 * i], vyi = e->ped_v[2 * i + 1];
            const double v0 = sf_desired_speed(c, gid, i);
            double gx = e->ped_aux[3 * i], gy = e->ped_aux[3 * i + 1];
            double gdx = gx - xi, gdy = gy - yi, gd2 = fma(gdx, gdx, gdy * gdy);
            if (gd2 <= eps2) {                                   /* goal reached: the next one of this pedestrian's sequence */
                const uint32_t m = (uint32_t)e->ped_aux[3 * i + 2] + 1u;
                sf_goal(c, gid, i, m, &gx, &gy);
                e->ped_aux[3 * i] = gx; e->ped_aux[3 * i + 1] = gy; e->ped_aux[3 * i + 2] = (double)m;
                gdx = gx - xi; gdy = gy - yi; gd2 = fma(gdx, gdx, gdy * gdy);
            }
            double ex = 0.0, ey = 0.0;
            if (gd2 > 0.0) { const double ginv = 1.0 / sqrt(gd2); ex = gdx * ginv; ey = gdy * ginv; }
            double ax = (v0 * ex - vxi) / tau, ay = (v0 * ey - vyi) / tau;
            double sx = 0.0, sy = 0.0;                           /* the pedestrians' repulsion: an exact sum (sf_quant) */
            for (int j = 0; j < P; ++j) {
                if (j == i) continue;
                const double ddx = xi - e->ped_p[2 * j], ddy = yi - e->ped_p[2 * j + 1];
                const double d2 = fma(ddx, ddx, ddy * ddy);
                if (!(d2 > 0.0)) continue;
                const double d = sqrt(d2), arg = (2.0 * r - d) / B;
                if (arg < -12.0) continue;
                const double f = (A * cno_det_exp(arg)) * (1.0 / d);
                sx += sf_quant(f * ddx); sy += sf_quant(f * ddy);
            }
            ax = ax + sx; ay = ay + sy;
            {   /* walls: distance from the centre to the wall plane, pushing inwards */
                double arg = (r - (xi + H)) / Bw;
                if (!(arg < -12.0)) ax = ax + Aw * cno_det_exp(arg);
                arg = (r - (H - xi)) / Bw;
                if (!(arg < -12.0)) ax = ax - Aw * cno_det_exp(arg);
                arg = (r - (yi + H)) / Bw;
                if (!(arg < -12.0)) ay = ay + Aw * cno_det_exp(arg);
                arg = (r - (H - yi)) / Bw;
                if (!(arg < -12.0)) ay = ay - Aw * cno_det_exp(arg);
            }
            {   /* the robot, where the tick leaves it */
                const double ddx = xi - e->rx, ddy = yi - e->ry;
                const double d2 = fma(ddx, ddx, ddy * ddy);
                if (d2 > 0.0) {
                    const double d = sqrt(d2), arg = (Rr - d) / B;
                    if (!(arg < -12.0)) {
                        const double f = (A * cno_det_exp(arg)) * (1.0 / d);
                        ax = fma(f, ddx, ax); ay = fma(f, ddy, ay);
                    }
                }
            }
            double vx = fma(ax, hs, vxi), vy = fma(ay, hs, vyi);
            const double cap = 1.3 * v0, s2 = fma(vx, vx, vy * vy);
            if (s2 > cap * cap) { const double k = cap / sqrt(s2); vx = vx * k; vy = vy * k; }
            nxt[4 * i] = clampd(fma(vx, hs, xi), lo, hi); nxt[4 * i + 1] = clampd(fma(vy, hs, yi), lo, hi);
            nxt[4 * i + 2] = vx; nxt[4 * i + 3] = vy;
        }
        for (int i = 0; i < P; ++i) {
            e->ped_p[2 * i] = nxt[4 * i]; e->ped_p[2 * i + 1] = nxt[4 * i + 1];
            e->ped_v[2 * i] = nxt[4 * i + 2]; e->ped_v[2 * i + 1] = nxt[4 * i + 3];
        }
        t += h;
    }
    e->crowd_ms += ms;
    free(nxt);
}

static void sim_advance(const cno_sim* s, env_t* e, int64_t gid, int64_t ms)
{
    if (ms <= 0) return;
    if (s->cfg.ped_mode == 2) { sim_advance_sf(s, e, gid, ms); return; }
    if (s->cfg.ped_contact) { sim_advance_contact(s, e, gid, ms); return; }
    ped_advance(s, e, gid, e->crowd_ms, e->crowd_ms + ms);
    e->crowd_ms += ms;
    if (s->cfg.wheel_accel > 0.0) robot_advance_wheels(s, e, ms);
    else robot_advance(s, e, ms);
}

/* gazebo/reset_simulation (ENV:1228-1231): poses back to their initial values, twists zeroed.
 * The crowd node is a separate process and keeps its own schedule (crowd_ms is not reset). */
static void sim_reset(const cno_sim* s, env_t* e)
{
    const cno_config* c = &s->cfg;
    e->rx = c->spawn_x; e->ry = c->spawn_y; e->ryaw = c->spawn_yaw;
    e->rv = 0.0; e->rw = 0.0; e->cmd_v = 0.0; e->cmd_w = 0.0;
    memcpy(e->ped_p, e->ped_init, sizeof(double) * 2 * c->n_peds);
    memset(e->ped_v, 0, sizeof(double) * 2 * c->n_peds);
}

static void raycast_impl(const cno_config* c, const double* lc, const double* ls, double rx, double ry,
                         double ryaw, const double* ped, int P, double* ranges)
{
    double sy, cy;
    cno_det_sincos(ryaw, &sy, &cy);
    double ox = fma(c->lidar_offset_x, cy, rx), oy = fma(c->lidar_offset_x, sy, ry);
    const double h = c->room_half, rr = c->ped_radius * c->ped_radius;
    for (int k = 0; k < c->n_rays; ++k) {
        double dx = fma(cy, lc[k], -(sy * ls[k]));
        double dy = fma(sy, lc[k], cy * ls[k]);
        double t = INFINITY;
        if (dx > 0.0) t = fmin(t, (h - ox) / dx);
        else if (dx < 0.0) t = fmin(t, (-h - ox) / dx);
        if (dy > 0.0) t = fmin(t, (h - oy) / dy);
        else if (dy < 0.0) t = fmin(t, (-h - oy) / dy);
        if (t < c->lidar_min) t = c->lidar_min;
        for (int j = 0; j < P; ++j) {
            double ocx = ped[2 * j] - ox, ocy = ped[2 * j + 1] - oy;
            double b = fma(ocx, dx, ocy * dy);
            double cc = fma(ocx, ocx, fma(ocy, ocy, -rr));
            double disc = fma(b, b, -cc);
            if (disc >= 0.0) {
                double sq = sqrt(disc);
                double t2 = b + sq;
                if (t2 >= c->lidar_min) {
                    double t1 = fmax(b - sq, c->lidar_min);
                    t = fmin(t, t1);
                }
            }
        }
        ranges[k] = (t > c->lidar_max) ? INFINITY : t;
        if (c->scan_f32) ranges[k] = (double)(float)ranges[k];    /* sensor_msgs/LaserScan.ranges is float32[] (XACRO:172-175) */
    }
}

void cno_raycast(const cno_config* cfg, double rx, double ry, double ryaw, const double* ped_xy, int P,
                 double* ranges)
{
    int R = cfg->n_rays;
    double* lc = (double*)malloc(sizeof(double) * 2 * R);
    double* ls = lc + R;
    double step = cfg->lidar_span / (double)(R - 1);
    for (int k = 0; k < R; ++k) cno_det_sincos((double)k * step, &ls[k], &lc[k]);
    raycast_impl(cfg, lc, ls, rx, ry, ryaw, ped_xy, P, ranges);
    free(lc);
}

/* ------------------------------------------------------------------------------------------
 * utils.py restatements
 * ---------------------------------------------------------------------------------------- */

/* UTL:375-392 get_scan_ranges */
void cno_scan_sanitize(const double* ranges, int R, double max_range, double* scan)
{
    /* out[j] = clean(ranges[R-1-j]), j = 0..R-2 (reverse, drop last) */
    for (int j = 0; j < R - 1; ++j) {
        double r = ranges[R - 1 - j];
        double o;
        if (isinf(r) && r > 0) o = max_range;
        else if (isnan(r)) o = 0.0;
        else if (r == 0.0) o = max_range;
        else if (r > max_range) o = max_range;
        else o = r;
        scan[j] = o;
    }
}

/* UTL:113 angle_increment = max_angle / (resolution - 1) with Python-2 integer operands.
 * 360/359 == 1.  For R-1 > 360 Python 2 would give 0 (every ray at angle 0); that case is
 * outside the reference's operating range and is defined here as true division. */
static double angle_increment_deg(int R)
{
    if (R - 1 <= 360) return (double)(360 / (R - 1));
    return 360.0 / (double)(R - 1);
}

/* UTL:110-126 convert_laserscan_to_coordinate */
void cno_scan_to_points(const double* scan, int R, double px, double py, double yaw, double* pts)
{
    const double deg2rad = M_PI / 180.0; /* CPython math.radians: x * (pi/180) */
    double inc = angle_increment_deg(R);
    for (int i = 0; i < R - 1; ++i) {
        double ang = (double)i * inc;
        double a = ang * deg2rad - yaw;
        pts[2 * i] = cno_py_round(px + (scan[i] * cos(a)), 3);
        pts[2 * i + 1] = cno_py_round(py + (scan[i] * sin(a)) * -1.0, 3);
    }
}

/* UTL:405-419 */
double cno_bbox_size(const double* pts, int n)
{
    double sum = 0.0; /* Python sum(): left-to-right float adds starting from int 0 */
    for (int i = 0; i < n; ++i) {
        int j = (i == n - 1) ? 0 : i + 1;
        sum += cno_hypot(pts[2 * i] - pts[2 * j], pts[2 * i + 1] - pts[2 * j + 1]);
    }
    return sum / (double)n;
}

/* UTL:395-402 */
int cno_estimate_num_obs_scans(double d, double max_range, double min_range)
{
    return 3 + (int)floor(29 * (max_range - d) / (max_range - min_range));
}

/* UTL:422-460 is_associated / get_iou: IoU of two axis-aligned squares built from their corner
 * coordinates, rounded to 3 decimals. */
double cno_iou(double ax, double ay, double bx, double by, double half)
{
    double axp = ax + half, axm = ax - half, ayp = ay + half, aym = ay - half;
    double bxp = bx + half, bxm = bx - half, byp = by + half, bym = by - half;
    double ix = fmin(axp, bxp) - fmax(axm, bxm);
    double iy = fmin(ayp, byp) - fmax(aym, bym);
    double inter = (ix > 0.0 && iy > 0.0) ? ix * iy : 0.0;
    double area_a = (axp - axm) * (ayp - aym);
    double area_b = (bxp - bxm) * (byp - bym);
    double uni = area_a + area_b - inter;
    return cno_py_round(inter / uni, 3);
}

static void poly_tables(double* pc, double* ps)
{
    for (int k = 0; k < 64; ++k) {
        double a = -(double)k * M_PI / 32.0;
        pc[k] = cos(a);
        ps[k] = sin(a);
    }
}

/* shapely: Point(c).buffer(r).boundary.intersection(LineString([a, b])) -> points.
 * Edge k runs from vertex k to vertex k+1; a hit belongs to edge k iff its edge parameter u is in
 * [0, 1) so a vertex is reported once. */
static int ring_segment(const double* pc, const double* ps, double cx, double cy, double r, double ax,
                        double ay, double bx, double by, v2* out, int maxout)
{
    int cnt = 0;
    double rx = bx - ax, ry = by - ay;
    for (int k = 0; k < 64; ++k) {
        int k2 = (k + 1) & 63;
        double c0x = cx + r * pc[k], c0y = cy + r * ps[k];
        double c1x = cx + r * pc[k2], c1y = cy + r * ps[k2];
        double sx = c1x - c0x, sy = c1y - c0y;
        double den = rx * sy - ry * sx;
        if (den == 0.0) continue;
        double qx = c0x - ax, qy = c0y - ay;
        double t = (qx * sy - qy * sx) / den;
        double u = (qx * ry - qy * rx) / den;
        if (t >= 0.0 && t <= 1.0 && u >= 0.0 && u < 1.0) {
            if (cnt < maxout) { out[cnt].x = ax + t * rx; out[cnt].y = ay + t * ry; }
            ++cnt;
        }
    }
    return cnt;
}

static int waypoint_impl(const double* pc, const double* ps, double ax, double ay, double gx, double gy,
                         double radius, double* wp)
{
    /* UTL:296-314 */
    v2 hit[4];
    int cnt = ring_segment(pc, ps, ax, ay, radius, ax, ay, gx, gy, hit, 4);
    if (cnt == 1) { wp[0] = hit[0].x; wp[1] = hit[0].y; return 1; }
    wp[0] = -(gx + 0.0); wp[1] = gy + 0.0; /* UTL:310-312: x sign flipped */
    return 0;
}

int cno_waypoint(double ax, double ay, double gx, double gy, double radius, double* wp)
{
    double pc[64], ps[64];
    poly_tables(pc, ps);
    return waypoint_impl(pc, ps, ax, ay, gx, gy, radius, wp);
}

/* UTL:251-293 get_collision_point.  returns 1 and *dist when a distance exists, else 0 (None) */
static int collision_point_impl(const double* pc, const double* ps, double a0x, double a0y, double a1x,
                                double a1y, double ox, double oy, double radius, double* dist, int untyped_empty)
{
    double gradient;
    if (a1y == 0.0) gradient = 0.0; /* ZeroDivisionError branch, UTL:262-263 */
    else gradient = (a1x - a0x) / a1y - a0y; /* UTL:261 precedence as written */
    double b = a0x - (gradient * a0y);
    int hi = (int)ceil(a0x + 3.5), lo = (int)floor(a0x - 3.5);
    for (int x2 = hi; x2 > lo; --x2) {
        double y2 = ((double)x2 * gradient) + b;
        v2 hit[4];
        int cnt = ring_segment(pc, ps, ox, oy, radius, a0x, a0y, (double)x2, y2, hit, 4);
        if (cnt == 0) {
            /* GEOS >= 3.9: str(i) == 'LINESTRING EMPTY' -> else branch, keep scanning x2 (UTL:290-291).
             * GEOS <= 3.8 (geos_untyped_empty): 'GEOMETRYCOLLECTION EMPTY' != the literal -> try: i.geoms[0] raises
             * IndexError -> dist_to_cp = None; break (UTL:279-289): the FIRST candidate that misses ends the search */
            if (untyped_empty) return 0;
            continue;
        }
        if (cnt == 1) return 0;     /* Point has no .geoms -> except -> None, break */
        double d1 = cno_hypot(a0x - hit[0].x, a0y - hit[0].y);
        double d2 = cno_hypot(a0x - hit[1].x, a0y - hit[1].y);
        *dist = fmin(d1, d2);
        return 1;
    }
    return 0;
}

int cno_collision_point(double a0x, double a0y, double a1x, double a1y, double ox, double oy, double radius,
                        double* dist)
{
    double pc[64], ps[64];
    poly_tables(pc, ps);
    return collision_point_impl(pc, ps, a0x, a0y, a1x, a1y, ox, oy, radius, dist, 0);
}

int cno_collision_point_geos(double a0x, double a0y, double a1x, double a1y, double ox, double oy, double radius,
                             int untyped_empty, double* dist)
{
    double pc[64], ps[64];
    poly_tables(pc, ps);
    return collision_point_impl(pc, ps, a0x, a0y, a1x, a1y, ox, oy, radius, dist, untyped_empty);
}

/* UTL:317-323 compute_collision_prob(time_to_collision): min(1, 0.15 / ttc) -- negative for a negative ttc */
double cno_collision_prob(double ttc) { return fmin(1.0, 0.15 / ttc); }
/* UTL:326-345 compute_general_collision_prob(scan, max_range, min_range) */
double cno_general_collision_prob(double d, double max_range, double min_range)
{
    return (d > max_range) ? 0.0 : (max_range - d) / (max_range - min_range);
}
/* ENV:882-883: sorted(entries, key=cp, reverse=True)[-K:] -- stable, descending, keep the LAST K (the K lowest CPs).
 * idx_out[0..kept) = indices in output order; returns kept */
int cno_topk(const double* cp, int n, int K, int32_t* idx_out)
{
    int idx[CNO_MAX_TRACKS];
    if (n > CNO_MAX_TRACKS) n = CNO_MAX_TRACKS;
    for (int i = 0; i < n; ++i) idx[i] = i;
    for (int i = 1; i < n; ++i) { /* stable insertion sort, descending */
        int q = idx[i]; int j = i - 1;
        while (j >= 0 && cp[idx[j]] < cp[q]) { idx[j + 1] = idx[j]; --j; }
        idx[j + 1] = q;
    }
    int first = n > K ? n - K : 0, kk = 0;
    for (int i = first; i < n; ++i) idx_out[kk++] = idx[i];
    return kk;
}

/* ------------------------------------------------------------------------------------------
 * Env.get_state (ENV:245-1044)
 * ---------------------------------------------------------------------------------------- */

static double heading_to_goal(const cno_config* c, const env_t* e, double px, double py, double yaw)
{
    /* ENV:222-237: adds starting_point to the position (ENV:191-209 does not) */
    double cx = px + c->start_x, cy = py + c->start_y;
    double ga = atan2(e->wpy - cy, e->wpx - cx);
    double h = ga - yaw;
    if (h > M_PI) h -= 2 * M_PI;
    else if (h < -M_PI) h += 2 * M_PI;
    return h;
}

static double dist3(double ax, double ay, double bx, double by)
{
    /* np.linalg.norm(a - b) of a 3-vector (ENV:191-197) = sqrt(dot(d, d)), and numpy hands the dot to BLAS ddot, whose
     * kernel accumulates with fused multiply-adds: acc = dx*dx; acc = fma(dy, dy, acc); acc = fma(dz, dz, acc) with dz = 0.
     * Pinned by the function goldens hd_* / dist_out (256 of 256 values; the unfused sum matches 237). */
    double dx = ax - bx, dy = ay - by;
    return sqrt(fma(dy, dy, dx * dx));
}

static void waypoint_refresh(const cno_sim* s, env_t* e, double px, double py)
{
    double wp[2];
    waypoint_impl(s->poly_c, s->poly_s, px, py, s->cfg.goal_x, s->cfg.goal_y, s->cfg.waypoint_radius, wp);
    e->wpx = wp[0];
    e->wpy = wp[1];
}

static int in_box(double x, double y, double gx, double gy, double eps)
{
    /* ENV:1285-1319: half-open box, x <= g+eps and x > g-eps */
    double xp = gx + eps, xm = gx - eps, yp = gy + eps, ym = gy - eps;
    return (x <= xp) && (x > xm) && (y <= yp) && (y > ym);
}

typedef struct { int type; v2 pose; double dist; } cobj_t;

static void track_new(env_t* e, const cobj_t* o, double now)
{
    if (e->ntracks >= CNO_MAX_TRACKS) { e->status |= ST_TRACK_OVERFLOW; return; }
    track_t* t = &e->tracks[e->ntracks++];
    t->pose = o->pose; t->dist = o->dist;
    t->dq[0] = o->pose; t->dq_len = 1;
    t->t = now; t->speed = -1.0;
    t->vel.x = 0.0; t->vel.y = 0.0;
}

/* ENV:656-743 tracker (the same code block as RW:490-571): popleft, IoU arg-max per track with the first maximum winning,
 * "delete only while len(tracks) > i" with the index drift of `keys.pop(i)`, unmatched 'o' objects become tracks. */
static void tracker_update(env_t* e, const cobj_t* conf, int nconf, double now, int* tmp)
{
    if (e->ntracks == 0) {
        for (int j = 0; j < nconf; ++j) if (conf[j].type == TY_O) track_new(e, &conf[j], now);
    } else {
        int nt0 = e->ntracks;
        for (int i = 0; i < nt0; ++i) if (e->tracks[i].dq_len > 1) { /* ENV:678-680 popleft */
            e->tracks[i].dq[0] = e->tracks[i].dq[1]; e->tracks[i].dq_len = 1;
        }
        if (nconf == 0) {
            e->ntracks = 0; /* ENV:683-686 nets out to clearing every track */
        } else {
            int alive[CNO_MAX_TRACKS];
            int* checked = tmp;
            for (int j = 0; j < nconf; ++j) checked[j] = 0;
            int cur = nt0; /* len(self.tracked_obstacles) */
            for (int i = 0; i < nt0; ++i) {
                alive[i] = 1;
                track_t* t = &e->tracks[i];
                int bj = 0; double best = -1.0;
                for (int j = 0; j < nconf; ++j) { /* ENV:688-689, walls included */
                    double u = cno_iou(t->pose.x, t->pose.y, conf[j].pose.x, conf[j].pose.y, 0.0505);
                    if (u > best) { best = u; bj = j; } /* list.index(max): first maximum */
                }
                if (best > 0.0) { /* ENV:702-712 */
                    t->pose = conf[bj].pose; t->dist = conf[bj].dist;
                    t->dq[t->dq_len++] = conf[bj].pose;
                    t->t = now - t->t;
                    checked[bj] = 1;
                } else if (cur > i) { /* ENV:715-717 */
                    alive[i] = 0; cur -= 1;
                }
            }
            int m = 0;
            for (int i = 0; i < nt0; ++i) if (alive[i]) { if (m != i) e->tracks[m] = e->tracks[i]; ++m; }
            e->ntracks = m;
            for (int j = 0; j < nconf; ++j) /* ENV:723-743 */
                if (!checked[j] && conf[j].type == TY_O) track_new(e, &conf[j], now);
        }
    }
}

static void env_get_state(const cno_sim* s, env_t* e, const double* ranges, double px, double py, double yaw,
                          double v, double w, int step_counter, double now, double* state, int* done_out,
                          int32_t* topk_idx)
{
    const cno_config* c = &s->cfg;
    const int R = c->n_rays, n = R - 1, K = c->k_obstacles;
    const double MAXR = c->max_scan_range;

    /* ENV:246-253 */
    if (step_counter == 1) waypoint_refresh(s, e, px, py);
    /* ENV:255-257: distance is np.float64 -> numpy rounding; heading is a Python float */
    double distance_to_goal = round_np64(dist3(px, py, e->wpx, e->wpy), 2);     /* round(np.float64, 2), ENV:255 */
    double heading = cno_py_round(heading_to_goal(c, e, px, py, yaw), 2);
    /* ENV:259-265 */
    if (step_counter % 5 == 0 || distance_to_goal < e->prev_dist) waypoint_refresh(s, e, px, py);
    /* ENV:267-268: the angular velocity is used as the angle */
    double agent_vel_x = -1.0 * (v * cos(w));
    double agent_vel_y = v * sin(w);

    double* scan = (double*)malloc(sizeof(double) * (size_t)n * 12 + sizeof(int) * (size_t)n * 8);
    double* pts = scan + n;          /* 2n */
    double* d = pts + 2 * n;         /* n  */
    double* g = d + n;               /* n  */
    double* cg = g + n;              /* n  */
    double* ds = cg + n;             /* n  */
    double* psx = ds + n;            /* n  */
    double* psy = psx + n;           /* n  */
    double* gtp = psy + n;           /* 2n */
    int* gnone = (int*)(gtp + 2 * n + n); /* keep alignment slack */
    int* cnone = gnone + n;
    int* Ttype = cnone + n;
    int* Tsrc = Ttype + n;
    int* ty = Tsrc + n;
    int* order = ty + n;
    int* segend = order + n;         /* segend[k] = 1 if a segment closes after order[k] */
    int* tmp = segend + n;

    /* ENV:277-283 */
    cno_scan_sanitize(ranges, R, MAXR, scan);
    cno_scan_to_points(scan, R, px, py, yaw, pts);

    /* ENV:287-294 */
    if (step_counter == 0) {
        for (int i = 0; i < n; ++i) d[i] = MAXR; /* ground_truth_scans, ENV:116 */
        cno_scan_to_points(d, R, px, py, yaw, gtp);
        e->bb = cno_bbox_size(gtp, n);
        v2 p = { cno_py_round(px, 3), cno_py_round(py, 3) };
        if (e->agent_dq_len < 2) e->agent_dq[e->agent_dq_len++] = p;
        else { e->agent_dq[0] = e->agent_dq[1]; e->agent_dq[1] = p; } /* unreachable: deque is unbounded but never exceeds 2 */
    }
    /* ENV:297-305 is dead code (result unused) -> omitted */

    cobj_t* conf = NULL;
    int nconf = 0, gt_ego_hit = 0;
    const int gt = (c->risk_mode == 1);
    if (!gt) {
    /* ENV:317-327 */
    for (int i = 0; i < n; ++i) d[i] = cno_py_round(scan[i], 3);

    /* ENV:329-346 gradients between consecutive end points */
    for (int i = 0; i < n; ++i) {
        if (d[i] == 0.6) { gnone[i] = 1; g[i] = 0.0; continue; }
        int j = (i == n - 1) ? 0 : i + 1;
        double dy = pts[2 * i + 1] - pts[2 * j + 1];
        double gr;
        if (dy == 0) gr = 0.0;
        else gr = (pts[2 * i] - pts[2 * j]) / dy;
        gnone[i] = 0;
        g[i] = cno_py_round(gr, 3);
    }
    /* ENV:348-367 change of gradient */
    {
        int last_none = 1; double last = 0.0;
        for (int i = 0; i < n; ++i) {
            if (gnone[i]) { cnone[i] = 1; cg[i] = 0.0; continue; }
            if (n == 1 || i == n - 1) { cnone[i] = last_none; cg[i] = last; }
            else if (!gnone[i + 1]) {
                double ch = fabs(g[i] - g[i + 1]);
                last = ch; last_none = 0; cnone[i] = 0; cg[i] = ch;
            } else { last_none = 1; last = 0.0; cnone[i] = 1; cg[i] = 0.0; }
        }
    }
    /* ENV:372-410 object-type state machine.  T[i] is None or a reference to the list created
     * at some ray (its range and pose travel with it): (Ttype, Tsrc). */
    {
        int last_type = TY_NONE, last_src = -1, du = 0;
        for (int i = 0; i < n; ++i) { Ttype[i] = TY_NONE; Tsrc[i] = -1; }
        for (int i = 0; i < n; ++i) {
            if (cnone[i]) continue;
            if (i == n - 1) continue;
            if (cg[i] == 0) {
                Ttype[i] = TY_W; Tsrc[i] = i; last_type = TY_W; last_src = i;
            } else {
                Ttype[i] = TY_O; Tsrc[i] = i;
                if (du != 1) {
                    if (!cnone[i + 1] && cg[i + 1] == 0) {
                        Ttype[i] = TY_W; Tsrc[i] = i; last_type = TY_W; last_src = i; du = 0;
                    }
                    if (cnone[i + 1]) {
                        /* ENV:394-395 pass */
                    } else if (fabs(cg[i] - cg[i + 1]) == 0) {
                        Ttype[i] = TY_W; Tsrc[i] = i; last_type = TY_W; last_src = i; du = 0;
                    } else {
                        Ttype[i] = last_type; Tsrc[i] = last_src; du += 1;
                    }
                } else {
                    Ttype[i] = TY_O; Tsrc[i] = i; last_type = TY_O; last_src = i;
                    if (!cnone[i + 1] && cg[i + 1] == 0) du = 0;
                }
            }
        }
    }
    /* ENV:433-445 per-ray (type, distance, pose) */
    for (int i = 0; i < n; ++i) {
        if (Ttype[i] == TY_NONE) { ty[i] = TY_NONE; ds[i] = d[i]; psx[i] = pts[2 * i]; psy[i] = pts[2 * i + 1]; }
        else { int q = Tsrc[i]; ty[i] = Ttype[i]; ds[i] = d[q]; psx[i] = pts[2 * q]; psy[i] = pts[2 * q + 1]; }
    }
    /* ENV:448-485 segmentation by bounding-box association of consecutive rays */
    int nseg = 0;
    {
        int* brk = tmp; /* brk[i] = 1: a segment closes after ray i */
        for (int i = 0; i < n; ++i) {
            if (i == n - 1) brk[i] = 1; /* both branches of ENV:454-470 close the segment */
            else brk[i] = !(cno_iou(psx[i], psy[i], psx[i + 1], psy[i + 1], e->bb) > 0.0);
        }
        int first_end = 0;
        while (!brk[first_end]) ++first_end;
        int nsegs0 = 0;
        for (int i = 0; i < n; ++i) nsegs0 += brk[i];
        int last_start = n - 1;
        while (last_start > 0 && !brk[last_start - 1]) --last_start;
        int merge = 0;
        /* ENV:490-502 first <-> last with twice the box */
        if (nsegs0 > 1 &&
            cno_iou(psx[0], psy[0], psx[n - 1], psy[n - 1], e->bb * 2) > 0.0) merge = 1;
        int pos = 0;
        if (merge) {
            for (int i = 0; i <= first_end; ++i) { order[pos] = i; segend[pos] = 0; ++pos; }
            for (int i = last_start; i < n; ++i) { order[pos] = i; segend[pos] = 0; ++pos; }
            segend[pos - 1] = 1;
            for (int i = first_end + 1; i < last_start; ++i) { order[pos] = i; segend[pos] = brk[i]; ++pos; }
        } else {
            for (int i = 0; i < n; ++i) { order[i] = i; segend[i] = brk[i]; }
            pos = n;
        }
        /* ENV:508-566 split each segment where free space (0.6) meets occupied */
        int k0 = 0;
        while (k0 < n) {
            int k1 = k0;
            while (!segend[k1]) ++k1;
            int any_occ = 0;
            for (int k = k0; k <= k1; ++k) if (ds[order[k]] != 0.6) any_occ = 1;
            if (any_occ) {
                for (int k = k0; k < k1; ++k) {
                    int a06 = ds[order[k]] == 0.6, b06 = ds[order[k + 1]] == 0.6;
                    if (a06 != b06) segend[k] = 1;
                }
            }
            k0 = k1 + 1;
        }
        for (int k = 0; k < n; ++k) nseg += segend[k];
    }
    /* ENV:568-620 confirmation */
    int maxc = n / 4 + 2;
    conf = (cobj_t*)malloc(sizeof(cobj_t) * (size_t)maxc);
    nconf = 0;
    {
        int k0 = 0;
        while (k0 < n) {
            int k1 = k0;
            while (!segend[k1]) ++k1;
            int len = k1 - k0 + 1;
            int any_occ = 0;
            for (int k = k0; k <= k1; ++k) if (ds[order[k]] != 0.6) any_occ = 1;
            if (any_occ && len >= 4) {
                int m = order[k0 + len / 2]; /* ENV:577 Python-2 integer division */
                int est = cno_estimate_num_obs_scans(ds[m], c->max_scan_range, c->min_scan_range);
                int no = 0, nw = 0, nn = 0;
                for (int k = k0; k <= k1; ++k) {
                    int t = ty[order[k]];
                    no += (t == TY_O); nw += (t == TY_W); nn += (t == TY_NONE);
                }
                int mn = len < est ? len : est;
                double score = (double)no / (double)mn;
                int kinds = (no > 0) + (nw > 0) + (nn > 0);
                int obj = -1;
                if (kinds > 1) {
                    if (score >= 0.5) obj = (no > nw) ? TY_O : TY_W;
                    else if (len <= est) obj = (no > nw) ? TY_O : TY_W;
                    else obj = TY_W;
                } else {
                    int lim = nseg < est ? nseg : est; /* ENV:608,615: compares with the number of segments */
                    if (len <= lim) obj = -1;
                    else obj = (nw > 0) ? TY_W : TY_O;
                }
                if (obj >= 0 && nconf < maxc) {
                    conf[nconf].type = obj;
                    conf[nconf].pose.x = psx[m]; conf[nconf].pose.y = psy[m];
                    conf[nconf].dist = ds[m];
                    ++nconf;
                }
            }
            k0 = k1 + 1;
        }
    }
    e->n_confirmed = nconf;
    /* ENV:637-654 */
    int n_obst = 0;
    for (int j = 0; j < nconf; ++j) n_obst += (conf[j].type == TY_O);
    if (n_obst > 0) e->obst_steps += 1;

    tracker_update(e, conf, nconf, now, tmp);
    } else {
        /* risk_mode gt ("two risk-feature modes"; include/crowdnav.h): rows A21-A24 fed with the simulator's own
         * pedestrians instead of tracked lidar blobs.  An entry = a pedestrian that is within lidar range of the lidar origin
         * and in line of sight (no other disc cuts the segment origin -> its nearest surface point), in id order:
         *   pose  = the surface point nearest to the ROBOT (A18's "surface point" convention: end points are laid out from
         *           the robot position), rounded to 3 decimals like a scan end point;  dist = |c - p| - r, rounded
         *   vel   = -(true velocity): ENV:806-811 subtract new from old, so the reference's feature is the negated velocity
         *   speed = |true velocity| (ENV:787-793 uses the FIRST entry's for every obstacle)
         * then the reference's collision cone, CP and top-K arithmetic unchanged; topk_idx = pedestrian ids. */
        const int P = c->n_peds;
        const double r = c->ped_radius;
        double sy_, cy_;
        cno_det_sincos(yaw, &sy_, &cy_);
        const double ox = fma(c->lidar_offset_x, cy_, px), oy = fma(c->lidar_offset_x, sy_, py);
        const double lim = c->lidar_max + r;
        e->ntracks = 0;
        for (int i = 0; i < P; ++i) {
            const double cx = e->ped_p[2 * i], cyy = e->ped_p[2 * i + 1];
            const double ocx = cx - ox, ocy = cyy - oy;
            const double dd2 = fma(ocx, ocx, ocy * ocy);
            if (!(dd2 <= lim * lim)) continue;                       /* out of lidar range */
            const double dd = sqrt(dd2);
            int blocked = 0;
            if (dd > r) {                                            /* (origin inside the disc: nothing can be in front of it) */
                const double k_ = (dd - r) / dd;                     /* origin -> nearest surface point = k_ * oc */
                const double wx = k_ * ocx, wy = k_ * ocy;
                const double len2 = fma(wx, wx, wy * wy);
                for (int j = 0; j < P && !blocked; ++j) {
                    if (j == i) continue;
                    const double qx = e->ped_p[2 * j] - ox, qy = e->ped_p[2 * j + 1] - oy;
                    if (!(fma(qx, qx, qy * qy) <= lim * lim)) continue;
                    double tau = (len2 > 0.0) ? fma(qx, wx, qy * wy) / len2 : 0.0;
                    tau = fmin(fmax(tau, 0.0), 1.0);
                    const double ex = fma(tau, wx, -qx), ey = fma(tau, wy, -qy);
                    if (fma(ex, ex, ey * ey) < r * r) blocked = 1;
                }
            }
            if (blocked) continue;
            if (e->ntracks >= CNO_MAX_TRACKS) { e->status |= ST_TRACK_OVERFLOW; break; }
            const double dx = cx - px, dy = cyy - py;
            const double dp = sqrt(fma(dx, dx, dy * dy));
            track_t* t = &e->tracks[e->ntracks++];
            if (dp > 0.0) { t->pose.x = cno_py_round(cx - r * (dx / dp), 3); t->pose.y = cno_py_round(cyy - r * (dy / dp), 3); }
            else { t->pose.x = cno_py_round(cx, 3); t->pose.y = cno_py_round(cyy, 3); }
            t->dist = cno_py_round(dp - r, 3);
            const double vx = e->ped_v[2 * i], vy = e->ped_v[2 * i + 1];
            t->vel.x = -vx; t->vel.y = -vy;
            t->speed = sqrt(fma(vx, vx, vy * vy));
            t->dq_len = 0; t->t = now; t->id = i;
            if (t->dist < 0.140) gt_ego_hit = 1;
        }
        e->n_confirmed = e->ntracks;
        if (e->ntracks > 0) e->obst_steps += 1;
    }
    /* ENV:745-760 speed of matched tracks */
    for (int i = 0; i < e->ntracks; ++i) {
        track_t* t = &e->tracks[i];
        if (t->dq_len > 1) {
            double dc = cno_hypot(t->dq[0].y - t->dq[1].y, t->dq[0].x - t->dq[1].x);
            t->speed = dc / t->t;
        }
    }

    /* default K x [px, py, 0, 0] (ENV:273) */
    double* feat = state + n + 7;
    for (int k = 0; k < K; ++k) { feat[4 * k] = px; feat[4 * k + 1] = py; feat[4 * k + 2] = 0.0; feat[4 * k + 3] = 0.0; }
    for (int k = 0; k < K; ++k) topk_idx[k] = -1;
    e->n_entries = 0;

    /* ENV:769-996 collision cone / collision probability */
    if (e->agent_dq_len == 2) {
        double ts = e->agent_vel_timestep;
        if (ts == 0.0) e->status |= ST_DT_ZERO;
        /* UTL:227-236 */
        double vx_ = (e->agent_dq[1].x - e->agent_dq[0].x) / ts;
        double vy_ = (e->agent_dq[1].y - e->agent_dq[0].y) / ts;
        double agent_vel = sqrt(pow(vx_, 2) + pow(vy_, 2));
        double obstacle_vel = (e->ntracks == 0) ? 0.0 : e->tracks[0].speed; /* ENV:787-793 */
        double vo_x = e->agent_dq[1].x, vo_y = e->agent_dq[1].y;
        for (int i = 0; i < e->ntracks; ++i) { /* ENV:800-815: the last track's value survives */
            track_t* t = &e->tracks[i];
            double chx = 0, chy = 0;
            if (gt) { chx = t->vel.x * ts; chy = t->vel.y * ts; }       /* displacement over the agent's timestep, old - new */
            else if (t->dq_len > 1) {
                chx = t->dq[0].x - t->dq[1].x; chy = t->dq[0].y - t->dq[1].y;
                t->vel.x = chx / ts; t->vel.y = chy / ts;
            }
            vo_x = e->agent_dq[1].x + chx; vo_y = e->agent_dq[1].y + chy;
        }
        double cp[CNO_MAX_TRACKS];
        double ego_prev = 0.0, ego_max = 0.0;
        int ne = 0;
        for (int i = 0; i < e->ntracks; ++i) { /* ENV:818-860 */
            track_t* t = &e->tracks[i];
            double dcp;
            int has = collision_point_impl(s->poly_c, s->poly_s, e->agent_dq[0].x, e->agent_dq[0].y, vo_x, vo_y,
                                           t->pose.x, t->pose.y, 0.178, &dcp, c->geos_untyped_empty);
            double rv = agent_vel - obstacle_vel;
            double gcp = cno_general_collision_prob(t->dist, c->max_scan_range, c->min_scan_range);
            double ego, cpv;
            if (has) {
                if (rv == 0) { cpv = 1.0 * gcp; ego = ego_prev; }
                else {
                    double ttc = dcp / rv;
                    if (ttc == 0.0) { e->status |= ST_TTC_ZERO; ego = 1.0; } /* Python raises; measure-zero */
                    else ego = cno_collision_prob(ttc); /* UTL:319 min(1, 0.15/ttc); may be negative */
                    cpv = 0.5 * ego + 0.5 * gcp;
                }
            } else { ego = 0.0; cpv = 0.5 * 0.0 + 0.5 * gcp; }
            ego_prev = ego;
            e->entry_cp[ne] = cpv; e->entry_ego[ne] = ego;
            cp[ne] = cpv;
            if (ne == 0 || ego > ego_max) ego_max = ego;
            ++ne;
        }
        e->n_entries = ne;
        if (ne == 0) { /* ENV:862-876 */
            e->collision_prob = 0.0; e->ego_score_cp = 0.0;
        } else { /* ENV:878-905 */
            e->ego_score_cp = ego_max;
            int32_t keep[CNO_MAX_TRACKS];
            int kept = cno_topk(cp, ne, K, keep);   /* [-K:] keeps the K lowest when ne > K */
            e->collision_prob = cp[keep[0]];
            for (int kk = 0; kk < kept; ++kk) {
                track_t* t = &e->tracks[keep[kk]];
                feat[4 * kk] = t->pose.x; feat[4 * kk + 1] = t->pose.y;
                feat[4 * kk + 2] = t->vel.x; feat[4 * kk + 3] = t->vel.y;
                topk_idx[kk] = gt ? t->id : keep[kk];
            }
        }
        /* ENV:990-996 */
        e->agent_dq[0] = e->agent_dq[1]; e->agent_dq_len = 1;
        for (int i = 0; i < e->ntracks; ++i) e->tracks[i].t = now;
    }
    /* ENV:998-1005 safety counters */
    for (int j = 0; j < nconf; ++j)
        if (conf[j].type == TY_O && conf[j].dist < 0.140) { e->ego_viol += 1; break; }
    if (gt && gt_ego_hit) e->ego_viol += 1;
    if (e->ego_score_cp > 0.4) e->social_viol += 1;

    /* ENV:1011-1023 done */
    if (!e->done) {
        double mn = scan[0];
        for (int i = 1; i < n; ++i) if (scan[i] < mn) mn = scan[i];
        if (mn < c->min_scan_range) e->done = 1;
        if (in_box(px, py, c->goal_x, c->goal_y, c->goal_eps)) e->done = 1;
        if (step_counter >= c->max_steps) e->done = 1;
    }
    /* ENV:1025-1042 observation */
    for (int i = 0; i < n; ++i) state[i] = scan[i];
    state[n] = heading;
    state[n + 1] = distance_to_goal;
    state[n + 2] = cno_py_round(px, 3);
    state[n + 3] = cno_py_round(py, 3);
    state[n + 4] = cno_py_round(yaw, 3);
    state[n + 5] = cno_py_round(agent_vel_x, 3);
    state[n + 6] = cno_py_round(agent_vel_y, 3);
    for (int i = 0; i < s->D; ++i) state[i] = cno_np_around(state[i], 3);
    *done_out = e->done;
    free(conf);
    free(scan);
}

/* ENV:1046-1162 */
static double env_compute_reward(const cno_sim* s, env_t* e, const double* state, double px, double py, int done)
{
    const cno_config* c = &s->cfg;
    const int n = s->n;
    double cur_head = state[n], cur_dist = state[n + 1];
    double dd = cur_dist - e->prev_dist, hd = cur_head - e->prev_head;
    int step_reward = -2, htg = 0, dtg = 0, wp = 0;
    if (dd > 0) dtg = 0;
    if (dd < 0) dtg = 1;
    double ph = e->prev_head;
    if (hd > 0) {
        if (cur_head > 0 && ph < 0) htg = 1;
        if (cur_head < 0 && ph < 0) htg = 1;
        if (cur_head < 0 && ph > 0) htg = 1;
        if (cur_head > 0 && ph > 0) htg = 0;
    }
    if (hd < 0) {
        if (cur_head < 0 && ph > 0) htg = 1;
        if (cur_head > 0 && ph > 0) htg = 1;
        if (cur_head > 0 && ph < 0) htg = 1;
        if (cur_head < 0 && ph < 0) htg = 0;
    }
    if (in_box(px, py, e->wpx, e->wpy, c->goal_eps)) { /* ENV:1109-1125 */
        waypoint_refresh(s, e, px, py);
        wp = c->waypoint_reward;                                   /* ENV:1116: 200 */
        if (in_box(e->wpx, e->wpy, c->goal_x, c->goal_y, c->goal_eps)) { e->wpx = c->goal_x; e->wpy = c->goal_y; }
    }
    double reward = (double)(step_reward + dtg + htg + wp);
    e->prev_dist = cur_dist;
    e->prev_head = cur_head;
    if (done) {
        if (in_box(px, py, c->goal_x, c->goal_y, c->goal_eps)) { e->ep_failure = 0; e->ep_success = 1; reward = 200 + reward; }
        else { e->ep_failure = 1; e->ep_success = 0; reward = -200 + reward; }
    }
    return reward;
}

/* ------------------------------------------------------------------------------------------
 * obs_layout 1: environment_stage_1_original.py ("ORIG"), the 363-input environment of the SAC / DQN /
 * Q-learning / SARSA trainers (start_sac_training.py:13,105-116): no waypoints, no tracker.
 * ---------------------------------------------------------------------------------------- */

static double orig_heading(const cno_config* c, double px, double py, double yaw)
{
    /* ORIG:244-260: no starting_pose offset, straight to desired_point */
    double ga = atan2(c->goal_y - py, c->goal_x - px);
    double h = ga - yaw;
    if (h > M_PI) h -= 2 * M_PI;
    else if (h < -M_PI) h += 2 * M_PI;
    return h;
}

/* ORIG:278-322.  state = [round(range, 3)] * (R-1) + [heading, distance] + [round(x, 3), round(y, 3)] */
static void orig_get_state(const cno_sim* s, env_t* e, const double* ranges, double px, double py, double yaw,
                           int step_counter, double* state, int* done_out)
{
    const cno_config* c = &s->cfg;
    const int R = c->n_rays, n = R - 1;
    double dist = round_np64(dist3(px, py, c->goal_x, c->goal_y), 2);    /* round(np.float64, 2), ORIG:280 */
    double head = cno_py_round(orig_heading(c, px, py, yaw), 2);          /* ORIG:281 */
    const double min_range = 0.105;                                       /* ORIG:282 */
    double mn = INFINITY;
    for (int j = 0; j < n; ++j) {
        double r = ranges[R - 1 - j], v;                                  /* ORIG:289-300: reverse, drop the last */
        if (isinf(r)) v = 0.6;                                            /* ORIG:290-291 (the literal, not max_scan_range) */
        else if (isnan(r)) v = 0.0;
        else v = r;
        if (v < mn) mn = v;                                               /* Python min(): first smallest, NaN-free here */
        state[j] = cno_py_round(v, 3);                                    /* ORIG:317 */
    }
    if (!e->done) {
        if (min_range > mn && mn > 0) e->done = 1;                        /* ORIG:303-305 */
        if (in_box(px, py, c->goal_x, c->goal_y, 0.20)) e->done = 1;     /* ORIG:307-309, epsilon default 0.20 (ORIG:500) */
        if (step_counter >= c->max_steps) e->done = 1;                    /* ORIG:311-313 */
    }
    state[n] = head; state[n + 1] = dist;
    state[n + 2] = cno_py_round(px, 3); state[n + 3] = cno_py_round(py, 3); /* ORIG:315 */
    *done_out = e->done;
}

/* ORIG:324-402.  The layout quirk is the reference's: state[-1] is the rounded y and state[-2] the rounded x, and those
 * are what it calls current_distance / current_heading. */
static double orig_compute_reward(const cno_sim* s, env_t* e, const double* state, double px, double py, int done)
{
    const cno_config* c = &s->cfg;
    const int n = s->n;
    double cur_dist = state[n + 3], cur_head = state[n + 2];
    double dd = cur_dist - e->prev_dist, hd = cur_head - e->prev_head;
    int htg = 0, dtg = 0;
    if (dd < 0) dtg = 1;
    double ph = e->prev_head;
    if (hd > 0) {
        if (cur_head > 0 && ph < 0) htg = 1;
        if (cur_head < 0 && ph < 0) htg = 1;
        if (cur_head < 0 && ph > 0) htg = 1;
        if (cur_head > 0 && ph > 0) htg = 0;
    }
    if (hd < 0) {
        if (cur_head < 0 && ph > 0) htg = 1;
        if (cur_head > 0 && ph > 0) htg = 1;
        if (cur_head > 0 && ph < 0) htg = 1;
        if (cur_head < 0 && ph < 0) htg = 0;
    }
    double reward = (double)(dtg + htg);                                  /* step_reward = 0, action_reward unused (ORIG:333-374) */
    e->prev_dist = cur_dist;
    e->prev_head = cur_head;
    if (done) {
        if (in_box(px, py, c->goal_x, c->goal_y, 0.20)) { e->ep_failure = 0; e->ep_success = 1; reward = 200 + reward; }
        else { e->ep_failure = 1; e->ep_success = 0; reward = -200 + reward; }
    }
    return reward;
}

/* ------------------------------------------------------------------------------------------
 * Env.reset / Env.step flows
 * ---------------------------------------------------------------------------------------- */

/* ------------------------------------------------------------------------------------------
 * obs_layout 2: environment_stage_1_nobonus_realworld.py ("RW"), the 370-input physical-robot variant:
 * 359 UNROUNDED sanitised ranges + heading + distance + rounded (x, y) + the constant yaw 3.14 + rounded twist features +
 * pose and velocity of the ONE tracked obstacle with the highest collision probability.  Its segmentation is an older
 * pipeline than ENV's: free-space rays are filtered out BEFORE the gradients, no way-points, the collision cone is cast
 * against a ring of radius min_scan_range, and Env.step holds the command for 0.05 s but books 0.15 s (RW:880-883).
 * ---------------------------------------------------------------------------------------- */
static double rw_distance(const cno_config* c, double px, double py)
{   /* RW:161-173: starting_point is added to the position here too */
    return dist3(px + c->start_x, py + c->start_y, c->goal_x, c->goal_y);
}
static double rw_heading(const cno_config* c, double px, double py, double yaw)
{   /* RW:186-200 */
    double cx = px + c->start_x, cy = py + c->start_y;
    double h = atan2(c->goal_y - cy, c->goal_x - cx) - yaw;
    if (h > M_PI) h -= 2 * M_PI;
    else if (h < -M_PI) h += 2 * M_PI;
    return h;
}

static void rw_get_state(const cno_sim* s, env_t* e, const double* ranges, double px, double py, double yaw, double v,
                         double w, int step_counter, double now, double* state, int* done_out)
{
    const cno_config* c = &s->cfg;
    const int R = c->n_rays, n = R - 1;
    const double MAXR = c->max_scan_range;
    double distance_to_goal = round_np64(rw_distance(c, px, py), 2);      /* RW:209 round(np.float64, 2) */
    double heading = cno_py_round(rw_heading(c, px, py, yaw), 2);         /* RW:210 */
    double agent_vel_x = -1.0 * (v * cos(w)), agent_vel_y = v * sin(w);   /* RW:211-212 */
    v2 closest_pose = { px, py }, closest_vel = { 0.0, 0.0 };             /* RW:215-216 */

    double* scan = (double*)malloc(sizeof(double) * (size_t)n * 10 + sizeof(int) * (size_t)n * 8);
    double* pts = scan + n;            /* 2n */
    double* gtp = pts + 2 * n;         /* 2n */
    double* fr = gtp + 2 * n;          /* n: filtered ranges */
    double* g = fr + n;                /* n */
    double* cg = g + n;                /* n */
    double* ed = cg + n;               /* n: estimated (flattened) distances */
    int* fi = (int*)(ed + n + 1);      /* filtered -> ray index */
    int* cnone = fi + n;
    int* Ttype = cnone + n;            /* 0 = None */
    int* Tsrc = Ttype + n;             /* filtered index whose range / pose the entry carries */
    int* et = Tsrc + n;                /* estimated types */
    int* es = et + n;                  /* estimated source (filtered index) */
    int* segend = es + n;
    int* tmp = segend + n;

    cno_scan_sanitize(ranges, R, MAXR, scan);                              /* RW:220 */
    cno_scan_to_points(scan, R, px, py, yaw, pts);                         /* RW:225 */
    if (step_counter == 0) {                                               /* RW:229-237 */
        for (int i = 0; i < n; ++i) g[i] = MAXR;
        cno_scan_to_points(g, R, px, py, yaw, gtp);
        e->bb = cno_bbox_size(gtp, n);
        v2 p = { cno_py_round(px, 3), cno_py_round(py, 3) };
        if (e->agent_dq_len < 2) e->agent_dq[e->agent_dq_len++] = p;
        else { e->agent_dq[0] = e->agent_dq[1]; e->agent_dq[1] = p; }
    }
    /* RW:239-247: obstacle regions / per-ray deques -- results never read.  RW:249-266: keep the rays that are not free space */
    int F = 0;
    for (int i = 0; i < n; ++i)
        if (!(1.0 * MAXR <= scan[i] && scan[i] <= 1.0 * MAXR)) { fi[F] = i; fr[F] = scan[i]; ++F; }
#define FPX(k) pts[2 * fi[k]]
#define FPY(k) pts[2 * fi[k] + 1]
    for (int i = 0; i < F; ++i) {                                          /* RW:268-282 gradients over the FILTERED list */
        int j = (i == F - 1) ? 0 : i + 1;
        double dy = FPY(i) - FPY(j);
        double gr = (dy == 0) ? 0.0 : (FPX(i) - FPX(j)) / dy;
        g[i] = cno_py_round(gr, 3);
    }
    {                                                                      /* RW:284-295 change of gradient */
        int last_none = 1; double last = 0.0;
        for (int i = 0; i < F; ++i) {
            if (F == 1 || i == F - 1) { cnone[i] = last_none; cg[i] = last; }
            else { double ch = fabs(g[i] - g[i + 1]); last = ch; last_none = 0; cnone[i] = 0; cg[i] = ch; }
        }
    }
    {                                                                      /* RW:300-333 object-type machine */
        int last_type = 0, last_src = -1, du = 0;
        for (int i = 0; i < F; ++i) {
            Ttype[i] = 0; Tsrc[i] = i;
            if (i == F - 1) continue;
            if (!cnone[i] && cg[i] == 0) { Ttype[i] = TY_W; last_type = TY_W; last_src = i; continue; }
            int nz = !cnone[i + 1] && cg[i + 1] == 0;
            if (du != 1) {
                if (nz) { Ttype[i] = TY_W; last_type = TY_W; last_src = i; du = 0; }
                else if (!cnone[i] && !cnone[i + 1] && fabs(cg[i] - cg[i + 1]) == 0) { Ttype[i] = TY_W; last_type = TY_W; last_src = i; du = 0; }
                else { Ttype[i] = last_type; Tsrc[i] = last_src; du += 1; }   /* = last_type: carries THAT ray's range and pose */
            } else {
                Ttype[i] = TY_O; last_type = TY_O; last_src = i;
                if (nz) du = 0;
            }
        }
    }
    /* RW:335-366: groups of consecutive typed entries (the last entry is never typed), flattened again */
    int M = 0;
    for (int i = 0; i + 1 < F; ++i)
        if (Ttype[i] != 0) { et[M] = Ttype[i]; es[M] = Tsrc[i]; ed[M] = cno_py_round(fr[Tsrc[i]], 3); ++M; }
#define EPX(k) FPX(es[k])
#define EPY(k) FPY(es[k])
    /* RW:368-403 segmentation by association of consecutive entries */
    int nseg = 0;
    for (int i = 0; i < M; ++i) {
        if (i == M - 1) { segend[i] = 1; ++nseg; }                          /* both branches of RW:375-390 close the segment */
        else if (cno_iou(EPX(i), EPY(i), EPX(i + 1), EPY(i + 1), e->bb) > 0.0) segend[i] = 0;
        else { segend[i] = 1; ++nseg; }
    }
    /* RW:408-420: first and last segment joined (first ++ last) when their outer ends associate with twice the box */
    int* order = tmp;          /* estimated index by position in the (possibly re-ordered) sequence */
    int* oend = tmp + n;       /* hmm: tmp has n ints; use segend for the re-ordered ends instead */
    (void)oend;
    int merged = 0, first_end = -1, last_start = 0;
    if (nseg > 1) {
        for (int i = 0; i < M; ++i) if (segend[i]) { first_end = i; break; }
        for (int i = M - 2; i >= 0; --i) if (segend[i]) { last_start = i + 1; break; }
        if (cno_iou(EPX(0), EPY(0), EPX(M - 1), EPY(M - 1), e->bb * 2) > 0.0) merged = 1;
    }
    int L = 0;
    if (merged) {
        for (int i = 0; i <= first_end; ++i) order[L++] = i;
        for (int i = last_start; i < M; ++i) order[L++] = i;
        for (int i = first_end + 1; i < last_start; ++i) order[L++] = i;
    } else for (int i = 0; i < M; ++i) order[L++] = i;
    if (merged) nseg -= 1;
    /* RW:426-468 confirmation */
    int maxc = M + 1;
    cobj_t* conf = (cobj_t*)malloc(sizeof(cobj_t) * (size_t)maxc);
    int nconf = 0;
    {
        int k0 = 0;
        while (k0 < L) {
            int k1 = k0;                                                     /* [k0, k1]: one segment in `order` space */
            if (merged && k0 == 0) k1 = first_end + (M - last_start);        /* the joined first segment */
            else while (!segend[order[k1]]) ++k1;
            int len = k1 - k0 + 1, no = 0, nw = 0;
            for (int k = k0; k <= k1; ++k) { no += (et[order[k]] == TY_O); nw += (et[order[k]] == TY_W); }
            int ce = order[k0 + len / 2];                                    /* Python-2 integer division */
            double dm = ed[ce];
            int est = cno_estimate_num_obs_scans(dm, c->max_scan_range, c->min_scan_range);
            int mn = len < est ? len : est;
            double score = (double)no / (double)mn;
            int obj = -1;
            if (no > 0 && nw > 0) {
                if (score >= 0.5) obj = (no > nw) ? TY_O : TY_W;
                else if (len <= est) obj = (no > nw) ? TY_O : TY_W;
                else obj = TY_W;
            } else {
                int lim = nseg < est ? nseg : est;
                if (len > lim) obj = (nw > 0) ? TY_W : TY_O;
            }
            if (obj >= 0) { conf[nconf].type = obj; conf[nconf].pose.x = EPX(ce); conf[nconf].pose.y = EPY(ce); conf[nconf].dist = dm; ++nconf; }
            k0 = k1 + 1;
        }
    }
    e->n_confirmed = nconf;
    /* RW:478-571 tracker: the same block as ENV:656-743 */
    tracker_update(e, conf, nconf, now, tmp);
    for (int i = 0; i < e->ntracks; ++i) {                                   /* RW:573-589 */
        track_t* t = &e->tracks[i];
        if (t->dq_len > 1) t->speed = cno_hypot(t->dq[0].y - t->dq[1].y, t->dq[0].x - t->dq[1].x) / t->t;
    }
    e->n_entries = 0;
    if (e->agent_dq_len == 2) {                                              /* RW:595-700 */
        double ts = e->agent_vel_timestep;
        if (ts == 0.0) e->status |= ST_DT_ZERO;
        double vx_ = (e->agent_dq[1].x - e->agent_dq[0].x) / ts, vy_ = (e->agent_dq[1].y - e->agent_dq[0].y) / ts;
        double agent_vel = sqrt(pow(vx_, 2) + pow(vy_, 2));
        double obstacle_vel = (e->ntracks == 0) ? 0.0 : e->tracks[0].speed;
        double vo_x = e->agent_dq[1].x, vo_y = e->agent_dq[1].y;
        for (int i = 0; i < e->ntracks; ++i) {
            track_t* t = &e->tracks[i];
            double chx = 0, chy = 0;
            if (t->dq_len > 1) { chx = t->dq[0].x - t->dq[1].x; chy = t->dq[0].y - t->dq[1].y; t->vel.x = chx / ts; t->vel.y = chy / ts; }
            vo_x = e->agent_dq[1].x + chx; vo_y = e->agent_dq[1].y + chy;
        }
        int ne = 0, best = -1; double bestcp = 0.0;
        for (int i = 0; i < e->ntracks; ++i) {
            track_t* t = &e->tracks[i];
            double dcp;
            int has = collision_point_impl(s->poly_c, s->poly_s, e->agent_dq[0].x, e->agent_dq[0].y, vo_x, vo_y, t->pose.x,
                                           t->pose.y, c->min_scan_range, &dcp, c->geos_untyped_empty);   /* RW:637: radius = min_scan_range */
            double rv = agent_vel - obstacle_vel;
            double gcp = cno_general_collision_prob(t->dist, c->max_scan_range, c->min_scan_range);
            double cpv;
            if (has) {
                if (rv == 0) cpv = 1.0 * gcp;
                else {
                    double ttc = dcp / rv;
                    if (ttc == 0.0) { e->status |= ST_TTC_ZERO; cpv = 0.5 * 1.0 + 0.5 * gcp; }
                    else cpv = 0.5 * cno_collision_prob(ttc) + 0.5 * gcp;
                }
            } else cpv = 0.5 * 0.0 + 0.5 * gcp;
            e->entry_cp[ne] = cpv; e->entry_ego[ne] = 0.0;
            if (ne == 0 || cpv >= bestcp) { bestcp = cpv; best = i; }          /* max((val, idx)): the LAST of equal maxima */
            ++ne;
        }
        e->n_entries = ne;
        if (ne == 0) e->collision_prob = 0.0;                                /* RW:663-666 */
        else {
            e->collision_prob = fmax(0.0, bestcp);                            /* RW:669 max(0.0, max(cp)) */
            closest_pose = e->tracks[best].pose; closest_vel = e->tracks[best].vel;
        }
        /* RW:673-690 goal-reaching probability: computed, never read */
        e->agent_dq[0] = e->agent_dq[1]; e->agent_dq_len = 1;              /* RW:693-699 */
        for (int i = 0; i < e->ntracks; ++i) e->tracks[i].t = now;
    }
    for (int j = 0; j < nconf; ++j)                                          /* RW:702-706 */
        if (conf[j].type == TY_O && conf[j].dist < 0.140) { e->ego_viol += 1; break; }
    if (e->collision_prob > 0.4) e->social_viol += 1;                       /* RW:708 (None > 0.4 is False in Python 2) */
    if (!e->done) {                                                          /* RW:715-728 */
        double mn = scan[0];
        for (int i = 1; i < n; ++i) if (scan[i] < mn) mn = scan[i];
        if (mn < c->min_scan_range) e->done = 1;
        if (in_box(px, py, c->goal_x, c->goal_y, 0.20)) e->done = 1;
        if (step_counter >= c->max_steps) e->done = 1;
    }
    for (int i = 0; i < n; ++i) state[i] = scan[i];                          /* RW:730-747: nothing is rounded again */
    state[n] = heading; state[n + 1] = distance_to_goal;
    state[n + 2] = cno_py_round(px, 3); state[n + 3] = cno_py_round(py, 3);
    state[n + 4] = cno_py_round(3.14, 3);                                    /* round(self.yaw, 3): the constructor's constant */
    state[n + 5] = cno_py_round(agent_vel_x, 3); state[n + 6] = cno_py_round(agent_vel_y, 3);
    state[n + 7] = closest_pose.x; state[n + 8] = closest_pose.y;
    state[n + 9] = closest_vel.x; state[n + 10] = closest_vel.y;
    *done_out = e->done;
#undef FPX
#undef FPY
#undef EPX
#undef EPY
    free(conf);
    free(scan);
}

/* RW:751-849: no way-point bonus; state[359] is the heading and state[360] the distance (labelled correctly here) */
static double rw_compute_reward(const cno_sim* s, env_t* e, const double* state, double px, double py, int done)
{
    const cno_config* c = &s->cfg;
    const int n = s->n;
    double cur_head = state[n], cur_dist = state[n + 1];
    double dd = cur_dist - e->prev_dist, hd = cur_head - e->prev_head;
    int htg = 0, dtg = 0;
    if (dd < 0) dtg = 1;
    double ph = e->prev_head;
    if (hd > 0) {
        if (cur_head > 0 && ph < 0) htg = 1;
        if (cur_head < 0 && ph < 0) htg = 1;
        if (cur_head < 0 && ph > 0) htg = 1;
        if (cur_head > 0 && ph > 0) htg = 0;
    }
    if (hd < 0) {
        if (cur_head < 0 && ph > 0) htg = 1;
        if (cur_head > 0 && ph > 0) htg = 1;
        if (cur_head > 0 && ph < 0) htg = 1;
        if (cur_head < 0 && ph < 0) htg = 0;
    }
    double reward = (double)(-2 + dtg + htg);
    e->prev_dist = cur_dist; e->prev_head = cur_head;
    if (done) {
        if (in_box(px, py, c->goal_x, c->goal_y, 0.20)) { e->ep_failure = 0; e->ep_success = 1; reward = 200 + reward; }
        else { e->ep_failure = 1; e->ep_success = 0; reward = -200 + reward; }
    }
    return reward;
}

static void env_init(const cno_sim* s, env_t* e)
{
    const cno_config* c = &s->cfg;
    /* Env.__init__ (ENV:43-168) */
    e->wpx = c->goal_x; e->wpy = c->goal_y;
    e->prev_dist = 0.0; e->prev_head = 0.0;
    e->done = 0;
    e->agent_dq_len = 0; e->agent_vel_timestep = 0.0;
    e->bb = (c->obs_layout == 2) ? 0.0210 : 0.0;                  /* RW:103 */
    e->ntracks = 0;
    e->ego_score_cp = 0.0; e->collision_prob = (c->obs_layout == 2) ? -INFINITY : 0.0;   /* RW:80 None: compares below any number in Python 2 */
    e->ego_viol = e->social_viol = e->obst_steps = 0;
    e->ep_success = e->ep_failure = 0;
    e->ep_step = 0; e->ep_return = 0.0; e->last_return = 0.0;
    e->clock = 0.0; e->crowd_ms = 0;
    e->status = 0;
}

/* Env.reset (ENV:1227-1263) followed by the trainer's sleep and `env.done = False` (TRAIN:113-116) */
static void env_reset_flow(const cno_sim* s, env_t* e, int64_t gid, double* obs)
{
    const cno_config* c = &s->cfg;
    int32_t idx[64];
    int done;
    sim_reset(s, e);
    e->clock += (double)c->scan_latency_ms / 1000.0; /* wait_for_message('scan') */
    sim_advance(s, e, gid, c->scan_latency_ms);
    raycast_impl(c, s->lidar_c, s->lidar_s, e->rx, e->ry, e->ryaw, e->ped_p, c->n_peds, e->ranges);
    if (c->obs_layout == 1) {                                      /* ORIG:456-486, then SAC:106-107 */
        e->prev_dist = dist3(e->rx, e->ry, c->goal_x, c->goal_y);  /* ORIG:472 (unrounded) */
        e->prev_head = orig_heading(c, e->rx, e->ry, e->ryaw);     /* ORIG:473 */
        orig_get_state(s, e, e->ranges, e->rx, e->ry, e->ryaw, 0, obs, &done);
    } else if (c->obs_layout == 2) {                               /* RW:910-944 */
        e->prev_dist = rw_distance(c, e->rx, e->ry);
        e->prev_head = rw_heading(c, e->rx, e->ry, e->ryaw);
        rw_get_state(s, e, e->ranges, e->rx, e->ry, e->ryaw, e->rv, e->rw, 0, e->clock, obs, &done);
    } else {
    e->prev_dist = dist3(e->rx, e->ry, e->wpx, e->wpy);           /* ENV:1243 (unrounded) */
    e->prev_head = heading_to_goal(c, e, e->rx, e->ry, e->ryaw);   /* ENV:1244 */
    env_get_state(s, e, e->ranges, e->rx, e->ry, e->ryaw, e->rv, e->rw, 0, e->clock, obs, &done, idx);
    }
    e->social_viol = 0; e->ego_viol = 0; e->obst_steps = 0;       /* ENV:1260-1262 */
    e->clock += (double)c->settle_ms / 1000.0;                     /* TRAIN:114 time.sleep(0.1) */
    sim_advance(s, e, gid, c->settle_ms);
    e->done = 0;                                                   /* TRAIN:116 */
    e->ep_step = 0; e->ep_return = 0.0;
    e->pending_reset = 0;
}

/* Env.step (ENV:1164-1225), continuous mode */
static void env_step_flow(const cno_sim* s, env_t* e, int64_t gid, double v, double w, int step_counter,
                          double* obs, double* reward, int* done, int32_t* topk_idx)
{
    const cno_config* c = &s->cfg;
    double t0 = e->clock;
    if (c->wheel_accel > 0.0) { e->cmd_v = v; e->cmd_w = w; }     /* pub_cmd_vel.publish (ENV:1200): the wheels ramp towards it */
    else { e->rv = v; e->rw = w; }                                 /* kinematic robot: the command is the twist */
    e->clock += (double)c->dt_ms / 1000.0;                         /* time.sleep(0.15) (ENV:1201) */
    sim_advance(s, e, gid, c->dt_ms);
    double end_timestep = e->clock - t0;                           /* ENV:1202 */
    if (c->obs_layout == 2) {
        /* RW:876-883: no sleep before the measurement, so end_timestep = 0 < 0.05 -> time.sleep(0.05 - 0) (the dt_ms of this
         * layout) and `end_timestep += 0.05 - end_timestep + 0.1`: the command is held 0.05 s and booked as 0.15000000000000002 */
        const double dts = (double)c->dt_ms / 1000.0;
        end_timestep = 0.0 + ((dts - 0.0) + 0.1);
    }
    v2 p = { cno_py_round(e->rx, 3), cno_py_round(e->ry, 3) };   /* ENV:1208 */
    if (e->agent_dq_len < 2) e->agent_dq[e->agent_dq_len++] = p;
    else { e->agent_dq[0] = e->agent_dq[1]; e->agent_dq[1] = p; }
    e->agent_vel_timestep = end_timestep;                          /* ENV:1209 */
    e->clock += (double)c->scan_latency_ms / 1000.0;               /* wait_for_message('scan') (ENV:1218) */
    sim_advance(s, e, gid, c->scan_latency_ms);
    raycast_impl(c, s->lidar_c, s->lidar_s, e->rx, e->ry, e->ryaw, e->ped_p, c->n_peds, e->ranges);
    if (c->obs_layout == 1) {                                      /* ORIG:404-454 */
        orig_get_state(s, e, e->ranges, e->rx, e->ry, e->ryaw, step_counter, obs, done);
        *reward = orig_compute_reward(s, e, obs, e->rx, e->ry, *done);
        for (int k = 0; k < c->k_obstacles; ++k) topk_idx[k] = -1;
    } else if (c->obs_layout == 2) {                               /* RW:898-901 */
        rw_get_state(s, e, e->ranges, e->rx, e->ry, e->ryaw, e->rv, e->rw, step_counter, e->clock, obs, done);
        *reward = rw_compute_reward(s, e, obs, e->rx, e->ry, *done);
        for (int k = 0; k < c->k_obstacles; ++k) topk_idx[k] = -1;
    } else {
    env_get_state(s, e, e->ranges, e->rx, e->ry, e->ryaw, e->rv, e->rw, step_counter, e->clock, obs, done, topk_idx);
    *reward = env_compute_reward(s, e, obs, e->rx, e->ry, *done);
    }
    if (*done) {                                                   /* pub_cmd_vel.publish(Twist()) (ENV:1160) */
        if (c->wheel_accel > 0.0) { e->cmd_v = 0.0; e->cmd_w = 0.0; } else { e->rv = 0.0; e->rw = 0.0; }
    }
}

/* ------------------------------------------------------------------------------------------
 * public API
 * ---------------------------------------------------------------------------------------- */

static void default_ped_init(const cno_sim* s, int env, double* xy)
{
    const cno_config* c = &s->cfg;
    int64_t gid = c->env_index_base + env;
    double lo = -c->room_half + 0.1, span = 2.0 * c->room_half - 0.2;
    for (int i = 0; i < c->n_peds; ++i) {
        uint32_t att = 0;
        for (;;) {
            double x = fma(span, cno_rng_u01(c->seed, gid, 2u, (uint32_t)i, 2 * att), lo);
            double y = fma(span, cno_rng_u01(c->seed, gid, 2u, (uint32_t)i, 2 * att + 1), lo);
            double dx = x - c->spawn_x, dy = y - c->spawn_y;
            ++att;
            if (fma(dx, dx, dy * dy) >= 0.16 || att > 1000) { xy[2 * i] = x; xy[2 * i + 1] = y; break; }
        }
    }
}

int cno_create(const cno_config* cfg, cno_sim** out)
{
    if (!cfg || !out) return -1;
    if (cfg->n_envs < 1 || cfg->n_peds < 0 || cfg->n_rays < 8 || cfg->k_obstacles < 1 || cfg->k_obstacles > 16)
        return -2;
    if (cfg->ped_cycle_ms < 1 || cfg->dt_ms < 1) return -2;
    if (cfg->ped_mode < 0 || cfg->ped_mode > 2) return -2;
    if (cfg->ped_mode == 2 && (cfg->ped_contact || !(cfg->sf_tau > 0.0) || !(cfg->sf_B > 0.0) || !(cfg->sf_wall_B > 0.0) || cfg->sf_tick_ms < 0)) return -2;
    if (cfg->ped_mode == 2 && !((double)cfg->n_peds * fabs(cfg->sf_A) * exp(2.0 * cfg->ped_radius / cfg->sf_B) < 32768.0)) return -2;   /* sf_quant: exact sums */
    if (!(cfg->wheel_accel >= 0.0) || (cfg->wheel_accel > 0.0 && (cfg->ped_contact || cfg->ped_mode == 2 || cfg->obs_layout != 0 ||
                                                                    !(cfg->wheel_separation > 0.0)))) return -2;
    if (cfg->scan_f32 < 0 || cfg->scan_f32 > 1) return -2;
    cno_sim* s = (cno_sim*)calloc(1, sizeof(cno_sim));
    s->cfg = *cfg;
    s->n = cfg->n_rays - 1;
    if (cfg->obs_layout < 0 || cfg->obs_layout > 2) return -2;
    s->D = cfg->obs_layout == 1 ? s->n + 4 : (cfg->obs_layout == 2 ? s->n + 11 : s->n + 7 + 4 * cfg->k_obstacles);
    int R = cfg->n_rays, P = cfg->n_peds;
    s->lidar_c = (double*)malloc(sizeof(double) * 2 * R);
    s->lidar_s = s->lidar_c + R;
    double step = cfg->lidar_span / (double)(R - 1);
    for (int k = 0; k < R; ++k) cno_det_sincos((double)k * step, &s->lidar_s[k], &s->lidar_c[k]);
    poly_tables(s->poly_c, s->poly_s);
    s->envs = (env_t*)calloc((size_t)cfg->n_envs, sizeof(env_t));
    for (int e = 0; e < cfg->n_envs; ++e) {
        env_t* en = &s->envs[e];
        en->ped_p = (double*)calloc((size_t)(11 * (P > 0 ? P : 1)) + R, sizeof(double));
        en->ped_v = en->ped_p + 2 * P;
        en->ped_init = en->ped_v + 2 * P;
        en->ped_preset = en->ped_init + 2 * P;
        en->ped_aux = en->ped_preset + 2 * P;
        en->ranges = en->ped_aux + 3 * P;
        env_init(s, en);
        default_ped_init(s, e, en->ped_init);
        if (cfg->ped_mode == 2)
            for (int i = 0; i < P; ++i) sf_goal(cfg, cfg->env_index_base + e, i, 0u, &en->ped_aux[3 * i], &en->ped_aux[3 * i + 1]);
        en->rx = cfg->spawn_x; en->ry = cfg->spawn_y; en->ryaw = cfg->spawn_yaw;
        memcpy(en->ped_p, en->ped_init, sizeof(double) * 2 * P);
    }
    *out = s;
    return 0;
}

void cno_destroy(cno_sim* s)
{
    if (!s) return;
    for (int e = 0; e < s->cfg.n_envs; ++e) free(s->envs[e].ped_p);
    free(s->envs);
    free(s->lidar_c);
    free(s);
}

int cno_obs_dim(const cno_sim* s) { return s->D; }

int cno_set_ped_init(cno_sim* s, const double* xy)
{
    int P = s->cfg.n_peds;
    for (int e = 0; e < s->cfg.n_envs; ++e) {
        memcpy(s->envs[e].ped_init, xy + (size_t)e * 2 * P, sizeof(double) * 2 * P);
        memcpy(s->envs[e].ped_p, s->envs[e].ped_init, sizeof(double) * 2 * P);
    }
    return 0;
}

int cno_get_ped_init(const cno_sim* s, double* xy)
{
    int P = s->cfg.n_peds;
    for (int e = 0; e < s->cfg.n_envs; ++e) memcpy(xy + (size_t)e * 2 * P, s->envs[e].ped_init, sizeof(double) * 2 * P);
    return 0;
}

int cno_set_ped_preset_vel(cno_sim* s, const double* vxy)
{
    int P = s->cfg.n_peds;
    for (int e = 0; e < s->cfg.n_envs; ++e)
        memcpy(s->envs[e].ped_preset, vxy + (size_t)e * 2 * P, sizeof(double) * 2 * P);
    return 0;
}

int cno_set_num_threads(int n)
{
    g_threads = n < 1 ? 1 : n;
    return g_threads;
}

int cno_reset(cno_sim* s, const uint8_t* mask, double* obs)
{
    int N = s->cfg.n_envs;
    g_py2 = s->cfg.py2_round;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int e = 0; e < N; ++e) {
        if (mask && !mask[e]) continue;
        g_py2 = s->cfg.py2_round;
        env_reset_flow(s, &s->envs[e], s->cfg.env_index_base + e, obs + (size_t)e * s->D);
    }
    return 0;
}

int cno_step(cno_sim* s, const double* action, const int32_t* step_counter, int auto_reset, double* obs,
             double* final_obs, double* reward, uint8_t* done, int32_t* topk_idx)
{
    int N = s->cfg.n_envs, K = s->cfg.k_obstacles;
    g_py2 = s->cfg.py2_round;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int e = 0; e < N; ++e) {
        env_t* en = &s->envs[e];
        int64_t gid = s->cfg.env_index_base + e;
        int32_t idx_local[16];
        g_py2 = s->cfg.py2_round;
        int32_t* idx = topk_idx ? topk_idx + (size_t)e * K : idx_local;
        double* o = obs + (size_t)e * s->D;
        double r; int d;
        if (auto_reset == 2 && en->pending_reset) { /* next-step reset: this call resets, action ignored */
            en->pending_reset = 0;
            env_reset_flow(s, en, gid, o);
            reward[e] = 0.0; done[e] = 0;
            for (int k = 0; k < K; ++k) idx[k] = -1;
            if (final_obs) memcpy(final_obs + (size_t)e * s->D, o, sizeof(double) * s->D);
            continue;
        }
        en->ep_step += 1;
        int sc = step_counter ? step_counter[e] : en->ep_step;
        env_step_flow(s, en, gid, action[2 * e], action[2 * e + 1], sc, o, &r, &d, idx);
        en->ep_return += r;
        reward[e] = r;
        done[e] = (uint8_t)d;
        if (final_obs) memcpy(final_obs + (size_t)e * s->D, o, sizeof(double) * s->D);
        if (d) {
            en->last_return = en->ep_return;
            en->episodes += 1;
            en->last_ego_viol = en->ego_viol; en->last_social_viol = en->social_viol;
            en->last_obst_steps = en->obst_steps; en->last_ep_steps = en->ep_step;
            if (auto_reset == 1) env_reset_flow(s, en, gid, o);
            else if (auto_reset == 2) en->pending_reset = 1;
        }
    }
    return 0;
}

int cno_get_counters(const cno_sim* s, int32_t* out)
{
    for (int e = 0; e < s->cfg.n_envs; ++e) {
        const env_t* en = &s->envs[e];
        out[6 * e + 0] = en->ego_viol; out[6 * e + 1] = en->social_viol; out[6 * e + 2] = en->obst_steps;
        out[6 * e + 3] = en->ep_step; out[6 * e + 4] = en->ep_success; out[6 * e + 5] = en->ep_failure;
    }
    return 0;
}

int cno_get_returns(const cno_sim* s, double* out)
{
    for (int e = 0; e < s->cfg.n_envs; ++e) out[e] = s->envs[e].last_return;
    return 0;
}

int cno_get_sim_state(const cno_sim* s, int env, double* robot5, double* ped_p, double* ped_v, double* ranges)
{
    const env_t* e = &s->envs[env];
    int P = s->cfg.n_peds;
    if (robot5) { robot5[0] = e->rx; robot5[1] = e->ry; robot5[2] = e->ryaw; robot5[3] = e->rv; robot5[4] = e->rw; }
    if (ped_p) memcpy(ped_p, e->ped_p, sizeof(double) * 2 * P);
    if (ped_v) memcpy(ped_v, e->ped_v, sizeof(double) * 2 * P);
    if (ranges) memcpy(ranges, e->ranges, sizeof(double) * s->cfg.n_rays);
    return 0;
}

int cno_get_debug(const cno_sim* s, int env, cno_debug* o)
{
    const env_t* e = &s->envs[env];
    memset(o, 0, sizeof(*o));
    o->n_confirmed = e->n_confirmed; o->n_tracks = e->ntracks; o->n_entries = e->n_entries; o->status = e->status;
    o->bb = e->bb; o->collision_prob = e->collision_prob; o->ego_score = e->ego_score_cp;
    o->wpx = e->wpx; o->wpy = e->wpy;
    for (int i = 0; i < e->n_entries && i < CNO_MAX_TRACKS; ++i) { o->entry_cp[i] = e->entry_cp[i]; o->entry_ego[i] = e->entry_ego[i]; }
    for (int i = 0; i < e->ntracks; ++i) {
        const track_t* t = &e->tracks[i];
        o->track_pose[i][0] = t->pose.x; o->track_pose[i][1] = t->pose.y;
        o->track_dist[i] = t->dist; o->track_speed[i] = t->speed;
        o->track_vel[i][0] = t->vel.x; o->track_vel[i][1] = t->vel.y;
        o->track_t[i] = t->t; o->track_dqlen[i] = t->dq_len;
    }
    return 0;
}

/* Golden replay: everything Gazebo/ROS supplied comes from the caller. */
int cno_ext_call(cno_sim* s, int env, const cno_ext_in* in, const double* ranges, double* obs, double* reward,
                 uint8_t* done, int32_t* topk_idx)
{
    env_t* e = &s->envs[env];
    const cno_config* c = &s->cfg;
    int32_t idx_local[16];
    int32_t* idx = topk_idx ? topk_idx : idx_local;
    int d = 0;
    g_py2 = c->py2_round;
    if (c->obs_layout == 1) {
        if (in->is_reset) {
            e->prev_dist = dist3(in->px, in->py, c->goal_x, c->goal_y);
            e->prev_head = orig_heading(c, in->px, in->py, in->yaw);
            orig_get_state(s, e, ranges, in->px, in->py, in->yaw, 0, obs, &d);
            if (reward) *reward = 0.0;
        } else {
            orig_get_state(s, e, ranges, in->px, in->py, in->yaw, in->step_counter, obs, &d);
            double r = orig_compute_reward(s, e, obs, in->px, in->py, d);
            if (reward) *reward = r;
        }
        if (done) *done = (uint8_t)d;
        for (int k = 0; k < c->k_obstacles; ++k) idx[k] = -1;
        return 0;
    }
    if (c->obs_layout == 2) {
        if (in->is_reset) {
            e->prev_dist = rw_distance(c, in->px, in->py);
            e->prev_head = rw_heading(c, in->px, in->py, in->yaw);
            rw_get_state(s, e, ranges, in->px, in->py, in->yaw, in->v, in->w, 0, in->now, obs, &d);
            e->social_viol = 0; e->ego_viol = 0;
            if (reward) *reward = 0.0;
        } else {
            v2 p = { cno_py_round(in->deque_x, 3), cno_py_round(in->deque_y, 3) };
            if (e->agent_dq_len < 2) e->agent_dq[e->agent_dq_len++] = p;
            else { e->agent_dq[0] = e->agent_dq[1]; e->agent_dq[1] = p; }
            e->agent_vel_timestep = in->end_timestep;
            rw_get_state(s, e, ranges, in->px, in->py, in->yaw, in->v, in->w, in->step_counter, in->now, obs, &d);
            double r = rw_compute_reward(s, e, obs, in->px, in->py, d);
            if (reward) *reward = r;
        }
        if (done) *done = (uint8_t)d;
        for (int k = 0; k < c->k_obstacles; ++k) idx[k] = -1;
        return 0;
    }
    if (in->is_reset) {
        e->prev_dist = dist3(in->px, in->py, e->wpx, e->wpy);
        e->prev_head = heading_to_goal(c, e, in->px, in->py, in->yaw);
        env_get_state(s, e, ranges, in->px, in->py, in->yaw, in->v, in->w, 0, in->now, obs, &d, idx);
        e->social_viol = 0; e->ego_viol = 0; e->obst_steps = 0;
        if (reward) *reward = 0.0;
        if (done) *done = (uint8_t)d;
        return 0;
    }
    v2 p = { cno_py_round(in->deque_x, 3), cno_py_round(in->deque_y, 3) };
    if (e->agent_dq_len < 2) e->agent_dq[e->agent_dq_len++] = p;
    else { e->agent_dq[0] = e->agent_dq[1]; e->agent_dq[1] = p; }
    e->agent_vel_timestep = in->end_timestep;
    env_get_state(s, e, ranges, in->px, in->py, in->yaw, in->v, in->w, in->step_counter, in->now, obs, &d, idx);
    double r = env_compute_reward(s, e, obs, in->px, in->py, d);
    if (reward) *reward = r;
    if (done) *done = (uint8_t)d;
    return 0;
}

void cno_ext_set_done(cno_sim* s, int env, int done) { s->envs[env].done = done; }

/* function-level entry points on a handle (golden vectors of ENV:191-237, 1046-1162, 1285-1319) */
double cno_heading_to_goal(cno_sim* s, int env, double wpx, double wpy, double px, double py, double yaw)
{
    env_t* e = &s->envs[env];
    e->wpx = wpx; e->wpy = wpy;
    g_py2 = s->cfg.py2_round;
    return heading_to_goal(&s->cfg, e, px, py, yaw);
}
double cno_distance_to_goal(double px, double py, double wpx, double wpy) { return dist3(px, py, wpx, wpy); }
int cno_in_box(double x, double y, double gx, double gy, double eps) { return in_box(x, y, gx, gy, eps); }
/* Env.compute_reward(state, step_counter, done) with previous_heading / previous_distance / way-point preset */
double cno_compute_reward(cno_sim* s, int env, double cur_head, double cur_dist, double prev_head, double prev_dist,
                          double wpx, double wpy, double px, double py, int done)
{
    env_t* e = &s->envs[env];
    double* st = (double*)calloc((size_t)s->D, sizeof(double));
    st[s->n] = cur_head; st[s->n + 1] = cur_dist;
    e->prev_head = prev_head; e->prev_dist = prev_dist; e->wpx = wpx; e->wpy = wpy;
    double r = env_compute_reward(s, e, st, px, py, done);
    free(st);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * Simulator-only entry points for oracle/harness (which plays Gazebo for the reference's Python)
 * ---------------------------------------------------------------------------------------- */
int cno_hsim_reset(cno_sim* s, int env) { sim_reset(s, &s->envs[env]); return 0; }

int cno_hsim_advance(cno_sim* s, int env, int32_t ms, double v, double w)
{
    env_t* e = &s->envs[env];
    if (s->cfg.wheel_accel > 0.0) { e->cmd_v = v; e->cmd_w = w; } else { e->rv = v; e->rw = w; }
    sim_advance(s, e, s->cfg.env_index_base + env, ms);
    return 0;
}

int cno_hsim_scan(cno_sim* s, int env, double* ranges)
{
    env_t* e = &s->envs[env];
    raycast_impl(&s->cfg, s->lidar_c, s->lidar_s, e->rx, e->ry, e->ryaw, e->ped_p, s->cfg.n_peds, ranges);
    return 0;
}

int cno_set_robot(cno_sim* s, int env, double x, double y, double yaw)
{
    env_t* e = &s->envs[env];
    e->rx = x; e->ry = y; e->ryaw = yaw;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * State exchange in the PRODUCT's snapshot layout (include/crowdnav.h: CN_SD_*, CN_SI_*, CN_TF_*): lets a test
 * seed this oracle from a GPU snapshot and step both side by side (tools/bisect_divergence.py), and read the oracle's state
 * back field by field.  sd[24] f64, si[16] i32, ped_p / ped_v [P][2], trk [trk_cap][12], ped_init / ped_preset [P][2],
 * ped_aux [P][3].  Any pointer may be NULL.
 * ---------------------------------------------------------------------------------------- */
enum { SD_RX = 0, SD_RY, SD_RYAW, SD_RV, SD_RW, SD_CLOCK, SD_WPX, SD_WPY, SD_PREV_DIST, SD_PREV_HEAD, SD_DQ0X, SD_DQ0Y, SD_DQ1X,
       SD_DQ1Y, SD_TS, SD_BB, SD_EGO, SD_CPROB, SD_EP_RETURN, SD_LAST_RETURN, SD_LAST_EGO_VIOL, SD_LAST_SOCIAL_VIOL,
       SD_LAST_OBST_STEPS, SD_LAST_EP_STEPS, SD_COUNT = 24 };
enum { SI_DONE = 0, SI_DQ_LEN, SI_NTRACKS, SI_EGO_VIOL, SI_SOCIAL_VIOL, SI_OBST_STEPS, SI_SUCCESS, SI_FAILURE, SI_EP_STEP, SI_STATUS,
       SI_NCONF, SI_NENTRIES, SI_CROWD_LO, SI_CROWD_HI, SI_PENDING_RESET, SI_EPISODES, SI_COUNT = 16 };
enum { TF_PX = 0, TF_PY, TF_DIST, TF_D0X, TF_D0Y, TF_D1X, TF_D1Y, TF_T, TF_SPEED, TF_VX, TF_VY, TF_DQLEN, TF_COUNT = 12 };

int cno_set_state(cno_sim* s, int env, const double* sd, const int32_t* si, const double* ped_p, const double* ped_v,
                  const double* trk, int trk_cap, const double* ped_init, const double* ped_preset, const double* ped_aux)
{
    if (!s || env < 0 || env >= s->cfg.n_envs) return -1;
    env_t* e = &s->envs[env];
    const int P = s->cfg.n_peds;
    if (sd) {
        e->rx = sd[SD_RX]; e->ry = sd[SD_RY]; e->ryaw = sd[SD_RYAW]; e->rv = sd[SD_RV]; e->rw = sd[SD_RW];
        e->clock = sd[SD_CLOCK]; e->wpx = sd[SD_WPX]; e->wpy = sd[SD_WPY];
        e->prev_dist = sd[SD_PREV_DIST]; e->prev_head = sd[SD_PREV_HEAD];
        e->agent_dq[0].x = sd[SD_DQ0X]; e->agent_dq[0].y = sd[SD_DQ0Y]; e->agent_dq[1].x = sd[SD_DQ1X]; e->agent_dq[1].y = sd[SD_DQ1Y];
        e->agent_vel_timestep = sd[SD_TS]; e->bb = sd[SD_BB]; e->ego_score_cp = sd[SD_EGO]; e->collision_prob = sd[SD_CPROB];
        e->ep_return = sd[SD_EP_RETURN]; e->last_return = sd[SD_LAST_RETURN];
        e->last_ego_viol = (int)sd[SD_LAST_EGO_VIOL]; e->last_social_viol = (int)sd[SD_LAST_SOCIAL_VIOL];
        e->last_obst_steps = (int)sd[SD_LAST_OBST_STEPS]; e->last_ep_steps = (int)sd[SD_LAST_EP_STEPS];
    }
    if (si) {
        e->done = si[SI_DONE]; e->agent_dq_len = si[SI_DQ_LEN]; e->ntracks = si[SI_NTRACKS];
        e->ego_viol = si[SI_EGO_VIOL]; e->social_viol = si[SI_SOCIAL_VIOL]; e->obst_steps = si[SI_OBST_STEPS];
        e->ep_success = si[SI_SUCCESS]; e->ep_failure = si[SI_FAILURE]; e->ep_step = si[SI_EP_STEP]; e->status = si[SI_STATUS];
        e->n_confirmed = si[SI_NCONF]; e->n_entries = si[SI_NENTRIES];
        e->crowd_ms = (int64_t)(((uint64_t)(uint32_t)si[SI_CROWD_HI] << 32) | (uint32_t)si[SI_CROWD_LO]);
        e->pending_reset = si[SI_PENDING_RESET]; e->episodes = si[SI_EPISODES];
        if (e->ntracks < 0 || e->ntracks > CNO_MAX_TRACKS) return -2;
    }
    if (ped_p) memcpy(e->ped_p, ped_p, sizeof(double) * 2 * P);
    if (ped_v) memcpy(e->ped_v, ped_v, sizeof(double) * 2 * P);
    if (ped_init) memcpy(e->ped_init, ped_init, sizeof(double) * 2 * P);
    if (ped_preset) memcpy(e->ped_preset, ped_preset, sizeof(double) * 2 * P);
    if (ped_aux) memcpy(e->ped_aux, ped_aux, sizeof(double) * 3 * P);
    if (trk) {
        if (e->ntracks > trk_cap) return -2;
        for (int i = 0; i < e->ntracks; ++i) {
            const double* r = trk + (size_t)i * TF_COUNT;
            track_t* t = &e->tracks[i];
            t->pose.x = r[TF_PX]; t->pose.y = r[TF_PY]; t->dist = r[TF_DIST];
            t->dq[0].x = r[TF_D0X]; t->dq[0].y = r[TF_D0Y]; t->dq[1].x = r[TF_D1X]; t->dq[1].y = r[TF_D1Y];
            t->t = r[TF_T]; t->speed = r[TF_SPEED]; t->vel.x = r[TF_VX]; t->vel.y = r[TF_VY]; t->dq_len = (int)r[TF_DQLEN];
            t->id = (s->cfg.risk_mode == 1) ? (int)r[TF_T] : 0;      /* gt mode keeps the pedestrian id in the T slot */
        }
    }
    return 0;
}

int cno_get_state(const cno_sim* s, int env, double* sd, int32_t* si, double* ped_p, double* ped_v, double* trk, int trk_cap,
                  double* ped_aux)
{
    if (!s || env < 0 || env >= s->cfg.n_envs) return -1;
    const env_t* e = &s->envs[env];
    const int P = s->cfg.n_peds;
    if (sd) {
        memset(sd, 0, sizeof(double) * SD_COUNT);
        sd[SD_RX] = e->rx; sd[SD_RY] = e->ry; sd[SD_RYAW] = e->ryaw; sd[SD_RV] = e->rv; sd[SD_RW] = e->rw;
        sd[SD_CLOCK] = e->clock; sd[SD_WPX] = e->wpx; sd[SD_WPY] = e->wpy;
        sd[SD_PREV_DIST] = e->prev_dist; sd[SD_PREV_HEAD] = e->prev_head;
        sd[SD_DQ0X] = e->agent_dq[0].x; sd[SD_DQ0Y] = e->agent_dq[0].y; sd[SD_DQ1X] = e->agent_dq[1].x; sd[SD_DQ1Y] = e->agent_dq[1].y;
        sd[SD_TS] = e->agent_vel_timestep; sd[SD_BB] = e->bb; sd[SD_EGO] = e->ego_score_cp; sd[SD_CPROB] = e->collision_prob;
        sd[SD_EP_RETURN] = e->ep_return; sd[SD_LAST_RETURN] = e->last_return;
        sd[SD_LAST_EGO_VIOL] = e->last_ego_viol; sd[SD_LAST_SOCIAL_VIOL] = e->last_social_viol;
        sd[SD_LAST_OBST_STEPS] = e->last_obst_steps; sd[SD_LAST_EP_STEPS] = e->last_ep_steps;
    }
    if (si) {
        memset(si, 0, sizeof(int32_t) * SI_COUNT);
        si[SI_DONE] = e->done; si[SI_DQ_LEN] = e->agent_dq_len; si[SI_NTRACKS] = e->ntracks;
        si[SI_EGO_VIOL] = e->ego_viol; si[SI_SOCIAL_VIOL] = e->social_viol; si[SI_OBST_STEPS] = e->obst_steps;
        si[SI_SUCCESS] = e->ep_success; si[SI_FAILURE] = e->ep_failure; si[SI_EP_STEP] = e->ep_step; si[SI_STATUS] = e->status;
        si[SI_NCONF] = e->n_confirmed; si[SI_NENTRIES] = e->n_entries;
        si[SI_CROWD_LO] = (int32_t)(uint32_t)((uint64_t)e->crowd_ms & 0xffffffffull);
        si[SI_CROWD_HI] = (int32_t)(uint32_t)((uint64_t)e->crowd_ms >> 32);
        si[SI_PENDING_RESET] = e->pending_reset; si[SI_EPISODES] = e->episodes;
    }
    if (ped_p) memcpy(ped_p, e->ped_p, sizeof(double) * 2 * P);
    if (ped_v) memcpy(ped_v, e->ped_v, sizeof(double) * 2 * P);
    if (ped_aux) memcpy(ped_aux, e->ped_aux, sizeof(double) * 3 * P);
    if (trk) {
        memset(trk, 0, sizeof(double) * (size_t)trk_cap * TF_COUNT);
        for (int i = 0; i < e->ntracks && i < trk_cap; ++i) {
            double* r = trk + (size_t)i * TF_COUNT;
            const track_t* t = &e->tracks[i];
            r[TF_PX] = t->pose.x; r[TF_PY] = t->pose.y; r[TF_DIST] = t->dist;
            r[TF_D0X] = t->dq[0].x; r[TF_D0Y] = t->dq[0].y; r[TF_D1X] = t->dq[1].x; r[TF_D1Y] = t->dq[1].y;
            r[TF_T] = (s->cfg.risk_mode == 1) ? (double)t->id : t->t; r[TF_SPEED] = t->speed;
            r[TF_VX] = t->vel.x; r[TF_VY] = t->vel.y; r[TF_DQLEN] = (double)t->dq_len;
        }
    }
    return 0;
}
