// crowdnav_kernel.hip -- the fused environment-step kernel of libcrowdnav.so (gfx950 / MI355X).
//
// One 64-lane wavefront = one environment (one block of 64 threads).  A launch advances every
// environment by one control period:
//     crowd velocity process -> pedestrian integration -> diff-drive -> 360-ray lidar
//  -> Env.get_state (scan sanitise, end points, gradient/type machine, segmentation, confirmation,
//     obstacle tracker, collision-cone CP, top-K) -> Env.compute_reward -> [Env.reset on done]
// with the whole per-env working set (end points, ranges, tracker table, pedestrians) in LDS.
// HBM is touched once to read the env state and once to write it back plus the observation.
//
// Reference lines: ENV = environment_stage_1_nobonus.py, UTL = utils.py, CROWD =
// crowd_behaviors/simulate_crowd.py, TRAIN = start_td3_training.py (under
// /root/reference/turtlebot3_rl_sim/src).  Gazebo owns the physics in the reference; the
// simulator here is defined in DESIGN.md and restated independently in oracle/cn_oracle.c.
#include "crowdnav_device.h"
#include "crowdnav_kernel.h"

#define TY_NONE 0
#define TY_W 1
#define TY_O 2

namespace {

struct EnvRegs {  // per-env scalars, uniform across the wave
    double rx, ry, ryaw, rv, rw, clock, wpx, wpy, prev_dist, prev_head;
    double dq0x, dq0y, dq1x, dq1y, ts, bb, ego, cprob, ep_ret, last_ret;
    long long crowd_ms;
    int done, dq_len, ntracks, ego_viol, social_viol, obst_steps, succ, fail, ep_step, status, nconf, nent;
};

struct Lds {
    double* ptx; double* pty; double* dd; double* g; double* cg;
    int* flags; int* tmpi; int* tinfo; int* segend; int* brk;   // flags+tmpi adjacent (reused as n doubles)
    double* ped;      // [2P] positions
    double* trk;      // [CN_TF_COUNT][CN_MAX_TRACKS]
    double* cfx; double* cfy; double* cfd; int* cft; int* checked;   // confirmed objects
    double* cpv;      // [CN_MAX_TRACKS]
    double* tail;     // [7 + 4K]
    int* kidx;        // [K]
};

__device__ __forceinline__ double heading_to_goal(const CnKParams& p, const EnvRegs& e, double px, double py, double yaw)
{
    // ENV:222-237 (adds starting_point to the position; ENV:191-209 does not)
    double cx = px + p.start_x, cy = py + p.start_y;
    double ga = atan2(e.wpy - cy, e.wpx - cx);
    double h = ga - yaw;
    if (h > CN_PI) h -= 2 * CN_PI;
    else if (h < -CN_PI) h += 2 * CN_PI;
    return h;
}

__device__ __forceinline__ double dist3(double ax, double ay, double bx, double by)
{
    double dx = ax - bx, dy = ay - by;
    return sqrt(dx * dx + dy * dy + 0.0);  // np.linalg.norm of the 3-vector, ENV:191-197
}

__device__ __forceinline__ bool in_box(double x, double y, double gx, double gy, double eps)
{
    // ENV:1285-1319 half-open box
    double xp = gx + eps, xm = gx - eps, yp = gy + eps, ym = gy - eps;
    return (x <= xp) && (x > xm) && (y <= yp) && (y > ym);
}

// Ring (64-gon of radius r about (cx,cy), vertex k at angle -k*pi/32) against segment a->b.
// Lane k owns edge k.  Division-free membership test: for den != 0,
//   0 <= tn/den <= 1  <=>  (den > 0 ? 0 <= tn <= den : den <= tn <= 0)   (also for the rounded quotient)
//   0 <= un/den <  1  likewise with a strict upper bound.
// Returns the ballot of hit lanes; hit lanes get their intersection point.
__device__ __forceinline__ unsigned long long ring_segment(const CnKParams& p, int lane, double cx, double cy, double r,
                                                           double ax, double ay, double bx, double by,
                                                           double* hx, double* hy)
{
    int k2 = (lane + 1) & 63;
    double c0x = cx + r * p.poly_c[lane], c0y = cy + r * p.poly_s[lane];
    double c1x = cx + r * p.poly_c[k2], c1y = cy + r * p.poly_s[k2];
    double rx = bx - ax, ry = by - ay;
    double sx = c1x - c0x, sy = c1y - c0y;
    double den = rx * sy - ry * sx;
    double qx = c0x - ax, qy = c0y - ay;
    double tn = qx * sy - qy * sx;
    double un = qx * ry - qy * rx;
    bool hit = false;
    if (den > 0.0) hit = (tn >= 0.0) && (tn <= den) && (un >= 0.0) && (un < den);
    else if (den < 0.0) hit = (tn <= 0.0) && (tn >= den) && (un <= 0.0) && (un > den);
    if (hit) {
        double t = tn / den;
        double u = un / den;
        hit = (t >= 0.0) && (t <= 1.0) && (u >= 0.0) && (u < 1.0);  // the oracle's test, on the quotients
        *hx = ax + t * rx;
        *hy = ay + t * ry;
    }
    return __ballot(hit);
}

__device__ __forceinline__ double bcast_d(double v, int src)
{
    int lo = __shfl(__double2loint(v), src, 64), hi = __shfl(__double2hiint(v), src, 64);
    return __hiloint2double(hi, lo);
}

// UTL:296-314 get_local_goal_waypoints
__device__ __forceinline__ void waypoint_refresh(const CnKParams& p, EnvRegs& e, int lane, double px, double py)
{
    double hx = 0.0, hy = 0.0;
    unsigned long long m = ring_segment(p, lane, px, py, p.waypoint_radius, px, py, p.goal_x, p.goal_y, &hx, &hy);
    if (__popcll(m) == 1) {
        int src = __ffsll((long long)m) - 1;
        e.wpx = bcast_d(hx, src);
        e.wpy = bcast_d(hy, src);
    } else {
        e.wpx = -(p.goal_x + 0.0);  // UTL:310-312: x sign flipped
        e.wpy = p.goal_y + 0.0;
    }
}

// ---- simulator -------------------------------------------------------------------------------
__device__ void ped_advance(const CnKParams& p, int env, int lane, double* ped_p, double* ped_v, long long t0, long long t1)
{
    const double lo = -p.room_half + p.ped_radius, hi = p.room_half - p.ped_radius;
    const long long T = p.ped_cycle_ms;
    const long long gid = p.env_index_base + env;
    const double* preset = p.ped_preset + (size_t)env * 2 * p.P;
    for (int i = lane; i < p.P; i += 64) {
        double x = ped_p[2 * i], y = ped_p[2 * i + 1];
        double vx = ped_v[2 * i], vy = ped_v[2 * i + 1];
        long long offs = (long long)i * p.ped_stagger_ms;
        long long m = (t0 <= offs) ? 0 : (t0 - offs + T - 1) / T;
        long long a = offs + m * T;
        long long tc = t0;
        while (a < t1) {
            if (a > tc) {
                double ds = (double)(a - tc) / 1000.0;
                x = cn_clamp(fma(vx, ds, x), lo, hi);
                y = cn_clamp(fma(vy, ds, y), lo, hi);
                tc = a;
            }
            if (p.ped_mode == 0) {  // CROWD:101-102
                double u0 = cn_rng_u01(p.seed, gid, 1u, (uint32_t)i, (uint32_t)(2 * m));
                double u1 = cn_rng_u01(p.seed, gid, 1u, (uint32_t)i, (uint32_t)(2 * m + 1));
                vx = fma(2.0 * p.ped_vmax, u0, -p.ped_vmax);
                vy = fma(2.0 * p.ped_vmax, u1, -p.ped_vmax);
            } else {
                vx = preset[2 * i];
                vy = preset[2 * i + 1];
            }
            a += T;
            m += 1;
        }
        if (t1 > tc) {
            double ds = (double)(t1 - tc) / 1000.0;
            x = cn_clamp(fma(vx, ds, x), lo, hi);
            y = cn_clamp(fma(vy, ds, y), lo, hi);
        }
        ped_p[2 * i] = x; ped_p[2 * i + 1] = y;
        ped_v[2 * i] = vx; ped_v[2 * i + 1] = vy;
    }
}

// mid-point diff-drive step (turtlebot3_fake.cpp:156-162)
__device__ __forceinline__ void robot_advance(const CnKParams& p, EnvRegs& e, int ms)
{
    double dts = (double)ms / 1000.0;
    double ds = e.rv * dts, dth = e.rw * dts;
    double sn, cs;
    cn_det_sincos(fma(0.5, dth, e.ryaw), &sn, &cs);
    double lim = p.room_half - p.robot_clearance;
    e.rx = cn_clamp(fma(ds, cs, e.rx), -lim, lim);
    e.ry = cn_clamp(fma(ds, sn, e.ry), -lim, lim);
    double th = e.ryaw + dth;
    if (th > CN_PI) th -= 2.0 * CN_PI;
    else if (th <= -CN_PI) th += 2.0 * CN_PI;
    e.ryaw = th;
}

__device__ __forceinline__ void sim_advance(const CnKParams& p, EnvRegs& e, int env, int lane, double* ped_p, double* ped_v, int ms)
{
    if (ms <= 0) return;
    ped_advance(p, env, lane, ped_p, ped_v, e.crowd_ms, e.crowd_ms + ms);
    e.crowd_ms += ms;
    robot_advance(p, e, ms);
}

// ---- Env.get_state (ENV:245-1044) ----------------------------------------------------------------
// ped_p: pedestrian positions in LDS.  Writes the observation (float32 and optionally float64).
__device__ void observe(const CnKParams& p, EnvRegs& e, const Lds& L, int env, int lane, int step_counter,
                        float* obs32, float* fin32, double* obs64, int* done_out)
{
    const int R = p.R, n = R - 1, K = p.K, D = n + 7 + 4 * K;
    const double MAXR = p.max_scan_range;
    const double px = e.rx, py = e.ry, yaw = e.ryaw, v = e.rv, w = e.rw, now = e.clock;

    // ENV:246-265
    if (step_counter == 1) waypoint_refresh(p, e, lane, px, py);
    double distance_to_goal = cn_np_around(dist3(px, py, e.wpx, e.wpy), 100.0);
    double heading = cn_py_round(heading_to_goal(p, e, px, py, yaw), 100.0);
    if (step_counter % 5 == 0 || distance_to_goal < e.prev_dist) waypoint_refresh(p, e, lane, px, py);
    // ENV:267-268: the angular velocity is used as the angle
    double agent_vel_x = -1.0 * (v * cos(w));
    double agent_vel_y = v * sin(w);

    // ---- lidar raycast (XACRO:150-178) + UTL:375-392 sanitise + UTL:110-126 end points ----------
    double sy, cy;
    cn_det_sincos(yaw, &sy, &cy);
    const double ox = fma(p.lidar_offset_x, cy, px), oy = fma(p.lidar_offset_x, sy, py);
    const double h = p.room_half, rr = p.ped_radius * p.ped_radius;
    const double deg2rad = CN_PI / 180.0;
    double smin = 1e300;
    float* o32 = obs32 + (size_t)env * D;
    float* f32 = fin32 ? fin32 + (size_t)env * D : nullptr;
    double* o64 = obs64 ? obs64 + (size_t)env * D : nullptr;
    for (int k = lane; k < R; k += 64) {
        double lc = p.lidar_c[k], ls = p.lidar_s[k];
        double dx = fma(cy, lc, -(sy * ls));
        double dy = fma(sy, lc, cy * ls);
        double t = INFINITY;
        if (dx > 0.0) t = fmin(t, (h - ox) / dx);
        else if (dx < 0.0) t = fmin(t, (-h - ox) / dx);
        if (dy > 0.0) t = fmin(t, (h - oy) / dy);
        else if (dy < 0.0) t = fmin(t, (-h - oy) / dy);
        if (t < p.lidar_min) t = p.lidar_min;
        for (int j = 0; j < ((p.ablate & 1) ? 0 : p.P); ++j) {
            double ocx = L.ped[2 * j] - ox, ocy = L.ped[2 * j + 1] - oy;
            double b = fma(ocx, dx, ocy * dy);
            double cc = fma(ocx, ocx, fma(ocy, ocy, -rr));
            double disc = fma(b, b, -cc);
            if (disc >= 0.0) {
                double sq = sqrt(disc);
                double t2 = b + sq;
                if (t2 >= p.lidar_min) {
                    double t1 = fmax(b - sq, p.lidar_min);
                    t = fmin(t, t1);
                }
            }
        }
        if (k >= 1) {
            int j = R - 1 - k;  // UTL:389-390 reverse, drop last
            double r = (t > p.lidar_max) ? INFINITY : t;
            double sc;
            if (isinf(r)) sc = MAXR;
            else if (r == 0.0) sc = MAXR;
            else if (r > MAXR) sc = MAXR;
            else sc = r;
            smin = fmin(smin, sc);
            double ang = (double)j * p.angle_inc_deg;
            double a = ang * deg2rad - yaw;
            double sa, ca;
            cn_det_sincos(a, &sa, &ca);
            L.ptx[j] = cn_py_round(px + (sc * ca), 1000.0);
            L.pty[j] = cn_py_round(py + (sc * sa) * -1.0, 1000.0);
            L.dd[j] = cn_py_round(sc, 1000.0);
            double so = cn_np_around(sc, 1000.0);  // ENV:1042
            o32[j] = (float)so;
            if (f32) f32[j] = (float)so;
            if (o64) o64[j] = so;
            if (step_counter == 0) {  // ENV:287-290 ground-truth end points of an all-max scan
                L.g[j] = cn_py_round(px + (MAXR * ca), 1000.0);
                L.cg[j] = cn_py_round(py + (MAXR * sa) * -1.0, 1000.0);
            }
        }
    }
    smin = cn_wave_min_d(smin);
    __syncthreads();

    if (step_counter == 0) {  // UTL:405-419 + ENV:294
        double* hs = (double*)L.flags;  // flags+tmpi = 2n ints = n doubles
        for (int i = lane; i < n; i += 64) {
            int j = (i == n - 1) ? 0 : i + 1;
            hs[i] = hypot(L.g[i] - L.g[j], L.cg[i] - L.cg[j]);
        }
        __syncthreads();
        double sum = 0.0;
        for (int i = 0; i < n; ++i) sum += hs[i];  // Python sum(): left to right
        e.bb = sum / (double)n;
        double qx = cn_py_round(px, 1000.0), qy = cn_py_round(py, 1000.0);
        if (e.dq_len < 2) { if (e.dq_len == 0) { e.dq0x = qx; e.dq0y = qy; } else { e.dq1x = qx; e.dq1y = qy; } e.dq_len += 1; }
        else { e.dq0x = e.dq1x; e.dq0y = e.dq1y; e.dq1x = qx; e.dq1y = qy; }
        __syncthreads();
    }

    // ENV:329-346 gradients
    for (int i = lane; i < n; i += 64) {
        double gr = 0.0; int none = 1;
        if (L.dd[i] != 0.6) {
            int j = (i == n - 1) ? 0 : i + 1;
            double dy = L.pty[i] - L.pty[j];
            double q = (dy == 0) ? 0.0 : (L.ptx[i] - L.ptx[j]) / dy;
            gr = cn_py_round(q, 1000.0);
            none = 0;
        }
        L.g[i] = gr;
        L.flags[i] = none;
    }
    __syncthreads();
    // ENV:348-367 change of gradient (i < n-1), then the `last_grad` value for i == n-1
    int lastnn = -1;
    for (int i = lane; i < n - 1; i += 64) {
        int none = L.flags[i] | L.flags[i + 1];
        L.cg[i] = none ? 0.0 : fabs(L.g[i] - L.g[i + 1]);
        L.tmpi[i] = none;
        if (!L.flags[i]) lastnn = i;
    }
    lastnn = cn_wave_max_i(lastnn);
    __syncthreads();
    if (lane == 0) {
        int none = 1; double val = 0.0;
        if (!L.flags[n - 1] && lastnn >= 0) { none = L.tmpi[lastnn]; val = L.cg[lastnn]; }
        L.cg[n - 1] = none ? 0.0 : val;
        L.tmpi[n - 1] = none;
    }
    __syncthreads();
    // per-ray flag word for the type machine: bit0 change is None, bit1 change == 0, bit2 |c[i]-c[i+1]| == 0
    for (int i = lane; i < n; i += 64) {
        int cn0 = L.tmpi[i];
        int f = cn0 | ((!cn0 && L.cg[i] == 0) ? 2 : 0);
        if (i < n - 1 && !cn0 && !L.tmpi[i + 1] && fabs(L.cg[i] - L.cg[i + 1]) == 0) f |= 4;
        L.flags[i] = f;
        L.tinfo[i] = TY_NONE;
    }
    __syncthreads();
    // ENV:372-410 object-type state machine (loop-carried: last_type, du_count); uniform serial
    {
        int last_t = TY_NONE, last_s = 0, du = 0;
        int fi = L.flags[0];
        for (int i = 0; i < ((p.ablate & 2) ? 0 : n - 1); ++i) {
            int fn = L.flags[i + 1];
            if (!(fi & 1)) {
                int ty, src = i;
                if (fi & 2) { ty = TY_W; last_t = TY_W; last_s = i; }
                else {
                    ty = TY_O;
                    if (du != 1) {
                        bool nnone = fn & 1, nzero = fn & 2;
                        if (nzero) { ty = TY_W; last_t = TY_W; last_s = i; du = 0; }
                        if (nnone) { /* ENV:394-395 pass */ }
                        else if (fi & 4) { ty = TY_W; last_t = TY_W; last_s = i; du = 0; }
                        else { ty = last_t; src = last_s; du += 1; }
                    } else {
                        ty = TY_O; last_t = TY_O; last_s = i;
                        if (fn & 2) du = 0;
                    }
                }
                if (lane == 0) L.tinfo[i] = ty | (src << 2);
            }
            fi = fn;
        }
    }
    __syncthreads();
    // ENV:433-445: a typed ray carries the range and pose of the ray its list was created at.
    // Written out of place: Ad/Ax/Ay reuse the gradient arrays and the flag words (all dead now).
    double* Ad = L.g; double* Ax = L.cg; double* Ay = (double*)L.flags;
    {
        int q = 0;
        (void)q;
        for (int i = lane; i < n; i += 64) {
            int t = L.tinfo[i];
            int s = (t & 3) ? (t >> 2) : i;
            Ad[i] = L.dd[s]; Ax[i] = L.ptx[s]; Ay[i] = L.pty[s];
        }
    }
    __syncthreads();
    // ENV:448-485 association of consecutive rays; brk[i] = a segment closes after ray i
    int fe = n, lb = -1, nsegs0 = 0;
    for (int i = lane; i < n; i += 64) {
        int brk = 1;
        if (i < n - 1) brk = !(cn_iou3(Ax[i], Ay[i], Ax[i + 1], Ay[i + 1], e.bb) > 0.0);
        L.brk[i] = brk;
        if (brk) { fe = min(fe, i); nsegs0 += 1; if (i < n - 1) lb = max(lb, i); }
    }
    fe = cn_wave_min_i(fe);           // end of the first segment
    lb = cn_wave_max_i(lb);           // last break before ray n-1
    nsegs0 = cn_wave_sum_i(nsegs0);
    const int ls = lb + 1;            // start of the last segment
    // ENV:490-502 first <-> last with twice the box
    bool merge = (nsegs0 > 1) && (cn_iou3(Ax[0], Ay[0], Ax[n - 1], Ay[n - 1], e.bb * 2) > 0.0);
    __syncthreads();
    // order-space: position k -> ray.  merged: [0..fe] ++ [ls..n-1] ++ [fe+1..ls-1]
    const int nl = n - ls;  // length of the last segment
#define ORDER(k) (merge ? ((k) <= fe ? (k) : ((k) <= fe + nl ? ls + ((k) - fe - 1) : (k) - nl)) : (k))
    for (int k = lane; k < n; k += 64) {
        int ray = ORDER(k);
        int se;
        if (!merge) se = L.brk[ray];
        else if (k <= fe + nl) se = (k == fe + nl);
        else se = L.brk[ray];
        L.segend[k] = se;
    }
    __syncthreads();
    // ENV:508-566 split where free space (0.6) meets occupied; count segments
    int nseg = 0;
    for (int k = lane; k < n; k += 64) {   // each position only rewrites its own flag
        int se = L.segend[k];
        if (!se && k < n - 1) {
            int a06 = Ad[ORDER(k)] == 0.6, b06 = Ad[ORDER(k + 1)] == 0.6;
            if (a06 != b06) { se = 1; L.segend[k] = 1; }
        }
        nseg += se;
    }
    nseg = cn_wave_sum_i(nseg);
    __syncthreads();
    // ENV:568-620 confirmation; uniform serial accumulate over order-space
    int nconf = 0;
    {
        int k0 = 0, no = 0, nw = 0, nn = 0, occ = 0;
        for (int k = 0; k < ((p.ablate & 4) ? 0 : n); ++k) {
            int ray = ORDER(k);
            int t = L.tinfo[ray] & 3;
            no += (t == TY_O); nw += (t == TY_W); nn += (t == TY_NONE);
            occ |= (Ad[ray] != 0.6);
            if (L.segend[k]) {
                int len = k - k0 + 1;
                if (occ && len >= 4) {
                    int m = ORDER(k0 + len / 2);  // ENV:577 Python-2 integer division
                    double dm = Ad[m];
                    int est = 3 + (int)floor(29 * (p.max_scan_range - dm) / (p.max_scan_range - p.min_scan_range));
                    int mn = len < est ? len : est;
                    double score = (double)no / (double)mn;
                    int kinds = (no > 0) + (nw > 0) + (nn > 0);
                    int obj = -1;
                    if (kinds > 1) {
                        if (score >= 0.5) obj = (no > nw) ? TY_O : TY_W;
                        else if (len <= est) obj = (no > nw) ? TY_O : TY_W;
                        else obj = TY_W;
                    } else {
                        int lim = nseg < est ? nseg : est;  // ENV:608,615
                        if (len > lim) obj = (nw > 0) ? TY_W : TY_O;
                    }
                    if (obj >= 0) {
                        if (nconf < p.max_conf) {
                            if (lane == 0) { L.cft[nconf] = obj; L.cfx[nconf] = Ax[m]; L.cfy[nconf] = Ay[m]; L.cfd[nconf] = dm; }
                            ++nconf;
                        } else e.status |= CN_ST_CONF_OVERFLOW;
                    }
                }
                k0 = k + 1; no = nw = nn = occ = 0;
            }
        }
    }
#undef ORDER
    e.nconf = nconf;
    __syncthreads();

    // ENV:637-654
    int n_obst = 0, ego_hit = 0;
    for (int j = lane; j < nconf; j += 64) {
        if (L.cft[j] == TY_O) { n_obst += 1; if (L.cfd[j] < 0.140) ego_hit = 1; }
    }
    n_obst = cn_wave_sum_i(n_obst);
    ego_hit = cn_wave_max_i(ego_hit);
    if (n_obst > 0) e.obst_steps += 1;

    // ---- ENV:656-743 tracker -----------------------------------------------------------------------
    double* T = L.trk;
#define TRK(f, i) T[(f) * CN_MAX_TRACKS + (i)]
    bool add_unchecked = false;
    if (p.ablate & 16) { e.ntracks = 0; nconf = 0; }
    if (e.ntracks == 0) {
        for (int j = lane; j < nconf; j += 64) L.checked[j] = 0;
        add_unchecked = true;  // every 'o' object becomes a track
    } else {
        const int nt0 = e.ntracks;
        if (lane < nt0 && TRK(CN_TF_DQLEN, lane) > 1.0) {  // ENV:678-680 popleft
            TRK(CN_TF_D0X, lane) = TRK(CN_TF_D1X, lane); TRK(CN_TF_D0Y, lane) = TRK(CN_TF_D1Y, lane);
            TRK(CN_TF_DQLEN, lane) = 1.0;
        }
        for (int j = lane; j < nconf; j += 64) L.checked[j] = 0;
        __syncthreads();
        if (nconf == 0) {
            e.ntracks = 0;  // ENV:683-686 nets out to clearing every track
        } else {
            unsigned long long alive = 0ull;
            int cur = nt0;
            for (int i = 0; i < nt0; ++i) {
                double tx = TRK(CN_TF_PX, i), ty_ = TRK(CN_TF_PY, i);
                double best = -1.0; int bj = 0x7fffffff;
                for (int j = lane; j < nconf; j += 64) {  // ENV:688-689 (walls included)
                    double u = cn_iou3(tx, ty_, L.cfx[j], L.cfy[j], 0.0505);
                    if (u > best) { best = u; bj = j; }
                }
                // wave arg-max, first maximum wins (list.index(max))
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    double ob = cn_shfl_xor_d(best, m);
                    int oj = __shfl_xor(bj, m, 64);
                    if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
                }
                if (best > 0.0) {  // ENV:702-712
                    alive |= (1ull << i);
                    if (lane == 0) {
                        double cxj = L.cfx[bj], cyj = L.cfy[bj];
                        TRK(CN_TF_PX, i) = cxj; TRK(CN_TF_PY, i) = cyj; TRK(CN_TF_DIST, i) = L.cfd[bj];
                        if (TRK(CN_TF_DQLEN, i) < 1.5) { TRK(CN_TF_D1X, i) = cxj; TRK(CN_TF_D1Y, i) = cyj; TRK(CN_TF_DQLEN, i) = 2.0; }
                        TRK(CN_TF_T, i) = now - TRK(CN_TF_T, i);
                        L.checked[bj] = 1;
                    }
                } else if (cur > i) {  // ENV:715-717
                    cur -= 1;
                } else {
                    alive |= (1ull << i);
                }
            }
            __syncthreads();
            // compact the survivors, order preserved
            double rec[CN_TF_COUNT];
            bool mine = (lane < nt0) && ((alive >> lane) & 1ull);
            if (mine) {
#pragma unroll
                for (int f = 0; f < CN_TF_COUNT; ++f) rec[f] = TRK(f, lane);
            }
            __syncthreads();
            if (mine) {
                int slot = __popcll(alive & ((1ull << lane) - 1ull));
#pragma unroll
                for (int f = 0; f < CN_TF_COUNT; ++f) TRK(f, slot) = rec[f];
            }
            e.ntracks = __popcll(alive);
            add_unchecked = true;  // ENV:723-743
        }
    }
    __syncthreads();
    if (add_unchecked) {
        for (int j0 = 0; j0 < nconf; j0 += 64) {
            int j = j0 + lane;
            bool want = (j < nconf) && !L.checked[j] && (L.cft[j] == TY_O);
            unsigned long long m = __ballot(want);
            int slot = e.ntracks + __popcll(m & ((1ull << lane) - 1ull));
            if (want) {
                if (slot < CN_MAX_TRACKS) {
                    double cxj = L.cfx[j], cyj = L.cfy[j];
                    TRK(CN_TF_PX, slot) = cxj; TRK(CN_TF_PY, slot) = cyj; TRK(CN_TF_DIST, slot) = L.cfd[j];
                    TRK(CN_TF_D0X, slot) = cxj; TRK(CN_TF_D0Y, slot) = cyj; TRK(CN_TF_D1X, slot) = 0.0; TRK(CN_TF_D1Y, slot) = 0.0;
                    TRK(CN_TF_T, slot) = now; TRK(CN_TF_SPEED, slot) = -1.0;
                    TRK(CN_TF_VX, slot) = 0.0; TRK(CN_TF_VY, slot) = 0.0; TRK(CN_TF_DQLEN, slot) = 1.0;
                }
            }
            int total = e.ntracks + __popcll(m);
            if (total > CN_MAX_TRACKS) { e.status |= CN_ST_TRACK_OVERFLOW; total = CN_MAX_TRACKS; }
            e.ntracks = total;
        }
    }
    __syncthreads();
    // ENV:745-760 speed of the tracks matched in this call
    if (lane < e.ntracks && TRK(CN_TF_DQLEN, lane) > 1.5) {
        double dc = hypot(TRK(CN_TF_D0Y, lane) - TRK(CN_TF_D1Y, lane), TRK(CN_TF_D0X, lane) - TRK(CN_TF_D1X, lane));
        TRK(CN_TF_SPEED, lane) = dc / TRK(CN_TF_T, lane);
    }
    __syncthreads();

    // default K x [px, py, 0, 0] (ENV:273)
    for (int i = lane; i < 4 * K; i += 64) {
        int c = i & 3;
        L.tail[7 + i] = (c == 0) ? px : (c == 1 ? py : 0.0);
    }
    if (lane < K) L.kidx[lane] = -1;
    e.nent = 0;
    __syncthreads();

    // ---- ENV:769-996 collision cone / collision probability / top-K -------------------------------
    if (e.dq_len == 2 && !(p.ablate & 8)) {
        const double ts = e.ts;
        if (ts == 0.0) e.status |= CN_ST_DT_ZERO;
        const int nt = e.ntracks;
        double vx_ = (e.dq1x - e.dq0x) / ts, vy_ = (e.dq1y - e.dq0y) / ts;  // UTL:227-236
        double agent_vel = sqrt(vx_ * vx_ + vy_ * vy_);
        double obstacle_vel = (nt == 0) ? 0.0 : TRK(CN_TF_SPEED, 0);  // ENV:787-793
        // ENV:800-815: per-track velocity; the relative-motion end point of the LAST track survives
        if (lane < nt && TRK(CN_TF_DQLEN, lane) > 1.5) {
            double chx = TRK(CN_TF_D0X, lane) - TRK(CN_TF_D1X, lane), chy = TRK(CN_TF_D0Y, lane) - TRK(CN_TF_D1Y, lane);
            TRK(CN_TF_VX, lane) = chx / ts; TRK(CN_TF_VY, lane) = chy / ts;
        }
        double vo_x = e.dq1x, vo_y = e.dq1y;
        if (nt > 0) {
            int l = nt - 1;
            double chx = 0.0, chy = 0.0;
            if (TRK(CN_TF_DQLEN, l) > 1.5) { chx = TRK(CN_TF_D0X, l) - TRK(CN_TF_D1X, l); chy = TRK(CN_TF_D0Y, l) - TRK(CN_TF_D1Y, l); }
            vo_x = e.dq1x + chx; vo_y = e.dq1y + chy;
        }
        __syncthreads();
        // UTL:251-293 collision point per track; lanes = the 64 ring edges
        const double a0x = e.dq0x, a0y = e.dq0y;
        double gradient = (vo_y == 0.0) ? 0.0 : (vo_x - a0x) / vo_y - a0y;  // UTL:261 precedence as written
        double bb0 = a0x - (gradient * a0y);
        int hi = (int)ceil(a0x + 3.5), lo = (int)floor(a0x - 3.5);
        double ego_prev = 0.0, ego_max = 0.0;
        for (int i = 0; i < nt; ++i) {  // ENV:818-860
            double tx = TRK(CN_TF_PX, i), ty_ = TRK(CN_TF_PY, i), td = TRK(CN_TF_DIST, i);
            int has = 0; double dcp = 0.0;
            for (int x2 = hi; x2 > lo; --x2) {
                double y2 = ((double)x2 * gradient) + bb0;
                double hx = 0.0, hy = 0.0;
                unsigned long long m = ring_segment(p, lane, tx, ty_, 0.178, a0x, a0y, (double)x2, y2, &hx, &hy);
                int cnt = __popcll(m);
                if (cnt == 0) continue;
                if (cnt == 1) break;  // Point has no .geoms -> None
                int l1 = __ffsll((long long)m) - 1;
                unsigned long long m2 = m & (m - 1ull);
                int l2 = __ffsll((long long)m2) - 1;
                double d1 = hypot(a0x - bcast_d(hx, l1), a0y - bcast_d(hy, l1));
                double d2 = hypot(a0x - bcast_d(hx, l2), a0y - bcast_d(hy, l2));
                dcp = fmin(d1, d2); has = 1;
                break;
            }
            double rv = agent_vel - obstacle_vel;
            double gcp = (td > p.max_scan_range) ? 0.0 : (p.max_scan_range - td) / (p.max_scan_range - p.min_scan_range);
            double ego, cpv;
            if (has) {
                if (rv == 0) { cpv = 1.0 * gcp; ego = ego_prev; }
                else {
                    double ttc = dcp / rv;
                    if (ttc == 0.0) { e.status |= CN_ST_TTC_ZERO; ego = 1.0; }
                    else ego = fmin(1.0, 0.15 / ttc);  // UTL:319
                    cpv = 0.5 * ego + 0.5 * gcp;
                }
            } else { ego = 0.0; cpv = 0.5 * 0.0 + 0.5 * gcp; }
            ego_prev = ego;
            if (lane == 0) L.cpv[i] = cpv;
            if (i == 0 || ego > ego_max) ego_max = ego;
        }
        e.nent = nt;
        __syncthreads();
        if (nt == 0) { e.cprob = 0.0; e.ego = 0.0; }  // ENV:862-876
        else {  // ENV:878-905: stable descending sort, keep the LAST K
            e.ego = ego_max;
            int first = nt > K ? nt - K : 0;
            int rank = -1; double mycp = 0.0;
            if (lane < nt) {
                mycp = L.cpv[lane];
                rank = 0;
                for (int j = 0; j < nt; ++j) {
                    double c = L.cpv[j];
                    rank += (c > mycp) || (c == mycp && j < lane);
                }
                if (rank >= first) {
                    int kk = rank - first;
                    L.tail[7 + 4 * kk + 0] = TRK(CN_TF_PX, lane); L.tail[7 + 4 * kk + 1] = TRK(CN_TF_PY, lane);
                    L.tail[7 + 4 * kk + 2] = TRK(CN_TF_VX, lane); L.tail[7 + 4 * kk + 3] = TRK(CN_TF_VY, lane);
                    L.kidx[kk] = lane;
                }
            }
            unsigned long long mf = __ballot(rank == first);
            e.cprob = bcast_d(mycp, __ffsll((long long)mf) - 1);
        }
        // ENV:990-996
        e.dq0x = e.dq1x; e.dq0y = e.dq1y; e.dq_len = 1;
        if (lane < nt) TRK(CN_TF_T, lane) = now;
    }
#undef TRK
    // ENV:998-1005 safety counters
    if (ego_hit) e.ego_viol += 1;
    if (e.ego > 0.4) e.social_viol += 1;
    // ENV:1011-1023 done
    if (!e.done) {
        if (smin < p.min_scan_range) e.done = 1;
        if (in_box(px, py, p.goal_x, p.goal_y, p.goal_eps)) e.done = 1;
        if (step_counter >= p.max_steps) e.done = 1;
    }
    // ENV:1025-1042 observation tail
    if (lane == 0) {
        L.tail[0] = heading; L.tail[1] = distance_to_goal;
        L.tail[2] = cn_py_round(px, 1000.0); L.tail[3] = cn_py_round(py, 1000.0);
        L.tail[4] = cn_py_round(yaw, 1000.0);
        L.tail[5] = cn_py_round(agent_vel_x, 1000.0); L.tail[6] = cn_py_round(agent_vel_y, 1000.0);
    }
    __syncthreads();
    for (int i = lane; i < 7 + 4 * K; i += 64) {
        double so = cn_np_around(L.tail[i], 1000.0);
        L.tail[i] = so;
        o32[n + i] = (float)so;
        if (f32) f32[n + i] = (float)so;
        if (o64) o64[n + i] = so;
    }
    __syncthreads();
    *done_out = e.done;
}

// ENV:1046-1162 compute_reward; state[n] = heading, state[n+1] = distance are in L.tail[0..1]
__device__ double compute_reward(const CnKParams& p, EnvRegs& e, const Lds& L, int lane, int done)
{
    double cur_head = L.tail[0], cur_dist = L.tail[1];
    double dd = cur_dist - e.prev_dist, hd = cur_head - e.prev_head;
    int htg = 0, dtg = 0, wp = 0;
    if (dd < 0) dtg = 1;
    double ph = e.prev_head;
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
    if (in_box(e.rx, e.ry, e.wpx, e.wpy, p.goal_eps)) {  // ENV:1109-1125
        waypoint_refresh(p, e, lane, e.rx, e.ry);
        wp = 200;
        if (in_box(e.wpx, e.wpy, p.goal_x, p.goal_y, p.goal_eps)) { e.wpx = p.goal_x; e.wpy = p.goal_y; }
    }
    double reward = (double)(-2 + dtg + htg + wp);
    e.prev_dist = cur_dist;
    e.prev_head = cur_head;
    if (done) {
        if (in_box(e.rx, e.ry, p.goal_x, p.goal_y, p.goal_eps)) { e.fail = 0; e.succ = 1; reward = 200 + reward; }
        else { e.fail = 1; e.succ = 0; reward = -200 + reward; }
    }
    return reward;
}

}  // namespace

extern "C" __global__ void __launch_bounds__(64) cn_env_kernel(CnKParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int env = blockIdx.x, lane = threadIdx.x;
    if (env >= p.N) return;
    if (p.mode == CN_MODE_RESET && p.mask && !p.mask[env]) return;
    const int R = p.R, n = R - 1, P = p.P, K = p.K;

    Lds L;
    {
        char* q = smem;
        L.ptx = (double*)q; q += 8 * (size_t)n;
        L.pty = (double*)q; q += 8 * (size_t)n;
        L.dd = (double*)q; q += 8 * (size_t)n;
        L.g = (double*)q; q += 8 * (size_t)n;
        L.cg = (double*)q; q += 8 * (size_t)n;
        L.ped = (double*)q; q += 8 * (size_t)(2 * P + 2);
        L.trk = (double*)q; q += 8 * (size_t)(CN_TF_COUNT * CN_MAX_TRACKS);
        L.cfx = (double*)q; q += 8 * (size_t)p.max_conf;
        L.cfy = (double*)q; q += 8 * (size_t)p.max_conf;
        L.cfd = (double*)q; q += 8 * (size_t)p.max_conf;
        L.cpv = (double*)q; q += 8 * (size_t)CN_MAX_TRACKS;
        L.tail = (double*)q; q += 8 * (size_t)(7 + 4 * K + 1);
        L.flags = (int*)q; q += 4 * (size_t)n;   // flags and tmpi must stay adjacent and 8-byte aligned (reused as n doubles)
        L.tmpi = (int*)q; q += 4 * (size_t)n;
        L.tinfo = (int*)q; q += 4 * (size_t)n;
        L.segend = (int*)q; q += 4 * (size_t)n;
        L.brk = (int*)q; q += 4 * (size_t)n;
        L.cft = (int*)q; q += 4 * (size_t)p.max_conf;
        L.checked = (int*)q; q += 4 * (size_t)p.max_conf;
        L.kidx = (int*)q; q += 4 * (size_t)CN_MAX_K;
    }

    // ---- load env state -------------------------------------------------------------------------
    double* sd = p.sd + (size_t)env * CN_SD_COUNT;
    int* si = p.si + (size_t)env * CN_SI_COUNT;
    EnvRegs e;
    e.rx = sd[CN_SD_RX]; e.ry = sd[CN_SD_RY]; e.ryaw = sd[CN_SD_RYAW]; e.rv = sd[CN_SD_RV]; e.rw = sd[CN_SD_RW];
    e.clock = sd[CN_SD_CLOCK]; e.wpx = sd[CN_SD_WPX]; e.wpy = sd[CN_SD_WPY];
    e.prev_dist = sd[CN_SD_PREV_DIST]; e.prev_head = sd[CN_SD_PREV_HEAD];
    e.dq0x = sd[CN_SD_DQ0X]; e.dq0y = sd[CN_SD_DQ0Y]; e.dq1x = sd[CN_SD_DQ1X]; e.dq1y = sd[CN_SD_DQ1Y];
    e.ts = sd[CN_SD_TS]; e.bb = sd[CN_SD_BB]; e.ego = sd[CN_SD_EGO]; e.cprob = sd[CN_SD_CPROB];
    e.ep_ret = sd[CN_SD_EP_RETURN]; e.last_ret = sd[CN_SD_LAST_RETURN];
    e.done = si[CN_SI_DONE]; e.dq_len = si[CN_SI_DQ_LEN]; e.ntracks = si[CN_SI_NTRACKS];
    e.ego_viol = si[CN_SI_EGO_VIOL]; e.social_viol = si[CN_SI_SOCIAL_VIOL]; e.obst_steps = si[CN_SI_OBST_STEPS];
    e.succ = si[CN_SI_SUCCESS]; e.fail = si[CN_SI_FAILURE]; e.ep_step = si[CN_SI_EP_STEP]; e.status = si[CN_SI_STATUS];
    e.nconf = si[CN_SI_NCONF]; e.nent = si[CN_SI_NENTRIES];
    e.crowd_ms = (long long)(((unsigned long long)(unsigned)si[CN_SI_CROWD_HI] << 32) | (unsigned)si[CN_SI_CROWD_LO]);

    double* gped_p = p.ped_p + (size_t)env * 2 * P;
    double* gped_v = p.ped_v + (size_t)env * 2 * P;
    const double* gped_init = p.ped_init + (size_t)env * 2 * P;
    double* pedv = L.g;  // velocities are only needed while advancing; L.g is free until the gradients
    for (int i = lane; i < 2 * P; i += 64) { L.ped[i] = gped_p[i]; pedv[i] = gped_v[i]; }
    double* gtrk = p.trk + (size_t)env * CN_TF_COUNT * CN_MAX_TRACKS;
    if (lane < e.ntracks) {
#pragma unroll
        for (int f = 0; f < CN_TF_COUNT; ++f) L.trk[f * CN_MAX_TRACKS + lane] = gtrk[f * CN_MAX_TRACKS + lane];
    }
    __syncthreads();

    int done = 0;
    bool need_reset = (p.mode == CN_MODE_RESET);
    if (p.mode == CN_MODE_STEP) {
        // ---- Env.step (ENV:1164-1225), continuous mode ---------------------------------------------
        e.ep_step += 1;
        const int sc = p.step_counter ? p.step_counter[env] : e.ep_step;
        const double v = (double)p.action[2 * env], w = (double)p.action[2 * env + 1];
        const double t0 = e.clock;
        e.rv = v; e.rw = w;                                   // pub_cmd_vel.publish (ENV:1200)
        e.clock += (double)p.dt_ms / 1000.0;                  // time.sleep(0.15) (ENV:1201)
        sim_advance(p, e, env, lane, L.ped, pedv, p.dt_ms);
        const double end_timestep = e.clock - t0;             // ENV:1202
        {
            double qx = cn_py_round(e.rx, 1000.0), qy = cn_py_round(e.ry, 1000.0);  // ENV:1208
            if (e.dq_len == 0) { e.dq0x = qx; e.dq0y = qy; e.dq_len = 1; }
            else if (e.dq_len == 1) { e.dq1x = qx; e.dq1y = qy; e.dq_len = 2; }
            else { e.dq0x = e.dq1x; e.dq0y = e.dq1y; e.dq1x = qx; e.dq1y = qy; }
        }
        e.ts = end_timestep;                                  // ENV:1209
        e.clock += (double)p.scan_latency_ms / 1000.0;        // wait_for_message('scan') (ENV:1218)
        sim_advance(p, e, env, lane, L.ped, pedv, p.scan_latency_ms);
        __syncthreads();
        for (int i = lane; i < 2 * P; i += 64) gped_v[i] = pedv[i];  // L.g is about to be reused
        __syncthreads();
        observe(p, e, L, env, lane, sc, p.obs, p.final_obs, p.obs_f64, &done);
        double r = compute_reward(p, e, L, lane, done);
        e.ep_ret += r;
        if (lane == 0) {
            p.reward[env] = (float)r;
            p.done[env] = (uint8_t)done;
        }
        if (p.topk_idx && lane < K) p.topk_idx[(size_t)env * K + lane] = L.kidx[lane];
        if (done) {
            e.rv = 0.0; e.rw = 0.0;                           // pub_cmd_vel.publish(Twist()) (ENV:1160)
            e.last_ret = e.ep_ret;
            need_reset = (p.auto_reset != 0);
        }
        __syncthreads();
    }
    if (need_reset) {
        // ---- Env.reset (ENV:1227-1263) + TRAIN:114-116 -----------------------------------------------
        // gazebo/reset_simulation: poses back to their initial values, twists zeroed (crowd clock keeps running)
        e.rx = p.spawn_x; e.ry = p.spawn_y; e.ryaw = p.spawn_yaw; e.rv = 0.0; e.rw = 0.0;
        for (int i = lane; i < 2 * P; i += 64) { L.ped[i] = gped_init[i]; pedv[i] = 0.0; }
        __syncthreads();
        e.clock += (double)p.scan_latency_ms / 1000.0;        // wait_for_message('scan') (ENV:1238)
        sim_advance(p, e, env, lane, L.ped, pedv, p.scan_latency_ms);
        __syncthreads();
        // the settle interval needs the velocities again after observe() has reused L.g: park them in HBM
        for (int i = lane; i < 2 * P; i += 64) gped_v[i] = pedv[i];
        e.prev_dist = dist3(e.rx, e.ry, e.wpx, e.wpy);        // ENV:1243 (unrounded)
        e.prev_head = heading_to_goal(p, e, e.rx, e.ry, e.ryaw);  // ENV:1244
        __syncthreads();
        int d2 = 0;
        observe(p, e, L, env, lane, 0, p.obs, nullptr, p.obs_f64, &d2);
        e.social_viol = 0; e.ego_viol = 0; e.obst_steps = 0;  // ENV:1260-1262
        __syncthreads();
        for (int i = lane; i < 2 * P; i += 64) pedv[i] = gped_v[i];
        __syncthreads();
        e.clock += (double)p.settle_ms / 1000.0;              // TRAIN:114 time.sleep(0.1)
        sim_advance(p, e, env, lane, L.ped, pedv, p.settle_ms);
        e.done = 0;                                           // TRAIN:116
        e.ep_step = 0; e.ep_ret = 0.0;
        __syncthreads();
        for (int i = lane; i < 2 * P; i += 64) gped_v[i] = pedv[i];
    }

    // ---- write env state back ---------------------------------------------------------------------
    for (int i = lane; i < 2 * P; i += 64) gped_p[i] = L.ped[i];
    if (lane < e.ntracks) {
#pragma unroll
        for (int f = 0; f < CN_TF_COUNT; ++f) gtrk[f * CN_MAX_TRACKS + lane] = L.trk[f * CN_MAX_TRACKS + lane];
    }
    if (lane == 0) {
        sd[CN_SD_RX] = e.rx; sd[CN_SD_RY] = e.ry; sd[CN_SD_RYAW] = e.ryaw; sd[CN_SD_RV] = e.rv; sd[CN_SD_RW] = e.rw;
        sd[CN_SD_CLOCK] = e.clock; sd[CN_SD_WPX] = e.wpx; sd[CN_SD_WPY] = e.wpy;
        sd[CN_SD_PREV_DIST] = e.prev_dist; sd[CN_SD_PREV_HEAD] = e.prev_head;
        sd[CN_SD_DQ0X] = e.dq0x; sd[CN_SD_DQ0Y] = e.dq0y; sd[CN_SD_DQ1X] = e.dq1x; sd[CN_SD_DQ1Y] = e.dq1y;
        sd[CN_SD_TS] = e.ts; sd[CN_SD_BB] = e.bb; sd[CN_SD_EGO] = e.ego; sd[CN_SD_CPROB] = e.cprob;
        sd[CN_SD_EP_RETURN] = e.ep_ret; sd[CN_SD_LAST_RETURN] = e.last_ret;
        si[CN_SI_DONE] = e.done; si[CN_SI_DQ_LEN] = e.dq_len; si[CN_SI_NTRACKS] = e.ntracks;
        si[CN_SI_EGO_VIOL] = e.ego_viol; si[CN_SI_SOCIAL_VIOL] = e.social_viol; si[CN_SI_OBST_STEPS] = e.obst_steps;
        si[CN_SI_SUCCESS] = e.succ; si[CN_SI_FAILURE] = e.fail; si[CN_SI_EP_STEP] = e.ep_step; si[CN_SI_STATUS] = e.status;
        si[CN_SI_NCONF] = e.nconf; si[CN_SI_NENTRIES] = e.nent;
        si[CN_SI_CROWD_LO] = (int)(unsigned)((unsigned long long)e.crowd_ms & 0xffffffffull);
        si[CN_SI_CROWD_HI] = (int)(unsigned)((unsigned long long)e.crowd_ms >> 32);
    }
}

// float32 views of the per-env returns (for the RCCL all-gather of episode returns) and counters
extern "C" __global__ void cn_gather_kernel(CnKParams p, float* last_ret, float* run_ret, int32_t* counters)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.N) return;
    const double* sd = p.sd + (size_t)i * CN_SD_COUNT;
    const int* si = p.si + (size_t)i * CN_SI_COUNT;
    if (last_ret) last_ret[i] = (float)sd[CN_SD_LAST_RETURN];
    if (run_ret) run_ret[i] = (float)sd[CN_SD_EP_RETURN];
    if (counters) {
        int32_t* c = counters + (size_t)i * 8;
        c[0] = si[CN_SI_EGO_VIOL]; c[1] = si[CN_SI_SOCIAL_VIOL]; c[2] = si[CN_SI_OBST_STEPS]; c[3] = si[CN_SI_EP_STEP];
        c[4] = si[CN_SI_SUCCESS]; c[5] = si[CN_SI_FAILURE]; c[6] = si[CN_SI_STATUS]; c[7] = si[CN_SI_NTRACKS];
    }
}
