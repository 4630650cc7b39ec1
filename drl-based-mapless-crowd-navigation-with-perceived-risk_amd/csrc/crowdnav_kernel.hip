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
// simulator here is defined in DESIGN.md and restated independently by the CPU oracle (test infrastructure).
#include "crowdnav_device.h"
#include "crowdnav_kernel.h"

// One wavefront per workgroup: the LDS processes a wave's DS instructions in issue order, so a
// cross-lane hand-off through LDS needs no s_barrier and no vmcnt/lgkmcnt drain -- only the compiler
// must not reorder the accesses.  (A real __syncthreads() here also waits for every outstanding
// global store, which put ~30 memory-latency stalls on each env-step's critical path.)
#define CN_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                       __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// Stage time stamps (PROFILING BUILD ONLY: build.sh timing -> libcrowdnav_timing.so; tools/stage_timing.py)
// ---- issue arbitration between the wavefronts of a SIMD (tools/wave_fairness.py, profiles/r03/arbitration.txt) ------------
// A SIMD's arbiter serves its OLDEST wavefront first.  With four environments per SIMD that start together (one launch of 4096
// environments) their lifetimes come out 62.7k / 69.3k / 77.9k / 88.1k ticks by entry rank: the launch lasts as long as the
// wave that was starved, and that wave runs its last quarter almost alone on a SIMD that is then mostly idle.
//  * FAIR kernels (cn_env_kernel_fair; cn_handle arbitration CN_ARB_FAIR) lower a wave's s_setprio level as it advances --
//    3 until the gradients are done, 2 until the association, 1 until the tracker, 0 for the cone / reward / write-back -- so
//    the wave that is BEHIND gets the issue slots and the four finish together (71.4k / 75.0k / 78.7k / 82.6k): one launch
//    per step 86.6 -> 93.5 M env-steps/s.  It costs throughput when launches OVERLAP on a SIMD (4 stream groups: 108 -> 102 M;
//    the oldest-first order is the better pipeline there), hence a per-handle switch and not the default for small launches.
//  * the sequence kernels rotate the levels over the four wave slots every control period (slot + t) & 3: every wave spends a
//    quarter of its steps at each level, so all four finish their T steps together instead of 0 / 9 / 20 / 29 % apart:
//    cn_step_sequence 99.5 -> 112 M at T = 1000, 92.9 -> 102 M at T = 20.
// Priorities change WHEN a wave's instructions issue, never what they compute: every parity test runs on both.
__device__ __forceinline__ void cn_setprio_uniform(int v)      // v: wave-uniform, 0..3 (s_setprio takes an immediate)
{
    if (v == 0) __builtin_amdgcn_s_setprio(0); else if (v == 1) __builtin_amdgcn_s_setprio(1);
    else if (v == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
}
// at stage stamp k (a literal; FAIR is a template parameter of the enclosing function): the four level changes of a FAIR kernel
// (SFENCE, a constexpr of the enclosing function: the oldest-first 720-ray kernel gets a scheduling fence where its fair sibling has an
//  s_setprio -- the only difference between the two bodies, and the sibling allocates with 13 scalar spills where this one had 66)
#define CN_FAIR_AT(k) do { if constexpr (FAIR) { if ((k) == 0) __builtin_amdgcn_s_setprio(3); else if ((k) == 7) __builtin_amdgcn_s_setprio(2); \
                           else if ((k) == 11) __builtin_amdgcn_s_setprio(1); else if ((k) == 15) __builtin_amdgcn_s_setprio(0); } \
                           else if constexpr (SFENCE) { if ((k) == 0 || (k) == 7 || (k) == 11 || (k) == 15) __builtin_amdgcn_s_setprio(0); } } while (0)
#ifdef CN_TIMING
#define CN_ABLATE(bit) (p->ablate & (bit))   /* stage-skipping mask of tools/ablate.py: timing build only */
/* bits 8..15 of the mask: stamp number + 1 at which every wavefront ENDS (tools/stage_instr.py: the PMC counters of a launch cut
   at stamp k are the instructions issued up to k, and the differences between consecutive cuts the dynamic ledger of the stages);
   bit 16: only the wavefronts of environments with (env & 3) != 0 end there (tools/pack_bound_probe.py: an upper bound for what
   sharing the narrow stages between the four environments of a workgroup could buy) */
#define CN_T(k) do { CN_FAIR_AT(k); if (p->timing && lane == 0) p->timing[(size_t)env * 32 + (k)] = (long long)__builtin_amdgcn_s_memtime(); \
                     if (((p->ablate >> 8) & 0xff) == (k) + 1 && (!((p->ablate >> 16) & 1) || (env & 3) != 0)) __builtin_amdgcn_endpgm(); } while (0)
#else
#define CN_ABLATE(bit) 0
#define CN_T(k) CN_FAIR_AT(k)
#endif

// TWO WAVEFRONTS PER ENVIRONMENT (cn_env_kernel_s360_x2; small grids: at most two wavefronts per SIMD would be resident anyway).
// Wave 0 runs the step as always; wave 1 takes the pedestrians' advance off its hands while it does the robot's, then half of every
// lane = ray stage (ray loop, gradients, flag words, association).  They meet at s_barrier with only the LDS counter drained (the
// hand-offs are all through LDS; a __syncthreads() would also wait for the observation's global stores).  `mb`: the pair's mailbox,
// 256 bytes of LDS behind the working set.
#define CN_XBAR() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)
struct XMail { double px, py, sy, cy, ox, oy, smin1, pad; int nnear, wall_x, wall_y, nocc, fast_assoc, k1, fe1, lb1, ns1; };
#define TY_NONE 0
#define TY_W 1
#define TY_O 2
#define CN_NAN __longlong_as_double(0x7ff8000000000000LL)
typedef unsigned long long u64;
enum { M_NONE = 0, M_ZERO, M_EQ, M_NNONE, M_NZERO, M_ISW, M_ISO, M_ALIAS, M_BRK, M_SEG, M_KW, M_KO, M_OCC, M_COUNT };

namespace {
// The kernel parameters (~110 dwords) are read from the KERNARG SEGMENT where they are needed, through a constant-address-
// space pointer, instead of through the by-value kernel argument: the compiler preloads a by-value argument into SGPRs at
// entry and, with 106 SGPRs per wave, spills most of it into VGPR lanes -- 1037 v_readlane reloads, a sixth of the kernel's
// VALU instructions and a third of the ray loop.  Read in place they are scalar loads next to their uses.
typedef const __attribute__((address_space(4))) CnKParams* KP;
#define PY2 (&p->py2_round)      /* cn_config.py2_round, dereferenced only inside a rounding's exact-tie branch (crowdnav_device.h) */

struct EnvRegs {  // per-env scalars, uniform across the wave
    double rx, ry, ryaw, rv, rw, clock, wpx, wpy, prev_dist, prev_head;
    double dq0x, dq0y, dq1x, dq1y, ts, bb, ego, cprob, ep_ret, last_ret;
    double cv, cw;        // SIM 3 (cn_config.wheel_accel): the twist /cmd_vel last carried; (rv, rw) is then the wheels' real twist
    long long crowd_ms;
    int done, dq_len, ntracks, ego_viol, social_viol, obst_steps, succ, fail, ep_step, status, nconf, nent, pending, episodes;
};

struct Lds {
    int* ptx; int* pty; int* gq;          // [n] end points / gradients in integer thousandths
    unsigned short* dmil;                 // [n] rounded range in thousandths (600 = free space)
    u64* w64;         // [M_COUNT][CN_MAXW] 64-ray bit words
    unsigned short* srcidx;  // [n] source ray of an aliased type entry
    int* wbase;       // [3][CN_MAXW]
    double* nearp;    // [4 (P+1)] pedestrians within lidar reach: centre relative to the lidar origin (x, y), |c|^2 - r^2, block bits
    double* ped;      // [2P] positions
    double* pedv;     // [2P] velocities
    double* trk;      // [CN_TF_COUNT][tcap]
    int tcap;         // tracker slots (32 or 64)
    double* cfx; double* cfy; double* cfd; int* cft; int* checked;   // confirmed objects
    double* cpv;      // [CN_MAX_TRACKS] collision probability per track
    double* stage;    // [64] staging buffer of the bbox-size sum (reset only)
    double* gtrk;     // this env's tracker table in HBM
    int wstride;      // words per mask
    double* tail;     // [7 + 4K]
    int* kidx;        // [K]
    // COMPACT layout (the 720-ray shape kernels; see "LDS map" in env_kernel_body): end points as int16 thousandths, confirmed objects
    // as integer thousandths + byte flags.  The values are the same integers either way (x == cn_div1000(mil) bit for bit).
    short* ptx16; short* pty16;
    int* cfxi; int* cfyi; unsigned short* cfdm; unsigned char* cft8; unsigned char* chk8;
};
// typed access to the arrays that have two representations (CMP = compact)
template <bool CMP> __device__ __forceinline__ int ptx_at(const Lds& L, int i) { if constexpr (CMP) return (int)L.ptx16[i]; else return L.ptx[i]; }
template <bool CMP> __device__ __forceinline__ int pty_at(const Lds& L, int i) { if constexpr (CMP) return (int)L.pty16[i]; else return L.pty[i]; }
template <bool CMP> __device__ __forceinline__ void pt_set(const Lds& L, int i, int x, int y)
{
    if constexpr (CMP) { L.ptx16[i] = (short)x; L.pty16[i] = (short)y; } else { L.ptx[i] = x; L.pty[i] = y; }
}
template <bool CMP> __device__ __forceinline__ double cfx_at(const Lds& L, int j) { if constexpr (CMP) return cn_div1000((double)L.cfxi[j]); else return L.cfx[j]; }
template <bool CMP> __device__ __forceinline__ double cfy_at(const Lds& L, int j) { if constexpr (CMP) return cn_div1000((double)L.cfyi[j]); else return L.cfy[j]; }
template <bool CMP> __device__ __forceinline__ double cfd_at(const Lds& L, int j) { if constexpr (CMP) return cn_div1000((double)L.cfdm[j]); else return L.cfd[j]; }
template <bool CMP> __device__ __forceinline__ int cft_at(const Lds& L, int j) { if constexpr (CMP) return (int)L.cft8[j]; else return L.cft[j]; }
template <bool CMP> __device__ __forceinline__ int chk_at(const Lds& L, int j) { if constexpr (CMP) return (int)L.chk8[j]; else return L.checked[j]; }
template <bool CMP> __device__ __forceinline__ void chk_set(const Lds& L, int j, int v) { if constexpr (CMP) L.chk8[j] = (unsigned char)v; else L.checked[j] = v; }
// a confirmed object from its centre ray's end point / range in thousandths
template <bool CMP> __device__ __forceinline__ void conf_set(const Lds& L, int slot, int obj, int xmil, int ymil, int dmil_)
{
    if constexpr (CMP) { L.cft8[slot] = (unsigned char)obj; L.cfxi[slot] = xmil; L.cfyi[slot] = ymil; L.cfdm[slot] = (unsigned short)dmil_; }
    else { L.cft[slot] = obj; L.cfx[slot] = cn_div1000((double)xmil); L.cfy[slot] = cn_div1000((double)ymil); L.cfd[slot] = cn_div1000((double)dmil_); }
}

__device__ __forceinline__ double heading_to_goal(KP p, const EnvRegs& e, double px, double py, double yaw)
{
    // ENV:222-237 (adds starting_point to the position; ENV:191-209 does not)
    double cx = px + p->start_x, cy = py + p->start_y;
    double ga = cn_atan2_t(p->trig, e.wpy - cy, e.wpx - cx);
    double h = ga - yaw;
    if (h > CN_PI) h -= 2 * CN_PI;
    else if (h < -CN_PI) h += 2 * CN_PI;
    return h;
}

__device__ __forceinline__ double dist3(double ax, double ay, double bx, double by)
{
    // np.linalg.norm of the 3-vector (ENV:191-197) = sqrt(ddot(d, d)): BLAS accumulates with fma (pinned by the goldens)
    double dx = ax - bx, dy = ay - by;
    return cn_sqrt(fma(dy, dy, dx * dx));
}

__device__ __forceinline__ bool in_box(double x, double y, double gx, double gy, double eps)
{
    // ENV:1285-1319 half-open box
    double xp = gx + eps, xm = gx - eps, yp = gy + eps, ym = gy - eps;
    return (x <= xp) & (x > xm) & (y <= yp) & (y > ym);        // (bitwise: four compares, no short-circuit branches)
}

// Ring (64-gon of radius r about (cx,cy), vertex k at angle -k*pi/32) against segment a->b.
// Lane k owns edge k.  Division-free membership test: for den != 0,
//   0 <= tn/den <= 1  <=>  (den > 0 ? 0 <= tn <= den : den <= tn <= 0)   (also for the rounded quotient)
//   0 <= un/den <  1  likewise with a strict upper bound.
// Returns the ballot of hit lanes; hit lanes get their intersection point.
struct Poly { double c0, s0, c1, s1; };  // this lane's edge: unit-circle vertices `lane` and `lane + 1`

__device__ __forceinline__ unsigned long long ring_segment(const Poly& pg, int lane, double cx, double cy, double r,
                                                           double ax, double ay, double bx, double by,
                                                           double* hx, double* hy)
{
    double c0x = cx + r * pg.c0, c0y = cy + r * pg.s0;
    double c1x = cx + r * pg.c1, c1y = cy + r * pg.s1;
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
        double t = cn_div(tn, den);      // hit lanes only: den != 0, |tn|, |un| <= |den|
        double u = cn_div(un, den);
        hit = (t >= 0.0) && (t <= 1.0) && (u >= 0.0) && (u < 1.0);  // the oracle's test, on the quotients
        *hx = ax + t * rx;
        *hy = ay + t * ry;
    }
    return __ballot(hit);
}

__device__ __forceinline__ double bcast_d(double v, int src)      // src: wave-uniform (a bit position of a ballot) -> v_readlane
{
    const int su = __builtin_amdgcn_readfirstlane(src);
    int lo = __builtin_amdgcn_readlane(__double2loint(v), su), hi = __builtin_amdgcn_readlane(__double2hiint(v), su);
    return __hiloint2double(hi, lo);
}

// a wave-uniform 64-bit value, forced into scalar registers
__device__ __forceinline__ u64 uni64(u64 v)
{
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}

// UTL:296-314 get_local_goal_waypoints
__device__ __forceinline__ void waypoint_refresh(KP p, const Poly& pg, EnvRegs& e, int lane, double px, double py)
{
    double hx = 0.0, hy = 0.0;
    unsigned long long m = ring_segment(pg, lane, px, py, p->waypoint_radius, px, py, p->goal_x, p->goal_y, &hx, &hy);
    if (__popcll(m) == 1) {
        int src = __ffsll((long long)m) - 1;
        e.wpx = bcast_d(hx, src);
        e.wpy = bcast_d(hy, src);
    } else {
        e.wpx = -(p->goal_x + 0.0);  // UTL:310-312: x sign flipped
        e.wpy = p->goal_y + 0.0;
    }
}

// ---- simulator -------------------------------------------------------------------------------
// Advances every pedestrian from crowd time t0 to t1.  `split` (0 < split < t1 - t0, or 0 for none) is an instant at which
// the integration is cut in two -- x <- clamp(fma(v, split - tc, x)), then on from there -- exactly as two back-to-back calls
// [t0, t0 + split], [t0 + split, t1] would: Env.step samples the world at +150 ms (deque) and +160 ms (scan), and the same
// fma chain cut at the same instants gives the same bits while the schedule prologue runs once.
// PACK (the shape kernels with 2 P <= 64): lane = (pedestrian, axis) -- the two coordinates of a pedestrian go through the same
// fma / clamp chain and the same generator on different data, so 2 P lanes do in one pass what P lanes did twice (20 of 64 lanes
// busy -> 40; same operations per coordinate, same bits).  The schedule arithmetic is per lane either way.
template <bool PACK = false>
__device__ __forceinline__ void ped_advance(KP p, int env, int lane, double* ped_p, double* ped_v, long long t0, long long t1, int split = 0)
{
    const double lo = -p->room_half + p->ped_radius, hi = p->room_half - p->ped_radius;
    const int T = p->ped_cycle_ms;
    const double invT = p->ped_inv_cycle;          // 1.0 / ped_cycle_ms, divided once on the host
    const long long gid = p->env_index_base + env;
    const double* preset = p->ped_preset + (size_t)env * 2 * p->P;
    // Per-env part of the schedule, once: t0 = cyc * T + ph.  Everything per pedestrian is then 32-bit
    // arithmetic on times RELATIVE to t0 (the 64-bit emulated integer ops were most of this stage).
    long long cyc = (long long)((double)t0 * invT);
    long long ph64 = t0 - cyc * (long long)T;
    if (ph64 < 0) { cyc -= 1; ph64 += T; }
    if (ph64 >= T) { cyc += 1; ph64 -= T; }
    const int ph = (int)ph64;
    const int dt = (int)(t1 - t0);
    if constexpr (PACK) {
        for (int l = lane; l < 2 * p->P; l += 64) {
            const int i = l >> 1, ax = l & 1;
            double x = ped_p[l], vx = ped_v[l];
            const int offs = i * p->ped_stagger_ms;
            const long long tt = cyc * (long long)T + (long long)(ph - offs);      // t0 - offs
            unsigned m; int a;
            if (tt <= 0) { m = 0u; a = (int)(-tt); }
            else {
                const int num = (ph - offs) + T - 1;
                int q = (int)floor((double)num * invT);
                int r = num - q * T;
                if (r < 0) { q -= 1; r += T; }
                if (r >= T) { q += 1; }
                m = (unsigned)cyc + (unsigned)q;
                a = q * T - (ph - offs);
            }
            int tc = 0;
            while (a < dt) {
                if (split > tc && a >= split) { x = cn_vclamp(fma(vx, cn_div1000((double)(split - tc)), x), lo, hi); tc = split; }
                if (a > tc) { x = cn_vclamp(fma(vx, cn_div1000((double)(a - tc)), x), lo, hi); tc = a; }
                if (p->ped_mode == 0) {
                    const uint64_t hbase = cn_mix64(p->seed ^ cn_mix64((uint64_t)gid));
                    const uint64_t h1 = cn_mix64(hbase ^ ((1ull << 32) | (uint64_t)(uint32_t)i));
                    const double u = (double)(cn_mix64(h1 ^ (uint64_t)(uint32_t)(2u * m + (unsigned)ax)) >> 11) * (1.0 / 9007199254740992.0);
                    vx = fma(2.0 * p->ped_vmax, u, -p->ped_vmax);
                } else vx = preset[l];
                a += T;
                m += 1u;
            }
            if (split > tc) { x = cn_vclamp(fma(vx, cn_div1000((double)(split - tc)), x), lo, hi); tc = split; }
            if (dt > tc) x = cn_vclamp(fma(vx, cn_div1000((double)(dt - tc)), x), lo, hi);
            ped_p[l] = x; ped_v[l] = vx;
        }
        return;
    }
    for (int i = lane; i < p->P; i += 64) {
        double x = ped_p[2 * i], y = ped_p[2 * i + 1];
        double vx = ped_v[2 * i], vy = ped_v[2 * i + 1];
        const int offs = i * p->ped_stagger_ms;
        // first update instant >= t0 is offs + m*T with m = max(0, ceil((t0 - offs) / T)); a = that instant - t0
        const long long tt = cyc * (long long)T + (long long)(ph - offs);      // t0 - offs
        unsigned m; int a;
        if (tt <= 0) { m = 0u; a = (int)(-tt); }
        else {
            const int num = (ph - offs) + T - 1;                               // ceil((ph - offs) / T), may be negative
            int q = (int)floor((double)num * invT);
            int r = num - q * T;
            if (r < 0) { q -= 1; r += T; }
            if (r >= T) { q += 1; }
            m = (unsigned)cyc + (unsigned)q;
            a = q * T - (ph - offs);
        }
        int tc = 0;
        while (a < dt) {
            if (split > tc && a >= split) {                  // the cut comes first (an update AT the cut belongs to the second half)
                double ds = cn_div1000((double)(split - tc));
                x = cn_vclamp(fma(vx, ds, x), lo, hi);
                y = cn_vclamp(fma(vy, ds, y), lo, hi);
                tc = split;
            }
            if (a > tc) {
                double ds = cn_div1000((double)(a - tc));    // == (a - tc) / 1000.0 exactly
                x = cn_vclamp(fma(vx, ds, x), lo, hi);
                y = cn_vclamp(fma(vy, ds, y), lo, hi);
                tc = a;
            }
            if (p->ped_mode == 0) {  // CROWD:101-102: cn_rng_u01(seed, gid, 1, i, 2m | 2m+1); the env part of the key is
                // only computed when some pedestrian of the wave really draws (most 10 ms advances have none)
                const uint64_t hbase = cn_mix64(p->seed ^ cn_mix64((uint64_t)gid));
                const uint64_t h1 = cn_mix64(hbase ^ ((1ull << 32) | (uint64_t)(uint32_t)i));
                const double u0 = (double)(cn_mix64(h1 ^ (uint64_t)(uint32_t)(2u * m)) >> 11) * (1.0 / 9007199254740992.0);
                const double u1 = (double)(cn_mix64(h1 ^ (uint64_t)(uint32_t)(2u * m + 1u)) >> 11) * (1.0 / 9007199254740992.0);
                vx = fma(2.0 * p->ped_vmax, u0, -p->ped_vmax);
                vy = fma(2.0 * p->ped_vmax, u1, -p->ped_vmax);
            } else {
                vx = preset[2 * i];
                vy = preset[2 * i + 1];
            }
            a += T;
            m += 1u;
        }
        if (split > tc) {
            double ds = cn_div1000((double)(split - tc));
            x = cn_vclamp(fma(vx, ds, x), lo, hi);
            y = cn_vclamp(fma(vy, ds, y), lo, hi);
            tc = split;
        }
        if (dt > tc) {
            double ds = cn_div1000((double)(dt - tc));
            x = cn_vclamp(fma(vx, ds, x), lo, hi);
            y = cn_vclamp(fma(vy, ds, y), lo, hi);
        }
        ped_p[2 * i] = x; ped_p[2 * i + 1] = y;
        ped_v[2 * i] = vx; ped_v[2 * i + 1] = vy;
    }
}

__device__ __forceinline__ double lane_d(double v, int src)   // lane `src` (a constant) of v as a wave-uniform value
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wrap_yaw(double th)
{
    if (th > CN_PI) th -= 2.0 * CN_PI;
    else if (th <= -CN_PI) th += 2.0 * CN_PI;
    return th;
}
// sin/cos an observation needs besides the simulator's own: the lidar frame (yaw) and ENV:267-268's cos(w), sin(w)
struct Trig { double sy, cy, sw, cw; };

// mid-point diff-drive step (turtlebot3_fake.cpp:156-162); sn, cs = sin/cos of the mid-point heading fma(0.5, w dt, yaw)
__device__ __forceinline__ void robot_advance_sc(KP p, EnvRegs& e, int ms, double sn, double cs)
{
    double dts = cn_div1000((double)ms);
    double ds = e.rv * dts, dth = e.rw * dts;
    double lim = p->room_half - p->robot_clearance;
    e.rx = cn_vclamp(fma(ds, cs, e.rx), -lim, lim);
    e.ry = cn_vclamp(fma(ds, sn, e.ry), -lim, lim);
    e.ryaw = wrap_yaw(e.ryaw + dth);
}
// Env.step's two robot advances (dt, then the scan latency) with every sine and cosine the step needs evaluated in ONE pass:
// the mid-point headings of both advances, the final yaw and w itself are all known once the action is, so lanes 0..3 take
// one argument each through the same polynomial (the four evaluations were ~45 wave-uniform float64 instructions apiece,
// each a serial dependency chain on the critical path).  Same arithmetic per value as robot_advance + observe.
__device__ __forceinline__ void step_trig(KP p, const EnvRegs& e, int lane, double& s1, double& c1, double& s2, double& c2, Trig& tg)
{
    const double dth1 = e.rw * cn_div1000((double)p->dt_ms), dth2 = e.rw * cn_div1000((double)p->scan_latency_ms);
    const double a1 = fma(0.5, dth1, e.ryaw);
    const double y1 = wrap_yaw(e.ryaw + dth1);
    const double a2 = fma(0.5, dth2, y1);
    const double y2 = wrap_yaw(y1 + dth2);
    const double x = lane == 0 ? a1 : (lane == 1 ? a2 : (lane == 2 ? y2 : e.rw));
    double s_, c_;
    cn_det_sincos_t(p->trig, x, &s_, &c_);
    s1 = lane_d(s_, 0); c1 = lane_d(c_, 0); s2 = lane_d(s_, 1); c2 = lane_d(c_, 1);
    tg.sy = lane_d(s_, 2); tg.cy = lane_d(c_, 2); tg.sw = lane_d(s_, 3); tg.cw = lane_d(c_, 3);
}
__device__ __forceinline__ void robot_advance(KP p, EnvRegs& e, int ms)
{
    double dts = cn_div1000((double)ms);
    double ds = e.rv * dts, dth = e.rw * dts;
    double sn, cs;
    cn_det_sincos_t(p->trig, fma(0.5, dth, e.ryaw), &sn, &cs);
    double lim = p->room_half - p->robot_clearance;
    e.rx = cn_vclamp(fma(ds, cs, e.rx), -lim, lim);
    e.ry = cn_vclamp(fma(ds, sn, e.ry), -lim, lim);
    double th = e.ryaw + dth;
    if (th > CN_PI) th -= 2.0 * CN_PI;
    else if (th <= -CN_PI) th += 2.0 * CN_PI;
    e.ryaw = th;
}

// cn_config.wheel_accel > 0 (SIM 3; XACRO:57-72 libgazebo_ros_diff_drive.so, restated in include/crowdnav.h): plugin ticks of at most
// 10 ms; the wheel speeds move towards the commanded ones by at most wheel_accel * h per tick -- either wheel within 0.01 m/s of
// its target releases both -- and the tick's twist moves the robot by the mid-point rule.  Wave-uniform; the same operations in
// the same order as the oracle's robot_advance_wheels.
__device__ __forceinline__ void robot_advance_wheels(KP p, EnvRegs& e, int ms)
{
    const double a = p->wheel_accel, sep = p->wheel_sep, half = 0.5 * sep;
    const double tl = e.cv - e.cw * half, tr = e.cv + e.cw * half;
    double cl = e.rv - e.rw * half, cr = e.rv + e.rw * half;
    for (int tt = 0; tt < ms; ) {
        const int h = (ms - tt < 10) ? (ms - tt) : 10;
        const double ah = a * ((double)h / 1000.0);
        if (fabs(tl - cl) < 0.01 || fabs(tr - cr) < 0.01) { cl = tl; cr = tr; }
        else {
            cl += (tl >= cl) ? fmin(tl - cl, ah) : fmax(tl - cl, -ah);
            cr += (tr >= cr) ? fmin(tr - cr, ah) : fmax(tr - cr, -ah);
        }
        e.rv = (cl + cr) * 0.5;
        e.rw = (cr - cl) / sep;
        robot_advance(p, e, h);
        tt += h;
    }
}

// cn_config.ped_contact = 1 (row A2; WORLD:86-145: rigid frictionless cylinders): the world advances in physics ticks of at
// most 10 ms -- crowd velocity assignments falling inside a tick take effect at its start, the (kinematic) robot moves by the
// mid-point rule, every pedestrian integrates into the room, then contacts are resolved Jacobi style from that state:
// overlapping discs each back off half the penetration and give up half the closing speed (a head-on pair stops), a disc
// overlapping the robot (radius robot_clearance) backs off all of it and loses its closing speed relative to the robot
// (the robot pushes it).  lane = pedestrian; corrections are summed in ascending index order, exactly as the oracle does.
// `corr`: 4 P doubles of LDS scratch (regions A + B are idle while the simulator runs).
__device__ __forceinline__ void sim_advance_contact(KP p, EnvRegs& e, int env, int lane, double* ped_p, double* ped_v, double* corr, int ms)
{
    const int P = p->P, T = p->ped_cycle_ms;
    const double lo = -p->room_half + p->ped_radius, hi = p->room_half - p->ped_radius;
    const double r = p->ped_radius, rr2 = (2.0 * r) * (2.0 * r);
    const double Rr = r + p->robot_clearance, Rr2 = Rr * Rr;
    const long long gid = p->env_index_base + env;
    const double* preset = p->ped_preset + (size_t)env * 2 * P;
    const long long t0 = e.crowd_ms;
    const double invT = 1.0 / (double)T;
    long long cyc = (long long)((double)t0 * invT);
    long long ph64 = t0 - cyc * (long long)T;
    if (ph64 < 0) { cyc -= 1; ph64 += T; }
    if (ph64 >= T) { cyc += 1; ph64 -= T; }
    const int ph = (int)ph64;
    for (int tt = 0; tt < ms; ) {
        const int h = (ms - tt < 10) ? (ms - tt) : 10;
        const double hs = cn_div1000((double)h);
        // 1. assignments whose instant lies in [t0 + tt, t0 + tt + h): first instant >= t0 + tt is offs + m T
        for (int i = lane; i < P; i += 64) {
            const int offs = i * p->ped_stagger_ms;
            const long long tq = cyc * (long long)T + (long long)(ph - offs) + (long long)tt;     // (t0 + tt) - offs
            long long m = 0, a = -tq;                                                             // a: instant relative to t0 + tt
            if (tq > 0) { m = (tq + T - 1) / T; a = m * (long long)T - tq; }
            while (a < h) {
                if (p->ped_mode == 0) {
                    const uint64_t hbase = cn_mix64(p->seed ^ cn_mix64((uint64_t)gid));
                    const uint64_t h1 = cn_mix64(hbase ^ ((1ull << 32) | (uint64_t)(uint32_t)i));
                    const double u0 = (double)(cn_mix64(h1 ^ (uint64_t)(uint32_t)(2u * (unsigned)m)) >> 11) * (1.0 / 9007199254740992.0);
                    const double u1 = (double)(cn_mix64(h1 ^ (uint64_t)(uint32_t)(2u * (unsigned)m + 1u)) >> 11) * (1.0 / 9007199254740992.0);
                    ped_v[2 * i] = fma(2.0 * p->ped_vmax, u0, -p->ped_vmax);
                    ped_v[2 * i + 1] = fma(2.0 * p->ped_vmax, u1, -p->ped_vmax);
                } else { ped_v[2 * i] = preset[2 * i]; ped_v[2 * i + 1] = preset[2 * i + 1]; }
                a += T; m += 1;
            }
        }
        // 2. the robot (uniform), and its linear velocity in the world frame
        robot_advance(p, e, h);
        double syaw, cyaw;
        cn_det_sincos_t(p->trig, e.ryaw, &syaw, &cyaw);
        const double rvx = e.rv * cyaw, rvy = e.rv * syaw;
        // 3. integrate
        for (int i = lane; i < P; i += 64) {
            ped_p[2 * i] = cn_vclamp(fma(ped_v[2 * i], hs, ped_p[2 * i]), lo, hi);
            ped_p[2 * i + 1] = cn_vclamp(fma(ped_v[2 * i + 1], hs, ped_p[2 * i + 1]), lo, hi);
        }
        CN_SYNC();
        // 4. corrections from the post-integration state
        for (int i = lane; i < P; i += 64) {
            const double xi = ped_p[2 * i], yi = ped_p[2 * i + 1], vxi = ped_v[2 * i], vyi = ped_v[2 * i + 1];
            double ax = 0.0, ay = 0.0, bx = 0.0, by = 0.0;
            for (int j = 0; j < P; ++j) {
                const double ddx = xi - ped_p[2 * j], ddy = yi - ped_p[2 * j + 1];
                const double d2 = fma(ddx, ddx, ddy * ddy);
                if (j == i || !(d2 < rr2) || !(d2 > 0.0)) continue;
                const double d = sqrt(d2), nx = ddx / d, ny = ddy / d;
                const double pen = 2.0 * r - d;
                ax = fma(0.5 * pen, nx, ax); ay = fma(0.5 * pen, ny, ay);
                const double vn = fma(vxi - ped_v[2 * j], nx, (vyi - ped_v[2 * j + 1]) * ny);
                if (vn < 0.0) { bx = fma(-0.5 * vn, nx, bx); by = fma(-0.5 * vn, ny, by); }
            }
            {
                const double ddx = xi - e.rx, ddy = yi - e.ry;
                const double d2 = fma(ddx, ddx, ddy * ddy);
                if (d2 < Rr2 && d2 > 0.0) {
                    const double d = sqrt(d2), nx = ddx / d, ny = ddy / d;
                    const double pen = Rr - d;
                    ax = fma(pen, nx, ax); ay = fma(pen, ny, ay);
                    const double vn = fma(vxi - rvx, nx, (vyi - rvy) * ny);
                    if (vn < 0.0) { bx = fma(-vn, nx, bx); by = fma(-vn, ny, by); }
                }
            }
            corr[4 * i] = ax; corr[4 * i + 1] = ay; corr[4 * i + 2] = bx; corr[4 * i + 3] = by;
        }
        CN_SYNC();
        for (int i = lane; i < P; i += 64) {
            ped_p[2 * i] = cn_vclamp(ped_p[2 * i] + corr[4 * i], lo, hi);
            ped_p[2 * i + 1] = cn_vclamp(ped_p[2 * i + 1] + corr[4 * i + 1], lo, hi);
            ped_v[2 * i] += corr[4 * i + 2]; ped_v[2 * i + 1] += corr[4 * i + 3];
        }
        CN_SYNC();
        tt += h;
    }
    e.crowd_ms += ms;
}

// cn_config.ped_mode = 2 (BASELINE north_star "per-env pedestrian social-force integration"; the model is stated in
// include/crowdnav.h and restated by the oracle's sim_advance_sf): Helbing-Molnar goal attraction + exponential repulsion from the
// other pedestrians, the four walls and the robot, on physics ticks of at most 10 ms.  lane = pedestrian; every acceleration is
// evaluated from the tick-start pedestrian state (Jacobi) in the oracle's order -- goal, pedestrians by index, walls -x +x -y +y,
// robot -- then v (capped at 1.3 v0), then x, clamped into the room.  The pair loop reads pedestrian j's position as an LDS
// broadcast (one address for the whole wave).  `scr`: 8 P doubles of LDS scratch (+ P^2 doubles + P (P - 1) / 2 shorts when
// cn_create found room for the pair matrix) (regions A + B are idle while the simulator
// runs): next state [4 P] and this call's copy of the goal records + desired speeds [4 P] (goal x, y, counter, v0), so that the
// 16 ticks of a step pay no global-memory round trip and no RNG evaluation.
// A pedestrian-pedestrian force component enters the sum on a grid of 2^-36 m/s^2: every partial sum of such values is exact
// (cn_create bounds P A e^{2r/B} below 2^15), so the total is independent of the ORDER of the additions (round 5; the oracle's
// sf_quant).  That is what allows one evaluation per unordered pair with +- scattered by LDS atomics below.
__device__ __forceinline__ double cn_sf_quant(double v) { return rint(v * 68719476736.0) * (1.0 / 68719476736.0); }
__device__ __forceinline__ void cn_lds_add_f64(double* lds_ptr, double v)     // ds_add_f64: exact here, so the arrival order is immaterial
{
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) double*)lds_ptr;
    asm volatile("ds_add_f64 %0, %1" :: "v"(a), "v"(v) : "memory");
}
template <bool DENSE>      // DENSE (SIM 4): the per-lane near masks below, for crowds of up to 128 whose pair matrix does not fit in LDS
__device__ __forceinline__ void sim_advance_sf(KP p, EnvRegs& e, int env, int lane, double* ped_p, double* ped_v, double* scr, int ms)
{
    const int P = p->P;
    const double H = p->room_half, r = p->ped_radius, lo = -H + r, hi = H - r;
    const double A = p->sf_A, B = p->sf_B, Aw = p->sf_wall_A, Bw = p->sf_wall_B, tau = p->sf_tau;
    const double Rr = r + p->robot_clearance;
    const double cut = fma(12.0, B, 2.0 * r) + 1e-6, cut2 = cut * cut;        // beyond it the exponent is certainly below -12
    const long long gid = p->env_index_base + env;
    double* const gaux = p->ped_aux + (size_t)env * 3 * P;
    double* const nxt = scr;
    double* const aux = scr + 4 * P;
    // Small crowds (cn_create: sf_pair_matrix, P <= 22 at 360 rays): ALL P (P - 1) / 2 unordered pairs, lane = pair over all 64 lanes
    // (190 pairs = 3 passes at P = 20), each evaluated ONCE and scattered +- into acc by LDS atomics -- exact sums, see cn_sf_quant.
    // The static list is in round-robin order -- (i, i + k mod P) for k = 1 .. P / 2 -- so that the 64 pairs of a pass touch every
    // pedestrian at most a handful of times (in (i, j) order nineteen consecutive lanes would add to the same address).
    const bool pairm = !DENSE && p->sf_pair_matrix != 0;
    // DENSE, round 5: one evaluation per unordered near pair.  acc [2 P]: the pedestrians' summed repulsion (LDS atomics, exact);
    // nlist: the near pairs (i << 8) | j, i < j, compacted -- sf_pair_cap of them fit behind acc (0: no room, the per-lane walk runs)
    double* const acc = scr + 8 * P;
    unsigned short* const nlist = (unsigned short*)(acc + 2 * P);
    const int pcap = DENSE ? p->sf_pair_cap : 0;
    unsigned short* const plist = nlist;                             // pair t -> (i << 8) | j, round-robin order
    const int npairs = (P * (P - 1)) >> 1;
    for (int i = lane; i < P; i += 64) {
        aux[4 * i] = gaux[3 * i]; aux[4 * i + 1] = gaux[3 * i + 1]; aux[4 * i + 2] = gaux[3 * i + 2];
        aux[4 * i + 3] = p->ped_vmax * fma(0.5, cn_rng_u01(p->seed, gid, 4u, (uint32_t)i, 0u), 0.5);     // desired speed v0
        if (pairm) {      // round k = 1 .. P / 2 holds the pairs (i, (i + k) mod P): P of them, or P / 2 in the last round of an even P
            for (int k = 1; 2 * k <= P; ++k) {
                if (2 * k == P && i >= k) break;
                const int j = (i + k) % P;
                plist[(k - 1) * P + i] = (unsigned short)((i << 8) | j);
            }
        }
    }
    CN_SYNC();
    const int tick = p->sf_tick_ms;
    for (int tt = 0; tt < ms; ) {
        const int h = (ms - tt < tick) ? (ms - tt) : tick;
        const double hs = cn_div1000((double)h);
        robot_advance(p, e, h);
        if (pairm) {
            for (int i = lane; i < 2 * P; i += 64) acc[i] = 0.0;
            CN_SYNC();
            for (int t0 = 0; t0 < npairs; t0 += 64) {
                const int t = t0 + lane;
                if (t < npairs) {
                    const int ij = plist[t], i = ij >> 8, j = ij & 255;
                    const double ddx = ped_p[2 * i] - ped_p[2 * j], ddy = ped_p[2 * i + 1] - ped_p[2 * j + 1];
                    const double d2 = fma(ddx, ddx, ddy * ddy);
                    if (d2 > 0.0 && !(d2 > cut2)) {
                        const double d = sqrt(d2), arg = (2.0 * r - d) / B;
                        if (!(arg < -12.0)) {
                            const double f = (A * cn_det_exp(arg)) * (1.0 / d);
                            const double cx = cn_sf_quant(f * ddx), cy = cn_sf_quant(f * ddy);
                            cn_lds_add_f64(acc + 2 * i, cx); cn_lds_add_f64(acc + 2 * i + 1, cy);
                            cn_lds_add_f64(acc + 2 * j, -cx); cn_lds_add_f64(acc + 2 * j + 1, -cy);
                        }
                    }
                }
            }
            CN_SYNC();
        }
        bool use_pairs = false;
        if constexpr (DENSE) {
            if (pcap > 0) {
                // (1) each lane marks, for its pedestrian of either pass (i = lane, lane + 64), the near pedestrians ABOVE it: broadcast
                //     reads, a dozen instructions per j; (2) the pairs are counted -- if they fit, (3) compacted into nlist in (i, j)
                //     order and (4) evaluated lane = pair, 64 at a time: the expensive part (sqrt, two divides, the exponential) runs
                //     ceil(near pairs / 64) times per tick instead of (busiest lane's neighbour count) x 2 passes, every pair once.
                u64 mk[2][2] = {{0ull, 0ull}, {0ull, 0ull}};
                int cnt[2] = {0, 0};
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int i = 64 * ps + lane;
                    if (64 * ps < P) {
                        const bool act_ = i < P;
                        const double xi_ = act_ ? ped_p[2 * i] : 0.0, yi_ = act_ ? ped_p[2 * i + 1] : 0.0;
                        for (int j = 64 * ps + 1; j < P; ++j) {
                            const double ddx = xi_ - ped_p[2 * j], ddy = yi_ - ped_p[2 * j + 1];
                            const double d2 = fma(ddx, ddx, ddy * ddy);
                            const u64 nr = (act_ && j > i && d2 > 0.0 && !(d2 > cut2)) ? 1ull : 0ull;
                            if (j < 64) mk[ps][0] |= nr << j; else mk[ps][1] |= nr << (j - 64);
                        }
                        cnt[ps] = __popcll(mk[ps][0]) + __popcll(mk[ps][1]);
                    }
                }
                const int tot0 = cn_wave_sum_i(cnt[0]), tot1 = cn_wave_sum_i(cnt[1]);
                const int npairs_ = tot0 + tot1;
                use_pairs = npairs_ <= pcap;
                if (use_pairs) {
                    for (int i = lane; i < 2 * P; i += 64) acc[i] = 0.0;
#pragma unroll
                    for (int ps = 0; ps < 2; ++ps) {
                        int incl = cnt[ps];                       // inclusive prefix sum over the lanes
#pragma unroll
                        for (int d_ = 1; d_ < 64; d_ <<= 1) { const int o_ = __shfl_up(incl, d_, 64); if (lane >= d_) incl += o_; }
                        int k = (ps ? tot0 : 0) + incl - cnt[ps];
                        u64 a0 = mk[ps][0], a1 = mk[ps][1];
                        const int i = 64 * ps + lane;
                        while (a0 | a1) {
                            int j;
                            if (a0) { j = __builtin_ctzll(a0); a0 &= a0 - 1ull; } else { j = 64 + __builtin_ctzll(a1); a1 &= a1 - 1ull; }
                            nlist[k++] = (unsigned short)((i << 8) | j);
                        }
                    }
                    CN_SYNC();
                    // lane l takes the pairs l * passes + q, q = 0 .. passes - 1: consecutive lanes are `passes` entries apart in the
                    // (i, j)-ordered list, i.e. mostly on different pedestrians i (side by side they would queue on one LDS address)
                    const int passes = (npairs_ + 63) >> 6;
                    for (int q = 0; q < passes; ++q) {
                        const int t = lane * passes + q;
                        if (t < npairs_) {
                            const int ij = nlist[t], i = ij >> 8, j = ij & 255;
                            const double ddx = ped_p[2 * i] - ped_p[2 * j], ddy = ped_p[2 * i + 1] - ped_p[2 * j + 1];
                            const double d2 = fma(ddx, ddx, ddy * ddy);
                            const double d = sqrt(d2), arg = (2.0 * r - d) / B;
                            if (!(arg < -12.0)) {
                                const double f = (A * cn_det_exp(arg)) * (1.0 / d);
                                const double cx = cn_sf_quant(f * ddx), cy = cn_sf_quant(f * ddy);
                                cn_lds_add_f64(acc + 2 * i, cx); cn_lds_add_f64(acc + 2 * i + 1, cy);
                                cn_lds_add_f64(acc + 2 * j, -cx); cn_lds_add_f64(acc + 2 * j + 1, -cy);
                            }
                        }
                    }
                    CN_SYNC();
                }
            }
        }
        for (int i0 = 0; i0 < P; i0 += 64) {
            const int i = i0 + lane;
            const bool act = i < P;
            double xi = 0.0, yi = 0.0, vxi = 0.0, vyi = 0.0, v0 = 0.0, gx = 0.0, gy = 0.0;
            if (act) {
                xi = ped_p[2 * i]; yi = ped_p[2 * i + 1]; vxi = ped_v[2 * i]; vyi = ped_v[2 * i + 1];
                gx = aux[4 * i]; gy = aux[4 * i + 1]; v0 = aux[4 * i + 3];
            }
            double gdx = gx - xi, gdy = gy - yi, gd2 = fma(gdx, gdx, gdy * gdy);
            if (act && gd2 <= p->sf_goal_eps2) {                 // goal reached: the next one of this pedestrian's sequence (stream 3)
                const uint32_t m = (uint32_t)aux[4 * i + 2] + 1u;
                const double glo = -H + 0.1, gspan = 2.0 * H - 0.2;
                gx = fma(gspan, cn_rng_u01(p->seed, gid, 3u, (uint32_t)i, 2u * m), glo);
                gy = fma(gspan, cn_rng_u01(p->seed, gid, 3u, (uint32_t)i, 2u * m + 1u), glo);
                aux[4 * i] = gx; aux[4 * i + 1] = gy; aux[4 * i + 2] = (double)m;
                gdx = gx - xi; gdy = gy - yi; gd2 = fma(gdx, gdx, gdy * gdy);
            }
            double ex = 0.0, ey = 0.0;
            if (gd2 > 0.0) { const double ginv = 1.0 / sqrt(gd2); ex = gdx * ginv; ey = gdy * ginv; }
            double ax = (v0 * ex - vxi) / tau, ay = (v0 * ey - vyi) / tau;
            double sx = 0.0, sy = 0.0;                     // the other pedestrians' repulsion: an exact sum of cn_sf_quant()ed components
            if (pairm || (DENSE && use_pairs)) {
                if (act) { sx = acc[2 * i]; sy = acc[2 * i + 1]; }
            } else if constexpr (DENSE) {
                // Dense crowds (no room for the pair matrix): the expensive part of a pair term -- sqrt, two divides, the exponential,
                // ~80 float64 instructions -- is only owed for pedestrians within the cut-off, a quarter of a 100-pedestrian room.
                // Written as one loop over j the wavefront pays it for EVERY j (some lane always has a near pair).  So: a cheap pass
                // marks each lane's near pedestrians in a 128-bit mask (broadcast reads, a dozen instructions per j), then every lane
                // walks ITS OWN mask in ascending j -- the loop runs as long as the busiest lane's neighbour count (~35-40 of 100),
                // and each lane's sum keeps the oracle's order and arithmetic.
                u64 m0 = 0ull, m1 = 0ull;
                for (int j = 0; j < P; ++j) {
                    const double ddx = xi - ped_p[2 * j], ddy = yi - ped_p[2 * j + 1];
                    const double d2 = fma(ddx, ddx, ddy * ddy);
                    const u64 nr = (act && j != i && d2 > 0.0 && !(d2 > cut2)) ? 1ull : 0ull;
                    if (j < 64) m0 |= nr << j; else m1 |= nr << (j - 64);
                }
                while (__ballot((m0 | m1) != 0ull) != 0ull) {
                    if ((m0 | m1) != 0ull) {
                        int j;
                        if (m0) { j = __builtin_ctzll(m0); m0 &= m0 - 1ull; } else { j = 64 + __builtin_ctzll(m1); m1 &= m1 - 1ull; }
                        const double ddx = xi - ped_p[2 * j], ddy = yi - ped_p[2 * j + 1];
                        const double d2 = fma(ddx, ddx, ddy * ddy);
                        const double d = sqrt(d2), arg = (2.0 * r - d) / B;
                        if (!(arg < -12.0)) {
                            const double f = (A * cn_det_exp(arg)) * (1.0 / d);
                            sx += cn_sf_quant(f * ddx); sy += cn_sf_quant(f * ddy);
                        }
                    }
                }
            } else {
            for (int j = 0; j < P; ++j) {
                const double ddx = xi - ped_p[2 * j], ddy = yi - ped_p[2 * j + 1];
                const double d2 = fma(ddx, ddx, ddy * ddy);
                if (act && j != i && d2 > 0.0 && !(d2 > cut2)) {
                    const double d = sqrt(d2), arg = (2.0 * r - d) / B;
                    if (!(arg < -12.0)) {
                        const double f = (A * cn_det_exp(arg)) * (1.0 / d);
                        sx += cn_sf_quant(f * ddx); sy += cn_sf_quant(f * ddy);
                    }
                }
            }
            }
            ax = ax + sx; ay = ay + sy;
            if (act) {
                {   // walls: distance from the centre to the wall plane, pushing inwards
                    double arg = (r - (xi + H)) / Bw;
                    if (!(arg < -12.0)) ax = ax + Aw * cn_det_exp(arg);
                    arg = (r - (H - xi)) / Bw;
                    if (!(arg < -12.0)) ax = ax - Aw * cn_det_exp(arg);
                    arg = (r - (yi + H)) / Bw;
                    if (!(arg < -12.0)) ay = ay + Aw * cn_det_exp(arg);
                    arg = (r - (H - yi)) / Bw;
                    if (!(arg < -12.0)) ay = ay - Aw * cn_det_exp(arg);
                }
                {   // the robot, where this tick leaves it
                    const double ddx = xi - e.rx, ddy = yi - e.ry;
                    const double d2 = fma(ddx, ddx, ddy * ddy);
                    if (d2 > 0.0) {
                        const double d = sqrt(d2), arg = (Rr - d) / B;
                        if (!(arg < -12.0)) {
                            const double f = (A * cn_det_exp(arg)) * (1.0 / d);
                            ax = fma(f, ddx, ax); ay = fma(f, ddy, ay);
                        }
                    }
                }
                double vx = fma(ax, hs, vxi), vy = fma(ay, hs, vyi);
                const double cap = 1.3 * v0, s2 = fma(vx, vx, vy * vy);
                if (s2 > cap * cap) { const double k = cap / sqrt(s2); vx = vx * k; vy = vy * k; }
                nxt[4 * i] = cn_clamp(fma(vx, hs, xi), lo, hi); nxt[4 * i + 1] = cn_clamp(fma(vy, hs, yi), lo, hi);
                nxt[4 * i + 2] = vx; nxt[4 * i + 3] = vy;
            }
        }
        CN_SYNC();
        for (int i = lane; i < P; i += 64) {
            ped_p[2 * i] = nxt[4 * i]; ped_p[2 * i + 1] = nxt[4 * i + 1];
            ped_v[2 * i] = nxt[4 * i + 2]; ped_v[2 * i + 1] = nxt[4 * i + 3];
        }
        CN_SYNC();
        tt += h;
    }
    for (int i = lane; i < P; i += 64) { gaux[3 * i] = aux[4 * i]; gaux[3 * i + 1] = aux[4 * i + 1]; gaux[3 * i + 2] = aux[4 * i + 2]; }
    CN_SYNC();
    e.crowd_ms += ms;
}

// the tick-based simulators: SIM 1 = rigid contact (cn_config.ped_contact), SIM 2 = social force (cn_config.ped_mode 2)
template <int SIM>
__device__ __forceinline__ void sim_advance_ticks(KP p, EnvRegs& e, int env, int lane, double* ped_p, double* ped_v, double* scr, int ms)
{
    if constexpr (SIM == 2) sim_advance_sf<false>(p, e, env, lane, ped_p, ped_v, scr, ms);
    else if constexpr (SIM == 4) sim_advance_sf<true>(p, e, env, lane, ped_p, ped_v, scr, ms);
    else if constexpr (SIM == 3) {        // plain pedestrians (they do not see the robot), wheel-ramp robot
        if (ms <= 0) return;
        ped_advance(p, env, lane, ped_p, ped_v, e.crowd_ms, e.crowd_ms + ms);
        e.crowd_ms += ms;
        robot_advance_wheels(p, e, ms);
    }
    else sim_advance_contact(p, e, env, lane, ped_p, ped_v, scr, ms);
}

__device__ __forceinline__ void sim_advance(KP p, EnvRegs& e, int env, int lane, double* ped_p, double* ped_v, int ms)
{
    if (ms <= 0) return;
    ped_advance(p, env, lane, ped_p, ped_v, e.crowd_ms, e.crowd_ms + ms);
    e.crowd_ms += ms;
    robot_advance(p, e, ms);
}

// ---- Env.get_state (ENV:245-1044) ----------------------------------------------------------------
// ped_p: pedestrian positions in LDS.  Writes the observation (float32 and optionally float64).
// ---- lidar (XACRO:150-178), shared by both observation layouts ------------------------------------------
// Pedestrians that can return a range <= lidar_max: |centre - origin| <= lidar_max + radius (+ slack).
// A culled pedestrian could only produce t > lidar_max, which reads as "no return" anyway, so the
// result is identical to testing all P (the oracle does).
// Each listed pedestrian also carries a bit per 64-ray block: can ANY ray of that block reach its disc?  Block q spans
// the rays within beta = 32.5 steps of ray 64q + 32; a disc at distance d subtends asin(r / d); the two cones meet iff
// the angle between them is <= asin(r / d) + beta, i.e. (times d, cos decreasing on [0, pi])
//     oc . u_q  >=  cos(beta) sqrt(d^2 - r^2) - sin(beta) r
// with u_q the block's axis (robot frame: a host table) and oc rotated into the robot frame.  Evaluated with slack, so a
// cleared bit only ever skips a test that could not hit (ranges unchanged); a crowded env otherwise pays every near
// pedestrian in every block, and the launch waits for its most crowded env.
// blkw (optional): lane q receives block q's word in a register and the LDS copy is not written -- the ray loop of the same wavefront
// then takes it with two lane reads per block instead of an LDS round trip (one wavefront per environment, lidar-tracker layout).
__device__ __forceinline__ int near_peds(KP p, const Lds& L, int lane, double ox, double oy, double sy, double cy, u64* blkw = nullptr)
{
    // The list holds what the ray test needs (not indices): the inner loop then reads three independent values per
    // pedestrian instead of chasing index -> position through two dependent LDS reads for every ray block.
    int nnear = 0;
    const double lim = p->lidar_max + p->ped_radius + 1e-6, lim2 = lim * lim;
    const double rr = p->ped_radius * p->ped_radius;
    const int Wb = (p->R + 63) >> 6;
    const double* const bd = p->blk_dir;
    for (int j0 = 0; j0 < p->P; j0 += 64) {
        int j = j0 + lane;
        bool nr = false;
        double ocx = 0.0, ocy = 0.0;
        if (j < p->P && !(CN_ABLATE(1))) {
            ocx = L.ped[2 * j] - ox; ocy = L.ped[2 * j + 1] - oy;
            nr = fma(ocx, ocx, ocy * ocy) <= lim2;
        }
        u64 m = __ballot(nr);
        if (nr) {
            const int slot = nnear + __popcll(m & ((1ull << lane) - 1ull));
            const double cc = fma(ocx, ocx, fma(ocy, ocy, -rr));
            u64 tag = ~0ull;
            if (cc > 0.0) {
                const double rx_ = fma(cy, ocx, sy * ocy), ry_ = fma(cy, ocy, -(sy * ocx));     // oc in the robot frame
                const double thr = fma(p->blk_cb, cn_sqrt(cc), -(p->blk_sb * p->ped_radius)) - 1e-9;
                tag = 0ull;
                for (int q = 0; q < Wb; ++q)
                    if (fma(rx_, bd[2 * q], ry_ * bd[2 * q + 1]) >= thr) tag |= 1ull << q;
            }
            L.nearp[4 * slot] = ocx; L.nearp[4 * slot + 1] = ocy; L.nearp[4 * slot + 2] = cc;
            ((u64*)L.nearp)[4 * slot + 3] = tag;
        }
        nnear += __popcll(m);
    }
    CN_SYNC();
    // Transposed for the ray loop: one 64-bit word per block = the list slots (first 64) worth testing in that block, so the
    // loop walks set bits held in scalar registers instead of paying an LDS round trip per pedestrian per block to find out.
    // (The flag-word area is free until the ray loop is over.)
    u64 mytag = 0ull;
    if (lane < nnear) mytag = ((const u64*)L.nearp)[4 * lane + 3];
    u64 mine = 0ull;
    for (int q = 0; q < Wb; ++q) {
        const u64 bm = __ballot(((mytag >> q) & 1ull) != 0ull);
        if (blkw) mine = cn_writelane_u64(mine, bm, q);
        else if (lane == 0) L.w64[q] = bm;
    }
    if (blkw) *blkw = mine;
    CN_SYNC();
    return nnear;
}

// Range of ray k: the sensor's own reading (EXT), or the nearest hit among the room walls and the near pedestrians.
// lc, ls: ray k in the robot frame = the host table of cn_det_sincos(k * step), loaded by the caller.
template <bool EXT>
__device__ __forceinline__ double cast_ray(KP p, const Lds& L, int env, int k, double ox, double oy, double sy,
                                           double cy, int nnear, bool wall_x, bool wall_y, double lc, double ls, const u64* blkw = nullptr,
                                           const double* lmin_reg = nullptr, const bool defer_f32 = false)
{
    // lmin_reg / defer_f32 (the lidar-tracker layout's ray loop): lidar_min comes in a vector register the caller loaded before the
    // loop, and the float32 rounding of cn_config.scan_f32 is the caller's -- read in place, the two parameters and the caller's own
    // lidar_min_positive flag were three scalar loads with three waits in every 64-ray block
    if constexpr (EXT) {
        return p->ext_ranges[(size_t)env * p->R + k];   // Gazebo / a physical lidar
    } else {
        const double h = p->room_half;
        double t = INFINITY;
        double dx = fma(cy, lc, -(sy * ls));
        double dy = fma(sy, lc, cy * ls);
        // A wall farther than lidar_max from the origin can only give t > lidar_max ("no return"), so its
        // divide is skipped when the whole env is out of its reach (uniform test, result unchanged).
        // (one divide per axis: the facing wall is selected first -- a wave has rays of both signs)
        // Per ray the same argument again, before paying the divide: a / d <= lidar_max  <=>  a sign(d) <= lidar_max |d|
        // (with slack; a quotient beyond lidar_max cannot change a range that is reported as "no return" above it).
        // Rays of a 64-block point within 64 degrees of each other, so whole blocks skip the divide of a wall behind them.
        // (a sign(d) = h - o sign(d): the distance to the facing wall along the axis, two bit operations and a subtract)
        const double reach = p->lidar_max * (1.0 + 1e-9);
        if (wall_x && dx != 0.0 && h - cn_xorsign(ox, dx) <= fma(reach, fabs(dx), 1e-12)) t = cn_vmin(t, cn_div(copysign(h, dx) - ox, dx));
        if (wall_y && dy != 0.0 && h - cn_xorsign(oy, dy) <= fma(reach, fabs(dy), 1e-12)) t = cn_vmin(t, cn_div(copysign(h, dy) - oy, dy));
        t = lmin_reg ? cn_vmax(t, *lmin_reg) : cn_vmax_s(t, p->lidar_min);   // (one v_max_f64: t is +inf or a finite quotient here, never a NaN)
        const int q = k >> 6;                                  // this block of 64 rays (wave-uniform)
        auto test = [&](int c) {
            const double ocx = L.nearp[4 * c], ocy = L.nearp[4 * c + 1], cc = L.nearp[4 * c + 2];
            double b = fma(ocx, dx, ocy * dy);
            double disc = fma(b, b, -cc);
            if (disc >= 0.0) {
                double sq = cn_sqrt(disc);
                double t2 = b + sq;
                if (t2 >= (lmin_reg ? *lmin_reg : p->lidar_min)) {
                    double t1 = lmin_reg ? cn_vmax(b - sq, *lmin_reg) : cn_vmax_s(b - sq, p->lidar_min);
                    t = cn_vmin(t, t1);
                }
            }
        };
        // list slots some ray of this block can reach: lane q of near_peds' register copy, or the LDS word
        u64 bm = blkw ? cn_readlane_u64(*blkw, q) : uni64(L.w64[q]);
        while (bm) {
            const int c = __builtin_ctzll(bm);
            bm &= bm - 1ull;
            test(c);
        }
        for (int c = 64; c < nnear; ++c)                       // beyond 64 near pedestrians: per-slot bits
            if ((((const u64*)L.nearp)[4 * c + 3] >> q) & 1ull) test(c);
        t = (t > p->lidar_max) ? INFINITY : t;          // the simulated sensor reports no return beyond its range
        // cn_config.scan_f32: sensor_msgs/LaserScan.ranges is float32[] (XACRO:172-175) -- what Gazebo hands ENV:1218
        if (!defer_f32 && __builtin_expect(p->scan_f32 != 0, 0)) t = (double)(float)t;
        return t;
    }
}

// ---- obs_layout 1: environment_stage_1_original.py ("ORIG"), SURVEY 8f N3 --------------------------------
// ORIG:244-260: heading straight to desired_point, no starting_pose offset
__device__ __forceinline__ double orig_heading(KP p, double px, double py, double yaw)
{
    double ga = cn_atan2_t(p->trig, p->goal_y - py, p->goal_x - px);
    double h = ga - yaw;
    if (h > CN_PI) h -= 2 * CN_PI;
    else if (h < -CN_PI) h += 2 * CN_PI;
    return h;
}

// ORIG:278-322: state = [round(range, 3)] * (R-1) + [heading, distance] + [round(x, 3), round(y, 3)]; tail -> L.tail[0..3]
template <bool EXT>
__device__ __forceinline__ void observe_original(KP p, EnvRegs& e, const Lds& L, int env, int lane, int step_counter,
                                                 float* obs32, float* fin32, double* obs64, int* done_out)
{
    const int R = p->R, n = R - 1, D = n + 4;
    const double px = e.rx, py = e.ry, yaw = e.ryaw;
    double dist = cn_round_np64_2_t<false>(dist3(px, py, p->goal_x, p->goal_y), PY2);   // round(np.float64, 2), ORIG:280
    double head = cn_py_round2(orig_heading(p, px, py, yaw), PY2);        // ORIG:281
    double sy, cy;
    cn_det_sincos_t(p->trig, yaw, &sy, &cy);
    const double ox = fma(p->lidar_offset_x, cy, px), oy = fma(p->lidar_offset_x, sy, py);
    const double h = p->room_half;
    const int nnear = near_peds(p, L, lane, ox, oy, sy, cy);
    const bool wall_x = !(h - fabs(ox) > p->lidar_max + 1e-6);
    const bool wall_y = !(h - fabs(oy) > p->lidar_max + 1e-6);
    float* o32 = obs32 + (size_t)env * D;
    float* f32 = fin32 ? fin32 + (size_t)env * D : nullptr;
    double* o64 = obs64 ? obs64 + (size_t)env * D : nullptr;
    double smin = 1e300;
    const double* const lidc = p->lidar_c; const double* const lids = p->lidar_s;
    double lc_n = 0.0, ls_n = 0.0;
    if (!EXT && lane < R) { lc_n = lidc[lane]; ls_n = lids[lane]; }
    for (int k = lane; k < R; k += 64) {
        const double lc = lc_n, ls = ls_n;
        if (!EXT && k + 64 < R) { lc_n = lidc[k + 64]; ls_n = lids[k + 64]; }      // next block's directions: in flight during this one
        const double r = cast_ray<EXT>(p, L, env, k, ox, oy, sy, cy, nnear, wall_x, wall_y, lc, ls);
        if (k >= 1) {
            int j = R - 1 - k;                                        // ORIG:298-300 reverse, drop the last
            double v;
            if (isinf(r)) v = 0.6;                                    // ORIG:290-291: the literal, not max_scan_range
            else if (r != r) v = 0.0;
            else v = r;
            smin = cn_vmin(smin, v);
            double so = cn_py_round3(v, PY2);                              // ORIG:317
            o32[j] = (float)so;
            if (f32) f32[j] = (float)so;
            if (o64) o64[j] = so;
        }
    }
    smin = cn_wave_min_d(smin);
    if (!e.done) {
        if (0.105 > smin && smin > 0.0) e.done = 1;                   // ORIG:282,303-305
        if (in_box(px, py, p->goal_x, p->goal_y, 0.20)) e.done = 1;     // ORIG:307-309 (epsilon default, ORIG:500)
        if (step_counter >= p->max_steps) e.done = 1;                  // ORIG:311-313
    }
    const double x3 = cn_py_round3(px, PY2), y3 = cn_py_round3(py, PY2);        // ORIG:315
    if (lane < 4) {
        double tv = lane == 0 ? head : lane == 1 ? dist : lane == 2 ? x3 : y3;
        L.tail[lane] = tv;
        o32[n + lane] = (float)tv;
        if (f32) f32[n + lane] = (float)tv;
        if (o64) o64[n + lane] = tv;
    }
    CN_SYNC();
    *done_out = e.done;
}

// ORIG:324-402.  The quirk is the reference's: state[-1] (the rounded y) is its "current_distance" and state[-2]
// (the rounded x) its "current_heading".
__device__ __forceinline__ double compute_reward_original(KP p, EnvRegs& e, const Lds& L, int done)
{
    double cur_dist = L.tail[3], cur_head = L.tail[2];
    double dd = cur_dist - e.prev_dist, hd = cur_head - e.prev_head;
    int htg = 0, dtg = 0;
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
    double reward = (double)(dtg + htg);                              // step_reward 0, action_reward unused (ORIG:333-374)
    e.prev_dist = cur_dist;
    e.prev_head = cur_head;
    if (done) {
        if (in_box(e.rx, e.ry, p->goal_x, p->goal_y, 0.20)) { e.fail = 0; e.succ = 1; reward = 200 + reward; }
        else { e.fail = 1; e.succ = 0; reward = -200 + reward; }
    }
    return reward;
}

// UTL:405-419 compute_average_bounding_box_size on the end points of an all-max scan at pose (px, py, yaw) (ENV:287-294).
// stage: 64 doubles of LDS.  Python sum(): strictly left to right.
__device__ __forceinline__ double bbox_size(KP p, double* stage, int lane, int n, double px, double py, double yaw)
{
    const double MAXR = p->max_scan_range, deg2rad = CN_PI / 180.0;
    double sum = 0.0;
#pragma unroll 1
    for (int i0 = 0; i0 < n; i0 += 64) {      // (cold: resets away from the spawn pose only -- never unrolled, whatever n is known to be)
        int i = i0 + lane;
        if (i < n) {
            int j = (i == n - 1) ? 0 : i + 1;
            double s0, c0, s1, c1;
            cn_det_sincos_t(p->trig, ((double)i * p->angle_inc_deg) * deg2rad - yaw, &s0, &c0);
            cn_det_sincos_t(p->trig, ((double)j * p->angle_inc_deg) * deg2rad - yaw, &s1, &c1);
            double x0 = cn_py_round3(px + (MAXR * c0), PY2), y0 = cn_py_round3(py + (MAXR * s0) * -1.0, PY2);
            double x1 = cn_py_round3(px + (MAXR * c1), PY2), y1 = cn_py_round3(py + (MAXR * s1) * -1.0, PY2);
            stage[lane] = cn_hypot(x0 - x1, y0 - y1);
        }
        CN_SYNC();
        int cnt = min(64, n - i0);
        for (int c = 0; c < cnt; ++c) sum += stage[c];
        CN_SYNC();
    }
    return sum / (double)n;
}

// ---- ENV:656-743 tracker (the same block is RW:478-571), on the confirmed objects L.cfx / cfy / cfd / cft [nconf] ----------
#define TRK(f, i) T[(f) * L.tcap + (i)]
template <bool CMP = false>
__device__ __forceinline__ void tracker_stage(KP p, EnvRegs& e, const Lds& L, double* const T, int lane, int nconf, double now)
{
    // ---- ENV:656-743 tracker -----------------------------------------------------------------------
    // The tracker table shares LDS with the end-point arrays (dead from here on): bring it in now.
    if (lane < e.ntracks) {
        // HBM: one 96-byte record per track, moved as six 16-byte pieces (records are 16-byte aligned: 96 = 6 x 16)
        static_assert(CN_TF_COUNT % 2 == 0, "track record in 16-byte pieces");
        const double2* rec2 = (const double2*)(L.gtrk + lane * CN_TF_COUNT);
#pragma unroll
        for (int f = 0; f < CN_TF_COUNT; f += 2) { const double2 v = rec2[f >> 1]; T[f * L.tcap + lane] = v.x; T[(f + 1) * L.tcap + lane] = v.y; }
    }
    CN_SYNC();
    bool add_unchecked = false;
    if (CN_ABLATE(16)) { e.ntracks = 0; nconf = 0; }
    if (e.ntracks == 0) {
        for (int j = lane; j < nconf; j += 64) chk_set<CMP>(L, j, 0);
        add_unchecked = true;  // every 'o' object becomes a track
    } else {
        const int nt0 = e.ntracks;
        if (lane < nt0 && TRK(CN_TF_DQLEN, lane) > 1.0) {  // ENV:678-680 popleft
            TRK(CN_TF_D0X, lane) = TRK(CN_TF_D1X, lane); TRK(CN_TF_D0Y, lane) = TRK(CN_TF_D1Y, lane);
            TRK(CN_TF_DQLEN, lane) = 1.0;
        }
        for (int j = lane; j < nconf; j += 64) chk_set<CMP>(L, j, 0);
        CN_SYNC();
        if (nconf == 0) {
            e.ntracks = 0;  // ENV:683-686 nets out to clearing every track
        } else {
            unsigned long long alive = 0ull;
            int cur = nt0;
            // ENV:688-700: every track against every confirmed object (walls included), arg-max with the first maximum
            // winning (list.index(max)).  Tiled 8 tracks x 8 objects per pass, lane = track * 8 + object: a lane keeps the
            // best of ITS objects across the object tiles (ascending index, strict >), the 8 lanes of a track then reduce
            // with the lower index winning ties.  One pass in the usual case; a crowded env (> 8 tracks or objects) pays
            // ceil(nt / 8) * ceil(nconf / 8) passes instead of one serial wave-wide arg-max per track -- and a launch
            // lasts as long as its most crowded env.
            int mybj = 0; bool mymatch = false;                      // lane = track
            for (int t0 = 0; t0 < nt0; t0 += 8) {
                const int ti = t0 + (lane >> 3);
                double best = -1.0; int bj = 0x7fffffff;
                if (ti < nt0) {
                    const double tx = TRK(CN_TF_PX, ti), ty_ = TRK(CN_TF_PY, ti);
                    for (int oj = lane & 7; oj < nconf; oj += 8) {
                        const double u = cn_iou3(tx, ty_, cfx_at<CMP>(L, oj), cfy_at<CMP>(L, oj), 0.0505, PY2);
                        if (u > best) { best = u; bj = oj; }
                    }
                }
                // (only lane 0 of each 8-lane group is read below: a reduction tree into it -- lane + 1, lane + 2 by quad permutes,
                // lane + 4 by row_shl:4, all DPP inside the row -- instead of a 3-step xor butterfly of nine ds_bpermute round trips;
                // the order (larger IoU first, then the lower object index) is total, so the tree's shape does not matter)
#define CN_AM_STEP(CTRL) { const double ob = cn_dpp_d<CTRL, 0xf>(best, best); const int ojx = cn_dpp_i<CTRL, 0xf>(bj, bj); \
                           if (ob > best || (ob == best && ojx < bj)) { best = ob; bj = ojx; } }
                CN_AM_STEP(0xF5)      /* quad_perm [1,1,3,3] */
                CN_AM_STEP(0xEE)      /* quad_perm [2,3,2,3] */
                CN_AM_STEP(0x104)     /* row_shl:4 */
#undef CN_AM_STEP
                const u64 posm = __ballot(best > 0.0);             // bit 8 g (any lane of group g): track t0 + g matched
                const int wb = __shfl(bj, (lane & 7) * 8, 64);     // lane t0 + g <- group g's winner
                if ((lane >> 3) == (t0 >> 3)) { mybj = wb; mymatch = ((posm >> (8 * (lane & 7))) & 1ull) != 0ull; }
            }
            const u64 matchm = __ballot(mymatch && lane < nt0);
            for (int i = 0; i < nt0; ++i) {                         // ENV:702-717, the order-dependent part (scalar)
                if ((matchm >> i) & 1ull) alive |= (1ull << i);
                else if (cur > i) cur -= 1;
                else alive |= (1ull << i);
            }
            if (lane < nt0 && ((matchm >> lane) & 1ull)) {          // ENV:702-712
                const int i = lane;
                double cxj = cfx_at<CMP>(L, mybj), cyj = cfy_at<CMP>(L, mybj);
                TRK(CN_TF_PX, i) = cxj; TRK(CN_TF_PY, i) = cyj; TRK(CN_TF_DIST, i) = cfd_at<CMP>(L, mybj);
                if (TRK(CN_TF_DQLEN, i) < 1.5) { TRK(CN_TF_D1X, i) = cxj; TRK(CN_TF_D1Y, i) = cyj; TRK(CN_TF_DQLEN, i) = 2.0; }
                TRK(CN_TF_T, i) = now - TRK(CN_TF_T, i);
                chk_set<CMP>(L, mybj, 1);
            }
            CN_SYNC();
            // compact the survivors, order preserved
            double rec[CN_TF_COUNT];
            bool mine = (lane < nt0) && ((alive >> lane) & 1ull);
            if (mine) {
#pragma unroll
                for (int f = 0; f < CN_TF_COUNT; ++f) rec[f] = TRK(f, lane);
            }
            CN_SYNC();
            if (mine) {
                int slot = __popcll(alive & ((1ull << lane) - 1ull));
#pragma unroll
                for (int f = 0; f < CN_TF_COUNT; ++f) TRK(f, slot) = rec[f];
            }
            e.ntracks = __popcll(alive);
            add_unchecked = true;  // ENV:723-743
        }
    }
    CN_SYNC();
    if (add_unchecked) {
        for (int j0 = 0; j0 < nconf; j0 += 64) {
            int j = j0 + lane;
            bool want = (j < nconf) && !chk_at<CMP>(L, j) && (cft_at<CMP>(L, j) == TY_O);
            unsigned long long m = __ballot(want);
            int slot = e.ntracks + __popcll(m & ((1ull << lane) - 1ull));
            if (want) {
                if (slot < L.tcap) {
                    double cxj = cfx_at<CMP>(L, j), cyj = cfy_at<CMP>(L, j);
                    TRK(CN_TF_PX, slot) = cxj; TRK(CN_TF_PY, slot) = cyj; TRK(CN_TF_DIST, slot) = cfd_at<CMP>(L, j);
                    TRK(CN_TF_D0X, slot) = cxj; TRK(CN_TF_D0Y, slot) = cyj; TRK(CN_TF_D1X, slot) = 0.0; TRK(CN_TF_D1Y, slot) = 0.0;
                    TRK(CN_TF_T, slot) = now; TRK(CN_TF_SPEED, slot) = -1.0;
                    TRK(CN_TF_VX, slot) = 0.0; TRK(CN_TF_VY, slot) = 0.0; TRK(CN_TF_DQLEN, slot) = 1.0;
                }
            }
            int total = e.ntracks + __popcll(m);
            if (total > L.tcap) { e.status |= CN_ST_TRACK_OVERFLOW; total = L.tcap; }
            e.ntracks = total;
        }
    }
    CN_SYNC();
}

template <bool EXT, bool GT = false, bool FAIR = false, bool CMP = false, bool X2 = false, bool SF = false, bool HOIST = true>
__device__ __forceinline__ void observe(KP p, const Poly& pg, EnvRegs& e, const Lds& L, int env, int lane, int step_counter,
                        float* obs32, float* fin32, double* obs64, int* done_out, bool have_tg = false, Trig tg = Trig{0.0, 0.0, 0.0, 0.0},
                        const int wv = 0, XMail* const mb = nullptr)
{
    constexpr bool SFENCE = SF || (CMP && !FAIR);
    static_assert(!X2 || (!EXT && !GT), "two wavefronts per environment: simulated sensors, lidar-tracker mode");
    const bool w0 = !X2 || wv == 0;             // wave 0 (or the only wave): everything that is not split
    const int R = p->R, n = R - 1, K = p->K, D = n + 7 + 4 * K;
    const double MAXR = p->max_scan_range;
    double px = e.rx, py = e.ry;
    const double yaw = e.ryaw, v = e.rv, w = e.rw, now = e.clock;
    double distance_to_goal = 0.0, heading = 0.0, agent_vel_x = 0.0, agent_vel_y = 0.0;
    double sy = 0.0, cy = 0.0, ox = 0.0, oy = 0.0;
    const double h = p->room_half;
    int nnear = 0;
    u64 blkw = 0ull;
    bool wall_x = false, wall_y = false;
    if (w0) {
    // ENV:246-265
    CN_T(22);
    if (step_counter == 1) waypoint_refresh(p, pg, e, lane, px, py);
    distance_to_goal = cn_round_np64_2_t<!EXT>(dist3(px, py, e.wpx, e.wpy), PY2);   // round(np.float64, 2), ENV:255
    heading = cn_py_round2_t<!EXT>(heading_to_goal(p, e, px, py, yaw), PY2);
    CN_T(23);
    if (step_counter % 5 == 0 || distance_to_goal < e.prev_dist) waypoint_refresh(p, pg, e, lane, px, py);
    CN_T(24);
    // ENV:267-268: the angular velocity is used as the angle
    double sw_, cw_;
    if (have_tg) { sw_ = tg.sw; cw_ = tg.cw; } else cn_det_sincos_t(p->trig, w, &sw_, &cw_);
    agent_vel_x = -1.0 * (v * cw_);
    agent_vel_y = v * sw_;

    CN_T(2);
    // ---- lidar raycast (XACRO:150-178) + UTL:375-392 sanitise + UTL:110-126 end points ----------
    if (have_tg) { sy = tg.sy; cy = tg.cy; } else cn_det_sincos_t(p->trig, yaw, &sy, &cy);
    ox = fma(p->lidar_offset_x, cy, px); oy = fma(p->lidar_offset_x, sy, py);
    if constexpr (X2) { if (lane == 0) { mb->px = px; mb->py = py; mb->sy = sy; mb->cy = cy; mb->ox = ox; mb->oy = oy; } }
    }
    if constexpr (X2) CN_XBAR();          // wave 1 has advanced the pedestrians, wave 0 the robot (and published where the lidar is)
    if (w0) {
    nnear = near_peds(p, L, lane, ox, oy, sy, cy, X2 ? nullptr : &blkw);    // (X2: wave 1 reads the block words from LDS)
    wall_x = !(h - fabs(ox) > p->lidar_max + 1e-6);
    wall_y = !(h - fabs(oy) > p->lidar_max + 1e-6);
    if constexpr (X2) { if (lane == 0) { mb->nnear = nnear; mb->wall_x = wall_x; mb->wall_y = wall_y; } }
    }
    if constexpr (X2) {
        CN_XBAR();                        // the near-pedestrian list and its block words are in LDS
        if (!w0) {
            px = mb->px; py = mb->py; sy = mb->sy; cy = mb->cy; ox = mb->ox; oy = mb->oy;
            nnear = mb->nnear; wall_x = mb->wall_x != 0; wall_y = mb->wall_y != 0;
        }
    }
    CN_T(3);
    double smin = 1e300;
    float* o32 = obs32 + (size_t)env * D;
    float* f32 = fin32 ? fin32 + (size_t)env * D : nullptr;
    double* o64 = obs64 ? obs64 + (size_t)env * D : nullptr;
    // End points, rounded ranges and gradients are 3-decimal values: LDS keeps the integer thousandths
    // (x == cn_div1000(mil) bit for bit), which halves the working set and doubles the waves per CU.
    // The four table values a ray needs (its direction, and sin/cos of its end-point angle) are loaded at the top of its block.
    // Loading them one block ahead measured the same twice: in round 2, and in round 3 with __builtin_amdgcn_sched_barrier(0)
    // pinning the loads ahead of the previous block's body (the compiler otherwise sinks them back to their uses) -- 86.4 vs
    // 86.7 M one launch, 107.8 vs 107.6 M in groups, same box: the L2 round trips are not what the wavefronts wait for.
    const double* const lidc = p->lidar_c; const double* const lids = p->lidar_s;
    const double* const angs = p->ang_s; const double* const angc = p->ang_c;
    // lane d's entry of cn_create's association table, asked for now so that its L2 round trip is over when the association
    // stage wants it (one VGPR held across the ray loop)
    short assoc_pre = 0;
    if (!GT && p->assoc_fast && lane <= p->assoc_k1 + 1) assoc_pre = p->assoc_tab[lane];
    // the ray loop's three per-block parameters, once, in vector registers (see cast_ray): lidar_min, and the two rare switches
    // (float32 ranges, a zero lidar_min) as one word tested once per block
    // (HOIST = false: cn_policy_kernel_s720, whose 128-register cap has no room for them -- it reads the parameters in place)
    double lmin_reg = 0.0;
    int rare_reg = 0;
    if constexpr (HOIST) {
        lmin_reg = p->lidar_min;
        rare_reg = (p->scan_f32 != 0 ? 1 : 0) | (p->lidar_min_positive ? 0 : 2);
        asm volatile("" : "+v"(lmin_reg), "+v"(rare_reg));
    }
    for (int k = X2 ? lane + 64 * wv : lane; k < R; k += X2 ? 128 : 64) {     // (X2: the 64-ray blocks alternate between the two waves)
        // (unsigned 32-bit element offsets from the uniform table bases: one VALU instruction per address)
        double lc = 0.0, ls = 0.0, tS = 0.0, tC = 0.0;
        if (!EXT) { lc = cn_ldg(lidc, (unsigned)k); ls = cn_ldg(lids, (unsigned)k); }
        if (!GT) { tS = cn_ldg(angs, (unsigned)(R - 1 - k)); tC = cn_ldg(angc, (unsigned)(R - 1 - k)); }   // (ray 0 reads entry R - 1 and never uses it: no exec mask around two loads)
        const double t = cast_ray<EXT>(p, L, env, k, ox, oy, sy, cy, nnear, wall_x, wall_y, lc, ls, X2 ? nullptr : &blkw, HOIST ? &lmin_reg : nullptr, HOIST);
        if (k >= 1) {
            const unsigned j = (unsigned)(R - 1 - k);  // UTL:389-390 reverse, drop last
            double r = t;
            double sc;
            if constexpr (EXT) {
                if (isinf(r) && r > 0) sc = MAXR;
                else if (r != r) sc = 0.0;                // UTL:380-381 NaN -> 0 (external scans only)
                else if (r == 0.0) sc = MAXR;
                else if (r > MAXR) sc = MAXR;
                else sc = r;
            } else {
                // the simulated sensor returns +inf or a finite range >= lidar_min >= 0, never a NaN: the same chain, shorter.
                // With lidar_min > 0 (cn_create records it) a zero range cannot occur either, and +inf falls out of the min.
                sc = cn_vmin(r, MAXR);
                // (a scalar branch: one lane read, a compare, a branch; without the hoist cast_ray has done the float32 rounding)
                const int rare = HOIST ? __builtin_amdgcn_readfirstlane(rare_reg) : (p->lidar_min_positive ? 0 : 2);
                if (__builtin_expect(rare != 0, 0)) {              // scan_f32 (cast_ray left the rounding to us) and / or lidar_min == 0
                    if (rare & 1) r = (double)(float)r;
                    sc = cn_vmin(r, MAXR);
                    if ((rare & 2) && r == 0.0) sc = MAXR;
                }
            }
            smin = cn_vmin(smin, sc);
            // sin/cos(radians(j * inc) - yaw) by angle addition from a host table of sin/cos(radians(j * inc))
            if constexpr (!GT) {      // end points feed the segmentation only
            const double sa = fma(tS, cy, -(tC * sy)), ca = fma(tC, cy, tS * sy);
            pt_set<CMP>(L, j, (int)cn_round_scaled(px + (sc * ca), 1000.0, PY2), (int)cn_round_scaled(py + (sc * sa) * -1.0, 1000.0, PY2));
            L.dmil[j] = (unsigned short)(int)cn_round_scaled(sc, 1000.0, PY2);
            }
            double so = cn_np_around3_t<!EXT>(sc);  // ENV:1042
            cn_stg(o32, j, (float)so);
            if (f32) cn_stg(f32, j, (float)so);
            if (o64) cn_stg(o64, j, so);
        }
    }
    CN_T(4);
    // ENV:1011 is the only reader of the scan's minimum, and all it asks is whether ANY range lies below min_scan_range: one compare
    // and a wave vote on the per-lane minima instead of a six-step float64 DPP reduction (min(x) < c <=> some x < c; no NaNs here)
    bool too_close = __ballot(smin < p->min_scan_range) != 0ull;
    CN_SYNC();
    if constexpr (X2) {
        if (!w0 && lane == 0) mb->smin1 = too_close ? -1.0 : 1e300;
        CN_XBAR();                        // every ray's end point and range are in LDS
        if (w0) too_close = too_close || (mb->smin1 < 0.0);
    }

    if (w0 && step_counter == 0) {  // UTL:405-419 + ENV:287-294: mean spacing of the end points of an all-max scan
        // After a simulated reset the robot stands exactly at the spawn pose (reset_simulation, then 10 ms at zero twist),
        // so the value is a constant of the configuration: cn_create evaluates bbox_size() once on the device and the
        // 359 sincos pairs + the strictly serial Python sum() leave the reset path (they made a resetting wavefront 30 %
        // longer than a stepping one, and a launch lasts as long as its slowest wavefront).
        if (!EXT && p->bb_spawn_valid && px == p->spawn_x && py == p->spawn_y && yaw == p->spawn_yaw) e.bb = p->bb_spawn;
        else e.bb = bbox_size(p, L.stage, lane, n, px, py, yaw);
        double qx = cn_py_round3_t<!EXT>(px, PY2), qy = cn_py_round3_t<!EXT>(py, PY2);
        if (e.dq_len < 2) { if (e.dq_len == 0) { e.dq0x = qx; e.dq0y = qy; } else { e.dq1x = qx; e.dq1y = qy; } e.dq_len += 1; }
        else { e.dq0x = e.dq1x; e.dq0y = e.dq1y; e.dq1x = qx; e.dq1y = qy; }
    }

    CN_T(5);
    int ego_hit = 0;
    double* const T = L.trk;      // track / entry table: [CN_TF_COUNT][tcap], shares LDS with the end points
    if constexpr (!GT) {
    const int W = (n + 63) >> 6;  // 64-ray words; ray i = bit (i & 63) of word (i >> 6)
#define WORD(id, q) L.w64[__mul24((id), L.wstride) + (q)]   /* 24-bit multiply: full rate (v_mul_lo_u32 is quarter rate) */
#define BIT(id, i) ((WORD(id, (i) >> 6) >> ((i) & 63)) & 1ull)
#define PX(i) cn_div1000((double)ptx_at<CMP>(L, (i)))
#define PY(i) cn_div1000((double)pty_at<CMP>(L, (i)))
#define GNONE 0x7fffffff
    // ENV:329-346 gradients between consecutive end points, kept as integer thousandths (GNONE = None), and
    // ENV:348-367 the flag words of the type machine.  Only OCCUPIED rays (range != 0.6, typically a third of the scan)
    // have a gradient, and the machine only looks at rays whose own and next gradient exist -- so the occupied rays are
    // compacted into one list first and both computations run over that list (2 dense passes instead of 6 sparse ones).
    // Flag bits of list entries are merged into the ray-space words with LDS atomics; rays outside the list keep
    // none = 1 (their other flags are never read: the machine masks everything with ~none).
    unsigned short* occlist = L.srcidx;            // free until the type machine records alias sources
    int nocc = 0;
    u64 occraw = 0ull;                             // lane q keeps word q of the ray-space occupancy (range != 0.6)
    if (w0)
    for (int q = 0; q < W; ++q) {
        const int i = lane + 64 * q;
        const bool oc = (i < n) && (L.dmil[i] != 600);
        if (i < n) L.gq[i] = GNONE;
        const u64 bo = __ballot(oc);
        if (oc) occlist[nocc + __popcll(bo & ((1ull << lane) - 1ull))] = (unsigned short)i;
        occraw = cn_writelane_u64(occraw, bo, q);
        nocc += __popcll(bo);
    }
    if (w0 && lane < W) { WORD(M_NONE, lane) = ~0ull; WORD(M_ZERO, lane) = 0ull; WORD(M_EQ, lane) = 0ull; WORD(M_NNONE, lane) = 0ull; WORD(M_NZERO, lane) = 0ull; }
    CN_SYNC();
    if constexpr (X2) {
        if (w0 && lane == 0) mb->nocc = nocc;
        CN_XBAR();                        // the occupied-ray list
        if (!w0) nocc = mb->nocc;
    }
    // two list entries per lane and pass (128 occupied rays cover almost every scan): both entries' index and end-point reads
    // are issued before the first use, and the two divides overlap
    for (int c0 = 0; c0 < nocc; c0 += 128) {
        const int ca = X2 ? c0 + 64 * wv + lane : c0 + lane, cb = X2 ? nocc : ca + 64;      // (X2: one entry per lane, the second 64 are the other wave's)
        const bool two = !X2 && c0 + 64 < nocc;            // wave-uniform: is there a second entry for anyone?
        const bool va = ca < nocc, vb = cb < nocc;
        const int ia = va ? (int)occlist[ca] : 0, ib = vb ? (int)occlist[cb] : 0;
        const int ja = (ia == n - 1) ? 0 : ia + 1, jb = (ib == n - 1) ? 0 : ib + 1;
        const int xai = ptx_at<CMP>(L, ia), yai = pty_at<CMP>(L, ia), xaj = ptx_at<CMP>(L, ja), yaj = pty_at<CMP>(L, ja);
        int xbi = 0, ybi = 0, xbj = 0, ybj = 0;
        if (two) { xbi = ptx_at<CMP>(L, ib); ybi = pty_at<CMP>(L, ib); xbj = ptx_at<CMP>(L, jb); ybj = pty_at<CMP>(L, jb); }
        const double dya = cn_div1000((double)yai) - cn_div1000((double)yaj);
        const double qa = (dya == 0) ? 0.0 : cn_div(cn_div1000((double)xai) - cn_div1000((double)xaj), dya);   // |dy| >= 0.001 or the lane is discarded
        if (va) L.gq[ia] = (int)cn_round_scaled(qa, 1000.0, PY2);
        if (two) {
            const double dyb = cn_div1000((double)ybi) - cn_div1000((double)ybj);
            const double qb = (dyb == 0) ? 0.0 : cn_div(cn_div1000((double)xbi) - cn_div1000((double)xbj), dyb);
            if (vb) L.gq[ib] = (int)cn_round_scaled(qb, 1000.0, PY2);
        }
    }
    // last occupied ray before n-1 (ENV:356-366 `last_grad`)
    int lastnn = -1;
    if (nocc > 0) {
        lastnn = occlist[nocc - 1];
        if (lastnn == n - 1) lastnn = (nocc > 1) ? (int)occlist[nocc - 2] : -1;
    }
    CN_SYNC();
    if constexpr (X2) CN_XBAR();          // both waves' gradients
    // change of gradient c[i] = |g[i]-g[i+1]| (None if either is None), recomputed where needed;
    // ray n-1 takes `last_grad`, i.e. c[lastnn] of the last valid gradient before it.
#define CHG(a_, b_) (((a_) == GNONE || (b_) == GNONE) ? CN_NAN : fabs(cn_div1000((double)(a_)) - cn_div1000((double)(b_))))
    double clast = CN_NAN;
    if (L.gq[n - 1] != GNONE && lastnn >= 0) clast = CHG(L.gq[lastnn], L.gq[lastnn + 1]);
    CN_T(6);
    for (int c0 = 0; c0 < nocc; c0 += 128) {       // two list entries per lane and pass, reads batched
        const int ca = X2 ? c0 + 64 * wv + lane : c0 + lane, cb = X2 ? nocc : ca + 64;
        const int ia = (ca < nocc) ? (int)occlist[ca] : n, ib = (cb < nocc) ? (int)occlist[cb] : n;
        const bool va = ia < n - 1, vb = ib < n - 1;     // the machine never visits ray n-1 (ENV:380-381)
        int ga0 = GNONE, ga1 = GNONE, ga2 = GNONE, gb0 = GNONE, gb1 = GNONE, gb2 = GNONE;
        if (va) { ga0 = L.gq[ia]; ga1 = L.gq[ia + 1]; if (ia + 1 < n - 1) ga2 = L.gq[ia + 2]; }
        if (vb) { gb0 = L.gq[ib]; gb1 = L.gq[ib + 1]; if (ib + 1 < n - 1) gb2 = L.gq[ib + 2]; }
        auto flags = [&](bool v_, int i, int g0, int g1, int g2) {
            if (!v_) return;
            double c0_ = CHG(g0, g1);
            if (c0_ == c0_) {                             // none = 0: the only rays the machine reads flags of
                double c1 = (i + 1 < n - 1) ? CHG(g1, g2) : clast;
                const bool zero = (c0_ == 0), nnone = !(c1 == c1), nzero = (c1 == 0);
                const bool eq = !nnone && (fabs(c0_ - c1) == 0);
                const u64 bit = 1ull << (i & 63);
                const int qw = i >> 6;
                atomicAnd((unsigned long long*)&WORD(M_NONE, qw), ~bit);
                if (zero) atomicOr((unsigned long long*)&WORD(M_ZERO, qw), bit);
                if (eq) atomicOr((unsigned long long*)&WORD(M_EQ, qw), bit);
                if (nnone) atomicOr((unsigned long long*)&WORD(M_NNONE, qw), bit);
                if (nzero) atomicOr((unsigned long long*)&WORD(M_NZERO, qw), bit);
            }
        };
        flags(va, ia, ga0, ga1, ga2);
        if (!X2 && c0 + 64 < nocc) flags(vb, ib, gb0, gb1, gb2);
    }
#undef CHG
    CN_SYNC();
    if constexpr (X2) CN_XBAR();          // both waves' flag bits
    // ENV:372-410 object-type state machine, on the scalar unit, processed in RUNS instead of rays.
    // With z = change == 0, nz = next change == 0, nn = next change is None, eq = |change - next| == 0
    // the per-ray rules (ENV:383-410) are
    //   du == 1:  every occupied ray gets a fresh type ('w' if z else 'o') and becomes last_type;
    //             du returns to 0 after the first non-z ray whose next change is 0
    //   du == 0:  z -> 'w' fresh (last_type = it);  !z,!nz,nn -> 'o' fresh, state untouched;
    //             !z,!nz,!nn,eq -> 'w' fresh (last_type = it);
    //             !z,nz -> 'w' fresh, last_type = it, du = 1;   !z,!nz,!nn,!eq -> ALIAS of last_type, du = 1
    // so within a 64-ray word the state only changes at the rays that flip du; everything between two
    // flips is a handful of 64-bit mask operations.
    CN_T(7);
    if (w0) {
        // lane = word (W <= 16).  (Round 4 also ran this stage on the SCALAR unit, one word after the other with every mask in SGPRs:
        // 283 -> 164 vector instructions but + 670 scalar ones per observation, and 4-8 % SLOWER in every leg -- the scalar unit
        // is shared by the CU's wavefronts and its dependent 64-bit chains do not overlap; profiles/r04/stage_instr_scalar_type_machine.txt.)
        // Inside a word, du before each ray is a prefix parity of the class-A rays (they swap du)
        // combined with a fill-forward from the class-D rays (they set it to 1); ACROSS words the same two maps
        // compose (a word with a D ray outputs a constant, one without XORs its parity in), so the words' incoming du
        // -- and the last ray that set last_type below each word -- come from two 4-step shuffle scans.
        const int q = lane;
        u64 occ = 0, Z = 0, cA = 0, cB = 0, cC = 0, cD = 0;
        if (q < W && !(CN_ABLATE(2))) {
            occ = ~WORD(M_NONE, q);
            Z = WORD(M_ZERO, q) & occ;
            const u64 NZ = WORD(M_NZERO, q), NN = WORD(M_NNONE, q), E = WORD(M_EQ, q);
            const u64 nonz = occ & ~Z;
            cA = nonz & NZ;                    // 'w' fresh, du -> 1
            cB = nonz & ~NZ & NN;              // 'o' fresh, state untouched
            cC = nonz & ~NZ & ~NN & E;         // 'w' fresh
            cD = nonz & ~NZ & ~NN & ~E;        // alias, du -> 1
        }
        u64 PI = cA;                                  // inclusive prefix parity of the A rays
        PI ^= PI << 1; PI ^= PI << 2; PI ^= PI << 4; PI ^= PI << 8; PI ^= PI << 16; PI ^= PI << 32;
        const u64 PE = PI << 1;                       // exclusive
        u64 have = cD, F = PI & cD;                   // F: PI at the last D ray at or below each position
        F |= (F << 1) & ~have;  have |= have << 1;
        F |= (F << 2) & ~have;  have |= have << 2;
        F |= (F << 4) & ~have;  have |= have << 4;
        F |= (F << 8) & ~have;  have |= have << 8;
        F |= (F << 16) & ~have; have |= have << 16;
        F |= (F << 32) & ~have; have |= have << 32;
        // this word as a map of du: constant (fc = 1, fv) if it has a D ray, else du ^ fv
        int fc = cD != 0ull;
        int fv = (int)((PI >> 63) & 1ull);
        if (fc) fv = 1 ^ fv ^ (int)((PI >> (63 - __builtin_clzll(cD))) & 1ull);
        int sc_ = fc, sv_ = fv;                       // inclusive scan of the composition over the words below
        int du;                                       // du entering this word: the maps below applied to du = 0
        // (W <= 16 words sit in one row of 16 lanes: the scan steps are DPP row shifts -- one VALU instruction each -- instead of
        // ds_bpermute round trips; a lane without a lower neighbour reads the identity, so the `lane >= d` tests fall away)
        const bool rowscan = W <= 16;
        if (rowscan) {
#define CN_TM_STEP(D) { const int lc = cn_row_shr_i<D>(0, sc_), lv = cn_row_shr_i<D>(0, sv_); if (!sc_) { sc_ = lc; sv_ ^= lv; } }
            CN_TM_STEP(1) CN_TM_STEP(2) CN_TM_STEP(4) CN_TM_STEP(8)
#undef CN_TM_STEP
            du = cn_row_shr_i<1>(0, sv_);
        } else {
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) {
                const int lc = __shfl_up(sc_, d, 64), lv = __shfl_up(sv_, d, 64);
                if (lane >= d && !sc_) { sc_ = lc; sv_ ^= lv; }   // (this o lower): a constant map absorbs what is below it
            }
            du = __shfl_up(sv_, 1, 64);
            if (lane == 0) du = 0;
        }
        const u64 haveE = have << 1, FE = F << 1;     // ... strictly below
        const u64 DU = (haveE & ~(PE ^ FE)) | (~haveE & (du ? ~PE : PE));   // du before each ray
        const u64 du1 = DU & occ, du0 = ~DU & occ;
        const u64 setW = (du1 & Z) | (du0 & (Z | cA | cC));   // fresh 'w'; these rays also become last_type
        const u64 setO = du1 & ~Z;                             // fresh 'o' that becomes last_type (du == 1 only)
        u64 isw = setW, iso = setO | (du0 & cB), al = 0;
        const u64 S = setW | setO;
        // last ray that set last_type at or below each word: packed (valid, type, index), fill-forward over the lanes
        int pk = 0;
        if (S) { const int hb = 63 - __builtin_clzll(S); pk = (1 << 30) | ((((setW >> hb) & 1ull) ? TY_W : TY_O) << 16) | (64 * q + hb); }
        int below;                                    // ... strictly below this word
        if (rowscan) {
#define CN_PK_STEP(D) { const int lo_ = cn_row_shr_i<D>(0, pk); if (!pk) pk = lo_; }
            CN_PK_STEP(1) CN_PK_STEP(2) CN_PK_STEP(4) CN_PK_STEP(8)
#undef CN_PK_STEP
            below = cn_row_shr_i<1>(0, pk);
        } else {
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) {
                const int lo_ = __shfl_up(pk, d, 64);
                if (lane >= d && !pk) pk = lo_;
            }
            below = __shfl_up(pk, 1, 64);
            if (lane == 0) below = 0;
        }
        u64 alias = du0 & cD;                          // T[i] = last_type: carries that ray's range and pose
        while (alias) {
            const int t = __builtin_ctzll(alias);
            const u64 bit = 1ull << t;
            alias &= ~bit;
            const u64 prev = S & (bit - 1ull);
            int ty = below ? ((below >> 16) & 3) : TY_NONE, src = below & 0xffff;
            if (prev) { const int hb = 63 - __builtin_clzll(prev); ty = ((setW >> hb) & 1ull) ? TY_W : TY_O; src = 64 * q + hb; }
            if (ty == TY_W) isw |= bit;
            else if (ty == TY_O) iso |= bit;
            if (ty != TY_NONE) {
                // ENV:433-445: the aliased ray carries the range and pose of the ray its list was created at.  Copied right
                // here: sources are never aliased themselves, and both ends are occupied rays, so the occupancy words
                // (range != 0.6) gathered before the gradients stay valid -- no separate pass over all rays.
                const int i_ = 64 * q + t;
                L.dmil[i_] = L.dmil[src]; pt_set<CMP>(L, i_, ptx_at<CMP>(L, src), pty_at<CMP>(L, src));
            }
        }
        (void)al;
        // (the flag-word slot M_NNONE is dead now: it takes the ray-space occupancy for the order/split words)
        if (q < W) { WORD(M_ISW, q) = isw; WORD(M_ISO, q) = iso; WORD(M_NNONE, q) = occraw; }
    }
    CN_SYNC();
    CN_T(8);
    CN_T(9);
    // ENV:448-485 association of consecutive rays; brk bit i = a segment closes after ray i.
    // is_associated = round(IoU, 3) > 0 of two squares of half-size bb about consecutive end points.  The end points are
    // integer thousandths, so the decision is a function of (|dx|, |dy|) in thousandths alone as long as no pair sits within
    // floating-point noise (~1e-14) of one of the two cuts: the squares overlap iff |dx|, |dy| < T = 2000 bb, and the rounded
    // IoU is positive iff (T - |dx|)(T - |dy|) > T^2 2c / (1 + c), c = 0.0005.  Both cuts are tabulated once per
    // observation WITH A GUARD BAND of 1e-7 thousandths / 1e-6 relative -- lane d holds the largest |dy| that still
    // associates with |dx| = d -- and the per-ray test becomes integer: two differences, a table look-up, a compare
    // (about 12 instructions instead of about 200 float64 ones per 64 rays x 6 blocks).  If the guard band is touched
    // (T within 1e-7 of an integer, a table cut within 1e-6 of the threshold) or T > 254 the float test runs instead.
    int fe = n, lb = -1, nsegs0 = 0;
    bool fast_assoc = false;
    int K1 = 0;
    short* const amax = (short*)L.gq;                  // region B is free between the type machine and the confirmation
    if (!w0) { }
    else if (p->assoc_fast && e.bb == p->bb_spawn && !(CN_ABLATE(32))) {
        // the table for the spawn pose's box size (every env of a simulated run), built by cn_create: a copy
        // (its first 64 entries were requested before the ray loop, see assoc_pre)
        K1 = p->assoc_k1; fast_assoc = true;
        if (lane <= K1 + 1) amax[lane] = assoc_pre;
        for (int d = lane + 64; d <= K1 + 1; d += 64) amax[d] = p->assoc_tab[d];
        CN_SYNC();
    } else {
        const double Tm = 2000.0 * e.bb;
        K1 = (int)floor(Tm - 1e-7);
        const int K2 = (int)ceil(Tm + 1e-7);
        fast_assoc = (K2 == K1 + 1) && K1 >= 0 && K1 <= 254 && !(CN_ABLATE(32));
        if (fast_assoc) {
            const double c_hi = 0.0005 * (1.0 + 1e-6), c_lo = 0.0005 * (1.0 - 1e-6);
            const double T2 = Tm * Tm;
            const double Phi = T2 * (2.0 * c_hi) / (1.0 + c_hi), Plo = T2 * (2.0 * c_lo) / (1.0 + c_lo);
            int amb = 0;
            for (int d = lane; d <= K1 + 1; d += 64) {
                int m = -1;
                if (d <= K1) {
                    const double fx = Tm - (double)d;                     // > 0
                    double mm = fmin(ceil(Tm - Phi / fx) - 1.0, (double)K1);
                    m = mm < 0.0 ? -1 : (int)mm;
                    if (m >= 0 && !(fx * (Tm - (double)m) > Phi)) amb = 1;               // m associates, with margin
                    if (m + 1 <= K1 && !(fx * (Tm - (double)(m + 1)) < Plo)) amb = 1;     // m + 1 does not, with margin
                }
                amax[d] = (short)m;
            }
            if (__ballot(amb != 0) != 0ull) fast_assoc = false;
        }
        CN_SYNC();
    }
    if constexpr (X2) {
        if (w0 && lane == 0) { mb->fast_assoc = fast_assoc; mb->k1 = K1; }
        CN_XBAR();                        // wave 0's type machine (it rewrites aliased end points) and the association table
        if (!w0) { fast_assoc = mb->fast_assoc != 0; K1 = mb->k1; }
    }
    // (one wavefront per environment: the break words collect in a register, lane q = word q -- two v_writelane each -- and go to
    // LDS in one store after the last block; X2: each wave stores the words of its own blocks at once, as before)
    u64 brkw = 0ull;
    auto note_breaks = [&](int q, u64 bw) {
        if constexpr (X2) { if (lane == 0) WORD(M_BRK, q) = bw; }
        else brkw = cn_writelane_u64(brkw, bw, q);
        if (bw) {
            if (fe == n) fe = 64 * q + __builtin_ctzll(bw);
            u64 bl = (q == W - 1) ? (bw & ~(1ull << ((n - 1) & 63))) : bw;  // breaks before ray n-1
            if (bl) lb = 64 * q + 63 - __builtin_clzll(bl);
            nsegs0 += __popcll(bw);
        }
    };
    int q_done = 0;
    if (fast_assoc) {
        // The first six 64-ray blocks (all of them up to 385 rays) in three batched steps -- every end-point read issued
        // before the first use, then every table look-up, then the ballots -- instead of two dependent LDS round trips per
        // block: this stage is pure LDS latency (a dozen integer instructions per block).
        constexpr int QB = 6;
        int dxm[QB], dym[QB], lim[QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const int i = lane + 64 * q;
            dxm[q] = 0; dym[q] = 0;
            if (X2 && (q & 1) != wv) continue;        // (X2: even blocks are wave 0's, odd blocks wave 1's)
            if (q < W && i < n - 1) { dxm[q] = ptx_at<CMP>(L, i) - ptx_at<CMP>(L, i + 1); dym[q] = pty_at<CMP>(L, i) - pty_at<CMP>(L, i + 1); }
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) { dxm[q] = min(abs(dxm[q]), K1 + 1); lim[q] = (q < W) ? (int)amax[dxm[q]] : 0; }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            if (q < W && !(X2 && (q & 1) != wv)) {
                const int i = lane + 64 * q;
                const bool brk = (i < n) && ((i == n - 1) || abs(dym[q]) > lim[q]);
                note_breaks(q, __ballot(brk));
            }
        }
        q_done = W < QB ? W : QB;
    }
    for (int q = q_done; q < W && w0; ++q) {          // (X2: whatever the batched path does not cover stays with wave 0)
        int i = lane + 64 * q;
        bool brk = false;
        if (i < n) {
            brk = true;
            if (i < n - 1) {
                if (fast_assoc) {
                    const int dx_ = abs(ptx_at<CMP>(L, i) - ptx_at<CMP>(L, i + 1)), dy_ = abs(pty_at<CMP>(L, i) - pty_at<CMP>(L, i + 1));
                    brk = dy_ > (int)amax[min(dx_, K1 + 1)];
                } else brk = !cn_iou3_positive(PX(i), PY(i), PX(i + 1), PY(i + 1), e.bb, PY2);
            }
        }
        note_breaks(q, __ballot(brk));
    }
    if constexpr (!X2) { if (lane < W) WORD(M_BRK, lane) = brkw; }
    if constexpr (X2) {
        if (!w0 && lane == 0) { mb->fe1 = fe; mb->lb1 = lb; mb->ns1 = nsegs0; }
        CN_XBAR();                        // both waves' break words
        if (!w0) return;                  // wave 1 is done: everything from here on is lane = word / segment / track / edge
        fe = min(fe, mb->fe1); lb = max(lb, mb->lb1); nsegs0 += mb->ns1;
    }
    const int ls = lb + 1;            // start of the last segment
    // ENV:490-502 first <-> last with twice the box
    bool merge = (nsegs0 > 1) && cn_iou3_positive(PX(0), PY(0), PX(n - 1), PY(n - 1), e.bb * 2, PY2);
    CN_SYNC();
    CN_T(10);
    // order-space: position k -> ray.  merged: [0..fe] ++ [ls..n-1] ++ [fe+1..ls-1]
    const int nl = n - ls;  // length of the last segment
#define ORDER(k) (merge ? ((k) <= fe ? (k) : ((k) <= fe + nl ? ls + ((k) - fe - 1) : (k) - nl)) : (k))
    // ENV:508-566 split where free space (0.6) meets occupied; per-position words in order space.
    // Order space is the ray sequence with three ranges moved ([0..fe] ++ [ls..n-1] ++ [fe+1..ls-1]), so the words are
    // built by lane = word with 64-bit field moves (two source words + a funnel shift per range) instead of one
    // pass per word with five LDS reads per ray.
    int nseg = 0;
    u64 segw_keep = 0ull;
    {
        // 64 bits of mask `id` starting at ray `pos` (bits past the last word read as 0)
        auto extract = [&](int id, int pos) -> u64 {
            const int w = pos >> 6, bsh = pos & 63;
            u64 v = WORD(id, w) >> bsh;
            if (bsh && w + 1 < W) v |= WORD(id, w + 1) << (64 - bsh);
            return v;
        };
        // order positions [o0, o1] <- rays s0.., restricted to word q
        auto piece = [&](int id, int q, int o0, int o1, int s0) -> u64 {
            const int lo = max(o0, 64 * q), hi = min(o1, 64 * q + 63);
            if (lo > hi) return 0ull;
            const int len = hi - lo + 1;
            u64 bits = extract(id, s0 + (lo - o0));
            if (len < 64) bits &= (1ull << len) - 1ull;
            return bits << (lo - 64 * q);
        };
        auto gather = [&](int id, int q) -> u64 {
            if (!merge) return WORD(id, q);
            return piece(id, q, 0, fe, 0) | piece(id, q, fe + 1, fe + nl, ls) | piece(id, q, fe + nl + 1, n - 1, fe + 1);
        };
        // W <= 16 words (R <= 1025): lane group g = lane / 16 gathers one mask (occupancy, wall type, obstacle type,
        // breaks), word q = lane % 16
        const int g = lane >> 4, q = lane & 15;
        const int gid = g == 0 ? M_NNONE : (g == 1 ? M_ISW : (g == 2 ? M_ISO : M_BRK));
        const u64 mine = (q < W) ? gather(gid, q) : 0ull;
        // the break lanes (group 3) need their word's occupancy and the first occupancy bit of the next word
        const u64 occw = ((u64)(unsigned)__shfl((int)(unsigned)(mine >> 32), q, 64) << 32) | (u64)(unsigned)__shfl((int)(unsigned)mine, q, 64);
        const int next0 = __shfl((int)(mine & 1ull), (q + 1) & 15, 64);
        u64 segw = 0;
        if (q < W) {
            if (g == 3) {
                if (merge) {   // inside the merged first segment only its last position is a segment end
                    const int j = fe + nl;
                    const u64 after = (64 * q > j) ? ~0ull : ((64 * q + 63 <= j) ? 0ull : (~0ull << ((j - 64 * q) + 1)));
                    segw = (mine & after) | (((j >> 6) == q) ? (1ull << (j & 63)) : 0ull);
                } else segw = mine;
                const u64 occ_next = (occw >> 1) | ((u64)(unsigned)((q + 1 < W) ? next0 : 0) << 63);
                const int lastk = n - 2;                                // transitions are tested for k < n-1
                const u64 vm = (64 * q > lastk) ? 0ull : ((64 * q + 63 <= lastk) ? ~0ull : ((1ull << (lastk - 64 * q + 1)) - 1ull));
                segw |= (occw ^ occ_next) & vm;
                WORD(M_SEG, q) = segw;
            } else WORD(g == 0 ? M_OCC : (g == 1 ? M_KW : M_KO), q) = mine;
        }
        segw_keep = segw;                               // lane 48 + q keeps word q of the segment ends for the list pass below
    }
    CN_SYNC();
    CN_T(11);
    // per-word running totals: types seen before word q, last segment end before word q
    int segbase = 0;
    {   // lane = word: exclusive prefix sums / prefix max across the (at most 17) words by shuffles
        int pw = 0, po = 0, le = -1, ps = 0;
        if (lane < W) {
            const u64 sw = WORD(M_SEG, lane);
            pw = __popcll(WORD(M_KW, lane)); po = __popcll(WORD(M_KO, lane)); ps = __popcll(sw);
            if (sw) le = 64 * lane + 63 - __builtin_clzll(sw);
        }
        int sw_ = pw, so_ = po, sl_ = le, ss_ = ps;    // inclusive scans
        int pl;                                         // last segment end before this word
        if (W <= 16) {                                  // one row of 16 lanes: DPP row shifts (identity 0 / -1 where there is no lower lane)
#define CN_PF_STEP(D) { sw_ += cn_row_shr_i<D>(0, sw_); so_ += cn_row_shr_i<D>(0, so_); sl_ = max(sl_, cn_row_shr_i<D>(-1, sl_)); ss_ += cn_row_shr_i<D>(0, ss_); }
            CN_PF_STEP(1) CN_PF_STEP(2) CN_PF_STEP(4) CN_PF_STEP(8)
#undef CN_PF_STEP
            pl = cn_row_shr_i<1>(-1, sl_);
        } else {
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) {
                const int aw = __shfl_up(sw_, d, 64), ao = __shfl_up(so_, d, 64), al_ = __shfl_up(sl_, d, 64), as_ = __shfl_up(ss_, d, 64);
                if (lane >= d) { sw_ += aw; so_ += ao; sl_ = max(sl_, al_); ss_ += as_; }
            }
            pl = __shfl_up(sl_, 1, 64);
        }
        segbase = ss_ - ps;                             // lane q: segment ends before word q
        nseg = __builtin_amdgcn_readlane(ss_, W - 1);   // ... and the inclusive total of the last word = the number of segments
        if (lane < W) {
            L.wbase[lane] = sw_ - pw; L.wbase[L.wstride + lane] = so_ - po; L.wbase[2 * L.wstride + lane] = lane ? pl : -1;
        }
    }
    CN_SYNC();
    CN_T(12);
    // ENV:568-620 confirmation.  The segment ends (a few dozen at most) are first compacted into one list, in order,
    // so that ONE pass with lane = segment evaluates them all (instead of one pass per 64-ray word, each executing the
    // whole body for its handful of segment-end lanes).  The list lives in the flag words, dead since the type machine.
    int nconf = 0;
    unsigned short* seglist = (unsigned short*)&WORD(M_NONE, 0);
    const int segcap = min(64, 20 * W);
    for (int c0 = 0; c0 < ((CN_ABLATE(4)) ? 0 : nseg); c0 += segcap) {
        for (int q = 0; q < W; ++q) {
            // (word q of the segment ends: still in the register of the lane that built it -- two lane reads with a uniform index
            // instead of an LDS round trip per word)
            const u64 sw = cn_readlane_u64(segw_keep, 48 + q);
            const int r = __builtin_amdgcn_readlane(segbase, q) + __popcll(sw & ((1ull << lane) - 1ull)) - c0;   // (q is wave-uniform: v_readlane, no LDS permute)
            if (((sw >> lane) & 1ull) && r >= 0 && r < segcap) seglist[r] = (unsigned short)(lane + 64 * q);
        }
        CN_SYNC();
        int obj = -1, m = 0; double dm = 0.0; int dmi = 0;
        if (lane < min(segcap, nseg - c0)) {
            const int k = seglist[lane], q = k >> 6, bl = k & 63;
            const u64 sw = WORD(M_SEG, q);
            const u64 low = sw & ((1ull << bl) - 1ull);
            const int pe = low ? (64 * q + 63 - __builtin_clzll(low)) : L.wbase[2 * L.wstride + q];  // previous segment end
            const int k0 = pe + 1, len = k - pe;
            const u64 incl = (bl == 63) ? ~0ull : ((2ull << bl) - 1ull);
            int cw = L.wbase[q] + __popcll(WORD(M_KW, q) & incl);
            int co = L.wbase[L.wstride + q] + __popcll(WORD(M_KO, q) & incl);
            if (pe >= 0) {
                const int qp = pe >> 6, bp = pe & 63;
                const u64 inclp = (bp == 63) ? ~0ull : ((2ull << bp) - 1ull);
                cw -= L.wbase[qp] + __popcll(WORD(M_KW, qp) & inclp);
                co -= L.wbase[L.wstride + qp] + __popcll(WORD(M_KO, qp) & inclp);
            }
            const int no = co, nw = cw, nn = len - no - nw;
            const bool occ = (WORD(M_OCC, q) >> bl) & 1ull;  // segments are homogeneous after the split
            if (occ && len >= 4) {
                m = ORDER(k0 + len / 2);  // ENV:577 Python-2 integer division
                dmi = (int)L.dmil[m];
                dm = cn_div1000((double)dmi);
                int est = 3 + (int)floor(cn_div(29 * (p->max_scan_range - dm), p->max_scan_range - p->min_scan_range));   // cn_create: max > min
                int mn = len < est ? len : est;
                double score = cn_div((double)no, (double)mn);          // mn >= 3
                int kinds = (no > 0) + (nw > 0) + (nn > 0);
                if (kinds > 1) {
                    if (score >= 0.5) obj = (no > nw) ? TY_O : TY_W;
                    else if (len <= est) obj = (no > nw) ? TY_O : TY_W;
                    else obj = TY_W;
                } else {
                    int lim = nseg < est ? nseg : est;  // ENV:608,615: compares with the NUMBER of segments
                    if (len > lim) obj = (nw > 0) ? TY_W : TY_O;
                }
            }
        }
        const u64 cwd = __ballot(obj >= 0);
        if (obj >= 0) {
            int slot = nconf + __popcll(cwd & ((1ull << lane) - 1ull));
            if (slot < p->max_conf) conf_set<CMP>(L, slot, obj, ptx_at<CMP>(L, m), pty_at<CMP>(L, m), dmi);
        }
        nconf += __popcll(cwd);
        CN_SYNC();
    }
    if (nconf > p->max_conf) { nconf = p->max_conf; e.status |= CN_ST_CONF_OVERFLOW; }
#undef ORDER
#undef BIT
#undef WORD
#undef PX
#undef PY
#undef GNONE
    e.nconf = nconf;
    CN_SYNC();

    // ENV:637-654
    int n_obst = 0;
    for (int j = lane; j < nconf; j += 64) {
        if (cft_at<CMP>(L, j) == TY_O) { n_obst += 1; if (cfd_at<CMP>(L, j) < 0.140) ego_hit = 1; }
    }
    n_obst = cn_wave_sum_i(n_obst);
    ego_hit = cn_wave_max_i(ego_hit);
    if (n_obst > 0) e.obst_steps += 1;

    CN_T(13);
    tracker_stage<CMP>(p, e, L, T, lane, nconf, now);
    } else {
        // ---- risk_mode gt (SURVEY 7 "two risk-feature modes", include/crowdnav.h): rows A21-A24 fed with the simulator's own
        // pedestrians instead of tracked lidar blobs -- the north star's "K-nearest perceived-risk feature extraction".
        // An entry = a pedestrian within lidar range of the lidar origin and in line of sight (no other disc cuts the segment
        // origin -> its nearest surface point), in id order.  lane = pedestrian; the blockers are walked as a bit mask.
        //   pose = surface point nearest to the ROBOT, rounded like a scan end point; dist = |c - p| - r, rounded
        //   vel  = -(true velocity) (ENV:806-811 subtract new from old); speed = |true velocity|; slot T holds the pedestrian id
        const int P_ = p->P;
        const double r_ = p->ped_radius, lim_ = p->lidar_max + r_, lim2_ = lim_ * lim_;
        int nt_ = 0;
        u64 hitm = 0ull;
        for (int i0 = 0; i0 < P_; i0 += 64) {
            const int i = i0 + lane;
            bool inr = false, seg = false, blocked = false;
            double cx = 0.0, cyy = 0.0, wx = 0.0, wy = 0.0, len2 = 0.0;
            if (i < P_) {
                cx = L.ped[2 * i]; cyy = L.ped[2 * i + 1];
                const double ocx = cx - ox, ocy = cyy - oy;
                const double dd2 = fma(ocx, ocx, ocy * ocy);
                inr = dd2 <= lim2_;
                if (inr) {
                    const double dd = cn_sqrt(dd2);
                    if (dd > r_) {                            // (origin inside the disc: nothing can stand in front of it)
                        seg = true;
                        const double k_ = (dd - r_) / dd;     // origin -> nearest surface point = k_ * oc
                        wx = k_ * ocx; wy = k_ * ocy;
                        len2 = fma(wx, wx, wy * wy);
                    }
                }
            }
            for (int j0 = 0; j0 < P_; j0 += 64) {
                const int jl = j0 + lane;
                bool jin = false;
                if (jl < P_) { const double qx = L.ped[2 * jl] - ox, qy = L.ped[2 * jl + 1] - oy; jin = fma(qx, qx, qy * qy) <= lim2_; }
                u64 jm = __ballot(jin);
                while (jm) {
                    const int j = j0 + __builtin_ctzll(jm);
                    jm &= jm - 1ull;
                    const double qx = L.ped[2 * j] - ox, qy = L.ped[2 * j + 1] - oy;
                    if (seg && j != i) {
                        double tau = (len2 > 0.0) ? fma(qx, wx, qy * wy) / len2 : 0.0;
                        tau = fmin(fmax(tau, 0.0), 1.0);
                        const double ex = fma(tau, wx, -qx), ey = fma(tau, wy, -qy);
                        if (fma(ex, ex, ey * ey) < r_ * r_) blocked = true;
                    }
                }
            }
            const bool vis = inr && !blocked;
            const u64 vm = __ballot(vis);
            const int slot = nt_ + __popcll(vm & ((1ull << lane) - 1ull));
            bool close_ = false;
            if (vis && slot < L.tcap) {
                const double dx = cx - px, dy = cyy - py;
                const double dp = cn_sqrt(fma(dx, dx, dy * dy));
                double sx_ = cx, sy_ = cyy;
                if (dp > 0.0) { sx_ = cx - r_ * (dx / dp); sy_ = cyy - r_ * (dy / dp); }
                const double dist = cn_py_round3(dp - r_, PY2);
                const double vx = L.pedv[2 * i], vy = L.pedv[2 * i + 1];
                TRK(CN_TF_PX, slot) = cn_py_round3(sx_, PY2); TRK(CN_TF_PY, slot) = cn_py_round3(sy_, PY2); TRK(CN_TF_DIST, slot) = dist;
                TRK(CN_TF_D0X, slot) = 0.0; TRK(CN_TF_D0Y, slot) = 0.0; TRK(CN_TF_D1X, slot) = 0.0; TRK(CN_TF_D1Y, slot) = 0.0;
                TRK(CN_TF_T, slot) = (double)i; TRK(CN_TF_SPEED, slot) = cn_sqrt(fma(vx, vx, vy * vy));
                TRK(CN_TF_VX, slot) = -vx; TRK(CN_TF_VY, slot) = -vy; TRK(CN_TF_DQLEN, slot) = 0.0;
                close_ = dist < 0.140;
            }
            hitm |= __ballot(close_);
            nt_ += __popcll(vm);
        }
        if (nt_ > L.tcap) { e.status |= CN_ST_TRACK_OVERFLOW; nt_ = L.tcap; }
        e.ntracks = nt_; e.nconf = nt_;
        if (nt_ > 0) e.obst_steps += 1;
        ego_hit = hitm != 0ull;
        CN_SYNC();
    }
    CN_T(14);
    // ENV:745-760 speed of the tracks matched in this call
    if (lane < e.ntracks && TRK(CN_TF_DQLEN, lane) > 1.5) {
        double dc = cn_hypot(TRK(CN_TF_D0Y, lane) - TRK(CN_TF_D1Y, lane), TRK(CN_TF_D0X, lane) - TRK(CN_TF_D1X, lane));
        TRK(CN_TF_SPEED, lane) = cn_div_z(dc, TRK(CN_TF_T, lane));
    }
    CN_SYNC();

    // default K x [px, py, 0, 0] (ENV:273)
    for (int i = lane; i < 4 * K; i += 64) {
        int c = i & 3;
        L.tail[7 + i] = (c == 0) ? px : (c == 1 ? py : 0.0);
    }
    if (lane < K) L.kidx[lane] = -1;
    e.nent = 0;
    CN_SYNC();

    CN_T(15);
    // ---- ENV:769-996 collision cone / collision probability / top-K -------------------------------
    if (e.dq_len == 2 && !(CN_ABLATE(8))) {
        const double ts = e.ts;
        if (ts == 0.0) e.status |= CN_ST_DT_ZERO;
        const int nt = e.ntracks;
        double vx_ = cn_div_z(e.dq1x - e.dq0x, ts), vy_ = cn_div_z(e.dq1y - e.dq0y, ts);  // UTL:227-236
        double agent_vel = cn_sqrt(vx_ * vx_ + vy_ * vy_);
        double obstacle_vel = (nt == 0) ? 0.0 : TRK(CN_TF_SPEED, 0);  // ENV:787-793
        // ENV:800-815: per-track velocity; the relative-motion end point of the LAST track survives
        if (!GT && lane < nt && TRK(CN_TF_DQLEN, lane) > 1.5) {
            double chx = TRK(CN_TF_D0X, lane) - TRK(CN_TF_D1X, lane), chy = TRK(CN_TF_D0Y, lane) - TRK(CN_TF_D1Y, lane);
            TRK(CN_TF_VX, lane) = cn_div_z(chx, ts); TRK(CN_TF_VY, lane) = cn_div_z(chy, ts);
        }
        double vo_x = e.dq1x, vo_y = e.dq1y;
        if (nt > 0) {
            int l = nt - 1;
            double chx = 0.0, chy = 0.0;
            if (GT) { chx = TRK(CN_TF_VX, l) * ts; chy = TRK(CN_TF_VY, l) * ts; }   // displacement over the agent's timestep, old - new
            else if (TRK(CN_TF_DQLEN, l) > 1.5) { chx = TRK(CN_TF_D0X, l) - TRK(CN_TF_D1X, l); chy = TRK(CN_TF_D0Y, l) - TRK(CN_TF_D1Y, l); }
            vo_x = e.dq1x + chx; vo_y = e.dq1y + chy;
        }
        CN_SYNC();
        // UTL:251-293 collision point per track; lanes = the 64 ring edges
        const double a0x = e.dq0x, a0y = e.dq0y;
        double gradient = (vo_y == 0.0) ? 0.0 : cn_div(vo_x - a0x, vo_y) - a0y;  // UTL:261 precedence as written
        double bb0 = a0x - (gradient * a0y);
        int hi = (int)ceil(a0x + 3.5), lo = (int)floor(a0x - 3.5);
        double ego_max = 0.0;
        // The candidate segments agent -> (x2, y2), x2 = hi, hi-1, ... > lo (at most 8 of them, UTL:264-291), are the
        // same for every track.  The polygon lies inside its circumscribed circle, so a segment whose closest
        // point to the track's centre is farther than the radius (with slack) cannot touch any edge -> same
        // "empty" result as running the ring test.  That pre-rejection is evaluated for 8 tracks x 8 candidates
        // at once (lane = track * 8 + candidate).
        // Two phases per chunk of tracks (a launch lasts as long as its slowest wavefront, and crowded envs used to
        // spend ~2 k ticks PER TRACK here in one serial chain):
        //   1. per track, only what needs lane = polygon edge: the ring test of the surviving candidates, in x2 order;
        //      the two hit lanes drop their intersection points into LDS (the confirmed-object arrays are dead by now);
        //   2. lane = track: distances to the hit points, ttc, ego score, CP -- the hypots and divides of ALL tracks at once.
        // ENV:818-860 carries `ego` from one track to the next only when the relative speed is exactly 0, and then every
        // track's value is 0 by induction (it starts at 0 and a track without a collision point resets it to 0).
        const double rv = agent_vel - obstacle_vel;
        const int hcap = min(p->max_conf, 64);         // tracks per chunk: one lane each, 4 doubles of LDS each
        double* const hitp = L.cfx;                     // [4][hcap]: first hit x, y, second hit x, y (L.cfx = the start of the confirmed-object arrays in both layouts)
        for (int c0 = 0; c0 < nt; c0 += hcap) {
            const int c1 = min(nt, c0 + hcap);
            u64 nearm = 0, hasm = 0;
            for (int i = c0; i < c1; ++i) {  // ENV:818-860, UTL:251-293
                const int r_ = i - c0;
                if ((r_ & 7) == 0) {
                    const int ti = i + (lane >> 3), x2l = hi - (lane & 7);
                    bool nearc = false;
                    if (ti < c1 && x2l > lo) {
                        const double ctx = TRK(CN_TF_PX, ti), cty = TRK(CN_TF_PY, ti);
                        const double y2l = ((double)x2l * gradient) + bb0;
                        // squared distance from the centre to the segment, compared with r^2 without a divide
                        const double ex = (double)x2l - a0x, ey = y2l - a0y, fx = ctx - a0x, fy = cty - a0y;
                        const double ee = ex * ex + ey * ey, ff = fx * fx + fy * fy, num = fx * ex + fy * ey;
                        const double rr2 = 0.178 * 0.178 * 1.000001;
                        bool far_;
                        if (num <= 0.0) far_ = ff > rr2;                                   // closest point is the agent
                        else if (num >= ee) far_ = (fx - ex) * (fx - ex) + (fy - ey) * (fy - ey) > rr2;  // ... the far end
                        else far_ = (ff - rr2) * ee > num * num * 1.000001;                // ... the foot of the perpendicular
                        nearc = !far_;
                    }
                    nearm = __ballot(nearc);
                }
                unsigned cand = (unsigned)((nearm >> (8 * (r_ & 7))) & 0xffull);   // bit c <-> x2 = hi - c
                // cn_config.geos_untyped_empty (GEOS <= 3.8): a candidate that misses prints 'GEOMETRYCOLLECTION EMPTY',
                // UTL:279's comparison is true, `.geoms[0]` raises and the search ends with None -- only x2 = hi counts
                if (p->geos_untyped_empty) cand &= 1u;
                if (!cand) continue;
                const double tx = TRK(CN_TF_PX, i), ty_ = TRK(CN_TF_PY, i);
                while (cand) {
                    const int c = __builtin_ctz(cand);
                    cand &= cand - 1u;
                    const int x2 = hi - c;
                    double y2 = ((double)x2 * gradient) + bb0;
                    double hx = 0.0, hy = 0.0;
                    unsigned long long m = ring_segment(pg, lane, tx, ty_, 0.178, a0x, a0y, (double)x2, y2, &hx, &hy);
                    int cnt = __popcll(m);
                    if (cnt == 0) continue;
                    if (cnt == 1) break;  // Point has no .geoms -> None
                    const int l1 = __ffsll((long long)m) - 1;
                    const int l2 = __ffsll((long long)(m & (m - 1ull))) - 1;
                    if (lane == l1) { hitp[r_] = hx; hitp[hcap + r_] = hy; }
                    if (lane == l2) { hitp[2 * hcap + r_] = hx; hitp[3 * hcap + r_] = hy; }
                    hasm |= 1ull << r_;
                    break;
                }
            }
            CN_SYNC();
            {
                const int i = c0 + lane;
                double ego = 0.0;
                bool ttc0 = false;
                if (i < c1) {
                    const double td = TRK(CN_TF_DIST, i);
                    const double gcp = (td > p->max_scan_range) ? 0.0 : cn_div(p->max_scan_range - td, p->max_scan_range - p->min_scan_range);
                    double cpv;
                    if ((hasm >> lane) & 1ull) {
                        const double d1 = cn_hypot(a0x - hitp[lane], a0y - hitp[hcap + lane]);
                        const double d2 = cn_hypot(a0x - hitp[2 * hcap + lane], a0y - hitp[3 * hcap + lane]);
                        const double dcp = cn_vmin(d1, d2);
                        if (rv == 0) { cpv = 1.0 * gcp; ego = 0.0; }
                        else {
                            // rv != 0 here: a difference of two finite speeds in a simulated run (never denormal, never infinite).
                            // External clocks can repeat a time stamp -> an infinite track speed: IEEE division there.
                            double ttc = EXT ? dcp / rv : cn_div(dcp, rv);
                            if (ttc == 0.0) { ttc0 = true; ego = 1.0; }
                            else ego = fmin(1.0, EXT ? 0.15 / ttc : cn_div(0.15, ttc));  // UTL:319
                            cpv = 0.5 * ego + 0.5 * gcp;
                        }
                    } else { ego = 0.0; cpv = 0.5 * 0.0 + 0.5 * gcp; }
                    L.cpv[i] = cpv;
                }
                if (__ballot(ttc0) != 0ull) e.status |= CN_ST_TTC_ZERO;
                // ENV: `if i == 0 or ego > ego_max: ego_max = ego`, folded left to right (a NaN is only ever kept as the first value)
                const double first = lane_d(ego, 0);
                const bool rest = (i < c1) && !(c0 == 0 && lane == 0) && (ego == ego);
                const double mrest = cn_wave_max_d(rest ? ego : -INFINITY);
                if (c0 == 0) ego_max = first;
                if (mrest > ego_max) ego_max = mrest;
            }
            CN_SYNC();
        }
        e.nent = nt;
        CN_SYNC();
        // (both fields are written once, after the branch: a store in each arm gets merged into one store through a pointer phi,
        // which keeps the whole EnvRegs field pair on the stack)
        double cprob_new = 0.0, ego_new = 0.0;         // ENV:862-876: no tracks
        if (nt != 0) {  // ENV:878-905: stable descending sort, keep the LAST K
            ego_new = ego_max;
            int first = nt > K ? nt - K : 0;
            int rank = -1; double mycp = 0.0;
            if (lane < nt) mycp = L.cpv[lane];
            if (nt <= 8) {
                // up to 8 tracks (almost every call): every ordered pair at once, lane = 8 i + j -- one compare and a wave vote
                // instead of a serial pass over the tracks with an LDS read per iteration; track i's rank = the set bits of byte i
                const int ti = lane >> 3, tj = lane & 7;
                bool beats = false;
                if (ti < nt && tj < nt) { const double ci = L.cpv[ti], cj = L.cpv[tj]; beats = (cj > ci) || (cj == ci && tj < ti); }
                const u64 bm = __ballot(beats);
                if (lane < nt) rank = __popc((unsigned)(bm >> (8 * lane)) & 0xffu);
            } else if (lane < nt) {
                rank = 0;
                for (int j = 0; j < nt; ++j) {
                    double c = L.cpv[j];
                    rank += (c > mycp) || (c == mycp && j < lane);
                }
            }
            if (lane < nt) {
                if (rank >= first) {
                    int kk = rank - first;
                    L.tail[7 + 4 * kk + 0] = TRK(CN_TF_PX, lane); L.tail[7 + 4 * kk + 1] = TRK(CN_TF_PY, lane);
                    L.tail[7 + 4 * kk + 2] = TRK(CN_TF_VX, lane); L.tail[7 + 4 * kk + 3] = TRK(CN_TF_VY, lane);
                    L.kidx[kk] = GT ? (int)TRK(CN_TF_T, lane) : lane;     // gt: pedestrian id
                }
            }
            unsigned long long mf = __ballot(rank == first);
            cprob_new = bcast_d(mycp, __ffsll((long long)mf) - 1);
        }
        e.cprob = cprob_new; e.ego = ego_new;
        // ENV:990-996
        e.dq0x = e.dq1x; e.dq0y = e.dq1y; e.dq_len = 1;
        if (!GT && lane < nt) TRK(CN_TF_T, lane) = now;
    }
    CN_T(16);
    // ENV:998-1005 safety counters
    if (ego_hit) e.ego_viol += 1;
    if (e.ego > 0.4) e.social_viol += 1;
    // ENV:1011-1023 done
    if (!e.done) {
        if (too_close) e.done = 1;
        if (in_box(px, py, p->goal_x, p->goal_y, p->goal_eps)) e.done = 1;
        if (step_counter >= p->max_steps) e.done = 1;
    }
    // ENV:1025-1042 observation tail
    if (lane < 7) {   // one rounding pass, lane = tail slot (heading and distance are already rounded)
        const double v = lane == 2 ? px : lane == 3 ? py : lane == 4 ? yaw : lane == 5 ? agent_vel_x : agent_vel_y;
        const double r = cn_py_round3_t<!EXT>(v, PY2);
        L.tail[lane] = lane == 0 ? heading : lane == 1 ? distance_to_goal : r;
    }
    CN_SYNC();
    for (int i = lane; i < 7 + 4 * K; i += 64) {
        double so = cn_np_around3_t<!EXT>(L.tail[i]);
        L.tail[i] = so;
        o32[n + i] = (float)so;
        if (f32) f32[n + i] = (float)so;
        if (o64) o64[n + i] = so;
    }
    CN_T(17);
    // tracker table back to HBM (its LDS space is reused by the next observation's end points)
    if (lane < e.ntracks) {
#pragma unroll
        for (int f = 0; f < CN_TF_COUNT; f += 2)
            ((double2*)(L.gtrk + lane * CN_TF_COUNT))[f >> 1] = make_double2(L.trk[f * L.tcap + lane], L.trk[(f + 1) * L.tcap + lane]);
    }
    CN_SYNC();
    *done_out = e.done;
}

// ---- obs_layout 2: environment_stage_1_nobonus_realworld.py ("RW"), the 370-input physical-robot variant ------
// 359 UNROUNDED sanitised ranges + heading + distance + rounded (x, y) + the constant yaw 3.14 + rounded twist features + pose and
// velocity of the ONE tracked obstacle with the highest collision probability.  Its segmentation is an older pipeline than ENV's:
// free-space rays are filtered out BEFORE the gradients (so neighbours are the next OCCUPIED rays), no way-points, the cone is cast
// against a ring of radius min_scan_range, and the top-K rule is "the maximum, last index among ties".  This is the variant that
// reads a physical lidar through cn_observe_external, one robot at a time: the lane-parallel parts are the ones that fall out for
// free (ray cast, filter, gradients, association, tracker); the type machine and the confirmation walk their short lists on the
// scalar unit.  L.rw*: the layout's own LDS region (12 bytes per ray).
struct RwLds { unsigned short* fi; int* g; unsigned char* tt; unsigned short* ts; unsigned short* es; unsigned char* et; };

__device__ __forceinline__ double rw_heading(KP p, double px, double py, double yaw)
{   // RW:186-200 (starting_point added to the position)
    double cx = px + p->start_x, cy = py + p->start_y;
    double h = cn_atan2_t(p->trig, p->goal_y - cy, p->goal_x - cx) - yaw;
    if (h > CN_PI) h -= 2 * CN_PI;
    else if (h < -CN_PI) h += 2 * CN_PI;
    return h;
}
__device__ __forceinline__ double rw_distance(KP p, double px, double py)
{   // RW:161-173
    return dist3(px + p->start_x, py + p->start_y, p->goal_x, p->goal_y);
}

template <bool EXT>
__device__ __forceinline__ void observe_realworld(KP p, const Poly& pg, EnvRegs& e, const Lds& L, const RwLds& Q, int env, int lane,
                                                  int step_counter, float* obs32, float* fin32, double* obs64, int* done_out)
{
    const int R = p->R, n = R - 1, D = n + 11;
    const double MAXR = p->max_scan_range;
    const double px = e.rx, py = e.ry, yaw = e.ryaw, v = e.rv, w = e.rw, now = e.clock;
    const double distance_to_goal = cn_round_np64_2_t<false>(rw_distance(p, px, py), PY2);      // RW:209 round(np.float64, 2)
    const double heading = cn_py_round2(rw_heading(p, px, py, yaw), PY2);            // RW:210
    double sw_, cw_;
    cn_det_sincos_t(p->trig, w, &sw_, &cw_);
    const double agent_vel_x = -1.0 * (v * cw_), agent_vel_y = v * sw_;         // RW:211-212
    double clx = px, cly = py, clvx = 0.0, clvy = 0.0;                          // RW:215-216 closest obstacle pose / velocity

    // ---- lidar + RW:220-225: sanitise, end points; the observation carries the UNROUNDED range ----------------------------------
    double sy, cy;
    cn_det_sincos_t(p->trig, yaw, &sy, &cy);
    const double ox = fma(p->lidar_offset_x, cy, px), oy = fma(p->lidar_offset_x, sy, py);
    const double h = p->room_half;
    const int nnear = near_peds(p, L, lane, ox, oy, sy, cy);
    const bool wall_x = !(h - fabs(ox) > p->lidar_max + 1e-6);
    const bool wall_y = !(h - fabs(oy) > p->lidar_max + 1e-6);
    double smin = 1e300;
    float* o32 = obs32 + (size_t)env * D;
    float* f32 = fin32 ? fin32 + (size_t)env * D : nullptr;
    double* o64 = obs64 ? obs64 + (size_t)env * D : nullptr;
    for (int k = lane; k < R; k += 64) {
        double lc = 0.0, ls = 0.0;
        if (!EXT) { lc = p->lidar_c[k]; ls = p->lidar_s[k]; }
        const double t = cast_ray<EXT>(p, L, env, k, ox, oy, sy, cy, nnear, wall_x, wall_y, lc, ls);
        if (k >= 1) {
            const int j = R - 1 - k;
            double sc;
            if (isinf(t) && t > 0) sc = MAXR;
            else if (t != t) sc = 0.0;
            else if (t == 0.0) sc = MAXR;
            else if (t > MAXR) sc = MAXR;
            else sc = t;
            smin = cn_vmin(smin, sc);
            const double tS = p->ang_s[j], tC = p->ang_c[j];
            const double sa = fma(tS, cy, -(tC * sy)), ca = fma(tC, cy, tS * sy);
            L.ptx[j] = (int)cn_round_scaled(px + (sc * ca), 1000.0, PY2);
            L.pty[j] = (int)cn_round_scaled(py + (sc * sa) * -1.0, 1000.0, PY2);
            // bit 15: not a ground-truth (free-space) ray -- RW:249-255 tests the UNROUNDED range against max_scan_range
            L.dmil[j] = (unsigned short)((int)cn_round_scaled(sc, 1000.0, PY2) | ((sc != MAXR) ? 0x8000 : 0));
            o32[j] = (float)sc;
            if (f32) f32[j] = (float)sc;
            if (o64) o64[j] = sc;
        }
    }
    smin = cn_wave_min_d(smin);
    CN_SYNC();
    if (step_counter == 0) {                                                    // RW:229-237
        if (!EXT && p->bb_spawn_valid && px == p->spawn_x && py == p->spawn_y && yaw == p->spawn_yaw) e.bb = p->bb_spawn;
        else e.bb = bbox_size(p, L.stage, lane, n, px, py, yaw);
        double qx = cn_py_round3(px, PY2), qy = cn_py_round3(py, PY2);
        if (e.dq_len < 2) { if (e.dq_len == 0) { e.dq0x = qx; e.dq0y = qy; } else { e.dq1x = qx; e.dq1y = qy; } e.dq_len += 1; }
        else { e.dq0x = e.dq1x; e.dq0y = e.dq1y; e.dq1x = qx; e.dq1y = qy; }
    }
#define PX(i) cn_div1000((double)L.ptx[i])
#define PY(i) cn_div1000((double)L.pty[i])
    // ---- RW:257-266 the filtered list: rays that are not free space, in ray order -------------------------------------------------
    int F = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool oc = (i < n) && (L.dmil[i] & 0x8000);
        const u64 bo = __ballot(oc);
        if (oc) Q.fi[F + __popcll(bo & ((1ull << lane) - 1ull))] = (unsigned short)i;
        F += __popcll(bo);
    }
    CN_SYNC();
    // ---- RW:268-282 gradients between CONSECUTIVE FILTERED end points (the last one against the first) ---------------------------
    for (int c = lane; c < F; c += 64) {
        const int i = Q.fi[c], j = Q.fi[(c == F - 1) ? 0 : c + 1];
        const double dy = PY(i) - PY(j);
        const double q = (dy == 0) ? 0.0 : (PX(i) - PX(j)) / dy;
        Q.g[c] = (int)cn_round_scaled(q, 1000.0, PY2);
    }
    CN_SYNC();
    // ---- RW:284-333 change of gradient + the object-type machine ------------------------------------------------------------------
    // change c(i) = |g[i] - g[i+1]| for i < F-1 and c(F-1) = c(F-2) (`last_grad`); the machine never types entry F-1.
    // Round 5: lane = entry, 64 entries at a time (it was one serial loop on lane 0: ~3 600 instructions a step, the largest single
    // piece of this layout's 4 x cost).  With z = (c(i) == 0), nz = (c(i+1) == 0), eq = (|c(i) - c(i+1)| == 0) entry i maps the
    // machine's `du` (0 / 1) to
    //     z: du                         'w' fresh, becomes last_type
    //    !z, du = 0:  nz or eq -> 0     'w' fresh, becomes last_type;       else -> 1   ALIAS of last_type (its entry's range and pose)
    //    !z, du = 1:  nz -> 0, else 1   'o' fresh, becomes last_type
    // -- a map {0, 1} -> {0, 1} per entry: an inclusive scan of map composition over the lanes (6 shuffle steps) gives every entry
    // its incoming du; the last entry that set last_type below each one is a max-scan of (index, type) keys.
    {
        int du_in = 0, key_in = -1;                       // carried over the 64-entry blocks: du, and (index << 2 | type) of the last setter
        for (int i0 = 0; i0 + 1 < F; i0 += 64) {
            const int i = i0 + lane;
            const bool act = i + 1 < F;
            int f0 = 0, f1 = 1;                           // this entry's map (identity past the end)
            bool z = false, nzq = false;
            if (act) {
                const double g0 = cn_div1000((double)Q.g[i]), g1 = cn_div1000((double)Q.g[i + 1]);
                const double c0 = fabs(g0 - g1);
                const double c1 = (i + 2 < F) ? fabs(g1 - cn_div1000((double)Q.g[i + 2])) : c0;
                z = c0 == 0;
                const bool nz = c1 == 0, eq = fabs(c0 - c1) == 0;
                nzq = nz || eq;
                if (!z) { f0 = nzq ? 0 : 1; f1 = nz ? 0 : 1; }
            }
            int s0 = f0, s1 = f1;                         // inclusive composition of the maps of lanes 0 .. lane (earlier entries first)
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int p0 = __shfl_up(s0, d, 64), p1 = __shfl_up(s1, d, 64);        // the block of maps ending d lanes below
                if (lane >= d) { const int n0 = p0 ? s1 : s0, n1 = p1 ? s1 : s0; s0 = n0; s1 = n1; }   // (mine after theirs)(x) = mine(theirs(x))
            }
            int e0 = __shfl_up(s0, 1, 64), e1 = __shfl_up(s1, 1, 64);                  // exclusive: the maps strictly below this lane
            if (lane == 0) { e0 = 0; e1 = 1; }
            const int du = du_in ? e1 : e0;               // du on entry i
            const bool fresh_w = act && (z || (du == 0 && nzq));
            const bool fresh_o = act && !z && du == 1;
            const bool alias = act && !z && du == 0 && !nzq;
            int key = (fresh_w || fresh_o) ? ((i << 2) | (fresh_w ? TY_W : TY_O)) : -1;
            int mk = key;                                 // inclusive max-scan: the last setter at or below this lane
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int o_ = __shfl_up(mk, d, 64); if (lane >= d) mk = max(mk, o_); }
            int below = __shfl_up(mk, 1, 64);
            if (lane == 0) below = -1;
            below = max(below, key_in);
            if (act) {
                int ty = fresh_w ? TY_W : (fresh_o ? TY_O : 0), src = i;
                if (alias) { ty = below >= 0 ? (below & 3) : 0; src = below >= 0 ? (below >> 2) : 0; }
                Q.tt[i] = (unsigned char)ty; Q.ts[i] = (unsigned short)src;
            }
            const int so0 = __builtin_amdgcn_readlane(s0, 63), so1 = __builtin_amdgcn_readlane(s1, 63);
            du_in = du_in ? so1 : so0;
            key_in = max(key_in, __builtin_amdgcn_readlane(mk, 63));
        }
        if (lane == 0 && F >= 1) { Q.tt[F - 1] = 0; Q.ts[F - 1] = (unsigned short)(F - 1); }
    }
    CN_SYNC();
    // ---- RW:335-366 the typed entries, flattened in order: type, source entry (its rounded range and pose travel with it) --------
    int M = 0;
    for (int i0 = 0; i0 + 1 < F; i0 += 64) {
        const int i = i0 + lane;
        const bool ok = (i + 1 < F) && (Q.tt[i] != 0);
        const u64 bo = __ballot(ok);
        if (ok) { const int m = M + __popcll(bo & ((1ull << lane) - 1ull)); Q.et[m] = Q.tt[i]; Q.es[m] = Q.ts[i]; }
        M += __popcll(bo);
    }
    CN_SYNC();
#define ERAY(m) ((int)Q.fi[Q.es[m]])
    // ---- RW:368-403 segmentation: a segment closes after entry m unless m and m + 1 associate (bit words, entry space) -----------
    u64* const segw = L.w64;                       // ceil(M / 64) words
    const int Wm = (M + 63) >> 6;
    const bool rw_fast = p->assoc_fast && e.bb == p->bb_spawn;      // cn_create's table for the spawn pose's box size (simulated runs)
    int nseg = 0;
    for (int q = 0; q < Wm; ++q) {
        const int m = lane + 64 * q;
        bool brk = false;
        if (m < M) {
            brk = true;
            if (m < M - 1) {
                const int a = ERAY(m), b = ERAY(m + 1);
                if (rw_fast) {        // the main layout's integer association table (same box size, same function of |dx|, |dy| in thousandths)
                    const int dx_ = abs(L.ptx[a] - L.ptx[b]), dy_ = abs(L.pty[a] - L.pty[b]);
                    brk = dy_ > (int)p->assoc_tab[min(dx_, p->assoc_k1 + 1)];
                } else brk = !cn_iou3_positive(PX(a), PY(a), PX(b), PY(b), e.bb, PY2);
            }
        }
        const u64 bw = __ballot(brk);
        if (lane == 0) segw[q] = bw;
        nseg += __popcll(bw);
    }
    CN_SYNC();
    // RW:408-420 first ++ last when their outer ends associate with twice the box
    bool merged = false; int first_end = -1, last_start = 0;
    if (nseg > 1) {
        for (int q = 0; q < Wm; ++q) { const u64 bw = uni64(segw[q]); if (bw) { first_end = 64 * q + __builtin_ctzll(bw); break; } }
        for (int q = Wm - 1; q >= 0; --q) {
            u64 bw = uni64(segw[q]);
            if (q == Wm - 1) bw &= ~(1ull << ((M - 1) & 63));
            if (bw) { last_start = 64 * q + 64 - __builtin_clzll(bw); break; }
        }
        const int a = ERAY(0), b = ERAY(M - 1);
        merged = cn_iou3_positive(PX(a), PY(a), PX(b), PY(b), e.bb * 2, PY2);
    }
    if (merged) nseg -= 1;
    // ---- RW:426-468 confirmation, one segment at a time (scalar unit); order space = [0..first_end] ++ [last_start..M-1] ++ rest ---
    int nconf = 0;
    {
        const int nl = M - last_start;
#define ORD(k) (merged ? ((k) <= first_end ? (k) : ((k) <= first_end + nl ? last_start + ((k) - first_end - 1) : (k) - nl)) : (k))
        int k0 = 0;
        while (k0 < M) {
            int k1 = k0;
            if (merged && k0 == 0) k1 = first_end + nl;
            else {
                // the next segment end at or after k0: past the merged first segment order space is entry space shifted by nl, so it is
                // the next set bit of the break words (the last entry before the moved range always closes one) -- bit scans instead
                // of one loop iteration per entry
                const int off = merged ? nl : 0, m0 = k0 - off;
                int q = m0 >> 6;
                u64 bw = uni64(segw[q]) & (~0ull << (m0 & 63));
                while (!bw) { ++q; bw = uni64(segw[q]); }
                k1 = 64 * q + __builtin_ctzll(bw) + off;
            }
            const int len = k1 - k0 + 1;
            int no = 0, nw = 0;
            for (int k = k0 + lane; k <= k1; k += 64) { const int t_ = Q.et[ORD(k)]; no += (t_ == TY_O); nw += (t_ == TY_W); }
            no = cn_wave_sum_i(no); nw = cn_wave_sum_i(nw);
            const int ce = ORD(k0 + len / 2);                              // Python-2 integer division
            const int ray = ERAY(ce);
            const double dm = cn_div1000((double)(L.dmil[ray] & 0x7fff));
            const int est = 3 + (int)floor(29 * (p->max_scan_range - dm) / (p->max_scan_range - p->min_scan_range));
            const int mn = len < est ? len : est;
            const double score = (double)no / (double)mn;
            int obj = -1;
            if (no > 0 && nw > 0) {
                if (score >= 0.5) obj = (no > nw) ? TY_O : TY_W;
                else if (len <= est) obj = (no > nw) ? TY_O : TY_W;
                else obj = TY_W;
            } else {
                const int lim = nseg < est ? nseg : est;
                if (len > lim) obj = (nw > 0) ? TY_W : TY_O;
            }
            if (obj >= 0) {
                if (nconf < p->max_conf) { if (lane == 0) { L.cft[nconf] = obj; L.cfx[nconf] = PX(ray); L.cfy[nconf] = PY(ray); L.cfd[nconf] = dm; } nconf += 1; }
                else e.status |= CN_ST_CONF_OVERFLOW;
            }
            k0 = k1 + 1;
        }
#undef ORD
    }
#undef ERAY
#undef PX
#undef PY
    e.nconf = nconf;
    CN_SYNC();
    int ego_hit = 0;
    for (int j = lane; j < nconf; j += 64) if (L.cft[j] == TY_O && L.cfd[j] < 0.140) ego_hit = 1;     // RW:702-706
    ego_hit = __ballot(ego_hit != 0) != 0ull;
    // ---- RW:478-589 tracker and speeds: the same block as ENV:656-760 -------------------------------------------------------------
    double* const T = L.trk;
    tracker_stage(p, e, L, T, lane, nconf, now);
    if (lane < e.ntracks && TRK(CN_TF_DQLEN, lane) > 1.5) {
        double dc = cn_hypot(TRK(CN_TF_D0Y, lane) - TRK(CN_TF_D1Y, lane), TRK(CN_TF_D0X, lane) - TRK(CN_TF_D1X, lane));
        TRK(CN_TF_SPEED, lane) = dc / TRK(CN_TF_T, lane);
    }
    CN_SYNC();
    e.nent = 0;
    // ---- RW:595-700 collision cone against a ring of radius min_scan_range; the obstacle with the highest CP (last among ties) ----
    if (e.dq_len == 2) {
        const double ts = e.ts;
        if (ts == 0.0) e.status |= CN_ST_DT_ZERO;
        const int nt = e.ntracks;
        const double vx_ = (e.dq1x - e.dq0x) / ts, vy_ = (e.dq1y - e.dq0y) / ts;
        const double agent_vel = sqrt(vx_ * vx_ + vy_ * vy_);
        const double obstacle_vel = (nt == 0) ? 0.0 : TRK(CN_TF_SPEED, 0);
        if (lane < nt && TRK(CN_TF_DQLEN, lane) > 1.5) {
            double chx = TRK(CN_TF_D0X, lane) - TRK(CN_TF_D1X, lane), chy = TRK(CN_TF_D0Y, lane) - TRK(CN_TF_D1Y, lane);
            TRK(CN_TF_VX, lane) = chx / ts; TRK(CN_TF_VY, lane) = chy / ts;
        }
        double vo_x = e.dq1x, vo_y = e.dq1y;
        if (nt > 0) {
            const int l = nt - 1;
            double chx = 0.0, chy = 0.0;
            if (TRK(CN_TF_DQLEN, l) > 1.5) { chx = TRK(CN_TF_D0X, l) - TRK(CN_TF_D1X, l); chy = TRK(CN_TF_D0Y, l) - TRK(CN_TF_D1Y, l); }
            vo_x = e.dq1x + chx; vo_y = e.dq1y + chy;
        }
        CN_SYNC();
        const double a0x = e.dq0x, a0y = e.dq0y;
        const double gradient = (vo_y == 0.0) ? 0.0 : (vo_x - a0x) / vo_y - a0y;
        const double bb0 = a0x - (gradient * a0y);
        const int hi = (int)ceil(a0x + 3.5), lo = (int)floor(a0x - 3.5);
        const double rv = agent_vel - obstacle_vel;
        int best = -1; double bestcp = 0.0;
        for (int i = 0; i < nt; ++i) {
            const double tx = TRK(CN_TF_PX, i), ty_ = TRK(CN_TF_PY, i), td = TRK(CN_TF_DIST, i);
            int has = 0; double dcp = 0.0;
            // Round 5: the pre-rejection of the main layout's cone (ENV:818-860 above).  The 64-gon lies inside its circle, so a candidate
            // segment whose closest point to the centre is farther than the radius (with slack) cannot touch an edge: the same "empty"
            // result as the ring test, without its 64-lane intersection arithmetic (up to 8 candidates per track, most of them far).
            unsigned nearm;
            {
                const int x2l = hi - lane;
                bool nearc = false;
                if (lane < 8 && x2l > lo) {
                    const double y2l = ((double)x2l * gradient) + bb0;
                    const double ex = (double)x2l - a0x, ey = y2l - a0y, fx = tx - a0x, fy = ty_ - a0y;
                    const double ee = ex * ex + ey * ey, ff = fx * fx + fy * fy, num = fx * ex + fy * ey;
                    const double rr2 = p->min_scan_range * p->min_scan_range * 1.000001;
                    bool far_;
                    if (num <= 0.0) far_ = ff > rr2;
                    else if (num >= ee) far_ = (fx - ex) * (fx - ex) + (fy - ey) * (fy - ey) > rr2;
                    else far_ = (ff - rr2) * ee > num * num * 1.000001;
                    nearc = !far_;
                }
                nearm = (unsigned)(__ballot(nearc) & 0xffull);
            }
            for (int x2 = hi; x2 > lo; --x2) {
                if (!((nearm >> (hi - x2)) & 1u)) { if (p->geos_untyped_empty) break; continue; }       // certainly no intersection
                const double y2 = ((double)x2 * gradient) + bb0;
                double hx = 0.0, hy = 0.0;
                const u64 m = ring_segment(pg, lane, tx, ty_, p->min_scan_range, a0x, a0y, (double)x2, y2, &hx, &hy);
                const int cnt = __popcll(m);
                if (cnt == 0) { if (p->geos_untyped_empty) break; continue; }
                if (cnt == 1) break;
                const int l1 = __ffsll((long long)m) - 1, l2 = __ffsll((long long)(m & (m - 1ull))) - 1;
                const double d1 = cn_hypot(a0x - bcast_d(hx, l1), a0y - bcast_d(hy, l1));
                const double d2 = cn_hypot(a0x - bcast_d(hx, l2), a0y - bcast_d(hy, l2));
                dcp = cn_vmin(d1, d2); has = 1;
                break;
            }
            const double gcp = (td > p->max_scan_range) ? 0.0 : (p->max_scan_range - td) / (p->max_scan_range - p->min_scan_range);
            double cpv;
            if (has) {
                if (rv == 0) cpv = 1.0 * gcp;
                else {
                    const double ttc = dcp / rv;
                    if (ttc == 0.0) { e.status |= CN_ST_TTC_ZERO; cpv = 0.5 * 1.0 + 0.5 * gcp; }
                    else cpv = 0.5 * fmin(1.0, 0.15 / ttc) + 0.5 * gcp;
                }
            } else cpv = 0.5 * 0.0 + 0.5 * gcp;
            if (i == 0 || cpv >= bestcp) { bestcp = cpv; best = i; }          // max((val, idx)): the LAST of equal maxima
        }
        e.nent = nt;
        if (nt == 0) e.cprob = 0.0;
        else {
            e.cprob = fmax(0.0, bestcp);
            clx = TRK(CN_TF_PX, best); cly = TRK(CN_TF_PY, best); clvx = TRK(CN_TF_VX, best); clvy = TRK(CN_TF_VY, best);
        }
        e.dq0x = e.dq1x; e.dq0y = e.dq1y; e.dq_len = 1;
        if (lane < nt) TRK(CN_TF_T, lane) = now;
    }
    if (ego_hit) e.ego_viol += 1;
    if (e.cprob > 0.4) e.social_viol += 1;                                      // RW:708 (None > 0.4 is False in Python 2: cprob starts at -inf)
    if (!e.done) {                                                              // RW:715-728
        if (smin < p->min_scan_range) e.done = 1;
        if (in_box(px, py, p->goal_x, p->goal_y, 0.20)) e.done = 1;
        if (step_counter >= p->max_steps) e.done = 1;
    }
    if (lane < 11) {                                                            // RW:730-747: nothing is rounded a second time
        double tv;
        switch (lane) {
        case 0: tv = heading; break;
        case 1: tv = distance_to_goal; break;
        case 2: tv = cn_py_round3(px, PY2); break;
        case 3: tv = cn_py_round3(py, PY2); break;
        case 4: tv = cn_py_round3(3.14, PY2); break;                                 // round(self.yaw, 3): the constructor's constant
        case 5: tv = cn_py_round3(agent_vel_x, PY2); break;
        case 6: tv = cn_py_round3(agent_vel_y, PY2); break;
        case 7: tv = clx; break;
        case 8: tv = cly; break;
        case 9: tv = clvx; break;
        default: tv = clvy; break;
        }
        L.tail[lane] = tv;
        o32[n + lane] = (float)tv;
        if (f32) f32[n + lane] = (float)tv;
        if (o64) o64[n + lane] = tv;
    }
    if (lane < e.ntracks) {
#pragma unroll
        for (int f = 0; f < CN_TF_COUNT; f += 2)
            ((double2*)(L.gtrk + lane * CN_TF_COUNT))[f >> 1] = make_double2(L.trk[f * L.tcap + lane], L.trk[(f + 1) * L.tcap + lane]);
    }
    CN_SYNC();
    *done_out = e.done;
}
#undef TRK

// RW:751-849: -2 per step, +1 for getting closer, +1 for turning towards the goal, +-200 at the end (no way-point bonus);
// state[359] = heading and state[360] = distance are in L.tail[0..1]
__device__ __forceinline__ double compute_reward_realworld(KP p, EnvRegs& e, const Lds& L, int done)
{
    const double cur_head = L.tail[0], cur_dist = L.tail[1];
    const double dd = cur_dist - e.prev_dist, hd = cur_head - e.prev_head;
    int htg = 0, dtg = 0;
    if (dd < 0) dtg = 1;
    const double ph = e.prev_head;
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
    e.prev_dist = cur_dist; e.prev_head = cur_head;
    if (done) {
        if (in_box(e.rx, e.ry, p->goal_x, p->goal_y, 0.20)) { e.fail = 0; e.succ = 1; reward = 200 + reward; }
        else { e.fail = 1; e.succ = 0; reward = -200 + reward; }
    }
    return reward;
}

// ENV:1046-1162 compute_reward; state[n] = heading, state[n+1] = distance are in L.tail[0..1]
__device__ __forceinline__ double compute_reward(KP p, const Poly& pg, EnvRegs& e, const Lds& L, int lane, int done)
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
    if (in_box(e.rx, e.ry, e.wpx, e.wpy, p->goal_eps)) {  // ENV:1109-1125
        waypoint_refresh(p, pg, e, lane, e.rx, e.ry);
        wp = p->waypoint_reward;                           // ENV:1116: 200 (cn_config.waypoint_reward; 0 = the published log's reward)
        if (in_box(e.wpx, e.wpy, p->goal_x, p->goal_y, p->goal_eps)) { e.wpx = p->goal_x; e.wpy = p->goal_y; }
    }
    double reward = (double)(-2 + dtg + htg + wp);
    e.prev_dist = cur_dist;
    e.prev_head = cur_head;
    if (done) {
        if (in_box(e.rx, e.ry, p->goal_x, p->goal_y, p->goal_eps)) { e.fail = 0; e.succ = 1; reward = 200 + reward; }
        else { e.fail = 1; e.succ = 0; reward = -200 + reward; }
    }
    return reward;
}

}  // namespace

// `env`, `lane`: this wavefront's environment and lane; `smem`: its LDS working set (cn_lds_bytes).  The per-launch kernels pass
// blockIdx.x / threadIdx.x / the block's dynamic LDS; the multi-step kernel (FUSED, cn_env_kernel_seq below) calls this once per
// step, `t` steps into its launch, with the step's actions / outputs at slot t of the caller's buffers.  `act_here`: this
// environment's (v, w) where the policy kernel's actor left it (LDS) instead of the caller's action array.
template <bool EXT, bool TWO, int LAYOUT, bool GT = false, int SIM = 0, bool FUSED = false, bool FAIR = false, int SHAPE = 0, bool X2 = false>
__device__ __forceinline__ void env_kernel_body(const int env, const int lane, char* const smem, const long long t = 0,
                                                const float* const act_here = nullptr, const int wv = 0)
{
    static_assert(!X2 || (!EXT && !TWO && LAYOUT == 0 && !GT && SIM == 0 && !FUSED), "two wavefronts per environment: the plain step kernel");
    // (round 6: the oldest-first 360-ray step kernels as well -- 26 scalar spills without the fences, see tools/kernel_resources.sh)
    constexpr bool SFENCE = (SHAPE == 720 || SHAPE == 360) && !FAIR && !FUSED;
    KP p = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    if constexpr (FUSED) {
        // Inside the multi-step kernel's step loop everything below is loop-invariant as far as the compiler can see, and it
        // hoists it: ~100 kernel parameters and every lane-derived mask would stay live across the whole step (hundreds of VGPR
        // spills).  Laundering the kernarg pointer through an empty asm once per step keeps the loads next to their uses.
        unsigned long long pp = (unsigned long long)p;
        asm volatile("" : "+s"(pp));
        p = (KP)pp;
    }
    if constexpr (SHAPE != 0) {
        // A benchmark shape as COMPILE-TIME facts -- SHAPE 360: BASELINE configs[1] (360 rays, 20 pedestrians, K = 8), SHAPE 720:
        // configs[4] (720 rays, 100 pedestrians, K = 8) -- with cn_create's max_conf / tracker slots / LDS map for it: the kernarg
        // loads of these six fields fold to constants everywhere below (the loads are invariant, so one assumption covers every
        // use), which turns the LDS map into immediates, the word loops (W = 6 / 12) into straight-line code and frees the scalar
        // registers that held the map.  launch() only picks an _s360 / _s720 kernel for a handle whose configuration IS that shape.
        constexpr int SR = SHAPE == 360 ? 360 : 720, SP = SHAPE == 360 ? 20 : 100, SMC = SHAPE == 360 ? 91 : 181, STC = SHAPE == 360 ? 32 : 64,
                      SNS = SHAPE == 360 ? 1 : 0;
        static_assert(SHAPE == 360 || SHAPE == 720, "compiled shapes");
        __builtin_assume(p->R == SR); __builtin_assume(p->P == SP); __builtin_assume(p->K == 8);
        __builtin_assume(p->max_conf == SMC); __builtin_assume(p->trk_cap == STC); __builtin_assume(p->near_sep == SNS);
    }
    if constexpr (!FUSED) {
        if (env >= p->N) return;
        if (p->mode == CN_MODE_RESET && p->mask && !p->mask[env]) return;
    }
    const int R = p->R, n = R - 1, P = p->P, K = p->K;
    // COMPACT LDS layout (round 5; BASELINE configs[4]'s shape only): 18.5 KB per environment left 8 wavefronts on a CU (2 per SIMD);
    // 13.2 KB leaves 12.  Three changes, none of which touches a value: (1) end points as int16 thousandths (|x| <= 32.767 m: cn_create
    // checks the room), (2) confirmed objects as integer thousandths + byte flags (12 instead of 32 bytes each), (3) the pedestrians'
    // velocities live in region A, which is idle whenever the simulator runs, and go to the state record before the observation
    // overwrites them (a reset brings them back for its settle advance).  crowdnav_abi.hip lds_bytes_impl(compact) mirrors the carve.
    constexpr bool CMP = SHAPE == 720;
    static_assert(!CMP || (!EXT && !TWO && LAYOUT == 0 && !GT && SIM == 0), "the compact layout is the 720-ray shape kernels' own");
    // where this step's outputs go: the caller's buffers, or (FUSED) slot t of its trajectory buffers (stride 0 = in place)
    auto io_obs = [&]() -> float* { if constexpr (FUSED) return p->obs + (size_t)(t * p->roll_obs_stride); else return p->obs; };
    auto io_reward = [&]() -> float* { if constexpr (FUSED) return p->reward + (size_t)(t * p->roll_reward_stride); else return p->reward; };
    auto io_done = [&]() -> uint8_t* { if constexpr (FUSED) return p->done + (size_t)(t * p->roll_done_stride); else return p->done; };
    auto io_action = [&]() -> const float* { if constexpr (FUSED) return p->action + (size_t)(t * p->roll_action_in_stride); else return p->action; };
    auto io_topk = [&]() -> int32_t* { if constexpr (FUSED) return p->topk_idx ? p->topk_idx + (size_t)(t * p->roll_topk_stride) : nullptr; else return p->topk_idx; };

    Lds L;
    RwLds RQ = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    {
        // LDS map.  Region A: end points (integer thousandths) | tracker table.
        // Region B: gradients + alias sources | bbox staging | confirmed objects, CP, observation tail.
        const size_t szA_pts = (size_t)((CMP ? 6 : 10) * n + 7) & ~(size_t)7;
        const size_t szA_trk = 8 * (size_t)(CN_TF_COUNT * p->trk_cap);
        L.tcap = p->trk_cap;
        const size_t szA = szA_pts > szA_trk ? szA_pts : szA_trk;
        const size_t mc = (size_t)p->max_conf;
        const size_t szB_g = (size_t)(6 * n + 7) & ~(size_t)7;
        const size_t szC = CMP ? ((12 * mc + 7) & ~(size_t)7) : 32 * mc;          // the confirmed-object arrays
        const size_t szB_c = szC + 8 * 64 + 8 * (size_t)(8 + 4 * K) + 4 * (size_t)CN_MAX_K;
        size_t szB = szB_g > szB_c ? szB_g : szB_c;
        if (szB < 8 * 64) szB = 8 * 64;
        if (!p->near_sep && szB < 32 * (size_t)(P + 1)) szB = 32 * (size_t)(P + 1);   // near-pedestrian list overlaid on B
        char* A = smem;
        char* B = A + szA;
        char* Cw = B + szB;
        L.ptx = (int*)A; L.pty = (int*)(A + 4 * (size_t)n); L.dmil = (unsigned short*)(A + 8 * (size_t)n);
        L.ptx16 = (short*)A; L.pty16 = (short*)(A + 2 * (size_t)n);
        if constexpr (CMP) L.dmil = (unsigned short*)(A + 4 * (size_t)n);
        L.trk = (double*)A;
        L.gq = (int*)B; L.srcidx = (unsigned short*)(B + 4 * (size_t)n);
        L.stage = (double*)B;
        L.cfx = (double*)B; L.cfy = (double*)(B + 8 * mc); L.cfd = (double*)(B + 16 * mc);
        L.cft = (int*)(B + 24 * mc); L.checked = (int*)(B + 28 * mc);
        L.cfxi = (int*)B; L.cfyi = (int*)(B + 4 * mc); L.cfdm = (unsigned short*)(B + 8 * mc);
        L.cft8 = (unsigned char*)(B + 10 * mc); L.chk8 = (unsigned char*)(B + 11 * mc);
        L.cpv = (double*)(B + szC);
        L.tail = L.cpv + 64;
        L.kidx = (int*)(L.tail + (8 + 4 * K));
        const int Wn = (n + 63) >> 6;
        L.wstride = Wn;
        L.w64 = (u64*)Cw; Cw += 8 * (size_t)(M_COUNT * Wn);
        L.wbase = (int*)Cw; Cw += 8 * (size_t)((3 * Wn + 1) / 2);
        L.ped = (double*)Cw; Cw += 8 * (size_t)(2 * P + 2);
        if constexpr (CMP) L.pedv = (double*)A;         // (3) above: 8 (2 P + 2) <= szA
        else { L.pedv = (double*)Cw; Cw += 8 * (size_t)(2 * P + 2); }
        L.nearp = p->near_sep ? (double*)Cw : (double*)B;   // ray loop only: region B is dead until the gradients are written
        if (p->near_sep) Cw += 32 * (size_t)(P + 1);
        L.gtrk = p->trk + (size_t)env * CN_TF_COUNT * p->trk_cap;
        if constexpr (LAYOUT == 2) {   // the real-world layout's own lists: 12 bytes per ray (cn_lds_bytes adds them)
            char* Rw = (char*)(((size_t)Cw + 15) & ~(size_t)15);
            RQ.g = (int*)Rw; Rw += 4 * (size_t)n;
            RQ.fi = (unsigned short*)Rw; Rw += 2 * (size_t)n;
            RQ.ts = (unsigned short*)Rw; Rw += 2 * (size_t)n;
            RQ.es = (unsigned short*)Rw; Rw += 2 * (size_t)n;
            RQ.tt = (unsigned char*)Rw; Rw += (size_t)n;
            RQ.et = (unsigned char*)Rw;
        }
    }

    CN_T(0);
#ifdef CN_TIMING
    if (p->timing && lane == 0) {      // where this wavefront runs: HW_ID (wave slot, SIMD, CU, SE) and XCC_ID -- tools/wave_fairness.py
        p->timing[(size_t)env * 32 + 25] = (long long)__builtin_amdgcn_s_getreg(4 | (31 << 11));
        p->timing[(size_t)env * 32 + 26] = (long long)__builtin_amdgcn_s_getreg(20 | (31 << 11));
    }
#endif
    Poly pg;
    pg.c0 = p->poly_c[lane]; pg.s0 = p->poly_s[lane]; pg.c1 = p->poly_c[(lane + 1) & 63]; pg.s1 = p->poly_s[(lane + 1) & 63];

    // ---- load env state -------------------------------------------------------------------------
    char* rec = p->state + (size_t)env * (size_t)p->state_stride;
    double* sd = (double*)(rec + CN_ST_OFF_SD);
    int* si = (int*)(rec + CN_ST_OFF_SI);
    // Round 6: every global load the first phase waits for is ASKED FOR before the first wait.  The pedestrians (behind a branch on
    // the loaded track count) and the action (behind the branch on the loaded pending-reset flag) used to be a second and a third
    // memory round trip in a row on a wavefront's critical path -- HBM-side latency each, since a launch finds nothing in its L2;
    // now they travel with the state record: one coordinate per lane while 2 P <= 64, the action in two registers.
    double* const gped_p = (double*)(rec + CN_ST_OFF_PED_P);
    double* const gped_v = (double*)(rec + CN_ST_OFF_PED_V(p->P));
    // (up to four coordinates per lane: 2 P <= 256 covers BASELINE configs[4]'s 100 pedestrians)
    const int P2_ = 2 * p->P;
    const bool ped_pre = !X2 && P2_ <= 256;
    double pp_pre[4] = {0.0, 0.0, 0.0, 0.0}, pv_pre[4] = {0.0, 0.0, 0.0, 0.0};
    if (ped_pre) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (64 * c < P2_) { const int li = lane + 64 * c < P2_ ? lane + 64 * c : P2_ - 1; pp_pre[c] = gped_p[li]; pv_pre[c] = gped_v[li]; }
    }
    const bool act_pre = !EXT && !act_here && p->mode == CN_MODE_STEP;
    float a0_pre = 0.0f, a1_pre = 0.0f;
    if (act_pre) { const float* const ain_ = io_action() + 2 * (size_t)env; a0_pre = ain_[0]; a1_pre = ain_[1]; }
    EnvRegs e;
    e.rx = sd[CN_SD_RX]; e.ry = sd[CN_SD_RY]; e.ryaw = sd[CN_SD_RYAW]; e.rv = sd[CN_SD_RV]; e.rw = sd[CN_SD_RW];
    e.clock = sd[CN_SD_CLOCK]; e.wpx = sd[CN_SD_WPX]; e.wpy = sd[CN_SD_WPY];
    e.prev_dist = sd[CN_SD_PREV_DIST]; e.prev_head = sd[CN_SD_PREV_HEAD];
    e.dq0x = sd[CN_SD_DQ0X]; e.dq0y = sd[CN_SD_DQ0Y]; e.dq1x = sd[CN_SD_DQ1X]; e.dq1y = sd[CN_SD_DQ1Y];
    e.ts = sd[CN_SD_TS]; e.bb = sd[CN_SD_BB]; e.ego = sd[CN_SD_EGO]; e.cprob = sd[CN_SD_CPROB];
    e.ep_ret = sd[CN_SD_EP_RETURN]; e.last_ret = sd[CN_SD_LAST_RETURN];
    e.done = si[CN_SI_DONE]; e.dq_len = si[CN_SI_DQ_LEN]; e.ntracks = si[CN_SI_NTRACKS];
    e.ep_step = si[CN_SI_EP_STEP]; e.pending = si[CN_SI_PENDING_RESET];
    // The counters nothing reads before the observation is over (safety violations, obstacle-present steps, success / failure,
    // status bits, episode count) stay in ONE vector register for most of the call -- lane k holds si[k], one coalesced load --
    // instead of nine scalar values that live from the first instruction to the last (the kernel sits at its register caps).
    // Until cold_settle() the fields of `e` are this call's increments; cold_settle() folds the stored values in.
    const int cold_i = si[lane & (CN_SI_COUNT - 1)];
    static_assert((CN_SI_COUNT & (CN_SI_COUNT - 1)) == 0, "CN_SI_COUNT is a power of two");
    e.ego_viol = 0; e.social_viol = 0; e.obst_steps = 0; e.status = 0; e.succ = 0; e.fail = 0; e.episodes = 0; e.nconf = 0; e.nent = 0;
    bool cold_settled = false;
    auto cold_settle = [&]() {
        if (cold_settled) return;
        cold_settled = true;
        e.ego_viol += __builtin_amdgcn_readlane(cold_i, CN_SI_EGO_VIOL); e.social_viol += __builtin_amdgcn_readlane(cold_i, CN_SI_SOCIAL_VIOL);
        e.obst_steps += __builtin_amdgcn_readlane(cold_i, CN_SI_OBST_STEPS); e.status |= __builtin_amdgcn_readlane(cold_i, CN_SI_STATUS);
        e.succ = __builtin_amdgcn_readlane(cold_i, CN_SI_SUCCESS); e.fail = __builtin_amdgcn_readlane(cold_i, CN_SI_FAILURE);
        e.episodes = __builtin_amdgcn_readlane(cold_i, CN_SI_EPISODES);
    };
    e.crowd_ms = (long long)(((unsigned long long)(unsigned)si[CN_SI_CROWD_HI] << 32) | (unsigned)si[CN_SI_CROWD_LO]);
    e.cv = 0.0; e.cw = 0.0;       // the command lives inside one call: a step publishes its action first, a reset leaves it zero

    // Round 6: the kernel-argument block (1 048 bytes: every parameter is read in place, where it is used) spans 17 lines of the
    // scalar cache, which a launch finds empty: each line's FIRST touch -- the trig table in front of the first sine, the pedestrian
    // schedule, the goal geometry ... -- was a miss all the way to memory on the critical path of every wavefront of the CU (they
    // run the same stage at the same time): ~2 000 cycles per stage that opens a new line (tools/stage_timing_physics.py).  All 17
    // lines are touched here, behind the state record's loads that are already in flight: the misses overlap each other and the
    // HBM round trip this wavefront waits for anyway.  (One destination register for all of them: the values are not used.)
    if constexpr (!FUSED) {
        static_assert(sizeof(CnKParams) <= 17 * 64, "kernarg lines touched below");
        unsigned kp_warm_;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_load_dword %0, %1, 0x40\n\ts_load_dword %0, %1, 0x80\n\ts_load_dword %0, %1, 0xc0\n\t"
                     "s_load_dword %0, %1, 0x100\n\ts_load_dword %0, %1, 0x140\n\ts_load_dword %0, %1, 0x180\n\ts_load_dword %0, %1, 0x1c0\n\t"
                     "s_load_dword %0, %1, 0x200\n\ts_load_dword %0, %1, 0x240\n\ts_load_dword %0, %1, 0x280\n\ts_load_dword %0, %1, 0x2c0\n\t"
                     "s_load_dword %0, %1, 0x300\n\ts_load_dword %0, %1, 0x340\n\ts_load_dword %0, %1, 0x380\n\ts_load_dword %0, %1, 0x3c0\n\t"
                     "s_load_dword %0, %1, 0x400\n\ts_load_dword %0, %1, 0x414\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(kp_warm_) : "s"(p) : "memory");
    }
    // The tracker table is read from HBM in the middle of the observation (its LDS space holds the end points until then),
    // which would put a full memory round trip on the wavefront's critical path.  Touch its lines now -- one dword per
    // 128-byte line of the live records -- so that the real load, ~20 us later, hits the L2.
    int trk_warm = 0;
    if (LAYOUT == 0 && lane * 128 < e.ntracks * (CN_TF_COUNT * 8)) trk_warm = ((const volatile int*)L.gtrk)[lane * 32];
    const double* gped_init = p->ped_init + (size_t)env * 2 * P;
    double* pedv = L.pedv;  // velocities are only needed while advancing
    if constexpr (!X2) {    // (X2: wave 1 brings the pedestrians in, below, once it is known that this launch is a step)
    if (ped_pre) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (lane + 64 * c < 2 * P) { L.ped[lane + 64 * c] = pp_pre[c]; pedv[lane + 64 * c] = pv_pre[c]; }
    }
    else for (int i = lane; i < 2 * P; i += 64) { L.ped[i] = gped_p[i]; pedv[i] = gped_v[i]; }
    CN_SYNC();
    }
    XMail* const mb = X2 ? (XMail*)(smem + p->wave_lds) : nullptr;
    // compact layout: the velocities' LDS space becomes the end points during the observation
    auto pedv_save = [&]() { if constexpr (CMP) { for (int i = lane; i < 2 * P; i += 64) gped_v[i] = pedv[i]; } };
    auto pedv_restore = [&]() { if constexpr (CMP) { for (int i = lane; i < 2 * P; i += 64) pedv[i] = gped_v[i]; CN_SYNC(); } };

    CN_T(1);
    int done = 0;
    constexpr bool ext = EXT;
    const double* od = ext ? p->ext_odom + (size_t)env * 10 : nullptr;
    if (ext) {   // /odom callback (ENV:239-243) and time.time()
        e.rx = od[0]; e.ry = od[1]; e.ryaw = od[2]; e.rv = od[3]; e.rw = od[4]; e.clock = od[5];
    }
    if constexpr (!TWO) {
        // ---- ONE observation per wavefront: Env.step, or Env.reset (explicit cn_reset, or the deferred reset
        // of auto_reset == 2, the gymnasium NEXT_STEP convention: an env that finished in the previous launch
        // spends THIS launch on its reset -- action ignored, reward 0, done 0).  Keeping a single inlined copy of
        // observe() also halves the kernel's code size (the instruction cache is 64 KB per two CUs).
        bool do_reset = !(p->mode == CN_MODE_STEP || p->mode == CN_MODE_EXT_STEP);
        if (p->mode == CN_MODE_STEP && p->auto_reset == 2 && e.pending) {
            do_reset = true;
            e.pending = 0;
            if (!X2 || wv == 0) {
            if (lane == 0) { io_reward()[env] = 0.0f; io_done()[env] = 0; }
            if (io_topk() && lane < K) io_topk()[(size_t)env * K + lane] = -1;
            }
        }
        if constexpr (X2) {
            if (wv != 0) {
                // WAVE 1 of the pair: the pedestrians' advance (while wave 0 advances the robot and does the goal geometry), then its
                // half of the lane = ray stages inside observe(), which it leaves after the association.  A reset launch leaves the
                // pedestrians to wave 0 (initial poses, two short advances).
                if (!do_reset) {
                    for (int i = lane; i < 2 * P; i += 64) { L.ped[i] = gped_p[i]; pedv[i] = gped_v[i]; }
                    CN_SYNC();
                    ped_advance<SHAPE == 360>(p, env, lane, L.ped, pedv, e.crowd_ms, e.crowd_ms + p->dt_ms + p->scan_latency_ms, p->dt_ms);
                }
                CN_SYNC();
                int d1 = 0;
                observe<EXT, GT, FAIR, CMP, true, SFENCE>(p, pg, e, L, env, lane, 0, io_obs(), do_reset ? nullptr : p->final_obs, p->obs_f64, &d1, false,
                                                  Trig{0.0, 0.0, 0.0, 0.0}, 1, mb);
                return;
            }
        }
        int sc = 0;
        float* fin = nullptr;
        Trig trig = Trig{0.0, 0.0, 0.0, 0.0}; bool have_trig = false;
        double rs1 = 0.0, rc1 = 0.0, rs2 = 0.0, rc2 = 0.0;
        // cn_external_io.phase: which pieces of Env.step this launch runs (external data only; 0 = all of them)
        bool ph_pre = true, ph_obs = true, ph_rew = true;
        if constexpr (EXT) {
            if (p->ext_phase) { ph_pre = p->ext_phase & CN_PHASE_PRE; ph_obs = p->ext_phase & CN_PHASE_GET_STATE; ph_rew = p->ext_phase & CN_PHASE_REWARD; }
        }
        if (!do_reset) {
            // Env.step (ENV:1164-1225), continuous mode
            if (ph_pre) e.ep_step += 1;
            sc = p->step_counter ? p->step_counter[env] : e.ep_step;
            double deq_x, deq_y, end_timestep;
            if (!ext) {
                double v, w;
                if (act_pre) { v = (double)a0_pre; w = (double)a1_pre; }          // (asked for with the state record)
                else { const float* const ain = act_here ? act_here : io_action() + 2 * (size_t)env; v = (double)ain[0]; w = (double)ain[1]; }
                const double t0 = e.clock;
                if constexpr (SIM == 3) { e.cv = v; e.cw = w; } else { e.rv = v; e.rw = w; }   // pub_cmd_vel.publish (ENV:1200)
                e.clock += cn_div1000((double)p->dt_ms);              // time.sleep(0.15) (ENV:1201)
                if constexpr (SIM != 0) sim_advance_ticks<SIM>(p, e, env, lane, L.ped, pedv, (double*)smem, p->dt_ms);
                else {
                if constexpr (LAYOUT == 0) { step_trig(p, e, lane, rs1, rc1, rs2, rc2, trig); have_trig = true; }
                CN_T(27);
                // pedestrians: ONE pass over [0, dt + scan latency], cut at dt (they are only looked at by the scan)
                if constexpr (!X2)
                ped_advance<SHAPE == 360>(p, env, lane, L.ped, pedv, e.crowd_ms, e.crowd_ms + p->dt_ms + p->scan_latency_ms, p->dt_ms);
                CN_T(28);
                e.crowd_ms += p->dt_ms + p->scan_latency_ms;
                if (have_trig) robot_advance_sc(p, e, p->dt_ms, rs1, rc1); else robot_advance(p, e, p->dt_ms);
                }
                CN_T(20);
                end_timestep = e.clock - t0;                      // ENV:1202
                if constexpr (LAYOUT == 2)   // RW:876-883: held for dt (0.05 s), booked as `0.05 - 0 + 0.1` on top of the measured 0
                    end_timestep = 0.0 + ((cn_div1000((double)p->dt_ms) - 0.0) + 0.1);
                deq_x = e.rx; deq_y = e.ry;
                fin = p->final_obs;
            } else {                                              // the caller ran the sleep; /odom said where we are
                deq_x = od[6]; deq_y = od[7]; end_timestep = od[8];
            }
            if (ph_pre) {
            double qx = cn_py_round3_t<!EXT>(deq_x, PY2), qy = cn_py_round3_t<!EXT>(deq_y, PY2);  // ENV:1208
            if (e.dq_len == 0) { e.dq0x = qx; e.dq0y = qy; e.dq_len = 1; }
            else if (e.dq_len == 1) { e.dq1x = qx; e.dq1y = qy; e.dq_len = 2; }
            else { e.dq0x = e.dq1x; e.dq0y = e.dq1y; e.dq1x = qx; e.dq1y = qy; }
            e.ts = end_timestep;                                  // ENV:1209
            }
            if (!ext) {
                e.clock += cn_div1000((double)p->scan_latency_ms);    // wait_for_message('scan') (ENV:1218)
                if constexpr (SIM != 0) sim_advance_ticks<SIM>(p, e, env, lane, L.ped, pedv, (double*)smem, p->scan_latency_ms);
                else if (have_trig) robot_advance_sc(p, e, p->scan_latency_ms, rs2, rc2); else robot_advance(p, e, p->scan_latency_ms);
                CN_T(21);
            }
        } else {
            // Env.reset (ENV:1227-1263): gazebo/reset_simulation puts poses back and zeroes twists (the crowd clock keeps running)
            if (!ext) {
                e.rx = p->spawn_x; e.ry = p->spawn_y; e.ryaw = p->spawn_yaw; e.rv = 0.0; e.rw = 0.0; e.cv = 0.0; e.cw = 0.0;
                for (int i = lane; i < 2 * P; i += 64) { L.ped[i] = gped_init[i]; pedv[i] = 0.0; }
                CN_SYNC();
                e.clock += cn_div1000((double)p->scan_latency_ms);    // wait_for_message('scan') (ENV:1238)
                if constexpr (SIM != 0) sim_advance_ticks<SIM>(p, e, env, lane, L.ped, pedv, (double*)smem, p->scan_latency_ms);
                else sim_advance(p, e, env, lane, L.ped, pedv, p->scan_latency_ms);
            }
            if constexpr (LAYOUT == 1) {
                e.prev_dist = dist3(e.rx, e.ry, p->goal_x, p->goal_y);   // ORIG:472 (unrounded)
                e.prev_head = orig_heading(p, e.rx, e.ry, e.ryaw);     // ORIG:473
            } else if constexpr (LAYOUT == 2) {
                e.prev_dist = rw_distance(p, e.rx, e.ry);                // RW:925-926 (unrounded)
                e.prev_head = rw_heading(p, e.rx, e.ry, e.ryaw);
            } else {
            e.prev_dist = dist3(e.rx, e.ry, e.wpx, e.wpy);        // ENV:1243 (unrounded)
            e.prev_head = heading_to_goal(p, e, e.rx, e.ry, e.ryaw);  // ENV:1244
            }
        }
        pedv_save();
        CN_SYNC();
        if (ph_obs) {
            if constexpr (LAYOUT == 1) observe_original<EXT>(p, e, L, env, lane, sc, io_obs(), fin, p->obs_f64, &done);
            else if constexpr (LAYOUT == 2) observe_realworld<EXT>(p, pg, e, L, RQ, env, lane, sc, io_obs(), fin, p->obs_f64, &done);
            else observe<EXT, GT, FAIR, CMP, X2, SFENCE, !(FUSED && SHAPE == 720)>(p, pg, e, L, env, lane, sc, io_obs(), fin, p->obs_f64, &done, have_trig, trig, 0, mb);
        } else if constexpr (EXT) {
            // Env.compute_reward(state, step_counter, done) on its own (ENV:1046): heading and distance are state[n], state[n+1]
            // (LAYOUT 1: state[-2], state[-1] are what ORIG:324-330 reads), `done` is the caller's
            const int D_ = (LAYOUT == 1) ? n + 4 : (LAYOUT == 2 ? n + 11 : n + 7 + 4 * K);
            if (lane < 4) {
                const int src = (LAYOUT == 1) ? n + lane : n + (lane & 1);
                L.tail[lane] = p->obs_f64 ? p->obs_f64[(size_t)env * D_ + src] : (double)p->obs[(size_t)env * D_ + src];
            }
            done = io_done()[env] ? 1 : 0;
            e.nconf = __builtin_amdgcn_readlane(cold_i, CN_SI_NCONF); e.nent = __builtin_amdgcn_readlane(cold_i, CN_SI_NENTRIES);   // no observation ran
            CN_SYNC();
        }
        cold_settle();
        if (!do_reset && !ph_rew) {
            if (lane == 0) io_done()[env] = (uint8_t)done;           // get_state returns (state, self.done) (ENV:1044)
            if (ph_obs && io_topk() && lane < K) io_topk()[(size_t)env * K + lane] = LAYOUT != 0 ? -1 : L.kidx[lane];
        } else
        if (!do_reset) {
            double r;
            if constexpr (LAYOUT == 1) r = compute_reward_original(p, e, L, done);
            else if constexpr (LAYOUT == 2) r = compute_reward_realworld(p, e, L, done);
            else r = compute_reward(p, pg, e, L, lane, done);
            e.ep_ret += r;
            if (lane == 0) {
                io_reward()[env] = (float)r;
                io_done()[env] = (uint8_t)done;
            }
            if (ph_obs && io_topk() && lane < K) io_topk()[(size_t)env * K + lane] = LAYOUT != 0 ? -1 : L.kidx[lane];
            if (done) {
                if (!ext) { if constexpr (SIM == 3) { e.cv = 0.0; e.cw = 0.0; } else { e.rv = 0.0; e.rw = 0.0; } }   // pub_cmd_vel.publish(Twist()) (ENV:1160)
                e.last_ret = e.ep_ret;
                e.episodes += 1;
                e.pending = !ext && (p->auto_reset == 2);
                if (lane == 0) {   // the finished episode's counters as TRAIN:142-147 reads them (the reset zeroes the live ones)
                    sd[CN_SD_LAST_EGO_VIOL] = (double)e.ego_viol; sd[CN_SD_LAST_SOCIAL_VIOL] = (double)e.social_viol;
                    sd[CN_SD_LAST_OBST_STEPS] = (double)e.obst_steps; sd[CN_SD_LAST_EP_STEPS] = (double)e.ep_step;
                }
            }
        } else {
            e.social_viol = 0; e.ego_viol = 0; e.obst_steps = 0;  // ENV:1260-1262
            if (!ext) {
                e.clock += cn_div1000((double)p->settle_ms);          // TRAIN:114 time.sleep(0.1)
                pedv_restore();
                if constexpr (SIM != 0) sim_advance_ticks<SIM>(p, e, env, lane, L.ped, pedv, (double*)smem, p->settle_ms);
                else sim_advance(p, e, env, lane, L.ped, pedv, p->settle_ms);
                CN_SYNC();
                pedv_save();
            }
            e.done = 0;                                           // TRAIN:116
            e.ep_step = 0; e.ep_ret = 0.0; e.pending = 0;
        }
        CN_SYNC();
    } else {
    // ---- same-call reset (auto_reset == 1): Env.step, then Env.reset for an env that just finished ----
    bool need_reset = (p->mode == CN_MODE_RESET || p->mode == CN_MODE_EXT_RESET);
    // auto_reset == 2 ("next-step" reset, the gymnasium NEXT_STEP convention): an env that finished in
    // the previous launch spends THIS launch on Env.reset() -- its action is ignored, reward 0, done 0 --
    // so no wavefront ever runs two observations back to back and the launch's critical path halves.
    if (p->mode == CN_MODE_STEP && p->auto_reset == 2 && e.pending) {
        need_reset = true;
        e.pending = 0;
        if (lane == 0) { io_reward()[env] = 0.0f; io_done()[env] = 0; }
        if (io_topk() && lane < K) io_topk()[(size_t)env * K + lane] = -1;
    } else if (p->mode == CN_MODE_STEP || p->mode == CN_MODE_EXT_STEP) {
        // ---- Env.step (ENV:1164-1225), continuous mode ---------------------------------------------
        e.ep_step += 1;
        const int sc = p->step_counter ? p->step_counter[env] : e.ep_step;
        double deq_x, deq_y, end_timestep;
        if (!ext) {
            const double v = (double)io_action()[2 * env], w = (double)io_action()[2 * env + 1];
            const double t0 = e.clock;
            if constexpr (SIM == 3) { e.cv = v; e.cw = w; } else { e.rv = v; e.rw = w; }   // pub_cmd_vel.publish (ENV:1200)
            e.clock += cn_div1000((double)p->dt_ms);              // time.sleep(0.15) (ENV:1201)
            if constexpr (SIM != 0) sim_advance_ticks<SIM>(p, e, env, lane, L.ped, pedv, (double*)smem, p->dt_ms);
            else {
            ped_advance(p, env, lane, L.ped, pedv, e.crowd_ms, e.crowd_ms + p->dt_ms + p->scan_latency_ms, p->dt_ms);
            e.crowd_ms += p->dt_ms + p->scan_latency_ms;
            robot_advance(p, e, p->dt_ms);
            }
            end_timestep = e.clock - t0;                      // ENV:1202
                if constexpr (LAYOUT == 2)   // RW:876-883: held for dt (0.05 s), booked as `0.05 - 0 + 0.1` on top of the measured 0
                    end_timestep = 0.0 + ((cn_div1000((double)p->dt_ms) - 0.0) + 0.1);
            deq_x = e.rx; deq_y = e.ry;
        } else {                                              // the caller ran the sleep; /odom said where we are
            deq_x = od[6]; deq_y = od[7]; end_timestep = od[8];
        }
        {
            double qx = cn_py_round3(deq_x, PY2), qy = cn_py_round3(deq_y, PY2);  // ENV:1208
            if (e.dq_len == 0) { e.dq0x = qx; e.dq0y = qy; e.dq_len = 1; }
            else if (e.dq_len == 1) { e.dq1x = qx; e.dq1y = qy; e.dq_len = 2; }
            else { e.dq0x = e.dq1x; e.dq0y = e.dq1y; e.dq1x = qx; e.dq1y = qy; }
        }
        e.ts = end_timestep;                                  // ENV:1209
        if (!ext) {
            e.clock += cn_div1000((double)p->scan_latency_ms);    // wait_for_message('scan') (ENV:1218)
            if constexpr (SIM != 0) sim_advance_ticks<SIM>(p, e, env, lane, L.ped, pedv, (double*)smem, p->scan_latency_ms);
            else robot_advance(p, e, p->scan_latency_ms);
        }
        CN_SYNC();
        double r;
        if constexpr (LAYOUT == 1) {
            observe_original<EXT>(p, e, L, env, lane, sc, p->obs, ext ? nullptr : p->final_obs, p->obs_f64, &done);
            cold_settle();
            r = compute_reward_original(p, e, L, done);
        } else if constexpr (LAYOUT == 2) {
            observe_realworld<EXT>(p, pg, e, L, RQ, env, lane, sc, p->obs, ext ? nullptr : p->final_obs, p->obs_f64, &done);
            cold_settle();
            r = compute_reward_realworld(p, e, L, done);
        } else {
            observe<EXT, GT, FAIR>(p, pg, e, L, env, lane, sc, p->obs, ext ? nullptr : p->final_obs, p->obs_f64, &done);
            cold_settle();
            r = compute_reward(p, pg, e, L, lane, done);
        }
        e.ep_ret += r;
        if (lane == 0) {
            io_reward()[env] = (float)r;
            io_done()[env] = (uint8_t)done;
        }
        if (io_topk() && lane < K) io_topk()[(size_t)env * K + lane] = LAYOUT != 0 ? -1 : L.kidx[lane];
        if (done) {
            if (!ext) { if constexpr (SIM == 3) { e.cv = 0.0; e.cw = 0.0; } else { e.rv = 0.0; e.rw = 0.0; } }   // pub_cmd_vel.publish(Twist()) (ENV:1160)
            e.last_ret = e.ep_ret;
            e.episodes += 1;
            need_reset = !ext && (p->auto_reset == 1);
            e.pending = !ext && (p->auto_reset == 2);
            if (lane == 0) {   // the finished episode's counters as TRAIN:142-147 reads them (the reset zeroes the live ones)
                sd[CN_SD_LAST_EGO_VIOL] = (double)e.ego_viol; sd[CN_SD_LAST_SOCIAL_VIOL] = (double)e.social_viol;
                sd[CN_SD_LAST_OBST_STEPS] = (double)e.obst_steps; sd[CN_SD_LAST_EP_STEPS] = (double)e.ep_step;
            }
        }
        CN_SYNC();
    }
    if (need_reset) {
        __syncthreads();  // rare path: drain the tracker-table stores of the step phase before they are re-read
        // ---- Env.reset (ENV:1227-1263) + TRAIN:114-116 -----------------------------------------------
        // gazebo/reset_simulation: poses back to their initial values, twists zeroed (crowd clock keeps running)
        if (!ext) {
        e.rx = p->spawn_x; e.ry = p->spawn_y; e.ryaw = p->spawn_yaw; e.rv = 0.0; e.rw = 0.0; e.cv = 0.0; e.cw = 0.0;
        for (int i = lane; i < 2 * P; i += 64) { L.ped[i] = gped_init[i]; pedv[i] = 0.0; }
        CN_SYNC();
        e.clock += cn_div1000((double)p->scan_latency_ms);        // wait_for_message('scan') (ENV:1238)
        if constexpr (SIM != 0) sim_advance_ticks<SIM>(p, e, env, lane, L.ped, pedv, (double*)smem, p->scan_latency_ms);
        else sim_advance(p, e, env, lane, L.ped, pedv, p->scan_latency_ms);
        }
        CN_SYNC();
        int d2 = 0;
        if constexpr (LAYOUT == 1) {
            e.prev_dist = dist3(e.rx, e.ry, p->goal_x, p->goal_y);   // ORIG:472 (unrounded)
            e.prev_head = orig_heading(p, e.rx, e.ry, e.ryaw);     // ORIG:473
            CN_SYNC();
            observe_original<EXT>(p, e, L, env, lane, 0, p->obs, nullptr, p->obs_f64, &d2);
        } else if constexpr (LAYOUT == 2) {
            e.prev_dist = rw_distance(p, e.rx, e.ry);
            e.prev_head = rw_heading(p, e.rx, e.ry, e.ryaw);
            CN_SYNC();
            observe_realworld<EXT>(p, pg, e, L, RQ, env, lane, 0, p->obs, nullptr, p->obs_f64, &d2);
        } else {
        e.prev_dist = dist3(e.rx, e.ry, e.wpx, e.wpy);        // ENV:1243 (unrounded)
        e.prev_head = heading_to_goal(p, e, e.rx, e.ry, e.ryaw);  // ENV:1244
        CN_SYNC();
        observe<EXT, GT, FAIR>(p, pg, e, L, env, lane, 0, p->obs, nullptr, p->obs_f64, &d2);
        }
        cold_settle();
        e.social_viol = 0; e.ego_viol = 0; e.obst_steps = 0;  // ENV:1260-1262
        if (!ext) {
        e.clock += cn_div1000((double)p->settle_ms);              // TRAIN:114 time.sleep(0.1)
        if constexpr (SIM != 0) sim_advance_ticks<SIM>(p, e, env, lane, L.ped, pedv, (double*)smem, p->settle_ms);
        else sim_advance(p, e, env, lane, L.ped, pedv, p->settle_ms);
        }
        e.done = 0;                                           // TRAIN:116
        e.ep_step = 0; e.ep_ret = 0.0; e.pending = 0;
        CN_SYNC();
    }

    }
    cold_settle();
    asm volatile("" :: "v"(trk_warm));   // keeps the warming load (its value is irrelevant)
    CN_T(18);
    // ---- write env state back ---------------------------------------------------------------------
    for (int i = lane; i < 2 * P; i += 64) { gped_p[i] = L.ped[i]; if constexpr (!CMP) gped_v[i] = pedv[i]; }     // (compact: pedv_save() did)
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
        si[CN_SI_NCONF] = e.nconf; si[CN_SI_NENTRIES] = e.nent; si[CN_SI_PENDING_RESET] = e.pending; si[CN_SI_EPISODES] = e.episodes;
        si[CN_SI_CROWD_LO] = (int)(unsigned)((unsigned long long)e.crowd_ms & 0xffffffffull);
        si[CN_SI_CROWD_HI] = (int)(unsigned)((unsigned long long)e.crowd_ms >> 32);
    }
    CN_T(19);
}

// The product kernel (simulated sensors) and its sibling for externally supplied /scan + /odom.  Two
// instantiations keep the external-data branch out of the hot kernel's registers.
// 4 waves per SIMD = 128 VGPRs: the allocator lands on 131 without the bound (3 waves per SIMD) and on 128 with it, no spills.
// (Not in the profiling build: there the bound trips an LLVM "even aligned vector registers" assertion.)
#ifdef CN_TIMING
#define CN_HOT_BOUNDS __launch_bounds__(64)
#define CN_S720_BOUNDS __launch_bounds__(64)
#else
#define CN_HOT_BOUNDS __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
// the 720-ray shape: 13.2 KB of LDS per environment puts 12 wavefronts on a CU (3 per SIMD) whatever the register count, so the
// allocator may use the 168 VGPRs that occupancy allows instead of squeezing into 128 (cn_env_kernel_s720: 66 -> SGPR spills below)
#define CN_S720_BOUNDS __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif
// Two translation units (csrc/build.sh): CN_TU 1 = every kernel except the sequence kernels, CN_TU 2 = the sequence kernels alone,
// compiled with -mllvm -disable-machine-licm.  Their step loop wraps the whole step body; MachineLICM hoists every constant and
// address the body materialises out of that loop and the register allocator then spills them (cn_env_kernel_seq: 155 SGPR + 6
// VGPR spills, 28 bytes of scratch; without the pass 6 / 0 / 0 and 3 % faster).  Unset = one unit with everything.
#if !defined(CN_TU) || CN_TU == 1
extern "C" __global__ void CN_HOT_BOUNDS cn_env_kernel(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void CN_HOT_BOUNDS cn_env_kernel_fair(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, false, 0, false, true>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void CN_HOT_BOUNDS cn_env_kernel_s360(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, false, 0, false, false, 360>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void CN_HOT_BOUNDS cn_env_kernel_fair_s360(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, false, 0, false, true, 360>(blockIdx.x, threadIdx.x, cn_smem); }
// FOUR environments per workgroup (256 threads; one wavefront is still one environment and the four never synchronise): a quarter of
// the workgroups for the dispatcher to create per launch -- the grid's start-up ramp is part of every step of a one-launch-per-step chain
#ifdef CN_TIMING
#define CN_HOT4_BOUNDS __launch_bounds__(1024)
#else
#define CN_HOT4_BOUNDS __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
extern "C" __global__ void CN_HOT4_BOUNDS cn_env_kernel_s360_w4(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; const int w_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); env_kernel_body<false, false, 0, false, 0, false, false, 360>(blockIdx.x * (blockDim.x >> 6) + w_, threadIdx.x & 63, cn_smem + (size_t)w_ * ((KP)__builtin_amdgcn_kernarg_segment_ptr())->wave_lds); }
extern "C" __global__ void CN_HOT4_BOUNDS cn_env_kernel_fair_s360_w4(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; const int w_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); env_kernel_body<false, false, 0, false, 0, false, true, 360>(blockIdx.x * (blockDim.x >> 6) + w_, threadIdx.x & 63, cn_smem + (size_t)w_ * ((KP)__builtin_amdgcn_kernarg_segment_ptr())->wave_lds); }
// TWO wavefronts per environment (128 threads): small grids -- up to two wavefronts per SIMD would be resident anyway (cn_create: n_envs
// <= 8 x CUs, BASELINE configs[3]'s 2048-env shard, the N = 1 `Env`) -- where a step is as long as ONE wavefront's dependent chain
#ifdef CN_TIMING
#define CN_X2_BOUNDS __launch_bounds__(128)
#else
#define CN_X2_BOUNDS __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
extern "C" __global__ void CN_X2_BOUNDS cn_env_kernel_s360_x2(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, false, 0, false, false, 360, true>(blockIdx.x, threadIdx.x & 63, cn_smem, 0, nullptr, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)); }
extern "C" __global__ void CN_S720_BOUNDS cn_env_kernel_s720(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, false, 0, false, false, 720>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void CN_S720_BOUNDS cn_env_kernel_fair_s720(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, false, 0, false, true, 720>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_same(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, true, 0>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_ext(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<true, false, 0>(blockIdx.x, threadIdx.x, cn_smem); }
#endif
#if !defined(CN_TU) || CN_TU == 2 || CN_TU == 3 || CN_TU == 5
// cn_step_sequence: T control periods per launch with OPEN-LOOP actions (resident in HBM: [T][N][2], or one [N][2] held for T
// periods).  One wavefront keeps its environment for the whole launch and walks its T steps at its own pace: no launch boundary,
// no device-wide join between steps -- the launch ends with its slowest wavefront's T steps, not with T x the slowest single
// step -- and after a few steps the wavefronts of a SIMD are out of phase (they stop contending for the same unit at the same
// time), which is what one launch per step can never be.  Each step is exactly cn_env_kernel's (next-step reset convention) and
// writes its observation / reward / done / indices to slot t of the caller's buffers (stride 0: in place).
template <bool GT, int SHAPE = 0, int SIM = 0, int LAYOUT = 0>
__device__ __forceinline__ void sequence_body()
{
    extern __shared__ __attribute__((aligned(16))) char cn_smem[];
    KP p = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    const long long T = p->roll_steps;
    const int wslot = (int)__builtin_amdgcn_s_getreg(4 | (1 << 11));     // HW_ID.wave_id & 3: this wave's slot on its SIMD
    for (long long t = 0; t < T; ++t) {
        int lane_ = threadIdx.x;
        asm volatile("" : "+v"(lane_));          // per-step laundering (see env_kernel_body): nothing is hoisted out of the step loop
        lane_ &= 63;
        cn_setprio_uniform((int)(t + wslot) & 3);      // see "issue arbitration" at the top: every slot gets every level in turn
        env_kernel_body<false, false, LAYOUT, GT, SIM, true, false, SHAPE>(blockIdx.x, lane_, cn_smem, t);
    }
}
#endif
#if !defined(CN_TU) || CN_TU == 2
extern "C" __global__ void CN_HOT_BOUNDS cn_env_kernel_seq(CnKParams p) { sequence_body<false>(); }
extern "C" __global__ void CN_HOT_BOUNDS cn_env_kernel_seq_s360(CnKParams p) { sequence_body<false, 360>(); }
extern "C" __global__ void CN_S720_BOUNDS cn_env_kernel_seq_s720(CnKParams p) { sequence_body<false, 720>(); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt_seq(CnKParams p) { sequence_body<true>(); }
#endif
#if !defined(CN_TU) || CN_TU == 3
// round 5: the same persistent-wavefront form for the other simulators (SIM 2 / 4: social-force pedestrians, pair matrix / dense;
// SIM 3: the diff-drive plugin's wheel ramp) -- the worlds one trains in "as Gazebo delivers it" -- in both risk modes
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_seq_sf(CnKParams p) { sequence_body<false, 0, 2>(); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_seq_sfd(CnKParams p) { sequence_body<false, 0, 4>(); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_seq_wa(CnKParams p) { sequence_body<false, 0, 3>(); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt_seq_sf(CnKParams p) { sequence_body<true, 0, 2>(); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt_seq_sfd(CnKParams p) { sequence_body<true, 0, 4>(); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt_seq_wa(CnKParams p) { sequence_body<true, 0, 3>(); }
#endif
#if !defined(CN_TU) || CN_TU == 5
// round 6: ... and for the worlds that were still refused -- the contact ticks (SIM 1, both risk modes) and the two older observation
// layouts (ORIG: 363 inputs, RW: 370 inputs).  Every configuration cn_create accepts now has both one-launch forms.
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_seq_ct(CnKParams p) { sequence_body<false, 0, 1>(); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt_seq_ct(CnKParams p) { sequence_body<true, 0, 1>(); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_seq_orig(CnKParams p) { sequence_body<false, 0, 0, 1>(); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_seq_rw(CnKParams p) { sequence_body<false, 0, 0, 2>(); }
#endif
#if !defined(CN_TU) || CN_TU == 1
// risk_mode gt: the perceived-risk features from the simulator's own pedestrians (no segmentation, no tracker)
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, true>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt_same(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, true, 0, true>(blockIdx.x, threadIdx.x, cn_smem); }
// ped_contact = 1: the simulator with rigid contacts (10 ms physics ticks); separate instantiations keep the default kernels lean
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_ct(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, false, 1>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_ct_same(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, true, 0, false, 1>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt_ct(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, true, 1>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt_ct_same(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, true, 0, true, 1>(blockIdx.x, threadIdx.x, cn_smem); }
// ped_mode = 2: social-force pedestrians (10 ms physics ticks), for both risk modes
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_sf(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, false, 2>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_sf_same(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, true, 0, false, 2>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt_sf(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, true, 2>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt_sf_same(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, true, 0, true, 2>(blockIdx.x, threadIdx.x, cn_smem); }
// ped_mode = 2 with a crowd too large for the pair matrix (up to 128 pedestrians): per-lane near masks (sim_advance_sf<true>)
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_sfd(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, false, 4>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_sfd_same(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, true, 0, false, 4>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt_sfd(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, true, 4>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt_sfd_same(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, true, 0, true, 4>(blockIdx.x, threadIdx.x, cn_smem); }
// wheel_accel > 0: the diff-drive plugin's wheel-speed ramp (10 ms plugin ticks for the robot), for both risk modes
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_wa(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, false, 3>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_wa_same(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, true, 0, false, 3>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt_wa(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 0, true, 3>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_gt_wa_same(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, true, 0, true, 3>(blockIdx.x, threadIdx.x, cn_smem); }
// obs_layout 2 (environment_stage_1_nobonus_realworld.py): the 370-input physical-robot variant
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_rw(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 2>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_rw_same(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, true, 2>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_rw_ext(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<true, false, 2>(blockIdx.x, threadIdx.x, cn_smem); }
// obs_layout 1 (environment_stage_1_original.py): same physics and lidar, no tracker
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_orig(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, false, 1>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_orig_same(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<false, true, 1>(blockIdx.x, threadIdx.x, cn_smem); }
extern "C" __global__ void __launch_bounds__(64) cn_env_kernel_orig_ext(CnKParams p) { extern __shared__ __attribute__((aligned(16))) char cn_smem[]; env_kernel_body<true, false, 1>(blockIdx.x, threadIdx.x, cn_smem); }

// cn_create: bbox_size() at the spawn pose, evaluated once by the same device code the step kernel would run
extern "C" __global__ void __launch_bounds__(64) cn_bbox_kernel(CnKParams pv, double* out)
{
    __shared__ double stage[64];
    KP p = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    const double v = bbox_size(p, stage, threadIdx.x, p->R - 1, p->spawn_x, p->spawn_y, p->spawn_yaw);
    if (threadIdx.x == 0) out[0] = v;
}

// float32 views of the per-env returns (for the RCCL all-gather of episode returns) and counters
extern "C" __global__ void cn_gather_kernel(CnKParams p, float* last_ret, float* run_ret, int32_t* counters)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.N) return;
    const char* rec = p.state + (size_t)i * (size_t)p.state_stride;
    const double* sd = (const double*)(rec + CN_ST_OFF_SD);
    const int* si = (const int*)(rec + CN_ST_OFF_SI);
    if (last_ret) last_ret[i] = (float)sd[CN_SD_LAST_RETURN];
    if (run_ret) run_ret[i] = (float)sd[CN_SD_EP_RETURN];
    if (counters) {
        int32_t* c = counters + (size_t)i * CN_COUNTER_COLS;
        c[0] = si[CN_SI_EGO_VIOL]; c[1] = si[CN_SI_SOCIAL_VIOL]; c[2] = si[CN_SI_OBST_STEPS]; c[3] = si[CN_SI_EP_STEP];
        c[4] = si[CN_SI_SUCCESS]; c[5] = si[CN_SI_FAILURE]; c[6] = si[CN_SI_STATUS]; c[7] = si[CN_SI_NTRACKS];
        c[8] = si[CN_SI_EPISODES]; c[9] = si[CN_SI_PENDING_RESET];
        c[10] = (int32_t)sd[CN_SD_LAST_EGO_VIOL]; c[11] = (int32_t)sd[CN_SD_LAST_SOCIAL_VIOL];
        c[12] = (int32_t)sd[CN_SD_LAST_OBST_STEPS]; c[13] = (int32_t)sd[CN_SD_LAST_EP_STEPS];
    }
}

// ---- policy tail of the TD3 actor (the caller of the hot path) ---------------------------
// One launch instead of ~10 elementwise ones: action heads sigmoid(l0)*max_v / tanh(l1)*max_w (TD3:103-104),
// Gaussian exploration noise N(0, sigma) (TD3:67-78, 209-211) from a counter-based RNG, clip to
// v in [0, max_v], w in [-max_w, max_w] (TD3:214-215).
extern "C" __global__ void cn_policy_tail_kernel(const float* __restrict__ logits, float* __restrict__ action, int n,
                                                 float max_v, float max_w, float sigma, uint64_t seed, uint64_t counter)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float l0 = logits[2 * i], l1 = logits[2 * i + 1];
    float v = max_v / (1.0f + __expf(-l0));
    float w = max_w * tanhf(l1);
    if (sigma > 0.0f) {
        uint64_t h = cn_mix64(seed ^ cn_mix64(counter));
        h = cn_mix64(h ^ (uint64_t)(uint32_t)i);
        float u1 = ((float)(uint32_t)(h >> 40) + 1.0f) * (1.0f / 16777217.0f);   // (0, 1]
        float u2 = (float)(uint32_t)((h >> 8) & 0xffffffu) * (1.0f / 16777216.0f);
        float r = sqrtf(-2.0f * __logf(u1)), s_, c_;
        __sincosf(6.28318530718f * u2, &s_, &c_);
        v += sigma * r * c_;
        w += sigma * r * s_;
    }
    action[2 * i] = fminf(fmaxf(v, 0.0f), max_v);
    action[2 * i + 1] = fminf(fmaxf(w, -max_w), max_w);
}

#endif   // CN_TU 1
// ---- fused TD3 actor: 3 x Linear(256) + ReLU + output stage in ONE launch (the caller of the hot path, A33) -------
// (device helpers: both translation units -- cn_actor_kernel is unit 1's, cn_policy_kernel unit 2's)
// Actor.forward (TD3:96-106) + Agent.act's noise and clip (TD3:209-215) for a tile of 16 environments per workgroup,
// on the f32-input matrix cores: v_mfma_f32_16x16x4_f32 (exact f32: a k-ordered fmaf chain, same precision as the
// reference's fp32 PyTorch actor).  Lane l feeds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15].
// Round 3: every wave runs the FULL K range of its own 256 / NW columns (NW = 8 waves: two interleaved column tiles per wave,
// col = 32 wave + 2 j + t, one 8-byte load of the K-major weights feeds both MFMAs), so there are no K-split partial sums
// to park in LDS and re-add: the tile needs X [16][Dp + 1] and one hidden buffer [16][257] -- 42 KB instead of 108 KB, 512
// threads instead of 1024 -- which is what lets an actor workgroup sit on a CU NEXT TO a dozen environment wavefronts
// (rollout_groups: one group's actor overlaps the others' env steps; before, it had to wait for 108 KB of LDS to drain).
// Activations never leave LDS; weights stream from L2 (670 KB, shared by all tiles).
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ACT_H 256
#define ACT_M 16
#define ACT_THREADS 512            /* cn_actor_kernel: 8 waves */

// Weights arrive PACKED in the order the matrix cores consume them (cn_actor_pack_weights / cn_actor_pack_kernel below): for a
// layer with K inputs (a multiple of 32) and 256 outputs, block b = 8 k-steps of 4, wave w = 32 columns, q = a pair of k-steps,
//   P[((((b 8 + w) 4 + q) 64 + lane) 4 + j] = W^T[k = 32 b + 4 (2 q + (j >> 1)) + (lane >> 4)][c = 32 w + 2 (lane & 15) + (j & 1)]
// so a lane's operands for one block are FOUR 16-byte loads 1 KB apart and a wave's are 4 KB contiguous.  With the K-major
// layout the same operands were eight 8-byte loads (16 lanes x 8 B on each of 4 rows per instruction); tools/micro/l2_stream.hip:
// a workgroup streaming a shared 688 KB array out of L2 gets 70-73 GB/s with global_load_dwordx2 and 114-139 GB/s with
// dwordx4 -- and the tile's two layers ran at exactly that dwordx2 pace (19.5 B/clk per CU in both), whatever the prefetch depth.
#define ACT_U 8                    /* k-steps per pipelined block */
struct ActW { float4 v[ACT_U / 2]; };
// A lane's operands of block `blk`: uniform base (SGPR pair, advanced per block on the scalar unit) + this lane's 32-bit
// offset + an immediate -- global_load_dwordx4 v, v_off, s[base] offset:1024 i -- so the loop has no 64-bit vector address
// arithmetic (with a per-lane pointer every load cost a v_add_co / v_addc pair and their s_nop).
struct ActWPtr { const float4* base; unsigned off; };
__device__ __forceinline__ void actor_wload(ActW& w, const ActWPtr bp, int blk)
{
    const float4* q = bp.base + (size_t)blk * (8 * 4 * 64);
#pragma unroll
    for (int i = 0; i < ACT_U / 2; ++i) w.v[i] = q[bp.off + (unsigned)(i * 64)];
}
__device__ __forceinline__ ActWPtr actor_wptr(const float* __restrict__ WP, int wave, int lane)
{
    return ActWPtr{reinterpret_cast<const float4*>(WP) + (size_t)wave * (4 * 64), (unsigned)lane};
}
#if !defined(CN_TU) || CN_TU == 1
extern "C" __global__ void cn_actor_pack_kernel(const float* __restrict__ wt, int K, float* __restrict__ packed)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;        // index into `packed`
    if (idx >= K * ACT_H) return;
    const int j = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) & 3, w = (idx >> 10) & 7, b = idx >> 13;
    const int k = 32 * b + 4 * (2 * q + (j >> 1)) + (lane >> 4), c = 32 * w + 2 * (lane & 15) + (j & 1);
    packed[idx] = wt[(size_t)k * ACT_H + c];
}
#endif

// One layer for this wave's 32 columns (two interleaved 16-column tiles: col = 32 wave + 2 j + t):
// out[r][c] = relu(sum_k A[r][k] W^T[k][c] + bias[c]), K a multiple of 32, k ascending.  `first`: the weights of block 0,
// requested by the caller BEFORE the barrier that releases A (their L2 round trip overlaps the staging / the previous layer's
// tail).  The loop keeps the NEXT block -- its weights from L2 AND its A operands from LDS -- in flight while this block's 16
// MFMAs issue, and its body is BRANCH-FREE on purpose: with `if (blk + 2 < nblk) load` in it the compiler's s_waitcnt counting
// merged the "loaded" and "not loaded" paths and waited for the block it had just requested.  The last pair re-requests the
// final block instead (clamped, never skipped).
struct ActA { float a[ACT_U]; };
__device__ __forceinline__ void actor_aload(ActA& x, const float* ap, int k0)
{
#pragma unroll
    for (int u = 0; u < ACT_U; ++u) x.a[u] = ap[k0 + 4 * u];
}
// FINAL (the second hidden layer): the activations are not written back -- linear3 (TD3:101) is folded into the epilogue: every
// lane multiplies its 4 rows x 2 columns of relu(h2) by linear3's weights of those columns, a DPP scan sums the 16 lanes
// (= 32 columns) of each row group, and lane 15 of the group leaves the wave's partial logits in out[(wave 16 + row) 2 + o];
// the caller adds the eight waves in wave order.  (Before: 16 x 256 activations through LDS, a barrier, 8 k LDS reads.)
__device__ __forceinline__ float actor_row_sum(float v)      // inclusive scan over the 16 lanes of a DPP row: lane 15 = the sum
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));   // row_shr:1
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false));   // row_shr:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, false));   // row_shr:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, false));   // row_shr:8
    return v;
}
template <bool FINAL = false>
__device__ __forceinline__ void actor_layer(const float* __restrict__ A, int lda, int K, const float* __restrict__ WP,
                                            const float* __restrict__ bias, float* __restrict__ out, int ldo, int wave, int lane,
                                            const ActW& first, const float* __restrict__ W3 = nullptr)
{
    const int ai = lane & 15, ak = lane >> 4;
    const int colb = 32 * wave + 2 * ai;
    float w3[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FINAL) { w3[0] = W3[colb]; w3[1] = W3[colb + 1]; w3[2] = W3[ACT_H + colb]; w3[3] = W3[ACT_H + colb + 1]; }
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const float* ap = A + ai * lda + ak;
    const ActWPtr bp = actor_wptr(WP, wave, lane);
    // two register blocks, ping-pong: block b's MFMAs run on one while the other receives block b + 1 (a copy `cur = nxt` at the
    // end of an iteration would wait for the loads it is supposed to hide).  The scheduling barriers keep the compiler from
    // sinking the loads below the MFMAs.  (A ring of four blocks, three requests ahead, measured the same 17.5 us: layer 1 7 %
    // faster, layer 2 10 % slower for its longer ramp -- the loads are not latency-bound any more.)
    ActW w0 = first, w1;
    ActA a0, a1;
    const int nblk = K / (4 * ACT_U);
    auto mma = [&](const ActW& w, const ActA& x) {
#pragma unroll
        for (int i = 0; i < ACT_U / 2; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.a[2 * i], w.v[i].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.a[2 * i], w.v[i].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.a[2 * i + 1], w.v[i].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.a[2 * i + 1], w.v[i].w, acc1, 0, 0, 0);
        }
    };
    actor_aload(a0, ap, 0);
    int blk = 0;
#ifndef ACT_ABLATE
#define ACT_ABLATE 0        /* experiments only: 1 = no weight loads in the loop, 2 = no A loads, 3 = neither */
#endif
    if (ACT_ABLATE & 1) w1 = w0;
    if (ACT_ABLATE & 2) a1 = a0;
    for (; blk + 1 < nblk; blk += 2) {
        if (!(ACT_ABLATE & 1)) actor_wload(w1, bp, blk + 1);
        if (!(ACT_ABLATE & 2)) actor_aload(a1, ap, (blk + 1) * 4 * ACT_U);
        __builtin_amdgcn_sched_barrier(0);
        mma(w0, a0);
        __builtin_amdgcn_sched_barrier(0);
        const int nb = min(blk + 2, nblk - 1);
        if (!(ACT_ABLATE & 1)) actor_wload(w0, bp, nb);
        if (!(ACT_ABLATE & 2)) actor_aload(a0, ap, nb * 4 * ACT_U);
        __builtin_amdgcn_sched_barrier(0);
        mma(w1, a1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (blk < nblk) mma(w0, a0);                        // odd block count: the last pair left block nblk - 1 in w0 / a0
    const int rowb = ak * 4;                            // C/D: col = lane & 15, row = (lane >> 4) * 4 + reg
    const float bv0 = bias[colb], bv1 = bias[colb + 1];
    if constexpr (!FINAL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            out[(rowb + r) * ldo + colb] = fmaxf(acc0[r] + bv0, 0.f);
            out[(rowb + r) * ldo + colb + 1] = fmaxf(acc1[r] + bv1, 0.f);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float h0 = fmaxf(acc0[r] + bv0, 0.f), h1 = fmaxf(acc1[r] + bv1, 0.f);
            const float l0 = actor_row_sum(fmaf(h1, w3[1], h0 * w3[0]));
            const float l1 = actor_row_sum(fmaf(h1, w3[3], h0 * w3[2]));
            if (ai == 15) { out[(wave * ACT_M + rowb + r) * 2] = l0; out[(wave * ACT_M + rowb + r) * 2 + 1] = l1; }
        }
    }
}

#if defined(CN_TIMING) && (!defined(CN_TU) || CN_TU == 1)
// profiling build: s_memtime stamps of workgroup b's wave 0 at [b][8] (tools/actor_timing.py)
__device__ long long* cn_actor_timing = nullptr;
extern "C" int cn_debug_set_actor_timing(long long* dev_buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(cn_actor_timing), &dev_buf, sizeof(dev_buf)) == hipSuccess ? 0 : -4;
}
#define ACT_T(k) do { if (cn_actor_timing && threadIdx.x == 0) cn_actor_timing[(size_t)blockIdx.x * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define ACT_T(k) do { } while (0)
#endif
// One tile of 16 environments through the actor (TD3:96-106 + 209-215), by the NW waves of a workgroup (all of its threads must
// call this).  obs / action: the tile's first row; n_live: rows of the tile that exist; act_sm: 16 (Dp + 1) + 16 * 257 floats of
// LDS.  Ends with the actions in global memory (the caller synchronises before anyone reads them).
// `active` (wave-uniform): the policy kernel's workgroups have 16 waves; the eight that do not take part in the tile only keep
// the barrier count.  action2: a second copy of the actions (LDS, or NULL).
template <int NW>        // NW = 8 (the packed weight layout is laid out for 8 waves x 32 columns)
__device__ __forceinline__ void actor_tile(const float* __restrict__ obs, int n_live, int row0, int D, int Dp,
        const float* __restrict__ W1T, const float* __restrict__ b1, const float* __restrict__ W2T,
        const float* __restrict__ b2, const float* __restrict__ W3, const float* __restrict__ b3,
        float* __restrict__ action, float* action2, float max_v, float max_w, float sigma, uint64_t seed, uint64_t counter,
        float* act_sm, const bool active = true, const int tid_in = -1)
{
    static_assert(NW == 8, "packed weights: 8 waves x 32 columns");
    const int ldx = Dp + 1, ldh = ACT_H + 1;
    float* X = act_sm;                 // [16][Dp + 1]; layer 2 writes its output here (the observations are dead by then)
    float* H = X + ACT_M * ldx;        // [16][257] hidden activations of layer 1
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    ActW w1, w2;
    ACT_T(0);
    if (active) {
    actor_wload(w1, actor_wptr(W1T, wave, lane), 0);                   // in flight while the observations are staged
    // Staging the tile: rows by wave, coalesced.  Every load of a chunk (8 x 64 columns of each of the wave's rows) is issued
    // before the first store: written as `X[c] = src[c]` the loop paid one L2 round trip per 64 columns, in series -- 7 to 14 of
    // them, half of the tile's latency.
    constexpr int RPW = ACT_M / NW, CH = 8;
    for (int c0 = 0; c0 < Dp; c0 += 64 * CH) {
        float v[RPW][CH];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int r = wave + q * NW;
            const float* src = obs + (size_t)r * D;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int c = c0 + lane + 64 * j;
                v[q][j] = (r < n_live && c < D) ? src[c] : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int r = wave + q * NW;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int c = c0 + lane + 64 * j;
                if (c < Dp) X[r * ldx + c] = v[q][j];
            }
        }
    }
    }
    ACT_T(1);
    __syncthreads();
    ACT_T(2);
    if (active) {
    actor_layer(X, ldx, Dp, W1T, b1, H, ldh, wave, lane, w1);
    actor_wload(w2, actor_wptr(W2T, wave, lane), 0);                   // ... and while the slowest wave finishes layer 1
    }
    ACT_T(3);
    __syncthreads();
    ACT_T(4);
    float* PL = X;                     // [8 waves][16 rows][2]: the waves' partial logits (the observations are dead by now)
    if (active) actor_layer<true>(H, ldh, ACT_H, W2T, b2, PL, 0, wave, lane, w2, W3);
    ACT_T(5);
    __syncthreads();
    ACT_T(6);
    if (tid < 2 * ACT_M)
    {   // heads, exploration noise, clip: thread = (env i, output o)
        const int i = tid >> 1, o = tid & 1, part = 0;
        float logit = b3[o];
#pragma unroll
        for (int w = 0; w < NW; ++w) logit += PL[(w * ACT_M + i) * 2 + o];
        const int e = row0 + i;
        float val = (o == 0) ? max_v / (1.0f + __expf(-logit)) : max_w * tanhf(logit);
        if (sigma > 0.0f) {   // same generator as cn_policy_tail_kernel: keyed by (seed, counter, env row)
            uint64_t hh = cn_mix64(seed ^ cn_mix64(counter));
            hh = cn_mix64(hh ^ (uint64_t)(uint32_t)e);
            float u1 = ((float)(uint32_t)(hh >> 40) + 1.0f) * (1.0f / 16777217.0f);
            float u2 = (float)(uint32_t)((hh >> 8) & 0xffffffu) * (1.0f / 16777216.0f);
            float rr_ = sqrtf(-2.0f * __logf(u1)), s_, c_;
            __sincosf(6.28318530718f * u2, &s_, &c_);
            val += sigma * rr_ * ((o == 0) ? c_ : s_);
        }
        val = (o == 0) ? fminf(fmaxf(val, 0.0f), max_v) : fminf(fmaxf(val, -max_w), max_w);
        if (part == 0 && i < n_live) {
            action[2 * (size_t)i + o] = val;
            if (action2) action2[2 * (size_t)i + o] = val;
        }
    }
    ACT_T(7);
}

#if !defined(CN_TU) || CN_TU == 1
extern "C" __global__ void __launch_bounds__(ACT_THREADS) cn_actor_kernel(const float* __restrict__ obs, int n, int D, int Dp,
        const float* __restrict__ W1T, const float* __restrict__ b1, const float* __restrict__ W2T,
        const float* __restrict__ b2, const float* __restrict__ W3, const float* __restrict__ b3,
        float* __restrict__ action, float max_v, float max_w, float sigma, uint64_t seed, uint64_t counter)
{
    extern __shared__ __attribute__((aligned(16))) float act_sm[];
    const int row0 = blockIdx.x * ACT_M;
    actor_tile<ACT_THREADS / 64>(obs + (size_t)row0 * D, min(ACT_M, n - row0), row0, D, Dp, W1T, b1, W2T, b2, W3, b3,
                                 action + 2 * (size_t)row0, nullptr, max_v, max_w, sigma, seed, counter, act_sm);
}
#endif

#if !defined(CN_TU) || CN_TU == 2 || CN_TU == 4 || CN_TU == 5
// ---- cn_rollout_policy: T control periods per launch with the POLICY IN THE LOOP --------------------------------------------
// A workgroup = 16 environments = 16 wavefronts (one CU's worth at 4 per SIMD).  Per control period: the first eight waves run
// the TD3 actor (actor_tile above: the arithmetic, noise keys and clip of cn_actor_forward) on the 16 observations the
// workgroup's environments wrote one period earlier and leave the 16 actions in LDS and in slot t of the caller's action array;
// a workgroup barrier; every wave advances its environment by one Env.step with its action (exactly cn_env_kernel's step,
// next-step reset convention) and writes observation / reward / done / indices to slot t; a workgroup barrier.  No launch and
// no device-wide join between periods -- the only joins are among the 16 waves of a CU -- and the observation -> actor hand-off
// never leaves the CU's L2 slice.  Bit-identical to T x (cn_actor_forward, cn_step(auto_reset 2)) with counters c, c + 1, ...
// The actor's LDS tile (42 KB) overlays the environments' working sets, which are dead between two steps (everything a step
// needs it reloads from the state record); only the 16 actions live outside them.
// Round 5: the workgroup is 16 OR 8 environments (blockDim.x / 64, cn_create picks: 8 where 16 working sets do not fit one CU's LDS --
// 720 rays x 100 pedestrians, dense social force -- then two workgroups share a CU and run out of phase by themselves; the actor
// tile keeps its 16 rows, the upper 8 are padding), and the step body is instantiated for every simulator.
#define POL_ENVS 16            /* the largest workgroup: launch bounds, the actor tile's rows */
#ifndef POL_ACTOR_WAVES
#define POL_ACTOR_WAVES(w) ((w) < 8)     /* experiments: (false) = barriers and heads only, to time the env phase alone */
#endif
#ifndef POL_FAIR
#define POL_FAIR 1            /* experiments: 0 = the sequence kernel's rotating levels instead of the falling ones */
#endif
template <int SHAPE, bool GT = false, int SIM = 0, int LAYOUT = 0>
__device__ __forceinline__ void policy_sequence_body()
{
    extern __shared__ __attribute__((aligned(16))) char cn_smem[];
    KP p0 = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    const int T = (int)p0->roll_steps;
    for (int t = 0; t < T; ++t) {
        unsigned long long pp = (unsigned long long)p0;
        asm volatile("" : "+s"(pp));             // per-period laundering (see env_kernel_body): the actor's ~20 parameters, the wave's
        KP p = (KP)pp;                           // index and everything derived from them are re-made each period instead of living in
        int tid_ = threadIdx.x;                  // registers across the whole step
        asm volatile("" : "+v"(tid_));
        const int lane_ = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
        const int PE = p->pol_envs;                                               // environments (= waves) per workgroup: 16 or 8
        const int row0 = blockIdx.x * PE, env = row0 + wave;
        const int ws = p->pol_wave_lds, D = p->pol_D;
        float* const act_lds = (float*)(cn_smem + (size_t)p->pol_act_off);          // [PE][2], past the working sets and the tile
        const float* ob = (t == 0 ? p->pol_obs0 : p->obs + (size_t)((t - 1) * p->roll_obs_stride)) + (size_t)row0 * D;
        float* ac = const_cast<float*>(p->action) + (size_t)(t * p->roll_action_in_stride) + 2 * (size_t)row0;
        // (Round 6, measured and not kept: a per-CU lock that makes the two 8-environment workgroups of a CU take turns in the actor
        //  phase.  They already run an actor phase apart by themselves -- tools/policy_phase_timing.py -- and it buys nothing, because a
        //  wave streaming v_mfma_f32_16x16x4_f32 holds the SIMD's vector issue for its 32 cycles: another wave of that SIMD gets ONE
        //  vector instruction in per MFMA (tools/micro/mfma_valu_mix.hip: 5.5 -> 37.8 cycles per v_fma_f64).  Matrix-core time and
        //  vector time ADD on a SIMD; one workgroup's tile beside another's Env.step overlaps nothing.  profiles/r06/.)
#ifdef CN_TIMING
#define POL_T(k) do { if (p->timing && tid_ == 0) p->timing[(size_t)row0 * 32 + 27 + (k)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define POL_T(k) do { } while (0)
#endif
        POL_T(0);
        POL_T(1);
        actor_tile<8>(ob, min(PE, p->N - row0), row0, D, p->pol_Dp, p->pol_w1p, p->pol_b1, p->pol_w2p, p->pol_b2, p->pol_w3, p->pol_b3,
                      ac, act_lds, p->pol_max_v, p->pol_max_w, p->pol_sigma, p->pol_seed, p->pol_counter + (uint64_t)t, (float*)cn_smem, POL_ACTOR_WAVES(wave), tid_);
        __syncthreads();
        POL_T(2);
        if (env < p->N)
        {
#if POL_FAIR == 0
            cn_setprio_uniform((t + (int)__builtin_amdgcn_s_getreg(4 | (1 << 11))) & 3);
#endif
            env_kernel_body<false, false, LAYOUT, GT, SIM, true, POL_FAIR != 0 && LAYOUT == 0, SHAPE>(env, lane_, cn_smem + (size_t)wave * ws, t, act_lds + 2 * wave);
        }
        POL_T(3);
        __syncthreads();
        POL_T(4);
    }
}
#endif
#if !defined(CN_TU) || CN_TU == 2
extern "C" __global__ void __launch_bounds__(64 * POL_ENVS) cn_policy_kernel(CnKParams p) { policy_sequence_body<0>(); }
extern "C" __global__ void __launch_bounds__(64 * POL_ENVS) cn_policy_kernel_s360(CnKParams p) { policy_sequence_body<360>(); }
extern "C" __global__ void __launch_bounds__(64 * POL_ENVS) cn_policy_kernel_gt(CnKParams p) { policy_sequence_body<0, true>(); }   // risk_mode gt
#endif
#if !defined(CN_TU) || CN_TU == 4
extern "C" __global__ void __launch_bounds__(64 * POL_ENVS) cn_policy_kernel_s720(CnKParams p) { policy_sequence_body<720>(); }      // BASELINE configs[4]: 8 per workgroup
extern "C" __global__ void __launch_bounds__(64 * POL_ENVS) cn_policy_kernel_sf(CnKParams p) { policy_sequence_body<0, false, 2>(); }
extern "C" __global__ void __launch_bounds__(64 * POL_ENVS) cn_policy_kernel_sfd(CnKParams p) { policy_sequence_body<0, false, 4>(); }
extern "C" __global__ void __launch_bounds__(64 * POL_ENVS) cn_policy_kernel_wa(CnKParams p) { policy_sequence_body<0, false, 3>(); }
extern "C" __global__ void __launch_bounds__(64 * POL_ENVS) cn_policy_kernel_gt_sf(CnKParams p) { policy_sequence_body<0, true, 2>(); }
extern "C" __global__ void __launch_bounds__(64 * POL_ENVS) cn_policy_kernel_gt_sfd(CnKParams p) { policy_sequence_body<0, true, 4>(); }
extern "C" __global__ void __launch_bounds__(64 * POL_ENVS) cn_policy_kernel_gt_wa(CnKParams p) { policy_sequence_body<0, true, 3>(); }
#endif
#if !defined(CN_TU) || CN_TU == 5
// round 6: the contact ticks and the two older observation layouts (their actors take 363 / 370 inputs: cn_actor_pack_weights pads
// any width to a multiple of 32)
extern "C" __global__ void __launch_bounds__(64 * POL_ENVS) cn_policy_kernel_ct(CnKParams p) { policy_sequence_body<0, false, 1>(); }
extern "C" __global__ void __launch_bounds__(64 * POL_ENVS) cn_policy_kernel_gt_ct(CnKParams p) { policy_sequence_body<0, true, 1>(); }
extern "C" __global__ void __launch_bounds__(64 * POL_ENVS) cn_policy_kernel_orig(CnKParams p) { policy_sequence_body<0, false, 0, 1>(); }
// (the RW observation needs 140 vector registers: 8 environments per workgroup -- two waves per SIMD -- so that the cap is 256, not 128)
extern "C" __global__ void __launch_bounds__(64 * 8) cn_policy_kernel_rw(CnKParams p) { policy_sequence_body<0, false, 0, 2>(); }
#endif

#if !defined(CN_TU) || CN_TU == 1

#ifdef CN_TIMING
// ---- device arithmetic under test (PROFILING BUILD ONLY; tests/test_gpu_parity.py::test_device_math_*): the hand-written
// replacements for libm / compiler expansions, one element per thread.  op: 0 cn_sqrt(x)  1 cn_div(x, y)  2 cn_hypot(x, y)
// 3 cn_atan2_t(x, y) = atan2 with x the ordinate (first argument) and y the abscissa  4, 5 sin, cos of cn_det_sincos_t(x)
__global__ void cn_math_kernel(int op, const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ out, int n,
                               const double* __restrict__ trig)
{
    cn_ktab tab = (cn_ktab)trig;           // the env kernels read this table from their kernel-argument block
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r = 0.0, s_, c_;
    switch (op) {
    case 0: r = cn_sqrt(x[i]); break;
    case 1: r = cn_div(x[i], y[i]); break;
    case 2: r = cn_hypot(x[i], y[i]); break;
    case 3: r = cn_atan2_t(tab, x[i], y[i]); break;
    case 4: cn_det_sincos_t(tab, x[i], &s_, &c_); r = s_; break;
    default: cn_det_sincos_t(tab, x[i], &s_, &c_); r = c_; break;
    }
    out[i] = r;
}
extern "C" int cn_debug_math(int op, const double* x, const double* y, double* out, int n, void* stream)
{
    static const double trig[CN_TRIG_COUNT] = CN_TRIG_TABLE;
    double* d = nullptr;
    if (hipMalloc(&d, sizeof(trig)) != hipSuccess) return -1;
    if (hipMemcpy(d, trig, sizeof(trig), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return -1; }
    hipLaunchKernelGGL(cn_math_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, op, x, y, out, n, (const double*)d);
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d);
    return e == hipSuccess ? 0 : -1;
}

// ---- PMC calibration (PROFILING BUILD ONLY, libcrowdnav_timing.so; tools/calib_pmc.py): known-byte streaming reads / writes at the access widths the
// env kernel uses, so FETCH_SIZE / WRITE_SIZE can be turned into bytes (MI355X_MICROARCH.md, HBM section).
template <typename T>
__global__ void cn_calib_read_kernel(const T* __restrict__ src, size_t n, T* __restrict__ out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    T acc = T(0);
    for (; i < n; i += stride) acc += src[i];
    if (acc == T(123456789)) out[0] = acc;  // never true for the zero-filled buffer; keeps the loads alive
}
template <typename T>
__global__ void cn_calib_write_kernel(T* __restrict__ dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = T(1);
}
extern "C" void cn_calib_launch(void* buf, size_t bytes, int width, int write, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    dim3 g(256 * 16), b(256);
    if (!write) {
        if (width == 4) hipLaunchKernelGGL(cn_calib_read_kernel<float>, g, b, 0, st, (const float*)buf, bytes / 4, (float*)buf);
        else hipLaunchKernelGGL(cn_calib_read_kernel<double>, g, b, 0, st, (const double*)buf, bytes / 8, (double*)buf);
    } else {
        if (width == 4) hipLaunchKernelGGL(cn_calib_write_kernel<float>, g, b, 0, st, (float*)buf, bytes / 4);
        else hipLaunchKernelGGL(cn_calib_write_kernel<double>, g, b, 0, st, (double*)buf, bytes / 8);
    }
}
#endif
#endif
