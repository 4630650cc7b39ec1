// crowdnav_abi.hip -- host side of libcrowdnav.so: the C-ABI declared in include/crowdnav.h.
// Owns the per-env SoA state in HBM, builds the constant tables, and enqueues the fused step
// kernel on the caller's stream.  No CPU fallback exists: without a HIP device cn_create fails.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "crowdnav_device.h"
#include "crowdnav_kernel.h"

extern "C" __global__ void cn_env_kernel(CnKParams p);
extern "C" __global__ void cn_env_kernel_fair(CnKParams p);
extern "C" __global__ void cn_env_kernel_ext(CnKParams p);
extern "C" __global__ void cn_env_kernel_same(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt_same(CnKParams p);
extern "C" __global__ void cn_env_kernel_ct(CnKParams p);
extern "C" __global__ void cn_env_kernel_ct_same(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt_ct(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt_ct_same(CnKParams p);
extern "C" __global__ void cn_env_kernel_sf(CnKParams p);
extern "C" __global__ void cn_env_kernel_sf_same(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt_sf(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt_sf_same(CnKParams p);
extern "C" __global__ void cn_env_kernel_rw(CnKParams p);
extern "C" __global__ void cn_env_kernel_rw_same(CnKParams p);
extern "C" __global__ void cn_env_kernel_rw_ext(CnKParams p);
extern "C" __global__ void cn_env_kernel_orig(CnKParams p);
extern "C" __global__ void cn_env_kernel_orig_ext(CnKParams p);
extern "C" __global__ void cn_env_kernel_orig_same(CnKParams p);
extern "C" __global__ void cn_env_kernel_sfd(CnKParams p);
extern "C" __global__ void cn_env_kernel_sfd_same(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt_sfd(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt_sfd_same(CnKParams p);
extern "C" __global__ void cn_env_kernel_wa(CnKParams p);
extern "C" __global__ void cn_env_kernel_wa_same(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt_wa(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt_wa_same(CnKParams p);
extern "C" __global__ void cn_env_kernel_seq(CnKParams p);
extern "C" __global__ void cn_env_kernel_s360(CnKParams p);
extern "C" __global__ void cn_env_kernel_fair_s360(CnKParams p);
extern "C" __global__ void cn_env_kernel_seq_s360(CnKParams p);
extern "C" __global__ void cn_env_kernel_s360_w4(CnKParams p);
extern "C" __global__ void cn_env_kernel_fair_s360_w4(CnKParams p);
extern "C" __global__ void cn_env_kernel_s360_x2(CnKParams p);

extern "C" __global__ void cn_env_kernel_seq_s720(CnKParams p);
extern "C" __global__ void cn_policy_kernel(CnKParams p);
extern "C" __global__ void cn_policy_kernel_s360(CnKParams p);
extern "C" __global__ void cn_policy_kernel_gt(CnKParams p);
extern "C" __global__ void cn_env_kernel_s720(CnKParams p);
extern "C" __global__ void cn_env_kernel_fair_s720(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt_seq(CnKParams p);
extern "C" __global__ void cn_env_kernel_seq_sf(CnKParams p);
extern "C" __global__ void cn_env_kernel_seq_sfd(CnKParams p);
extern "C" __global__ void cn_env_kernel_seq_wa(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt_seq_sf(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt_seq_sfd(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt_seq_wa(CnKParams p);
extern "C" __global__ void cn_policy_kernel_s720(CnKParams p);
extern "C" __global__ void cn_policy_kernel_sf(CnKParams p);
extern "C" __global__ void cn_policy_kernel_sfd(CnKParams p);
extern "C" __global__ void cn_policy_kernel_wa(CnKParams p);
extern "C" __global__ void cn_policy_kernel_gt_sf(CnKParams p);
extern "C" __global__ void cn_policy_kernel_gt_sfd(CnKParams p);
extern "C" __global__ void cn_policy_kernel_gt_wa(CnKParams p);
extern "C" __global__ void cn_env_kernel_seq_ct(CnKParams p);
extern "C" __global__ void cn_env_kernel_gt_seq_ct(CnKParams p);
extern "C" __global__ void cn_env_kernel_seq_orig(CnKParams p);
extern "C" __global__ void cn_env_kernel_seq_rw(CnKParams p);
extern "C" __global__ void cn_policy_kernel_ct(CnKParams p);
extern "C" __global__ void cn_policy_kernel_gt_ct(CnKParams p);
extern "C" __global__ void cn_policy_kernel_orig(CnKParams p);
extern "C" __global__ void cn_policy_kernel_rw(CnKParams p);
// every kernel launched with cn_create's dynamic LDS size (hipFuncAttributeMaxDynamicSharedMemorySize above 64 KiB)
static const void* const kDynamicLdsKernels[] = {
    (const void*)cn_env_kernel, (const void*)cn_env_kernel_fair, (const void*)cn_env_kernel_ext, (const void*)cn_env_kernel_same,
    (const void*)cn_env_kernel_gt, (const void*)cn_env_kernel_gt_same, (const void*)cn_env_kernel_ct, (const void*)cn_env_kernel_ct_same,
    (const void*)cn_env_kernel_gt_ct, (const void*)cn_env_kernel_gt_ct_same, (const void*)cn_env_kernel_sf, (const void*)cn_env_kernel_sf_same,
    (const void*)cn_env_kernel_gt_sf, (const void*)cn_env_kernel_gt_sf_same, (const void*)cn_env_kernel_rw, (const void*)cn_env_kernel_rw_same,
    (const void*)cn_env_kernel_rw_ext, (const void*)cn_env_kernel_orig, (const void*)cn_env_kernel_orig_ext, (const void*)cn_env_kernel_orig_same,
    (const void*)cn_env_kernel_sfd, (const void*)cn_env_kernel_sfd_same, (const void*)cn_env_kernel_gt_sfd, (const void*)cn_env_kernel_gt_sfd_same,
    (const void*)cn_env_kernel_wa, (const void*)cn_env_kernel_wa_same, (const void*)cn_env_kernel_gt_wa, (const void*)cn_env_kernel_gt_wa_same,
    (const void*)cn_env_kernel_seq, (const void*)cn_env_kernel_s360, (const void*)cn_env_kernel_fair_s360, (const void*)cn_env_kernel_seq_s360,
    (const void*)cn_env_kernel_seq_s720, (const void*)cn_env_kernel_s720, (const void*)cn_env_kernel_fair_s720, (const void*)cn_env_kernel_gt_seq,
    (const void*)cn_env_kernel_seq_sf, (const void*)cn_env_kernel_seq_sfd, (const void*)cn_env_kernel_seq_wa,
    (const void*)cn_env_kernel_gt_seq_sf, (const void*)cn_env_kernel_gt_seq_sfd, (const void*)cn_env_kernel_gt_seq_wa,
    (const void*)cn_env_kernel_s360_w4, (const void*)cn_env_kernel_fair_s360_w4, (const void*)cn_env_kernel_s360_x2,
    (const void*)cn_env_kernel_seq_ct, (const void*)cn_env_kernel_gt_seq_ct, (const void*)cn_env_kernel_seq_orig, (const void*)cn_env_kernel_seq_rw};
static const void* const kPolicyKernels[] = {
    (const void*)cn_policy_kernel, (const void*)cn_policy_kernel_s360, (const void*)cn_policy_kernel_gt, (const void*)cn_policy_kernel_s720,
    (const void*)cn_policy_kernel_sf, (const void*)cn_policy_kernel_sfd, (const void*)cn_policy_kernel_wa,
    (const void*)cn_policy_kernel_gt_sf, (const void*)cn_policy_kernel_gt_sfd, (const void*)cn_policy_kernel_gt_wa,
    (const void*)cn_policy_kernel_ct, (const void*)cn_policy_kernel_gt_ct, (const void*)cn_policy_kernel_orig, (const void*)cn_policy_kernel_rw};
extern "C" __global__ void cn_bbox_kernel(CnKParams p, double* out);
extern "C" __global__ void cn_gather_kernel(CnKParams p, float* last_ret, float* run_ret, int32_t* counters);

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) \
    return fail(CN_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct cn_env_s {
    cn_config cfg;
    int device;
    int D, max_conf, trk_cap;
    size_t lds;
    CnKParams kp;        // template with state/table pointers filled in
    double *d_lidar = nullptr, *d_poly = nullptr, *d_ped_init = nullptr, *d_ped_preset = nullptr, *d_trk = nullptr;
    double* d_ped_aux = nullptr;      // [N, P, 3] ped_mode 2: goal x, goal y, goal counter
    char* d_state = nullptr;          // N per-env records (crowdnav_kernel.h: sd | si | ped_p | ped_v | pad), `stride` bytes apart
    size_t stride;
    std::vector<double> ped_init;
    int arbitration = CN_ARB_AUTO;    // cn_set_arbitration
    int n_cus = 0;                    // compute units of `device` (CN_ARB_AUTO: fair from 2 wavefronts per SIMD = 8 x n_cus envs)
    size_t lds_shape = 0;             // dynamic LDS of the _s720 kernels (compact layout); 0 = this handle has none
    size_t pol_wave_lds = 0, pol_lds = 0;   // cn_rollout_policy: bytes between the environments' LDS working sets of a workgroup; the workgroup's total (0 = does not fit)
    int pol_envs = 0;                 // ... environments per workgroup: 16, or 8 where 16 working sets do not fit one CU's LDS
    size_t pol_act_off = 0;           // ... byte offset of the workgroup's actions (past the working sets and the actor tile)
    int64_t group_envs = 0;           // cn_set_group_envs: environments in flight together with this handle's (0 = alone)
    int x2 = -1;                      // cn_env_kernel_s360_x2 (two wavefronts per environment): -1 = by grid size, 0 / 1 = CN_X2 override
    int wpb4 = 0;                     // 4: the 360-ray step kernels run four environments per workgroup (launches of at most one round of wavefronts); 0: one
    bool shape360 = false;            // the headline shape (360 rays, 20 pedestrians, K = 8 and cn_create's sizes for it): the _s360 kernels
    bool shape720 = false;            // BASELINE configs[4] (720 rays, 100 pedestrians, K = 8): the _s720 kernels
};

// RAII: run on the handle's device even if the calling thread's current device is another one
struct DeviceScope {
    int prev = -1, want;
    explicit DeviceScope(int dev) : want(dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != want) (void)hipSetDevice(want); }
    ~DeviceScope() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
};

static void destroy_handle(cn_env_s* h)
{
    if (!h) return;
    DeviceScope scope(h->device);
    (void)hipFree(h->d_lidar); (void)hipFree(h->d_poly); (void)hipFree(h->d_state);
    (void)hipFree(h->d_ped_init); (void)hipFree(h->d_ped_preset);
    (void)hipFree(h->d_trk); (void)hipFree(h->d_ped_aux);
    delete h;
}
struct HandleDeleter { void operator()(cn_env_s* h) const { destroy_handle(h); } };

// Host restatement of the kernel's association table for bb = bb_spawn (crowdnav_kernel.hip, "ENV:448-485"): the same IEEE
// operations in the same order (+ - * / floor ceil; this file is compiled with -ffp-contract=off like the kernel), so the
// table a wavefront copies equals the one it would compute.  Pinned by the rollout parity tests: the oracle evaluates the IoU
// itself; simulated runs take the copied table, cn_observe_external replays (box size from the external scan) the computed one.
static void build_assoc_table(CnKParams& k, int16_t* tab)
{
    k.assoc_fast = 0; k.assoc_k1 = 0;
    const double Tm = 2000.0 * k.bb_spawn;
    const int K1 = (int)floor(Tm - 1e-7), K2 = (int)ceil(Tm + 1e-7);
    if (!((K2 == K1 + 1) && K1 >= 0 && K1 <= 254)) return;
    const double c_hi = 0.0005 * (1.0 + 1e-6), c_lo = 0.0005 * (1.0 - 1e-6);
    const double T2 = Tm * Tm;
    const double Phi = T2 * (2.0 * c_hi) / (1.0 + c_hi), Plo = T2 * (2.0 * c_lo) / (1.0 + c_lo);
    for (int d = 0; d <= K1 + 1; ++d) {
        int m = -1;
        if (d <= K1) {
            const double fx = Tm - (double)d;
            const double mm = fmin(ceil(Tm - Phi / fx) - 1.0, (double)K1);
            m = mm < 0.0 ? -1 : (int)mm;
            if (m >= 0 && !(fx * (Tm - (double)m) > Phi)) return;
            if (m + 1 <= K1 && !(fx * (Tm - (double)(m + 1)) < Plo)) return;
        }
        tab[d] = (int16_t)m;
    }
    k.assoc_k1 = K1; k.assoc_fast = 1;
}

// regions A + B of the carve below: the simulators' scratch (idle while the world advances)
static size_t lds_scratch_bytes(int R, int P, int K, int max_conf, int trk_cap, bool near_separate)
{
    size_t n = (size_t)(R - 1), mc = (size_t)max_conf;
    size_t szA_pts = (10 * n + 7) & ~(size_t)7, szA_trk = 8 * (size_t)(CN_TF_COUNT * trk_cap);
    size_t szA = szA_pts > szA_trk ? szA_pts : szA_trk;
    size_t szB_g = (6 * n + 7) & ~(size_t)7;
    size_t szB_c = 32 * mc + 8 * 64 + 8 * (size_t)(8 + 4 * K) + 4 * (size_t)CN_MAX_K;
    size_t szB = szB_g > szB_c ? szB_g : szB_c;
    if (szB < 8 * 64) szB = 8 * 64;
    if (!near_separate && szB < 32 * (size_t)(P + 1)) szB = 32 * (size_t)(P + 1);
    return szA + szB;
}
static size_t lds_bytes_impl(int R, int P, int K, int max_conf, int trk_cap, bool near_separate, int layout = 0, bool compact = false)
{
    // must mirror the carve in cn_env_kernel (compact: the 720-ray shape kernels' layout -- int16 end points, 12-byte confirmed
    // objects, the pedestrians' velocities overlaid on region A)
    size_t n = (size_t)(R - 1), mc = (size_t)max_conf;
    size_t szA_pts = ((compact ? 6 : 10) * n + 7) & ~(size_t)7, szA_trk = 8 * (size_t)(CN_TF_COUNT * trk_cap);
    size_t szA = szA_pts > szA_trk ? szA_pts : szA_trk;
    size_t szB_g = (6 * n + 7) & ~(size_t)7;
    size_t szC = compact ? ((12 * mc + 7) & ~(size_t)7) : 32 * mc;
    size_t szB_c = szC + 8 * 64 + 8 * (size_t)(8 + 4 * K) + 4 * (size_t)CN_MAX_K;
    size_t szB = szB_g > szB_c ? szB_g : szB_c;
    if (szB < 8 * 64) szB = 8 * 64;
    if (!near_separate && szB < 32 * (size_t)(P + 1)) szB = 32 * (size_t)(P + 1);   // near-pedestrian list overlaid on region B
    size_t Wn = (n + 63) >> 6;
    size_t b = szA + szB;
    b += 8 * (size_t)(CN_NMASK * Wn);             // bit words
    b += 8 * ((3 * Wn + 1) / 2);                  // wbase
    b += 8 * (size_t)(2 * P + 2) * (compact ? 1 : 2);   // ped, pedv (compact: pedv inside region A)
    if (compact && 8 * (size_t)(2 * P + 2) > szA) return (size_t)1 << 30;
    if (near_separate) b += 32 * (size_t)(P + 1); // near-pedestrian list in its own region
    if (layout == CN_LAYOUT_REALWORLD) b = ((b + 15) & ~(size_t)15) + 12 * n + 16;   // filtered list, gradients, types (12 bytes per ray)
    return (b + 15) & ~(size_t)15;
}

// The near-pedestrian list (ray loop only) gets its own LDS region when that costs no resident wavefront -- measured
// 1.5 % faster than sharing region B with the gradients -- and is overlaid on region B otherwise (100 pedestrians x 720
// rays: 8 instead of 7 wavefronts per CU).
static int waves_per_cu(size_t lds) { size_t w = (160 * 1024) / (lds ? lds : 1); return (int)(w > 16 ? 16 : w); }
int cn_near_separate(int R, int P, int K, int max_conf, int trk_cap)
{
    return waves_per_cu(lds_bytes_impl(R, P, K, max_conf, trk_cap, true)) ==
           waves_per_cu(lds_bytes_impl(R, P, K, max_conf, trk_cap, false));
}
size_t cn_lds_bytes(int R, int P, int K, int max_conf, int trk_cap)
{
    return lds_bytes_impl(R, P, K, max_conf, trk_cap, cn_near_separate(R, P, K, max_conf, trk_cap) != 0);
}

extern "C" int cn_abi_version(void) { return CN_ABI_VERSION; }
extern "C" const char* cn_last_error(void) { return g_err.c_str(); }

static void default_ped_init(const cn_config& c, int env, double* xy)
{
    // seeded uniform in the room, rejected within 0.4 m of the robot spawn
    int64_t gid = c.env_index_base + env;
    double lo = -c.room_half + 0.1, span = 2.0 * c.room_half - 0.2;
    for (int i = 0; i < c.n_peds; ++i) {
        uint32_t att = 0;
        for (;;) {
            double x = fma(span, cn_rng_u01(c.seed, gid, 2u, (uint32_t)i, 2 * att), lo);
            double y = fma(span, cn_rng_u01(c.seed, gid, 2u, (uint32_t)i, 2 * att + 1), lo);
            double dx = x - c.spawn_x, dy = y - c.spawn_y;
            ++att;
            if (fma(dx, dx, dy * dy) >= 0.16 || att > 1000) { xy[2 * i] = x; xy[2 * i + 1] = y; break; }
        }
    }
}

static double angle_increment_deg(int R)
{
    // UTL:113 `max_angle / (resolution - 1)` with Python-2 integer operands (360/359 == 1).
    // For R-1 > 360 Python 2 would give 0; defined as true division (outside the reference's range).
    if (R - 1 <= 360) return (double)(360 / (R - 1));
    return 360.0 / (double)(R - 1);
}

// One field of every env's record <-> a dense host array [N, width bytes]
static int field_to_device(cn_env_s* h, size_t off, const void* host, size_t width)
{
    if (!width) return CN_OK;
    HIPCHK(hipMemcpy2D(h->d_state + off, h->stride, host, width, width, (size_t)h->cfg.n_envs, hipMemcpyHostToDevice));
    return CN_OK;
}
static int field_to_host(cn_env_s* h, size_t off, void* host, size_t width)
{
    if (!width) return CN_OK;
    HIPCHK(hipMemcpy2D(host, width, h->d_state + off, h->stride, width, (size_t)h->cfg.n_envs, hipMemcpyDeviceToHost));
    return CN_OK;
}
#define FIELD(call) do { int rc_ = (call); if (rc_ != CN_OK) return rc_; } while (0)

static int upload_initial_state(cn_env_s* h)
{
    const cn_config& c = h->cfg;
    const int N = c.n_envs, P = c.n_peds;
    std::vector<double> sd((size_t)N * CN_SD_COUNT, 0.0);
    std::vector<int32_t> si((size_t)N * CN_SI_COUNT, 0);
    for (int e = 0; e < N; ++e) {  // Env.__init__ (ENV:43-168)
        double* s = &sd[(size_t)e * CN_SD_COUNT];
        s[CN_SD_RX] = c.spawn_x; s[CN_SD_RY] = c.spawn_y; s[CN_SD_RYAW] = c.spawn_yaw;
        s[CN_SD_WPX] = c.goal_x; s[CN_SD_WPY] = c.goal_y;
        if (c.obs_layout == CN_LAYOUT_REALWORLD) {   // RW:80 `collision_prob = None` (below every number in Python 2), RW:103 bbox constant
            s[CN_SD_CPROB] = -INFINITY; s[CN_SD_BB] = 0.0210;
        }
    }
    HIPCHK(hipMemset(h->d_state, 0, (size_t)N * h->stride));
    FIELD(field_to_device(h, CN_ST_OFF_SD, sd.data(), CN_SD_COUNT * 8));
    FIELD(field_to_device(h, CN_ST_OFF_SI, si.data(), CN_SI_COUNT * 4));
    if (P > 0) {
        HIPCHK(hipMemcpy(h->d_ped_init, h->ped_init.data(), (size_t)N * P * 16, hipMemcpyHostToDevice));
        FIELD(field_to_device(h, CN_ST_OFF_PED_P, h->ped_init.data(), (size_t)P * 16));
        HIPCHK(hipMemset(h->d_ped_preset, 0, (size_t)N * P * 16));
    }
    HIPCHK(hipMemset(h->d_trk, 0, (size_t)N * CN_TF_COUNT * h->trk_cap * 8));
    {   // ped_mode 2: every pedestrian's first goal (goal 0 of its counter-based sequence); zeros otherwise
        std::vector<double> aux((size_t)N * (P > 0 ? P : 1) * 3, 0.0);
        if (c.ped_mode == 2) {
            const double lo = -c.room_half + 0.1, span = 2.0 * c.room_half - 0.2;
            for (int e = 0; e < N; ++e)
                for (int i = 0; i < P; ++i) {
                    double* a = &aux[((size_t)e * P + i) * 3];
                    a[0] = fma(span, cn_rng_u01(c.seed, c.env_index_base + e, 3u, (uint32_t)i, 0u), lo);
                    a[1] = fma(span, cn_rng_u01(c.seed, c.env_index_base + e, 3u, (uint32_t)i, 1u), lo);
                }
        }
        HIPCHK(hipMemcpy(h->d_ped_aux, aux.data(), aux.size() * 8, hipMemcpyHostToDevice));
    }
    return CN_OK;
}

typedef void (*cn_kernel_fn)(CnKParams);
struct KernelChoice { cn_kernel_fn fn; const char* name; bool compact = false; int wpb = 1; bool x2 = false; };      // compact: launched with h->lds_shape; wpb: environments (waves) per workgroup
#define CN_KC(f) KernelChoice{f, #f}
#define CN_KCC(f) KernelChoice{f, #f, true}
static size_t lds_of(const cn_env_s* h, const KernelChoice& kc) { return kc.compact ? h->lds_shape : h->lds; }
static KernelChoice choose_policy_kernel(const cn_env_s* h);

extern "C" int cn_create(const cn_config* cfg, int device, cn_handle* out)
{
    if (!cfg || !out) return fail(CN_ERR_ARG, "cn_create: null argument");
    const cn_config& c = *cfg;
    if (c.n_envs < 1 || c.n_peds < 0 || c.n_peds > 4096 || c.n_rays < 8 || c.n_rays > 1025 || c.k_obstacles < 1 ||
        c.k_obstacles > CN_MAX_K || c.ped_cycle_ms < 1 || c.dt_ms < 1 || c.scan_latency_ms < 1 || c.settle_ms < 0 ||
        c.max_steps < 1 || !(c.track_capacity == 0 || c.track_capacity == 32 || c.track_capacity == 64) ||
        !(c.obs_layout == CN_LAYOUT_RISK || c.obs_layout == CN_LAYOUT_ORIGINAL || c.obs_layout == CN_LAYOUT_REALWORLD) ||
        !(c.geos_untyped_empty == 0 || c.geos_untyped_empty == 1) || !(c.ped_contact == 0 || c.ped_contact == 1) ||
        !(c.risk_mode == CN_RISK_LIDAR_TRACKER || c.risk_mode == CN_RISK_GT) || !(c.py2_round == 0 || c.py2_round == 1) ||
        c.ped_mode < 0 || c.ped_mode > 2 || !(c.scan_f32 == 0 || c.scan_f32 == 1))
        return fail(CN_ERR_CONFIG, "cn_create: config out of range");
    if (!(c.wheel_accel >= 0.0) || (c.wheel_accel > 0.0 && (c.ped_contact || c.ped_mode == 2 || c.obs_layout != CN_LAYOUT_RISK || !(c.wheel_separation > 0.0))))
        return fail(CN_ERR_CONFIG, "cn_create: wheel_accel must be >= 0 and needs obs_layout 0, the plain simulator (ped_contact 0, ped_mode 0 / 1) "
                                   "and a positive wheel_separation");
    if (c.ped_mode == 2 && (c.ped_contact || c.obs_layout != CN_LAYOUT_RISK || !(c.sf_tau > 0.0) || !(c.sf_B > 0.0) ||
                            !(c.sf_wall_B > 0.0) || !(c.sf_goal_eps >= 0.0) || c.sf_tick_ms < 0 || 64 * (size_t)c.n_peds > 16 * (size_t)(c.n_rays - 1)))
        return fail(CN_ERR_CONFIG, "cn_create: ped_mode 2 (social force) needs obs_layout 0, ped_contact 0, positive sf_tau / sf_B / "
                                   "sf_wall_B and n_peds <= (n_rays - 1) / 4");
    // the pedestrians' repulsion is summed on a 2^-36 grid (include/crowdnav.h): exact while the sum stays below 2^16
    if (c.ped_mode == 2 && !((double)c.n_peds * fabs(c.sf_A) * exp(2.0 * c.ped_radius / c.sf_B) < 32768.0))
        return fail(CN_ERR_CONFIG, "cn_create: ped_mode 2: n_peds * sf_A * exp(2 ped_radius / sf_B) must stay below 2^15 (exact force sums)");
    if (!(c.max_scan_range > c.min_scan_range))    // ENV:581, UTL:322 divide by their difference (ZeroDivisionError in the reference)
        return fail(CN_ERR_CONFIG, "cn_create: max_scan_range must exceed min_scan_range");
    if (c.obs_layout == CN_LAYOUT_REALWORLD && (c.n_rays - 1 > 65535 / 2))
        return fail(CN_ERR_CONFIG, "cn_create: config out of range");
    if (c.risk_mode == CN_RISK_GT && c.obs_layout != CN_LAYOUT_RISK)
        return fail(CN_ERR_CONFIG, "cn_create: risk_mode gt needs the risk observation layout (obs_layout 0)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(CN_ERR_NO_DEVICE, "cn_create: no HIP device (libcrowdnav has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(CN_ERR_ARG, "cn_create: bad device ordinal");
    DeviceScope scope(device);
    // every failure path below (HIPCHK returns) releases what was allocated so far
    std::unique_ptr<cn_env_s, HandleDeleter> guard(new cn_env_s());
    cn_env_s* h = guard.get();
    h->cfg = c;
    h->device = device;
    { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) h->n_cus = ncu; }
    const int N = c.n_envs, P = c.n_peds, R = c.n_rays, K = c.k_obstacles;
    h->D = c.obs_layout == CN_LAYOUT_ORIGINAL ? (R - 1) + 4 : (c.obs_layout == CN_LAYOUT_REALWORLD ? (R - 1) + 11 : (R - 1) + 7 + 4 * K);
    h->max_conf = (R - 1) / 4 + 2;
    // (risk_mode gt lists the simulator's own pedestrians in range: up to P entries, so 33-40 pedestrians take the larger table there --
    // tools/fuzz_parity.py found a 36-pedestrian gt world with more than 32 of them in range)
    h->trk_cap = c.track_capacity ? c.track_capacity : ((P <= 40 && !(c.risk_mode == CN_RISK_GT && P > 32)) ? 32 : 64);
    h->lds = lds_bytes_impl(R, P, K, h->max_conf, h->trk_cap, cn_near_separate(R, P, K, h->max_conf, h->trk_cap) != 0, c.obs_layout);
    if (c.ped_contact && c.obs_layout != CN_LAYOUT_RISK)
        return fail(CN_ERR_CONFIG, "cn_create: ped_contact is built for the risk observation layout (obs_layout 0) only");
    if (c.ped_contact && 32 * (size_t)P > 16 * (size_t)(R - 1))   // contact corrections: 4 doubles per pedestrian in LDS regions A + B
        return fail(CN_ERR_CONFIG, "cn_create: ped_contact needs n_peds <= (n_rays - 1) / 2");
    if (h->lds > 160 * 1024) { return fail(CN_ERR_CONFIG, "cn_create: per-env working set exceeds 160 KiB of LDS"); }
    // tables
    const int Wb = (R + 63) / 64;
    std::vector<double> lidar(4 * (size_t)R + 2 * (size_t)Wb), poly(128);
    double step = c.lidar_span / (double)(R - 1);
    for (int k = 0; k < R; ++k) cn_det_sincos((double)k * step, &lidar[R + k], &lidar[k]);
    for (int j = 0; j < R - 1; ++j) {  // UTL:121-123 math.radians(i * angle_increment)
        double a = ((double)j * angle_increment_deg(R)) * (M_PI / 180.0);
        lidar[2 * R + j] = sin(a); lidar[3 * R + j] = cos(a);
    }
    for (int q = 0; q < Wb; ++q) {   // axis of 64-ray block q: ray 64 q + 32 (robot frame)
        int kq = 64 * q + 32 < R ? 64 * q + 32 : R - 1;
        lidar[4 * (size_t)R + 2 * q] = lidar[kq]; lidar[4 * (size_t)R + 2 * q + 1] = lidar[R + kq];
    }
    for (int k = 0; k < 64; ++k) { double a = -(double)k * M_PI / 32.0; poly[k] = cos(a); poly[64 + k] = sin(a); }
    HIPCHK(hipMalloc(&h->d_lidar, lidar.size() * 8));
    HIPCHK(hipMalloc(&h->d_poly, poly.size() * 8 + 512));      // + the association table (256 shorts, below)
    HIPCHK(hipMemcpy(h->d_lidar, lidar.data(), lidar.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_poly, poly.data(), poly.size() * 8, hipMemcpyHostToDevice));
    size_t pb = (size_t)N * (P > 0 ? P : 1) * 16;
    h->stride = CN_ST_STRIDE(P);
    HIPCHK(hipMalloc(&h->d_state, (size_t)N * h->stride));
    HIPCHK(hipMalloc(&h->d_ped_init, pb));
    HIPCHK(hipMalloc(&h->d_ped_preset, pb));
    HIPCHK(hipMalloc(&h->d_trk, (size_t)N * CN_TF_COUNT * h->trk_cap * 8));
    HIPCHK(hipMalloc(&h->d_ped_aux, (size_t)N * (P > 0 ? P : 1) * 24));
    h->ped_init.resize((size_t)N * P * 2);
    for (int e = 0; e < N; ++e) default_ped_init(c, e, &h->ped_init[(size_t)e * P * 2]);
    int rc = upload_initial_state(h);
    if (rc != CN_OK) return rc;

    CnKParams& k = h->kp;
    memset(&k, 0, sizeof(k));
    k.N = N; k.P = P; k.R = R; k.K = K;
    k.max_steps = c.max_steps; k.ped_mode = c.ped_mode; k.dt_ms = c.dt_ms; k.scan_latency_ms = c.scan_latency_ms;
    k.settle_ms = c.settle_ms; k.ped_cycle_ms = c.ped_cycle_ms; k.ped_stagger_ms = c.ped_stagger_ms;
    k.geos_untyped_empty = c.geos_untyped_empty; k.ped_contact = c.ped_contact; k.risk_mode = c.risk_mode; k.py2_round = c.py2_round;
    k.lidar_min_positive = c.lidar_min > 0.0 ? 1 : 0;
    k.scan_f32 = c.scan_f32; k.waypoint_reward = c.waypoint_reward; k.wheel_accel = c.wheel_accel; k.wheel_sep = c.wheel_separation;
    k.sf_tau = c.sf_tau; k.sf_A = c.sf_A; k.sf_B = c.sf_B; k.sf_wall_A = c.sf_wall_A; k.sf_wall_B = c.sf_wall_B;
    k.sf_goal_eps2 = c.sf_goal_eps * c.sf_goal_eps; k.sf_tick_ms = c.sf_tick_ms > 0 ? c.sf_tick_ms : 10;
    // pair matrix G [P][P] + next state + goal records in the simulator's LDS scratch (regions A + B, 16 (R - 1) bytes)?
    {   // dense social force: the near-pair list behind next / aux [8 P] and acc [2 P] doubles of the scratch
        const size_t scr = lds_scratch_bytes(R, P, K, h->max_conf, h->trk_cap, cn_near_separate(R, P, K, h->max_conf, h->trk_cap) != 0);
        const size_t used = 80 * (size_t)P;
        size_t cap = scr > used ? (scr - used) / 2 : 0;
        k.sf_pair_cap = (c.ped_mode == 2 && P <= 128) ? (int32_t)(cap > 60000 ? 60000 : cap) : 0;
        k.sf_reserved = 0;
    }
    k.sf_pair_matrix = (c.ped_mode == 2 && P >= 2 && P < 256 && 8 * (size_t)P * P + 64 * (size_t)P + (size_t)P * (P - 1) + 16 <= 16 * (size_t)(R - 1)) ? 1 : 0;
    k.near_sep = cn_near_separate(R, P, K, h->max_conf, h->trk_cap);
    // the kernels compiled for the headline shape assume exactly these six values (crowdnav_kernel.hip, SHAPE == 360)
    h->shape360 = R == 360 && P == 20 && K == 8 && h->max_conf == 91 && h->trk_cap == 32 && k.near_sep == 1 && !getenv("CN_NO_SHAPE_KERNELS");
    h->shape720 = R == 720 && P == 100 && K == 8 && h->max_conf == 181 && h->trk_cap == 64 && k.near_sep == 0 && !getenv("CN_NO_SHAPE_KERNELS")
                  && c.room_half + c.lidar_max + 1.0 < 32.0;       // the _s720 kernels keep end points as int16 thousandths
    // Four environments per workgroup for the 360-ray step kernels when the whole launch is resident at once (n_envs <= 16 wavefronts
    // x CUs): a quarter of the workgroups for the dispatcher to create, + 2-3 % in every decomposition at 4096 envs; with several
    // rounds of wavefronts (16384 envs) the coarser release of LDS / wave slots costs 3.6 % instead, and the 720-ray shape (12
    // wavefronts per CU, 1.33 rounds at 4096) loses 7-25 % (profiles/r05/ab_wpb.txt).  Decided per launch in choose_kernel() from
    // max(n_envs, cn_set_group_envs).  CN_WPB=1 / 2 / 4 / 8 / 16 overrides for A/B runs.
    h->wpb4 = 4;
    if (getenv("CN_X2")) h->x2 = atoi(getenv("CN_X2")) ? 1 : 0;
    if (getenv("CN_WPB")) { h->wpb4 = atoi(getenv("CN_WPB")); if (h->wpb4 != 2 && h->wpb4 != 4 && h->wpb4 != 8 && h->wpb4 != 16) h->wpb4 = 0; }
    if (h->shape720) h->lds_shape = lds_bytes_impl(R, P, K, h->max_conf, h->trk_cap, false, c.obs_layout, true);
    k.max_conf = h->max_conf; k.trk_cap = h->trk_cap; k.env_index_base = c.env_index_base; k.seed = c.seed;
    k.room_half = c.room_half; k.ped_radius = c.ped_radius; k.ped_vmax = c.ped_vmax; k.robot_clearance = c.robot_clearance;
    k.lidar_min = c.lidar_min; k.lidar_max = c.lidar_max; k.lidar_offset_x = c.lidar_offset_x;
    k.ped_inv_cycle = 1.0 / (double)c.ped_cycle_ms;
    k.max_scan_range = c.max_scan_range; k.min_scan_range = c.min_scan_range; k.goal_x = c.goal_x; k.goal_y = c.goal_y;
    k.start_x = c.start_x; k.start_y = c.start_y; k.spawn_x = c.spawn_x; k.spawn_y = c.spawn_y; k.spawn_yaw = c.spawn_yaw;
    k.waypoint_radius = c.waypoint_radius; k.goal_eps = c.goal_eps; k.angle_inc_deg = angle_increment_deg(R); k.lidar_step = step;
    { static const double trig[CN_TRIG_COUNT] = CN_TRIG_TABLE; static_assert(sizeof(trig) == sizeof(k.trig), "CnKParams::trig"); memcpy(k.trig, trig, sizeof(trig)); }
    k.blk_dir = h->d_lidar + 4 * (size_t)R;
    {   // near_peds' block test  oc . u_q >= cos(beta) sqrt(d^2 - r^2) - sin(beta) r  is  theta <= beta + asin(r / d)  only while
        // beta + asin(r / d) <= pi (cos decreasing).  With few rays a block's half-width beta = 32.5 steps passes pi / 2 (R < 130) or
        // even pi (R < 66) and the test would clear bits of pedestrians a ray can hit (found by tools/fuzz_parity.py: 13 / 42 / 46
        // rays).  There the block bits are switched off: (cb, sb) = (-1, 1) makes the threshold -(sqrt(d^2 - r^2) + r) <= -d,
        // which every direction passes, so every listed pedestrian is tested in every block.
        const double beta = 32.5 * fabs(step);
        const bool cone_ok = beta < 1.5707;
        k.blk_cb = cone_ok ? cos(beta) : -1.0; k.blk_sb = cone_ok ? sin(beta) : 1.0;
    }
    k.lidar_c = h->d_lidar; k.lidar_s = h->d_lidar + R; k.ang_s = h->d_lidar + 2 * R; k.ang_c = h->d_lidar + 3 * R; k.poly_c = h->d_poly; k.poly_s = h->d_poly + 64;
    k.state = h->d_state; k.state_stride = (int64_t)h->stride; k.ped_init = h->d_ped_init;
    k.ped_preset = h->d_ped_preset; k.trk = h->d_trk; k.ped_aux = h->d_ped_aux;
    {   // the reset path's bounding-box size at the spawn pose (a constant of the configuration), from the device's own arithmetic
        double* d_out = nullptr;
        HIPCHK(hipMalloc(&d_out, sizeof(double)));
        hipLaunchKernelGGL(cn_bbox_kernel, dim3(1), dim3(64), 0, (hipStream_t)0, k, d_out);
        hipError_t e1 = hipGetLastError(), e2 = hipMemcpy(&k.bb_spawn, d_out, sizeof(double), hipMemcpyDeviceToHost);
        (void)hipFree(d_out);
        HIPCHK(e1); HIPCHK(e2);
        k.bb_spawn_valid = 1;
        int16_t tab[256] = {0};
        build_assoc_table(k, tab);
        k.assoc_tab = (const int16_t*)(h->d_poly + 128);
        HIPCHK(hipMemcpy((void*)k.assoc_tab, tab, sizeof(tab), hipMemcpyHostToDevice));
    }
    if (h->shape360 && h->wpb4) {     // the _w4 kernels are only ever launched for the headline shape (choose_kernel)
        const size_t w4 = (size_t)h->wpb4 * ((h->lds + 15) & ~(size_t)15);      // what launch() asks for
        if (w4 > 160 * 1024) h->wpb4 = 0;                                        // (a CN_WPB override that does not fit a CU's LDS: one per workgroup)
        else if (w4 > 64 * 1024) {
            HIPCHK(hipFuncSetAttribute((const void*)cn_env_kernel_s360_w4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w4));
            HIPCHK(hipFuncSetAttribute((const void*)cn_env_kernel_fair_s360_w4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w4));
        }
    }
    if (h->lds > 64 * 1024)
        for (const void* f : kDynamicLdsKernels) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds));
    {   // cn_rollout_policy: 16 (or 8) environments per workgroup, their working sets 16-byte aligned one after the other (the
        // actor's 16-row tile is laid out compactly over them between two steps), + the workgroup's actions
        // (whichever kernel choose_policy_kernel picks for this handle: the 720-ray shape's has the compact layout)
        h->pol_wave_lds = ((choose_policy_kernel(h).compact ? h->lds_shape : h->lds) + 15) & ~(size_t)15;
        const size_t tile = sizeof(float) * (16 * (size_t)(((h->D + 31) & ~31) + 1) + 16 * 257);
        const char* pe_env = getenv("CN_POL_ENVS");             // experiments: CN_POL_ENVS=8 forces the 8-environment workgroups
        // (cn_policy_kernel_rw is compiled for 8 waves: the RW observation needs more than the 128 vector registers a 16-wave workgroup leaves a wave)
        for (int pe = ((pe_env && atoi(pe_env) == 8) || h->cfg.obs_layout == CN_LAYOUT_REALWORLD) ? 8 : 16; pe >= 8 && !h->pol_lds; pe -= 8) {
            size_t off = (size_t)pe * h->pol_wave_lds;
            if (off < tile) off = (tile + 15) & ~(size_t)15;
            const size_t tot = off + (size_t)pe * 2 * sizeof(float);
            if (tot <= 160 * 1024) { h->pol_lds = tot; h->pol_envs = pe; h->pol_act_off = off; }
        }
        if (h->pol_lds > 64 * 1024)
            for (const void* f : kPolicyKernels) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->pol_lds));
    }
    *out = guard.release();
    return CN_OK;
}

extern "C" void cn_destroy(cn_handle h) { destroy_handle(h); }

#ifdef CN_TIMING
// PROFILING BUILD ONLY (libcrowdnav_timing.so; not in crowdnav.h, not in the product library): stage time stamps
// (tools/stage_timing.py) and the stage-skipping mask for time attribution (tools/ablate.py)
extern "C" int cn_debug_set_timing(cn_handle h, long long* dev_buf) { if (!h) return CN_ERR_ARG; h->kp.timing = dev_buf; return CN_OK; }
extern "C" int cn_debug_set_ablate(cn_handle h, int mask) { if (!h) return CN_ERR_ARG; h->kp.ablate = mask; return CN_OK; }
#endif

extern "C" int cn_obs_dim(cn_handle h) { return h ? h->D : fail(CN_ERR_ARG, "null handle"); }

extern "C" int cn_config_of(cn_handle h, cn_config* out)
{
    if (!h || !out) return fail(CN_ERR_ARG, "cn_config_of: null argument");
    *out = h->cfg;
    return CN_OK;
}

extern "C" int cn_set_ped_init(cn_handle h, const double* xy)
{
    if (!h || !xy) return fail(CN_ERR_ARG, "cn_set_ped_init: null argument");
    DeviceScope scope(h->device);
    size_t cnt = (size_t)h->cfg.n_envs * h->cfg.n_peds * 2;
    h->ped_init.assign(xy, xy + cnt);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(h->d_ped_init, xy, cnt * 8, hipMemcpyHostToDevice));
    FIELD(field_to_device(h, CN_ST_OFF_PED_P, xy, (size_t)h->cfg.n_peds * 16));
    return CN_OK;
}

extern "C" int cn_get_ped_init(cn_handle h, double* xy)
{
    if (!h || !xy) return fail(CN_ERR_ARG, "cn_get_ped_init: null argument");
    memcpy(xy, h->ped_init.data(), h->ped_init.size() * 8);
    return CN_OK;
}

extern "C" int cn_set_ped_preset_vel(cn_handle h, const double* vxy)
{
    if (!h || !vxy) return fail(CN_ERR_ARG, "cn_set_ped_preset_vel: null argument");
    DeviceScope scope(h->device);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(h->d_ped_preset, vxy, (size_t)h->cfg.n_envs * h->cfg.n_peds * 16, hipMemcpyHostToDevice));
    return CN_OK;
}

// Does a cn_step launch of this handle use the fair kernel?  (`overlapped`: the call is one of several in a cn_step_multi)
static bool fair_launch(const cn_env_s* h, bool overlapped)
{
    if (h->arbitration == CN_ARB_FAIR) return true;
    if (h->arbitration == CN_ARB_OLDEST_FIRST || overlapped) return false;
    return h->n_cus > 0 && h->cfg.n_envs >= 8 * h->n_cus;
}


// Which kernel a launch of this handle runs.  ext: externally supplied /scan + /odom; same: Env.step + same-call reset.
static KernelChoice choose_kernel(const cn_env_s* h, bool ext, bool same, bool overlapped)
{
    const cn_config& c = h->cfg;
    if (c.obs_layout == CN_LAYOUT_REALWORLD) return ext ? CN_KC(cn_env_kernel_rw_ext) : same ? CN_KC(cn_env_kernel_rw_same) : CN_KC(cn_env_kernel_rw);
    if (c.obs_layout == CN_LAYOUT_ORIGINAL) return ext ? CN_KC(cn_env_kernel_orig_ext) : same ? CN_KC(cn_env_kernel_orig_same) : CN_KC(cn_env_kernel_orig);
    if (ext) return c.risk_mode == CN_RISK_GT ? KernelChoice{nullptr, nullptr} : CN_KC(cn_env_kernel_ext);
    // simulated sensors: {lidar tracker, gt} x {plain, contact, social force, wheel ramp} x {one observation per launch, step + same-call reset}
    const bool gt = c.risk_mode == CN_RISK_GT, ct = c.ped_contact != 0, sf = c.ped_mode == 2, wa = c.wheel_accel > 0.0;
    if (wa) return gt ? (same ? CN_KC(cn_env_kernel_gt_wa_same) : CN_KC(cn_env_kernel_gt_wa)) : (same ? CN_KC(cn_env_kernel_wa_same) : CN_KC(cn_env_kernel_wa));
    const bool sfd = sf && !h->kp.sf_pair_matrix && c.n_peds <= 128;      // dense social-force crowd: per-lane near masks
    if (sfd) return gt ? (same ? CN_KC(cn_env_kernel_gt_sfd_same) : CN_KC(cn_env_kernel_gt_sfd)) : (same ? CN_KC(cn_env_kernel_sfd_same) : CN_KC(cn_env_kernel_sfd));
    if (gt) return sf ? (same ? CN_KC(cn_env_kernel_gt_sf_same) : CN_KC(cn_env_kernel_gt_sf))
                      : ct ? (same ? CN_KC(cn_env_kernel_gt_ct_same) : CN_KC(cn_env_kernel_gt_ct)) : (same ? CN_KC(cn_env_kernel_gt_same) : CN_KC(cn_env_kernel_gt));
    if (sf) return same ? CN_KC(cn_env_kernel_sf_same) : CN_KC(cn_env_kernel_sf);
    if (ct) return same ? CN_KC(cn_env_kernel_ct_same) : CN_KC(cn_env_kernel_ct);
    if (same) return CN_KC(cn_env_kernel_same);
    const bool fair = fair_launch(h, overlapped);
    const int64_t resident = h->group_envs > h->cfg.n_envs ? h->group_envs : (int64_t)h->cfg.n_envs;
    // small grids: two wavefronts per environment while all of them fit at two per SIMD (CN_X2=0 / 1 overrides for A/B runs)
    if (h->shape360 && h->n_cus > 0 && (h->x2 == 1 || (h->x2 < 0 && resident <= 8 * (int64_t)h->n_cus)))
        return KernelChoice{cn_env_kernel_s360_x2, "cn_env_kernel_s360_x2", false, 1, true};
    if (h->shape360 && h->wpb4 && h->n_cus > 0 && resident <= 16 * (int64_t)h->n_cus) return fair ? KernelChoice{cn_env_kernel_fair_s360_w4, "cn_env_kernel_fair_s360_w4", false, h->wpb4} : KernelChoice{cn_env_kernel_s360_w4, "cn_env_kernel_s360_w4", false, h->wpb4};
    if (h->shape360) return fair ? CN_KC(cn_env_kernel_fair_s360) : CN_KC(cn_env_kernel_s360);
    if (h->shape720) return fair ? CN_KCC(cn_env_kernel_fair_s720) : CN_KCC(cn_env_kernel_s720);
    return fair ? CN_KC(cn_env_kernel_fair) : CN_KC(cn_env_kernel);
}
// cn_step_sequence / cn_rollout_policy: every configuration cn_create accepts has both (round 6: the contact ticks and the two older
// observation layouts included) -- the same selection order as choose_kernel
static KernelChoice choose_sequence_kernel(const cn_env_s* h)
{
    const cn_config& c = h->cfg;
    if (c.obs_layout == CN_LAYOUT_REALWORLD) return CN_KC(cn_env_kernel_seq_rw);
    if (c.obs_layout == CN_LAYOUT_ORIGINAL) return CN_KC(cn_env_kernel_seq_orig);
    const bool gt = c.risk_mode == CN_RISK_GT, ct = c.ped_contact != 0, sf = c.ped_mode == 2, wa = c.wheel_accel > 0.0;
    const bool sfd = sf && !h->kp.sf_pair_matrix && c.n_peds <= 128;
    if (wa) return gt ? CN_KC(cn_env_kernel_gt_seq_wa) : CN_KC(cn_env_kernel_seq_wa);          // (choose_kernel: the wheel ramp comes first)
    if (sfd) return gt ? CN_KC(cn_env_kernel_gt_seq_sfd) : CN_KC(cn_env_kernel_seq_sfd);
    if (sf) return gt ? CN_KC(cn_env_kernel_gt_seq_sf) : CN_KC(cn_env_kernel_seq_sf);
    if (ct) return gt ? CN_KC(cn_env_kernel_gt_seq_ct) : CN_KC(cn_env_kernel_seq_ct);
    return gt ? CN_KC(cn_env_kernel_gt_seq)
         : h->shape360 ? CN_KC(cn_env_kernel_seq_s360) : h->shape720 ? CN_KCC(cn_env_kernel_seq_s720) : CN_KC(cn_env_kernel_seq);
}
static KernelChoice choose_policy_kernel(const cn_env_s* h)
{
    const cn_config& c = h->cfg;
    if (c.obs_layout == CN_LAYOUT_REALWORLD) return CN_KC(cn_policy_kernel_rw);
    if (c.obs_layout == CN_LAYOUT_ORIGINAL) return CN_KC(cn_policy_kernel_orig);
    const bool gt = c.risk_mode == CN_RISK_GT, ct = c.ped_contact != 0, sf = c.ped_mode == 2, wa = c.wheel_accel > 0.0;
    const bool sfd = sf && !h->kp.sf_pair_matrix && c.n_peds <= 128;
    if (wa) return gt ? CN_KC(cn_policy_kernel_gt_wa) : CN_KC(cn_policy_kernel_wa);
    if (sfd) return gt ? CN_KC(cn_policy_kernel_gt_sfd) : CN_KC(cn_policy_kernel_sfd);
    if (sf) return gt ? CN_KC(cn_policy_kernel_gt_sf) : CN_KC(cn_policy_kernel_sf);
    if (ct) return gt ? CN_KC(cn_policy_kernel_gt_ct) : CN_KC(cn_policy_kernel_ct);
    return gt ? CN_KC(cn_policy_kernel_gt)
         : h->shape360 ? CN_KC(cn_policy_kernel_s360) : h->shape720 ? CN_KCC(cn_policy_kernel_s720) : CN_KC(cn_policy_kernel);
}

extern "C" const char* cn_kernel_name(cn_handle h, int what)
{
    if (!h || what < 0 || what > 5) { fail(CN_ERR_ARG, "cn_kernel_name: bad argument"); return nullptr; }
    if (what == 5) { const char* n_ = choose_policy_kernel(h).name; if (!n_) fail(CN_ERR_CONFIG, "cn_kernel_name: cn_rollout_policy has no kernel for this configuration"); return n_; }
    if (what == 2) { const char* n_ = choose_sequence_kernel(h).name; if (!n_) fail(CN_ERR_CONFIG, "cn_kernel_name: cn_step_sequence has no kernel for this configuration"); return n_; }
    return choose_kernel(h, what == 3, what == 1, what == 4).name;
}

// Both counters are per-XCD (two one-thread kernels can land on different dies: their readings do not subtract), so the interval
// is taken INSIDE one wave: it reads both, idles on s_sleep until `span_ticks` of the 100 MHz counter have passed, reads both again.
__global__ void cn_clock_kernel(long long* out, long long span_ticks)
{
    const long long t0 = (long long)__builtin_amdgcn_s_memtime(), r0 = (long long)__builtin_amdgcn_s_memrealtime();
    long long r1 = r0;
    while (r1 - r0 < span_ticks) { __builtin_amdgcn_s_sleep(32); r1 = (long long)__builtin_amdgcn_s_memrealtime(); }
    out[0] = (long long)__builtin_amdgcn_s_memtime() - t0;
    out[1] = r1 - r0;
}
extern "C" int cn_device_clock(int64_t* out_dev, int span_us, int device, void* stream)
{
    if (!out_dev || span_us < 1 || span_us > 1000000) return fail(CN_ERR_ARG, "cn_device_clock: null argument or span outside 1 us .. 1 s");
    DeviceScope scope(device);
    hipLaunchKernelGGL(cn_clock_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)out_dev, (long long)span_us * 100);
    HIPCHK(hipGetLastError());
    return CN_OK;
}

static int launch(cn_handle h, const CnKParams& kp, hipStream_t st, bool overlapped = false)
{
    DeviceScope scope(h->device);
    const bool ext = kp.mode == CN_MODE_EXT_STEP || kp.mode == CN_MODE_EXT_RESET;
    const KernelChoice kc = choose_kernel(h, ext, kp.mode == CN_MODE_STEP && kp.auto_reset == 1, overlapped);
    if (!kc.fn)
        return fail(CN_ERR_CONFIG, "cn_observe_external: risk_mode gt reads the library simulator's pedestrians; "
                                   "external /scan + /odom only exist in lidar_tracker mode");
    if (kc.x2) {
        CnKParams k2 = kp;
        k2.wave_lds = (int32_t)((lds_of(h, kc) + 15) & ~(size_t)15);
        hipLaunchKernelGGL(kc.fn, dim3(kp.N), dim3(128), (size_t)k2.wave_lds + 256, st, k2);
    } else if (kc.wpb > 1) {
        CnKParams k4 = kp;
        k4.wave_lds = (int32_t)((lds_of(h, kc) + 15) & ~(size_t)15);
        hipLaunchKernelGGL(kc.fn, dim3((kp.N + kc.wpb - 1) / kc.wpb), dim3(64 * kc.wpb), (size_t)k4.wave_lds * kc.wpb, st, k4);
    } else
    hipLaunchKernelGGL(kc.fn, dim3(kp.N), dim3(64), lds_of(h, kc), st, kp);
    HIPCHK(hipGetLastError());
    return CN_OK;
}

extern "C" int cn_reset(cn_handle h, const uint8_t* mask, float* obs, double* obs_f64, void* stream)
{
    if (!h || !obs) return fail(CN_ERR_ARG, "cn_reset: null argument");
    CnKParams kp = h->kp;
    kp.mode = CN_MODE_RESET; kp.mask = mask; kp.obs = obs; kp.obs_f64 = obs_f64;
    return launch(h, kp, (hipStream_t)stream);
}

static int step_one(cn_handle h, const cn_step_io* io, void* stream, bool overlapped)
{
    if (!h || !io || !io->action || !io->obs || !io->reward || !io->done) return fail(CN_ERR_ARG, "cn_step: null argument");
    CnKParams kp = h->kp;
    kp.mode = CN_MODE_STEP; kp.auto_reset = io->auto_reset;
    kp.action = io->action; kp.step_counter = io->step_counter; kp.obs = io->obs; kp.final_obs = io->final_obs;
    kp.obs_f64 = io->obs_f64; kp.reward = io->reward; kp.done = io->done; kp.topk_idx = io->topk_idx;
    return launch(h, kp, (hipStream_t)stream, overlapped);
}

extern "C" int cn_step(cn_handle h, const cn_step_io* io, void* stream) { return step_one(h, io, stream, false); }

extern "C" int cn_set_arbitration(cn_handle h, int mode)
{
    if (!h) return fail(CN_ERR_ARG, "cn_set_arbitration: null handle");
    if (mode != CN_ARB_AUTO && mode != CN_ARB_OLDEST_FIRST && mode != CN_ARB_FAIR)
        return fail(CN_ERR_ARG, "cn_set_arbitration: mode must be CN_ARB_AUTO (0), CN_ARB_OLDEST_FIRST (1) or CN_ARB_FAIR (2)");
    h->arbitration = mode;
    return CN_OK;
}

extern "C" int cn_set_group_envs(cn_handle h, int64_t total_envs)
{
    if (!h || total_envs < 0) return fail(CN_ERR_ARG, "cn_set_group_envs: null handle or negative count");
    h->group_envs = total_envs;
    return CN_OK;
}

extern "C" int cn_get_arbitration(cn_handle h)
{
    if (!h) return fail(CN_ERR_ARG, "cn_get_arbitration: null handle");
    const bool has_variant = h->cfg.obs_layout == CN_LAYOUT_RISK && h->cfg.risk_mode != CN_RISK_GT && !h->cfg.ped_contact && h->cfg.ped_mode != 2 && !(h->cfg.wheel_accel > 0.0);
    return has_variant && fair_launch(h, false) ? CN_ARB_FAIR : CN_ARB_OLDEST_FIRST;
}

extern "C" int cn_step_multi(int n, const cn_handle* handles, const cn_step_io* ios, void* const* streams)
{
    if (n < 0 || (n > 0 && (!handles || !ios || !streams))) return fail(CN_ERR_ARG, "cn_step_multi: null argument");
    bool overlapped = false;        // more than one handle in the call: their launches are meant to overlap (CN_ARB_AUTO -> oldest-first)
    for (int i = 1; i < n; ++i) overlapped = overlapped || handles[i] != handles[0];
    for (int i = 0; i < n; ++i) {
        const int rc = step_one(handles[i], &ios[i], streams[i], overlapped);
        if (rc != CN_OK) return rc;
    }
    return CN_OK;
}

extern "C" int cn_observe_external(cn_handle h, const cn_external_io* io, void* stream)
{
    if (!h || !io || !io->ranges || !io->odom || !io->obs || !io->reward || !io->done)
        return fail(CN_ERR_ARG, "cn_observe_external: null argument");
    CnKParams kp = h->kp;
    if (io->phase < 0 || io->phase > 7 || (io->phase && io->is_reset))
        return fail(CN_ERR_ARG, "cn_observe_external: phase is a mask of CN_PHASE_* and only applies to the step flow");
    if ((io->phase & CN_PHASE_REWARD) && !(io->phase & CN_PHASE_GET_STATE) && !io->obs_f64)
        return fail(CN_ERR_ARG, "cn_observe_external: CN_PHASE_REWARD alone reads the float64 state (obs_f64)");
    kp.mode = io->is_reset ? CN_MODE_EXT_RESET : CN_MODE_EXT_STEP;
    kp.ext_phase = io->phase;
    kp.ext_ranges = io->ranges; kp.ext_odom = io->odom; kp.step_counter = io->step_counter;
    kp.obs = io->obs; kp.obs_f64 = io->obs_f64; kp.reward = io->reward; kp.done = io->done; kp.topk_idx = io->topk_idx;
    return launch(h, kp, (hipStream_t)stream);
}

extern "C" __global__ void cn_policy_tail_kernel(const float* logits, float* action, int n, float max_v, float max_w,
                                                 float sigma, uint64_t seed, uint64_t counter);

extern "C" int cn_policy_tail(const float* logits, float* action, int n, float max_v, float max_w, float sigma,
                              uint64_t seed, uint64_t counter, int device, void* stream)
{
    if (!logits || !action || n < 0) return fail(CN_ERR_ARG, "cn_policy_tail: bad argument");
    if (n == 0) return CN_OK;
    int dev = device;
    if (dev < 0) HIPCHK(hipGetDevice(&dev));
    DeviceScope scope(dev);
    hipLaunchKernelGGL(cn_policy_tail_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, action, n,
                       max_v, max_w, sigma, seed, counter);
    HIPCHK(hipGetLastError());
    return CN_OK;
}

extern "C" __global__ void cn_actor_kernel(const float* obs, int n, int D, int Dp, const float* W1T, const float* b1,
                                           const float* W2T, const float* b2, const float* W3, const float* b3, float* action,
                                           float max_v, float max_w, float sigma, uint64_t seed, uint64_t counter);

extern "C" __global__ void cn_actor_pack_kernel(const float* wt, int K, float* packed);
extern "C" int cn_actor_pack_weights(const float* wt, int k_rows, float* packed, int device, void* stream)
{
    if (!wt || !packed || wt == packed) return fail(CN_ERR_ARG, "cn_actor_pack_weights: null or aliasing argument");
    if (k_rows < 32 || (k_rows & 31)) return fail(CN_ERR_CONFIG, "cn_actor_pack_weights: k_rows must be a positive multiple of 32");
    int dev = device;
    if (dev < 0) HIPCHK(hipGetDevice(&dev));
    DeviceScope scope(dev);
    const int total = k_rows * 256;
    hipLaunchKernelGGL(cn_actor_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, wt, k_rows, packed);
    HIPCHK(hipGetLastError());
    return CN_OK;
}

extern "C" int cn_actor_forward(const cn_actor_weights* w, const float* obs, float* action, int n, float max_v, float max_w,
                                float sigma, uint64_t seed, uint64_t counter, int device, void* stream)
{
    if (!w || !obs || !action || n < 0 || !w->w1p || !w->b1 || !w->w2p || !w->b2 || !w->w3 || !w->b3)
        return fail(CN_ERR_ARG, "cn_actor_forward: null argument");
    if (w->hidden != 256 || w->obs_dim < 1 || w->obs_dim_padded < w->obs_dim || (w->obs_dim_padded & 31))
        return fail(CN_ERR_CONFIG, "cn_actor_forward: hidden must be 256 and obs_dim_padded a multiple of 32 (the packed layout of cn_actor_pack_weights)");
    if (n == 0) return CN_OK;
    const int Dp = w->obs_dim_padded;
    const size_t lds = sizeof(float) * (16 * (size_t)(Dp + 1) + 16 * 257);         // X (layer 2 reuses it), H
    if (lds > 160 * 1024) return fail(CN_ERR_CONFIG, "cn_actor_forward: observation too wide for one LDS tile");
    int dev = device;
    if (dev < 0) HIPCHK(hipGetDevice(&dev));
    DeviceScope scope(dev);
    if (lds > 64 * 1024) {   // hipFuncSetAttribute is per device: once per ordinal, under a lock
        static std::mutex mu;
        static bool attr_set[64] = {false};
        std::lock_guard<std::mutex> lk(mu);
        if (dev >= 64 || !attr_set[dev]) {
            HIPCHK(hipFuncSetAttribute((const void*)cn_actor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            if (dev < 64) attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL(cn_actor_kernel, dim3((n + 15) / 16), dim3(512), lds, (hipStream_t)stream, obs, n, w->obs_dim, Dp,
                       w->w1p, w->b1, w->w2p, w->b2, w->w3, w->b3, action, max_v, max_w, sigma, seed, counter);
    HIPCHK(hipGetLastError());
    return CN_OK;
}

extern "C" int cn_step_sequence(cn_handle h, const cn_sequence_io* io, void* stream)
{
    if (!h || !io || !io->action || !io->obs || !io->reward || !io->done) return fail(CN_ERR_ARG, "cn_step_sequence: null argument");
    if (io->n_steps < 0 || io->action_stride < 0 || io->obs_stride < 0 || io->reward_stride < 0 || io->done_stride < 0 || io->topk_stride < 0)
        return fail(CN_ERR_ARG, "cn_step_sequence: negative step count or stride");
    if (io->n_steps == 0) return CN_OK;
    CnKParams kp = h->kp;
    kp.mode = CN_MODE_STEP; kp.auto_reset = 2;
    kp.action = io->action; kp.step_counter = nullptr; kp.final_obs = nullptr; kp.obs_f64 = nullptr;
    kp.obs = io->obs; kp.reward = io->reward; kp.done = io->done; kp.topk_idx = io->topk_idx;
    kp.roll_steps = io->n_steps; kp.roll_action_in_stride = io->action_stride; kp.roll_obs_stride = io->obs_stride;
    kp.roll_reward_stride = io->reward_stride; kp.roll_done_stride = io->done_stride; kp.roll_topk_stride = io->topk_stride;
    DeviceScope scope(h->device);
    const KernelChoice kc = choose_sequence_kernel(h);
    hipLaunchKernelGGL(kc.fn, dim3(h->cfg.n_envs), dim3(64), lds_of(h, kc), (hipStream_t)stream, kp);
    HIPCHK(hipGetLastError());
    return CN_OK;
}

extern "C" int cn_rollout_policy(cn_handle h, const cn_actor_weights* w, const cn_policy_io* io, void* stream)
{
    if (!h || !w || !io || !io->obs0 || !io->action || !io->obs || !io->reward || !io->done || !w->w1p || !w->b1 || !w->w2p || !w->b2 || !w->w3 || !w->b3)
        return fail(CN_ERR_ARG, "cn_rollout_policy: null argument");
    if (!h->pol_lds)
        return fail(CN_ERR_CONFIG, "cn_rollout_policy: not even 8 environments of this shape fit one CU's LDS (160 KiB)");
    const int D = h->D;      // 366 + 4 K, or 363 / 370 at 360 rays for the two older layouts
    if (w->hidden != 256 || w->obs_dim != D || w->obs_dim_padded != ((D + 31) & ~31))
        return fail(CN_ERR_CONFIG, "cn_rollout_policy: the actor must be cn_actor_pack_weights' layout for this handle's observation width (hidden 256, obs_dim_padded = obs_dim rounded up to 32)");
    if (io->n_steps < 0 || io->action_stride < 0 || io->obs_stride < 0 || io->reward_stride < 0 || io->done_stride < 0 || io->topk_stride < 0)
        return fail(CN_ERR_ARG, "cn_rollout_policy: negative step count or stride");
    if (io->n_steps == 0) return CN_OK;
    CnKParams kp = h->kp;
    kp.mode = CN_MODE_STEP; kp.auto_reset = 2;
    kp.action = io->action; kp.step_counter = nullptr; kp.final_obs = nullptr; kp.obs_f64 = nullptr;
    kp.obs = io->obs; kp.reward = io->reward; kp.done = io->done; kp.topk_idx = io->topk_idx;
    kp.roll_steps = io->n_steps; kp.roll_action_in_stride = io->action_stride; kp.roll_obs_stride = io->obs_stride;
    kp.roll_reward_stride = io->reward_stride; kp.roll_done_stride = io->done_stride; kp.roll_topk_stride = io->topk_stride;
    kp.pol_w1p = w->w1p; kp.pol_b1 = w->b1; kp.pol_w2p = w->w2p; kp.pol_b2 = w->b2; kp.pol_w3 = w->w3; kp.pol_b3 = w->b3;
    kp.pol_obs0 = io->obs0; kp.pol_seed = io->seed; kp.pol_counter = io->counter;
    kp.pol_max_v = io->max_v; kp.pol_max_w = io->max_w; kp.pol_sigma = io->sigma;
    kp.pol_D = D; kp.pol_Dp = w->obs_dim_padded; kp.pol_wave_lds = (int32_t)h->pol_wave_lds;
    kp.pol_envs = h->pol_envs; kp.pol_act_off = (int32_t)h->pol_act_off;
    DeviceScope scope(h->device);
    cn_kernel_fn fn = choose_policy_kernel(h).fn;
    hipLaunchKernelGGL(fn, dim3((h->cfg.n_envs + h->pol_envs - 1) / h->pol_envs), dim3(64 * h->pol_envs), h->pol_lds, (hipStream_t)stream, kp);
    HIPCHK(hipGetLastError());
    return CN_OK;
}

extern "C" int cn_get_counters(cn_handle h, int32_t* out, void* stream)
{
    if (!h || !out) return fail(CN_ERR_ARG, "cn_get_counters: null argument");
    int N = h->cfg.n_envs;
    DeviceScope scope(h->device);
    hipLaunchKernelGGL(cn_gather_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->kp,
                       (float*)nullptr, (float*)nullptr, out);
    HIPCHK(hipGetLastError());
    return CN_OK;
}

extern "C" int cn_get_returns(cn_handle h, float* last_return, float* running_return, void* stream)
{
    if (!h) return fail(CN_ERR_ARG, "cn_get_returns: null handle");
    int N = h->cfg.n_envs;
    DeviceScope scope(h->device);
    hipLaunchKernelGGL(cn_gather_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->kp, last_return,
                       running_return, (int32_t*)nullptr);
    HIPCHK(hipGetLastError());
    return CN_OK;
}

extern "C" int cn_debug_env(cn_handle h, int env, double* scalars, double* robot_ped, double* tracks, int32_t* ints)
{
    if (!h || env < 0 || env >= h->cfg.n_envs) return fail(CN_ERR_ARG, "cn_debug_env: bad argument");
    DeviceScope scope(h->device);
    HIPCHK(hipDeviceSynchronize());
    const int P = h->cfg.n_peds;
    std::vector<double> sd(CN_SD_COUNT);
    const char* rec = h->d_state + (size_t)env * h->stride;
    HIPCHK(hipMemcpy(sd.data(), rec + CN_ST_OFF_SD, CN_SD_COUNT * 8, hipMemcpyDeviceToHost));
    if (scalars) memcpy(scalars, sd.data(), CN_SD_COUNT * 8);
    if (robot_ped) {
        memcpy(robot_ped, sd.data(), 5 * 8);
        if (P > 0) {
            HIPCHK(hipMemcpy(robot_ped + 5, rec + CN_ST_OFF_PED_P, (size_t)P * 32, hipMemcpyDeviceToHost));   // ped_p | ped_v
        }
    }
    if (tracks)
        HIPCHK(hipMemcpy(tracks, h->d_trk + (size_t)env * CN_TF_COUNT * h->trk_cap, CN_TF_COUNT * h->trk_cap * 8,
                         hipMemcpyDeviceToHost));
    if (ints) HIPCHK(hipMemcpy(ints, rec + CN_ST_OFF_SI, CN_SI_COUNT * 4, hipMemcpyDeviceToHost));
    return CN_OK;
}

// snapshot layout (include/crowdnav.h): cn_snapshot_header | sd | si | ped_p | ped_v | trk | ped_init | ped_preset | ped_aux
static size_t snapshot_payload(cn_handle h)
{
    size_t N = h->cfg.n_envs, P = h->cfg.n_peds;
    return N * (CN_SD_COUNT * 8 + CN_SI_COUNT * 4 + 2 * P * 16 + (size_t)CN_TF_COUNT * h->trk_cap * 8 + 2 * P * 16 + P * 24);
}
extern "C" size_t cn_snapshot_size(cn_handle h)
{
    if (!h) return 0;
    return sizeof(cn_snapshot_header) + snapshot_payload(h);
}

static void fill_header(cn_handle h, cn_snapshot_header* hd)
{
    memset(hd, 0, sizeof(*hd));
    hd->magic = CN_SNAPSHOT_MAGIC; hd->abi_version = CN_ABI_VERSION; hd->header_bytes = (int32_t)sizeof(cn_snapshot_header);
    hd->sd_count = CN_SD_COUNT; hd->si_count = CN_SI_COUNT; hd->tf_count = CN_TF_COUNT; hd->track_capacity = h->trk_cap;
    hd->total_bytes = sizeof(cn_snapshot_header) + snapshot_payload(h);
    hd->config = h->cfg;
}

extern "C" int cn_snapshot(cn_handle h, void* buf, size_t size)
{
    if (!h || !buf) return fail(CN_ERR_ARG, "cn_snapshot: null argument");
    if (size < cn_snapshot_size(h)) return fail(CN_ERR_SIZE, "cn_snapshot: buffer too small");
    DeviceScope scope(h->device);
    HIPCHK(hipDeviceSynchronize());
    size_t N = h->cfg.n_envs, P = h->cfg.n_peds;
    char* q = (char*)buf;
    cn_snapshot_header hd;
    fill_header(h, &hd);
    memcpy(q, &hd, sizeof(hd)); q += sizeof(hd);
    FIELD(field_to_host(h, CN_ST_OFF_SD, q, CN_SD_COUNT * 8)); q += N * CN_SD_COUNT * 8;
    FIELD(field_to_host(h, CN_ST_OFF_SI, q, CN_SI_COUNT * 4)); q += N * CN_SI_COUNT * 4;
    if (P) {
        FIELD(field_to_host(h, CN_ST_OFF_PED_P, q, P * 16)); q += N * P * 16;
        FIELD(field_to_host(h, CN_ST_OFF_PED_V(P), q, P * 16)); q += N * P * 16;
    }
    HIPCHK(hipMemcpy(q, h->d_trk, N * CN_TF_COUNT * h->trk_cap * 8, hipMemcpyDeviceToHost)); q += N * CN_TF_COUNT * h->trk_cap * 8;
    if (P) {
        HIPCHK(hipMemcpy(q, h->d_ped_init, N * P * 16, hipMemcpyDeviceToHost)); q += N * P * 16;
        HIPCHK(hipMemcpy(q, h->d_ped_preset, N * P * 16, hipMemcpyDeviceToHost)); q += N * P * 16;
        HIPCHK(hipMemcpy(q, h->d_ped_aux, N * P * 24, hipMemcpyDeviceToHost)); q += N * P * 24;
    }
    return CN_OK;
}

// first field of the two configurations that differs (nullptr: identical)
static const char* config_mismatch(const cn_config& a, const cn_config& b)
{
#define CN_CMP(f) if (memcmp(&a.f, &b.f, sizeof(a.f)) != 0) return #f;
    CN_CMP(n_envs) CN_CMP(n_peds) CN_CMP(n_rays) CN_CMP(k_obstacles) CN_CMP(max_steps) CN_CMP(ped_mode) CN_CMP(dt_ms)
    CN_CMP(scan_latency_ms) CN_CMP(settle_ms) CN_CMP(ped_cycle_ms) CN_CMP(ped_stagger_ms) CN_CMP(track_capacity) CN_CMP(obs_layout)
    CN_CMP(geos_untyped_empty) CN_CMP(ped_contact) CN_CMP(risk_mode) CN_CMP(py2_round) CN_CMP(sf_tick_ms) CN_CMP(scan_f32) CN_CMP(waypoint_reward)
    CN_CMP(env_index_base) CN_CMP(seed)
    CN_CMP(room_half) CN_CMP(ped_radius) CN_CMP(ped_vmax) CN_CMP(robot_clearance) CN_CMP(lidar_min) CN_CMP(lidar_max) CN_CMP(lidar_span)
    CN_CMP(lidar_offset_x) CN_CMP(max_scan_range) CN_CMP(min_scan_range) CN_CMP(goal_x) CN_CMP(goal_y) CN_CMP(start_x) CN_CMP(start_y)
    CN_CMP(spawn_x) CN_CMP(spawn_y) CN_CMP(spawn_yaw) CN_CMP(waypoint_radius) CN_CMP(goal_eps)
    CN_CMP(sf_tau) CN_CMP(sf_A) CN_CMP(sf_B) CN_CMP(sf_wall_A) CN_CMP(sf_wall_B) CN_CMP(sf_goal_eps) CN_CMP(wheel_accel) CN_CMP(wheel_separation)
#undef CN_CMP
    return nullptr;
}

extern "C" int cn_restore(cn_handle h, const void* buf, size_t size)
{
    if (!h || !buf) return fail(CN_ERR_ARG, "cn_restore: null argument");
    if (size < sizeof(cn_snapshot_header)) return fail(CN_ERR_SIZE, "cn_restore: buffer smaller than a snapshot header");
    cn_snapshot_header hd;
    memcpy(&hd, buf, sizeof(hd));
    if (hd.magic != CN_SNAPSHOT_MAGIC) return fail(CN_ERR_ARG, "cn_restore: not a libcrowdnav snapshot (bad magic)");
    if (hd.abi_version != CN_ABI_VERSION || hd.header_bytes != (int32_t)sizeof(cn_snapshot_header))
        return fail(CN_ERR_CONFIG, "cn_restore: snapshot written by ABI version " + std::to_string(hd.abi_version) +
                                   ", this library is ABI " + std::to_string(CN_ABI_VERSION));
    if (hd.sd_count != CN_SD_COUNT || hd.si_count != CN_SI_COUNT || hd.tf_count != CN_TF_COUNT || hd.track_capacity != h->trk_cap)
        return fail(CN_ERR_CONFIG, "cn_restore: snapshot record layout (sd / si / track fields, tracker slots) differs from this handle's");
    if (const char* f = config_mismatch(hd.config, h->cfg))
        return fail(CN_ERR_CONFIG, std::string("cn_restore: the snapshot was taken under another configuration: cn_config.") + f +
                                   " differs (a state only means something under the configuration that produced it)");
    if (hd.total_bytes != cn_snapshot_size(h) || size < hd.total_bytes) return fail(CN_ERR_SIZE, "cn_restore: truncated snapshot");
    DeviceScope scope(h->device);
    HIPCHK(hipDeviceSynchronize());
    size_t N = h->cfg.n_envs, P = h->cfg.n_peds;
    const char* q = (const char*)buf + sizeof(hd);
    FIELD(field_to_device(h, CN_ST_OFF_SD, q, CN_SD_COUNT * 8)); q += N * CN_SD_COUNT * 8;
    FIELD(field_to_device(h, CN_ST_OFF_SI, q, CN_SI_COUNT * 4)); q += N * CN_SI_COUNT * 4;
    if (P) {
        FIELD(field_to_device(h, CN_ST_OFF_PED_P, q, P * 16)); q += N * P * 16;
        FIELD(field_to_device(h, CN_ST_OFF_PED_V(P), q, P * 16)); q += N * P * 16;
    }
    HIPCHK(hipMemcpy(h->d_trk, q, N * CN_TF_COUNT * h->trk_cap * 8, hipMemcpyHostToDevice)); q += N * CN_TF_COUNT * h->trk_cap * 8;
    if (P) {
        h->ped_init.assign((const double*)q, (const double*)q + N * P * 2);
        HIPCHK(hipMemcpy(h->d_ped_init, q, N * P * 16, hipMemcpyHostToDevice)); q += N * P * 16;
        HIPCHK(hipMemcpy(h->d_ped_preset, q, N * P * 16, hipMemcpyHostToDevice)); q += N * P * 16;
        HIPCHK(hipMemcpy(h->d_ped_aux, q, N * P * 24, hipMemcpyHostToDevice)); q += N * P * 24;
    }
    return CN_OK;
}
