// crowdnav_td3.hip -- the TD3 update (td3.py:225-285 of the reference: Agent.learn) as a short chain of HIP kernels (gfx950).
//
// The caller of the hot path.  A vectorised environment makes the learner the bottleneck: through PyTorch one
// update is ~150 small kernels (1.2 ms as a hipGraph at batch 128).  Here the same arithmetic is 7 launches for the critic
// step and 5 more when the actor and the targets move (rounds 3-4: 10 + 11), all float32 like the reference:
//   prep        sample the replay on the device (counter-based indices and target-policy noise), gather [s|a], [s2|.], r, d
//   gemm F      Y = act(X W^T + b) on the f32 matrix cores (v_mfma_f32_16x16x4_f32), up to four networks per launch;
//               optional: the policy's last layer + heads evaluated in place of X's action columns, the critic's last layer
//               as per-tile partial sums of the activation just written, the first link of the actor-loss chain
//   gemm G      dX = (dY W) (.) [H > 0]   (back-propagation through a ReLU layer); optional: dY evaluated, not read -- the TD
//               target + MSE gradient + linear3 backward of a critic, or the heads' derivatives + linear3 backward of the actor;
//               the action gradient's partial sums
//   gemm H      dW = dY^T X folded into the Adam step of W (and of b) and the soft update of the target's copy: the gradient
//               never exists in memory; linear3's gradients are one- and two-row jobs of the same launch
// Launch order: prep | actor_t L1 (+ actor L1) | L2 (+ L2) | critics L1 (target actions from the head) | critics L2 (+ q partials,
// tick) | G (dq, dz2 evaluated) | H (six jobs)   and, every policy_delay-th update:  q1 L1 on (s, pi(s)) | q1 L2 (+ dz) | G (+ da
// partials) | G (dl, dz2a evaluated) | H (three jobs).
// The parameters are the caller's (PyTorch nn.Linear storages, weight [out][in]); Adam's moments and step counters live here.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <new>
#include <string>

#include "../../include/crowdnav.h"
#include "crowdnav_device.h"

typedef float f32x4_t __attribute__((ext_vector_type(4)));

namespace {

thread_local std::string g_td3_err;
int td3_fail(int code, const std::string& msg) { g_td3_err = msg; return code; }
#define TD3CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return td3_fail(CN_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// ---- the three GEMM kernels ---------------------------------------------------------------------------------------------
// C[i][j] = sum_r A(i, r) B(r, j) on v_mfma_f32_16x16x4_f32.  These layers are tiny (batch 128 x 256 units x 400 inputs =
// 26 MFLOP) and latency-bound, so the shape is: many small workgroups (a 16 x 16 / 16 x 32 / 32 x 32 tile of C each), the
// reduction split over the workgroup's four wavefronts, every operand loaded from global memory STRAIGHT into the MFMA operand
// registers as 8 / 16-byte vectors along the direction that is contiguous in memory -- no LDS staging, no barrier before the
// single one of the cross-wavefront sum.  What makes that possible: the order in which a reduction's terms are fed to the
// matrix core is free as long as both operands use the same order (F), and an operand whose contiguous direction is the OUTPUT
// index can feed several accumulators from one vector (G: two, H: two x two).  (Rounds 3-4 staged 32 x 32 tiles through LDS in
// 128-deep chunks, 32 workgroups a network: 9-13 us a launch, now 4-6.)
//   F  forward          i = row m, j = unit n, r = input k :  A = X[m][k],  B = W[n][k],   C = act(acc + bias[n])
//   G  backward (data)  i = row m, j = input k, r = unit n :  A = dY[m][n], B = W[n][k],   C = acc * [mask[m][k] > 0]
//   H  backward (weights) + Adam   i = unit n, j = input k, r = row m :  A = dY[m][n], B = X[m][k],  W[n][k] <- Adam(acc);
//      the workgroups of the first j-tile also reduce dY over the rows and step the bias
enum { GEMM_F = 0, GEMM_G = 1, GEMM_H = 2 };
// one thread: advance the update counter and the Adam step counters, publish this update's bias corrections.  Runs inside a forward
// launch (a kernel of its own cost a full launch, ~4.6 us, for six scalar operations): after td3_prep_kernel, which reads the
// counter, and before the first kernel that reads the corrections (td3_critic_head_bwd_kernel).
struct TickArgs {
    float* adam;                                   // [2 optimizers][2]: lr / (1 - beta1^t), sqrt(1 - beta2^t)
    float* steps;                                  // [2] step counters (critics, actor)
    double* pw;                                    // [2 optimizers][2]: beta1^t, beta2^t as running products (powf was most of this thread's time)
    unsigned long long* counter;                   // update counter (keys the sampling)
    int do_actor; float lr_critic, lr_actor, beta1, beta2;
};
__device__ __forceinline__ void td3_tick(const TickArgs& p)
{
    *p.counter += 1ull;
    p.steps[0] += 1.f;
    const double c1 = p.pw[0] * (double)p.beta1, c2 = p.pw[1] * (double)p.beta2;
    p.pw[0] = c1; p.pw[1] = c2;
    p.adam[0] = p.lr_critic / (float)(1.0 - c1);
    p.adam[1] = sqrtf((float)(1.0 - c2));
    if (p.do_actor) {
        p.steps[1] += 1.f;
        const double a1 = p.pw[2] * (double)p.beta1, a2 = p.pw[3] * (double)p.beta2;
        p.pw[2] = a1; p.pw[3] = a2;
        p.adam[2] = p.lr_actor / (float)(1.0 - a1);
        p.adam[3] = sqrtf((float)(1.0 - a2));
    }
}
struct GemmJob {
    const float* A; const float* B; float* C;      // H: C = the weight being stepped
    const float* bias;                             // F: bias[n]
    const float* mask;                             // G: activation the ReLU mask is taken from (same shape / ld as C)
    float* m; float* v;                            // H: Adam moments of the weight
    float* bparam; float* bm; float* bv;           // H: bias and its moments
    const float* adam;                             // H: {lr / (1 - beta1^t), sqrt(1 - beta2^t)} of this optimizer (device)
    int I, J, R;                                   // extents of i, j, r
    int lda, ldb, ldc;
    int relu;
    // F, optional: the first link of the actor-loss chain written next to the activation it is masked by (TD3:268-269,
    // -mean Q1(s, pi(s))): dz_out[m][n] = -(1 / dz_rows) dz_w3[n] [y > 0]  (was a kernel of its own: one more launch)
    const float* dz_w3; float* dz_out; float dz_rows;
    // H, optional (actor updates): the target network's copy of the weight / bias, soft-updated from the value just stepped
    // (TD3:287-299; was an 18-tensor launch of its own at the end of the update)
    float* tgt; float* btgt;
    // F, optional: the LAST TWO columns of A are not read but evaluated here -- they are a policy's action on the row,
    // Actor.forward's last layer and heads (TD3:101-105) on the policy's second hidden activation hd_h2 [I][hd_H]:
    //   logits = hd_h2 hd_W3^T + hd_b3,  action = (sigmoid max_v, tanh max_w) (+ hd_noise: the clipped target-policy noise, not
    //   re-clipped to the action bounds, TD3:244-247);  R counts the columns before them, B's row has R + 2.
    // Their share of the product is rank 2 and is added after the cross-wavefront sum.  (Was a launch of its own that wrote the
    // actions into A.)  hd_logits [I][2], optional: the logits, kept for the backward pass (written by the first column tile).
    const float* hd_h2; const float* hd_W3; const float* hd_b3; const float* hd_noise; float* hd_logits;
    int hd_H; float hd_max_v, hd_max_w;
    // F, optional: Critic.forward's last layer on the activation this job writes, q = C . qp_w3 + qp_b3 (TD3:139), as partial sums
    // over the tile's 16 units: qp_out[i * qp_nt + column tile] (the bias rides in tile 0; the consumer adds the tiles in order)
    const float* qp_w3; const float* qp_b3; float* qp_out; int qp_nt;
    // G, optional: the next link of the actor-loss chain, through the critic's first layer to the action: partial sums over the
    // tile's 32 units of C[i][j] da_w[j * da_ld + o], o = 0, 1 (the two action columns of W1): da_out[(2 i + o) * da_nt + column tile]
    const float* da_w; float* da_out; int da_ld, da_nt;
    // G, optional: A is not read but evaluated -- it is the gradient at a critic's second hidden layer (TD3:249-260),
    //   A[m][k] = dq[m] hb_w3[k] [hb_h2[m][k] > 0],   dq[m] = 2 (q[m] - y[m]) / I,   y = r + (1 - d) gamma min(q1_t, q2_t)[m],
    // q of network hb_net and of the two targets (networks 2, 3) from td3_fwd_kernel's per-tile partial sums hb_qpart.  dq scales a
    // whole row, so the reduction runs on hb_w3 (.) [h2 > 0] and the epilogue multiplies.  The first column tile also writes what
    // the weight-gradient launch needs: hb_dq [I] (linear3's gradient = dq^T h2: a job of that launch) and hb_dz2 [I][R] = A.
    // (Was td3_critic_head_bwd_kernel: a launch between the forward pass and this one.)
    const float* hb_h2; const float* hb_w3; const float* hb_qpart; const float* hb_r; const float* hb_d;
    float* hb_dq; float* hb_dz2; int hb_qnt, hb_net; float hb_gamma;
    // G, optional, the ACTOR's counterpart (TD3:268-269, -mean Q1(s, pi(s)) arriving at the policy's second hidden layer):
    //   A[m][k] = (dl[m][0] ab_w3[k] + dl[m][1] ab_w3[R + k]) [ab_h2[m][k] > 0],   dl[m][o] = da[m][o] (max_v s (1 - s), max_w (1 - t^2)),
    // da from this kernel's own per-tile partial sums of the launch before (ab_dapart), s / t from the policy's logits.  The first
    // column tile writes ab_dl [I][2] (linear3's gradient = dl^T h2: a two-row job of the weight-gradient launch) and ab_dz2 = A.
    // (Was td3_actor_head_bwd_kernel.)
    const float* ab_h2; const float* ab_w3; const float* ab_dapart; const float* ab_logits;
    float* ab_dl; float* ab_dz2; int ab_dant; float ab_max_v, ab_max_w;
    // H, optional: A is one column (I = 1) of per-row loss gradients dq; loss_out[0] = mean squared TD error = (R / 4) sum dq^2
    float* loss_out;
};
struct GemmArgs { GemmJob job[6]; float beta1, beta2, eps, tau; TickArgs tick; int do_tick; };
static_assert(sizeof(GemmArgs) <= 4096, "GemmArgs travels in the kernarg segment (4 KB)");
// target <- target (1 - tau) + local tau (TD3:297-299)
__device__ __forceinline__ float td3_soft(float target, float local, float tau) { return target * (1.f - tau) + local * tau; }

// vectors that are only 4-byte aligned (a row of 398 floats starts on an 8-byte boundary at best): global_load_dwordx2 / x4
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2u_t __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4u_t __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ f32x4_t td3_ld4(const float* __restrict__ p, int c, int n)     // p[c .. c + 3], zero at and past n
{
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    if (c < n) v[0] = p[c];
    if (c + 1 < n) v[1] = p[c + 1];
    if (c + 2 < n) v[2] = p[c + 2];
    if (c + 3 < n) v[3] = p[c + 3];
    return v;
}
__device__ __forceinline__ f32x2_t td3_ld2(const float* __restrict__ p, int c, int n)
{
    f32x2_t v = {0.f, 0.f};
    if (c < n) v[0] = p[c];
    if (c + 1 < n) v[1] = p[c + 1];
    return v;
}
#define TD3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// MFMA 16x16x4 operand / result layout, lane l: a = A[row l & 15][k l >> 4], b = B[k l >> 4][col l & 15], acc[q] = C[row 4 (l >> 4) + q][col l & 15]

// F: a 16 x 16 tile per workgroup; the reduction in blocks of 16 inputs, block t = wavefront t mod 4.  Lane (li, lk) loads
// X[i0 + li][16 t + 4 lk ..+3] and W[j0 + li][the same]: component e of the two vectors is the pair the lane feeds to MFMA e of
// the block (k = 16 t + 4 lk + e on both sides).
#define TD3_FKB 8          /* blocks a wavefront has in flight (16 dwordx4 loads) */
__global__ void __launch_bounds__(256) td3_fwd_kernel(GemmArgs args)
{
    const GemmJob& jb = args.job[blockIdx.z];
    const int I = jb.I, J = jb.J, R = jb.R;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    if (args.do_tick && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) td3_tick(args.tick);
    if (i0 >= I || j0 >= J) return;
    __shared__ float red[4][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const float* __restrict__ arow = jb.A + (size_t)min(i0 + li, I - 1) * jb.lda;       // (rows past the edge: loaded, never stored)
    const float* __restrict__ brow = jb.B + (size_t)min(j0 + li, J - 1) * jb.ldb;
    const int nfull = R >> 4;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const float pbias = jb.bias[min(j0 + li, J - 1)];
    const bool ragged = (R & 15) && (nfull & 3) == wave;          // the last, partial block: loaded first, multiplied last
    f32x4_t ta = {0.f, 0.f, 0.f, 0.f}, tb = {0.f, 0.f, 0.f, 0.f};
    if (ragged) { ta = td3_ld4(arow, 16 * nfull + 4 * lk, R); tb = td3_ld4(brow, 16 * nfull + 4 * lk, R); }
    // the policy head of this tile's 16 rows: the hidden units in runs of 4, run c of every 16 = (wavefront c >> 2, lane group c & 3)
    const bool head = jb.hd_h2 != nullptr;
    __shared__ float hred[4][2][16];
    float hp0 = 0.f, hp1 = 0.f, wa0 = 0.f, wa1 = 0.f;
    if (head) {
        const int HH = jb.hd_H;
        const float* __restrict__ hrow = jb.hd_h2 + (size_t)min(i0 + li, I - 1) * HH;
        const float* __restrict__ w3 = jb.hd_W3;
        wa0 = brow[R]; wa1 = brow[R + 1];
        if ((HH & 3) == 0) {
#pragma unroll 4
            for (int n = 4 * (4 * wave + lk); n < HH; n += 64) {
                const f32x4_t hv = *(const f32x4u_t*)(hrow + n), u0 = *(const f32x4u_t*)(w3 + n), u1 = *(const f32x4u_t*)(w3 + HH + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) { hp0 = fmaf(hv[e], u0[e], hp0); hp1 = fmaf(hv[e], u1[e], hp1); }
            }
        } else {
            for (int n = 4 * (4 * wave + lk); n < HH; n += 64) {
                const f32x4_t hv = td3_ld4(hrow, n, HH), u0 = td3_ld4(w3, n, HH), u1 = td3_ld4(w3 + HH, n, HH);
#pragma unroll
                for (int e = 0; e < 4; ++e) { hp0 = fmaf(hv[e], u0[e], hp0); hp1 = fmaf(hv[e], u1[e], hp1); }
            }
        }
    }
    for (int t0 = wave; t0 < nfull; t0 += 4 * TD3_FKB) {
        f32x4_t av[TD3_FKB], bv[TD3_FKB];
#pragma unroll
        for (int u = 0; u < TD3_FKB; ++u) {
            const int t = min(t0 + 4 * u, nfull - 1);         // (past the end: a block that is loaded and not used)
            av[u] = *(const f32x4u_t*)(arow + 16 * t + 4 * lk);
            bv[u] = *(const f32x4u_t*)(brow + 16 * t + 4 * lk);
        }
#pragma unroll
        for (int u = 0; u < TD3_FKB; ++u) {
            if (t0 + 4 * u < nfull) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = TD3_MFMA(av[u][e], bv[u][e], acc);
            }
        }
    }
    if (ragged) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = TD3_MFMA(ta[e], tb[e], acc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][q][lane] = acc[q];
    if (head) {
        hp0 += __shfl_xor(hp0, 16, 64); hp0 += __shfl_xor(hp0, 32, 64);
        hp1 += __shfl_xor(hp1, 16, 64); hp1 += __shfl_xor(hp1, 32, 64);
        if (lk == 0) { hred[wave][0][li] = hp0; hred[wave][1][li] = hp1; }
    }
    __syncthreads();
    const int q = wave, i = i0 + 4 * lk + q, j = j0 + li;      // thread -> one element of the tile
    const bool in = i < I && j < J;
    float y = ((red[0][q][lane] + red[1][q][lane]) + red[2][q][lane]) + red[3][q][lane];
    if (head && in) {
        const int r = 4 * lk + q;
        const float lg0 = (((hred[0][0][r] + hred[1][0][r]) + hred[2][0][r]) + hred[3][0][r]) + jb.hd_b3[0];
        const float lg1 = (((hred[0][1][r] + hred[1][1][r]) + hred[2][1][r]) + hred[3][1][r]) + jb.hd_b3[1];
        float a0 = jb.hd_max_v / (1.f + expf(-lg0)), a1 = jb.hd_max_w * tanhf(lg1);
        if (jb.hd_noise) { a0 += jb.hd_noise[2 * i]; a1 += jb.hd_noise[2 * i + 1]; }
        if (jb.hd_logits && blockIdx.x == 0 && li == 0) { jb.hd_logits[2 * i] = lg0; jb.hd_logits[2 * i + 1] = lg1; }
        y = fmaf(a1, wa1, fmaf(a0, wa0, y));
    }
    y += pbias;
    if (jb.relu) y = fmaxf(y, 0.f);
    if (in) {
        const size_t o = (size_t)i * jb.ldc + j;
        jb.C[o] = y;
        if (jb.dz_out) jb.dz_out[o] = y > 0.f ? -jb.dz_w3[j] / jb.dz_rows : 0.f;
    }
    if (jb.qp_out) {                                // (uniform) this tile's share of q[i]: the 16 lanes of a row, then tile x's slot
        float pq = in ? y * jb.qp_w3[j] : 0.f;
        pq += __shfl_xor(pq, 1, 64); pq += __shfl_xor(pq, 2, 64); pq += __shfl_xor(pq, 4, 64); pq += __shfl_xor(pq, 8, 64);
        if (li == 0 && i < I) jb.qp_out[(size_t)i * jb.qp_nt + blockIdx.x] = blockIdx.x == 0 ? pq + jb.qp_b3[0] : pq;
    }
}

// G: a 16 x 32 tile per workgroup.  dY's rows are contiguous along the reduction (as in F), W's along the OUTPUT: lane (li, lk)
// loads W[16 t + 4 lk + e][j0 + 2 li, + 1] for e = 0..3 -- two accumulators, columns j0 + 2 c and j0 + 2 c + 1.
#define TD3_GKB 4
__global__ void __launch_bounds__(256) td3_dgrad_kernel(GemmArgs args)
{
    const GemmJob& jb = args.job[blockIdx.z];
    const int I = jb.I, J = jb.J, R = jb.R;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 32;
    if (i0 >= I || j0 >= J) return;
    __shared__ float red[4][8][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const float* __restrict__ arow = jb.A + (size_t)min(i0 + li, I - 1) * jb.lda;
    const float* __restrict__ B = jb.B;
    const int jc = j0 + 2 * li;
    const int nb = (R + 15) >> 4;
    const bool inner = j0 + 32 <= J;               // (uniform) no ragged edge along j
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float pmask[2];                                // the ReLU mask of this thread's two elements, requested before the reduction
#pragma unroll
    for (int c = 0; c < 2; ++c) { const int i = i0 + 4 * lk + wave, j = jc + c; pmask[c] = (i < I && j < J) ? jb.mask[(size_t)i * jb.ldc + j] : 0.f; }
    float pdw[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    if (jb.da_out) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (jc + c < J) { pdw[c][0] = jb.da_w[(size_t)(jc + c) * jb.da_ld]; pdw[c][1] = jb.da_w[(size_t)(jc + c) * jb.da_ld + 1]; }
    }
    // head-backward mode: this lane's row of dq (14 loads in flight with the operands')
    const bool hb = jb.hb_h2 != nullptr;
    __shared__ float dqs[16];
    float dq_li = 0.f;
    if (hb) arow = jb.hb_h2 + (size_t)min(i0 + li, I - 1) * jb.lda;
    if (hb && (blockIdx.x == 0 || (wave == 0 && lk == 0))) {      // (the first column tile needs dq in every lane: it stores dz2)
        const int m = min(i0 + li, I - 1);
        float qs[3];                               // this critic, the two target critics
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const float* __restrict__ pp = jb.hb_qpart + ((size_t)(n == 0 ? jb.hb_net : 1 + n) * I + m) * jb.hb_qnt;
            float a_ = 0.f;
            int t = 0;
#pragma unroll 4
            for (; t + 4 <= jb.hb_qnt; t += 4) { const f32x4_t v = *(const f32x4u_t*)(pp + t); a_ += (v[0] + v[1]) + (v[2] + v[3]); }
            for (; t < jb.hb_qnt; ++t) a_ += pp[t];
            qs[n] = a_;
        }
        const float y = jb.hb_r[m] + (1.f - jb.hb_d[m]) * jb.hb_gamma * fminf(qs[1], qs[2]);
        dq_li = 2.f * (qs[0] - y) / (float)I;
        if (wave == 0 && lk == 0) {
            dqs[li] = dq_li;
            if (blockIdx.x == 0 && i0 + li < I) jb.hb_dq[i0 + li] = dq_li;
        }
    }
    const bool ab = jb.ab_h2 != nullptr;
    float dl0 = 0.f, dl1 = 0.f;
    if (ab) {
        const int m = min(i0 + li, I - 1);
        float da[2];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const float* __restrict__ pp = jb.ab_dapart + ((size_t)2 * m + o) * jb.ab_dant;
            float a_ = 0.f;
            int t = 0;
#pragma unroll 2
            for (; t + 4 <= jb.ab_dant; t += 4) { const f32x4_t v = *(const f32x4u_t*)(pp + t); a_ += (v[0] + v[1]) + (v[2] + v[3]); }
            for (; t < jb.ab_dant; ++t) a_ += pp[t];
            da[o] = a_;
        }
        const float lg0 = jb.ab_logits[2 * m], lg1 = jb.ab_logits[2 * m + 1];
        const float s_ = 1.f / (1.f + expf(-lg0)), th = tanhf(lg1);
        dl0 = da[0] * (jb.ab_max_v * s_ * (1.f - s_));
        dl1 = da[1] * (jb.ab_max_w * (1.f - th * th));
        if (blockIdx.x == 0 && wave == 0 && lk == 0 && i0 + li < I) { jb.ab_dl[2 * (i0 + li)] = dl0; jb.ab_dl[2 * (i0 + li) + 1] = dl1; }
        arow = jb.ab_h2 + (size_t)m * jb.lda;
    }
    for (int t0 = wave; t0 < nb; t0 += 4 * TD3_GKB) {
        f32x4_t av[TD3_GKB];
        f32x2_t bv[TD3_GKB][4];
#pragma unroll
        for (int u = 0; u < TD3_GKB; ++u) {
            const int t = t0 + 4 * u, k = 16 * t + 4 * lk;
            if (t < nb && inner && 16 * t + 16 <= R) {
                av[u] = *(const f32x4u_t*)(arow + k);
                if (hb) {
                    const f32x4_t wv = *(const f32x4u_t*)(jb.hb_w3 + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[u][e] = av[u][e] > 0.f ? wv[e] : 0.f;
                }
                if (ab) {
                    const f32x4_t w0 = *(const f32x4u_t*)(jb.ab_w3 + k), w1 = *(const f32x4u_t*)(jb.ab_w3 + R + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[u][e] = av[u][e] > 0.f ? fmaf(dl1, w1[e], dl0 * w0[e]) : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[u][e] = *(const f32x2u_t*)(B + (size_t)(k + e) * jb.ldb + jc);
            } else if (t < nb) {
                av[u] = td3_ld4(arow, k, R);
                if (hb) {
                    const f32x4_t wv = td3_ld4(jb.hb_w3, k, R);
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[u][e] = av[u][e] > 0.f ? wv[e] : 0.f;
                }
                if (ab) {
                    const f32x4_t w0 = td3_ld4(jb.ab_w3, k, R), w1 = td3_ld4(jb.ab_w3 + R, k, R);
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[u][e] = av[u][e] > 0.f ? fmaf(dl1, w1[e], dl0 * w0[e]) : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[u][e] = k + e < R ? td3_ld2(B + (size_t)(k + e) * jb.ldb, jc, J) : f32x2_t{0.f, 0.f};
            } else {
                av[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[u][e] = f32x2_t{0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < TD3_GKB; ++u) {
            if (t0 + 4 * u < nb) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc0 = TD3_MFMA(av[u][e], bv[u][e][0], acc0); acc1 = TD3_MFMA(av[u][e], bv[u][e][1], acc1); }
            }
        }
        if ((hb || ab) && blockIdx.x == 0 && i0 + li < I) {      // dz2 = A, for the weight-gradient launch
            const float sc = hb ? dq_li : 1.f;
            float* __restrict__ zb = (hb ? jb.hb_dz2 : jb.ab_dz2) + (size_t)(i0 + li) * jb.lda;
#pragma unroll
            for (int u = 0; u < TD3_GKB; ++u) {
                const int t = t0 + 4 * u, k = 16 * t + 4 * lk;
                if (t >= nb) continue;
                float* __restrict__ zr = zb + k;
                if (k + 4 <= R) *(f32x4u_t*)zr = av[u] * sc;
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (k + e < R) zr[e] = av[u][e] * sc;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { red[wave][q][lane] = acc0[q]; red[wave][4 + q][lane] = acc1[q]; }
    __syncthreads();
    const int q = wave, i = i0 + 4 * lk + q;
    const float rowscale = hb ? dqs[4 * lk + q] : 1.f;
    float dv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int j = jc + c;
        const float d = (((red[0][4 * c + q][lane] + red[1][4 * c + q][lane]) + red[2][4 * c + q][lane]) + red[3][4 * c + q][lane]) * rowscale;
        dv[c] = (i < I && j < J && pmask[c] > 0.f) ? d : 0.f;
        if (i < I && j < J) jb.C[(size_t)i * jb.ldc + j] = dv[c];
    }
    if (jb.da_out) {                                // (uniform)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            float pa = fmaf(dv[1], pdw[1][o], dv[0] * pdw[0][o]);
            pa += __shfl_xor(pa, 1, 64); pa += __shfl_xor(pa, 2, 64); pa += __shfl_xor(pa, 4, 64); pa += __shfl_xor(pa, 8, 64);
            if (li == 0 && i < I) jb.da_out[((size_t)2 * i + o) * jb.da_nt + blockIdx.x] = pa;
        }
    }
}

// H: a 32 x 32 tile per workgroup, the batch rows split over the four wavefronts in steps of 4 (step s = wavefront s mod 4).
// Both operands are contiguous along their output index: lane (li, lk) loads dY[4 s + lk][i0 + 2 li, + 1] and
// X[4 s + lk][j0 + 2 li, + 1] -- 2 x 2 accumulators, acc[a][b] = the (rows i0 + 2 r + a) x (columns j0 + 2 c + b) sub-lattice.
#define TD3_HKS 8          /* steps a wavefront has in flight (16 dwordx2 loads, 32 MFMAs) */
__global__ void __launch_bounds__(256) td3_wgrad_kernel(GemmArgs args)
{
    const GemmJob& jb = args.job[blockIdx.z];
    const int I = jb.I, J = jb.J, R = jb.R;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    if (i0 >= I || j0 >= J) return;
    __shared__ float red[4][16][64];
    __shared__ float bred[4][4][32];
    __shared__ float lred[4][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const float* __restrict__ A = jb.A;
    const float* __restrict__ B = jb.B;
    const int ic = i0 + 2 * li, jc = j0 + 2 * li;
    const int ns = (R + 3) >> 2;
    // a lane's pair of rows / columns: 2 = both inside, 1 = only the first (odd extents), 0 = past the edge (a ragged tile: the lane
    // loads a pair that IS inside and its products are never stored).  Pairs that straddle the edge take the element-wise path.
    const int amode = ic + 1 < I ? 2 : ic < I ? 1 : 0, bmode = jc + 1 < J ? 2 : jc < J ? 1 : 0;
    const bool vec = __all(bmode != 1) && J >= 2;       // (an odd last ROW of the tile -- or I = 1, the linear3 jobs -- loads one element)
    const int jcv = bmode == 2 ? jc : 0;
    const float adam0 = jb.adam[0], adam1 = jb.adam[1];
    f32x4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float bs0 = 0.f, bs1 = 0.f;                    // sums of dY over this lane's rows (the bias gradient, first j-tile only)
    float ls0 = 0.f;                               // ... and of dY[.][0]^2 (loss_out)
    // the Adam step's operands of this thread's four elements, requested now: their round trip overlaps the reduction's
    float pm[4], pv[4], pw[4], pt[4];
#pragma unroll
    for (int z = 0; z < 4; ++z) {
        const int i = i0 + (tid >> 5) + 8 * z, j = j0 + (tid & 31);
        const bool in = i < I && j < J;
        const size_t o = in ? (size_t)i * jb.ldc + j : 0;
        pm[z] = in ? jb.m[o] : 0.f; pv[z] = in ? jb.v[o] : 0.f; pw[z] = in ? jb.C[o] : 0.f; pt[z] = (in && jb.tgt) ? jb.tgt[o] : 0.f;
    }
    for (int s0 = wave; s0 < ns; s0 += 4 * TD3_HKS) {
        f32x2_t av[TD3_HKS], bv[TD3_HKS];
#pragma unroll
        for (int u = 0; u < TD3_HKS; ++u) {
            const int s = s0 + 4 * u, k = 4 * s + lk;
            if (s < ns && vec && 4 * s + 4 <= R) {
                av[u] = f32x2_t{0.f, 0.f};
                if (amode == 2) av[u] = *(const f32x2u_t*)(A + (size_t)k * jb.lda + ic);
                else if (amode == 1) av[u][0] = A[(size_t)k * jb.lda + ic];
                bv[u] = *(const f32x2u_t*)(B + (size_t)k * jb.ldb + jcv);
            } else if (s < ns && k < R) {
                av[u] = td3_ld2(A + (size_t)k * jb.lda, ic, I);
                bv[u] = td3_ld2(B + (size_t)k * jb.ldb, jc, J);
            } else { av[u] = f32x2_t{0.f, 0.f}; bv[u] = f32x2_t{0.f, 0.f}; }
        }
#pragma unroll
        for (int u = 0; u < TD3_HKS; ++u) {
            if (s0 + 4 * u < ns) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = TD3_MFMA(av[u][a], bv[u][b], acc[a][b]);
                bs0 += av[u][0]; bs1 += av[u][1]; ls0 = fmaf(av[u][0], av[u][0], ls0);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[wave][(2 * a + b) * 4 + q][lane] = acc[a][b][q];
    bred[wave][lk][2 * li] = bs0; bred[wave][lk][2 * li + 1] = bs1;
    if (li == 0) lred[wave][lk] = ls0;
    __syncthreads();
    if (jb.loss_out && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        float l_ = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int k = 0; k < 4; ++k) l_ += lred[w][k];
        jb.loss_out[0] = l_ * (0.25f * (float)R);
    }
    // thread -> four elements of the tile, 32 consecutive columns per half-wavefront: row ii = 2 (4 g + q) + a, column jj = 2 c + b
#pragma unroll
    for (int z = 0; z < 4; ++z) {
        const int ii = (tid >> 5) + 8 * z, jj = tid & 31;
        const int a = ii & 1, g = ii >> 3, q = (ii >> 1) & 3, c = jj >> 1, b = jj & 1;
        const int slot = (2 * a + b) * 4 + q, l = 16 * g + c;
        const int i = i0 + ii, j = j0 + jj;
        if (i >= I || j >= J) continue;
        const float gsum = ((red[0][slot][l] + red[1][slot][l]) + red[2][slot][l]) + red[3][slot][l];
        const size_t o = (size_t)i * jb.ldc + j;
        const float m = args.beta1 * pm[z] + (1.f - args.beta1) * gsum;
        const float v = args.beta2 * pv[z] + (1.f - args.beta2) * gsum * gsum;
        jb.m[o] = m; jb.v[o] = v;
        const float w = pw[z] - adam0 * m / (sqrtf(v) / adam1 + args.eps);
        jb.C[o] = w;
        if (jb.tgt) jb.tgt[o] = td3_soft(pt[z], w, args.tau);
    }
    if (blockIdx.x == 0 && tid < 32 && i0 + tid < I && jb.bparam) {
        const int i = i0 + tid;
        float gsum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
            for (int k = 0; k < 4; ++k) gsum += bred[w][k][tid];
        const float m = args.beta1 * jb.bm[i] + (1.f - args.beta1) * gsum;
        const float v = args.beta2 * jb.bv[i] + (1.f - args.beta2) * gsum * gsum;
        jb.bm[i] = m; jb.bv[i] = v;
        const float b = jb.bparam[i] - adam0 * m / (sqrtf(v) / adam1 + args.eps);
        jb.bparam[i] = b;
        if (jb.btgt) jb.btgt[i] = td3_soft(jb.btgt[i], b, args.tau);
    }
}


// ---- small kernels ------------------------------------------------------------------------------------------------------
struct PrepArgs {
    const float *rs, *ra, *rr, *rs2, *rd;          // replay ring (rows `obs_dim` / 2 / 1 wide) or the explicit batch
    const float* noise_in;                         // explicit target-policy noise [B][2] (unit variance, before the clip) or null
    const int64_t* size_dev;                       // live replay size (device) or null = the rows ARE the batch
    float *xs, *x2, *r, *d, *noise;                // outputs: [B][D + 2] x 2, [B], [B], [B][2]
    const unsigned long long* counter;             // update counter (keys the sampling; advanced by td3_tick)
    uint64_t seed;
    int B, D;
    float noise_std, noise_clip;
};
__global__ void __launch_bounds__(256) td3_prep_kernel(PrepArgs p)
{
    const int m = blockIdx.x, tid = threadIdx.x, Dc = p.D + 2;
    const unsigned long long cnt = *p.counter;       // (advanced by td3_tick inside a later launch on the stream)
    size_t row = (size_t)m;
    if (p.size_dev) {
        const unsigned long long size = (unsigned long long)(*p.size_dev > 0 ? *p.size_dev : 1);
        const uint64_t h = cn_mix64(cn_mix64(p.seed ^ cn_mix64(cnt)) ^ (uint64_t)(uint32_t)m);
        row = (size_t)(h % size);
    }
    const float* s = p.rs + row * (size_t)p.D;
    const float* s2 = p.rs2 + row * (size_t)p.D;
    for (int c = tid; c < p.D; c += blockDim.x) {
        p.xs[(size_t)m * Dc + c] = s[c];
        p.x2[(size_t)m * Dc + c] = s2[c];
    }
    if (tid < 2) {
        p.xs[(size_t)m * Dc + p.D + tid] = p.ra[row * 2 + tid];
        float z;
        if (p.noise_in) z = p.noise_in[(size_t)m * 2 + tid];
        else {   // Box-Muller on a counter-based pair, keyed by (seed, update counter, row)
            const uint64_t h = cn_mix64(cn_mix64(p.seed ^ cn_mix64(cnt ^ 0x5bd1e995u)) ^ (uint64_t)(uint32_t)m);
            const float u1 = ((float)(uint32_t)(h >> 40) + 1.0f) * (1.0f / 16777217.0f);
            const float u2 = (float)(uint32_t)((h >> 8) & 0xffffffu) * (1.0f / 16777216.0f);
            const float rr = sqrtf(-2.0f * logf(u1));
            z = tid == 0 ? rr * cosf(6.28318530718f * u2) : rr * sinf(6.28318530718f * u2);
        }
        p.noise[(size_t)m * 2 + tid] = fminf(fmaxf(z * p.noise_std, -p.noise_clip), p.noise_clip);     // TD3:241-242
    }
    if (tid == 2) p.r[m] = p.rr[row];
    if (tid == 3) p.d[m] = p.rd[row];
}
__device__ __forceinline__ float td3_wave_sum(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
// The actor-loss chain -mean Q1(s, pi(s)) (TD3:268-269) has no kernel of its own: its first link (d/dh2 of the critic) is
// td3_fwd_kernel's dz epilogue; through the critic's first layer to the action (the two action columns of W1) = per-tile partial
// sums in td3_dgrad_kernel's da epilogue; through the heads' derivatives and the actor's linear3 = the same kernel's ab mode on
// the next launch; dW3a = dl^T h2a = a two-row job of td3_wgrad_kernel.  (Rounds 3-4: td3_dlogit_kernel, one wavefront per row,
// 4.8 us, and td3_actor_head_bwd_kernel, 9 us; the critics had td3_q_head_kernel and td3_critic_head_bwd_kernel.)
// ---- the collection loop's bookkeeping (cn_replay_write, cn_episode_log_add) --------------------------------------------------
// inclusive scan of one int per thread over a 1024-thread workgroup (wave scans + a scan of the 16 wave totals)
__device__ __forceinline__ int cn_block_scan_1024(int v, int* __restrict__ wsum, int& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(v, d, 64); if (lane >= d) v += u; }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int x = wsum[w]; if (w < wave) before += x; tot += x; }
    __syncthreads();
    total = tot;
    return v + before;
}
// slots of the kept rows, in row order; the ring's position and fill level move at the end (one workgroup: they are read first)
__global__ void __launch_bounds__(1024) cn_replay_slot_kernel(const uint8_t* __restrict__ keep, int n, int64_t cap, int64_t* pos_dev,
                                                              int64_t* size_dev, int32_t* __restrict__ slot)
{
    __shared__ int wsum[16];
    const int64_t pos = *pos_dev, size = *size_dev;
    int64_t carry = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int k = i < n ? (keep ? (keep[i] != 0) : 1) : 0;
        int tot;
        const int c = cn_block_scan_1024(k, wsum, tot);
        if (i < n) slot[i] = k ? (int32_t)((pos + carry + c - 1) % cap) : -1;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        *pos_dev = (pos + carry) % cap;
        *size_dev = size + carry < cap ? size + carry : cap;
    }
}
struct ReplayCopyArgs { cn_replay_ring ring; const float *s, *a, *r, *s2; const uint8_t* done; const int32_t* slot; };
__global__ void __launch_bounds__(256) cn_replay_copy_kernel(ReplayCopyArgs p)
{
    const int i = blockIdx.x, D = p.ring.obs_dim;
    const int32_t sl = p.slot[i];
    if (sl < 0) return;
    const float* __restrict__ s = p.s + (size_t)i * D;
    const float* __restrict__ s2 = p.s2 + (size_t)i * D;
    float* __restrict__ ds = p.ring.s + (size_t)sl * D;
    float* __restrict__ ds2 = p.ring.s2 + (size_t)sl * D;
    for (int c = threadIdx.x; c < D; c += blockDim.x) { ds[c] = s[c]; ds2[c] = s2[c]; }
    if (threadIdx.x < 2) p.ring.a[(size_t)sl * 2 + threadIdx.x] = p.a[(size_t)i * 2 + threadIdx.x];
    if (threadIdx.x == 2) p.ring.r[sl] = p.r[i];
    if (threadIdx.x == 3) p.ring.d[sl] = p.done[i] ? 1.f : 0.f;
}
// the finished episodes' rows and the running totals, one workgroup
struct EpisodeLogArgs { cn_episode_log log; const uint8_t* done; const int32_t* counters; int cols; const float* ret; const uint8_t* trans; float launch; int n; };
__global__ void __launch_bounds__(1024) cn_episode_log_kernel(EpisodeLogArgs p)
{
    __shared__ int wsum[16];
    __shared__ double red[5][16];
    const int64_t n0 = *p.log.n_dev;
    int64_t carry = 0;
    double t[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int base = 0; base < p.n; base += 1024) {
        const int i = base + threadIdx.x;
        const int k = i < p.n ? (p.done[i] != 0) : 0;
        int tot;
        const int c = cn_block_scan_1024(k, wsum, tot);
        if (i < p.n) {
            const int32_t* __restrict__ cr = p.counters + (size_t)i * p.cols;
            if (k) {
                const int64_t at = n0 + carry + c - 1;
                if (at < p.log.max_rows) {
                    float* __restrict__ row = p.log.rows + (size_t)at * 8;
                    row[0] = (float)cr[4]; row[1] = (float)cr[5]; row[2] = p.ret[i]; row[3] = (float)cr[13];
                    row[4] = (float)cr[10]; row[5] = (float)cr[11]; row[6] = (float)cr[12]; row[7] = p.launch;
                }
                t[0] += 1.0; t[1] += (double)(float)cr[4]; t[2] += (double)p.ret[i]; t[3] += (double)(float)cr[13];
            }
            if (p.trans[i]) t[4] += 1.0;
        }
        carry += tot;
    }
    // totals: lanes, then wavefronts, in a fixed order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        double v = t[q];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0) red[q][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += red[threadIdx.x][w];
        p.log.tot_dev[threadIdx.x] += v;
    }
    if (threadIdx.x == 0) *p.log.n_dev = n0 + carry;
}

}  // namespace

// ---- host side ------------------------------------------------------------------------------------------------------------
struct cn_td3_s {
    cn_td3_config cfg;
    int device;
    int B, D, Dc, H;
    float* pool = nullptr;         // one allocation for the whole workspace
    // batch
    float *xs, *x2, *r, *d, *noise, *logits;
    float *t_h1, *t_h2;            // target actor
    float *c_h1[4], *c_h2[4];               // q1, q2, q1_t, q2_t
    float *dq[2];                           // the critics' loss gradients per row (td3_dgrad_kernel's head-backward mode)
    float *dl;                              // the actor's: dlogit [B][2]
    float *qpart, *dapart; int qnt, dant;   // partial sums of the critics' outputs [4][B][qnt] and of the action gradient [B][2][dant]
    float *a_h1, *a_h2;            // actor
    float *dz2[2], *dz1[2];
    float* loss;
    float* adam;                   // [4]
    float* steps;                  // [2]
    double* pw;                    // [4] running products beta^t
    unsigned long long* counter;
    // Adam moments: actor, q1, q2 x {w1, b1, w2, b2, w3, b3} x {m, v}
    float* mom[3][6][2];
};

namespace {
struct DevScope {
    int prev = -1, want;
    explicit DevScope(int dev) : want(dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != want) (void)hipSetDevice(want); }
    ~DevScope() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
};
const cn_td3_mlp& net_of(const cn_td3_config& c, int k) { return k == 0 ? c.actor : k == 1 ? c.q1 : c.q2; }
size_t param_count(const cn_td3_s* h, int net, int j)
{
    const size_t in1 = net == 0 ? (size_t)h->D : (size_t)h->Dc, out3 = net == 0 ? 2 : 1, H = (size_t)h->H;
    switch (j) { case 0: return H * in1; case 1: return H; case 2: return H * H; case 3: return H; case 4: return out3 * H; default: return out3; }
}
template <int MODE>
void launch_gemm(const GemmArgs& ga, int njobs, hipStream_t st)
{
    constexpr int TI = MODE == GEMM_H ? 32 : 16, TJ = MODE == GEMM_F ? 16 : 32;      // the kernel's tile of C
    int gx = 0, gy = 0;
    for (int z = 0; z < njobs; ++z) { const int x_ = (ga.job[z].J + TJ - 1) / TJ, y_ = (ga.job[z].I + TI - 1) / TI; gx = x_ > gx ? x_ : gx; gy = y_ > gy ? y_ : gy; }
    if (MODE == GEMM_F) hipLaunchKernelGGL(td3_fwd_kernel, dim3(gx, gy, njobs), dim3(256), 0, st, ga);
    else if (MODE == GEMM_G) hipLaunchKernelGGL(td3_dgrad_kernel, dim3(gx, gy, njobs), dim3(256), 0, st, ga);
    else hipLaunchKernelGGL(td3_wgrad_kernel, dim3(gx, gy, njobs), dim3(256), 0, st, ga);
}
}  // namespace

extern "C" const char* cn_td3_last_error(void) { return g_td3_err.c_str(); }

extern "C" int cn_td3_create(const cn_td3_config* cfg, int device, cn_td3_handle* out)
{
    if (!cfg || !out) return td3_fail(CN_ERR_ARG, "cn_td3_create: null argument");
    const cn_td3_config& c = *cfg;
    if (c.obs_dim < 1 || c.hidden < 1 || c.batch < 1 || c.batch > 4096 || c.hidden > 4096 || c.policy_delay < 1)
        return td3_fail(CN_ERR_CONFIG, "cn_td3_create: obs_dim / hidden / batch / policy_delay out of range");
    const cn_td3_mlp* nets[6] = {&c.actor, &c.actor_t, &c.q1, &c.q1_t, &c.q2, &c.q2_t};
    for (const cn_td3_mlp* n : nets)
        if (!n->w1 || !n->b1 || !n->w2 || !n->b2 || !n->w3 || !n->b3) return td3_fail(CN_ERR_ARG, "cn_td3_create: null parameter pointer");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return td3_fail(CN_ERR_NO_DEVICE, "cn_td3_create: no HIP device (libcrowdnav has no CPU fallback)");
    if (device < 0 || device >= ndev) return td3_fail(CN_ERR_ARG, "cn_td3_create: bad device ordinal");
    DevScope scope(device);
    cn_td3_s* h = new (std::nothrow) cn_td3_s();
    if (!h) return td3_fail(CN_ERR_ARG, "cn_td3_create: out of memory");
    h->cfg = c; h->device = device; h->B = c.batch; h->D = c.obs_dim; h->Dc = c.obs_dim + 2; h->H = c.hidden;
    const size_t B = h->B, Dc = h->Dc, H = h->H;
    size_t words = 2 * B * Dc + 2 * B + 2 * B + 2 * B          // xs, x2, r, d, noise, logits
                   + 2 * B * H + 4 * (2 * B * H) + 2 * B * H + 4 * B * H + 1 + 4 + 2 + 2 + 8  // t_h, c_h, a_h, dz, loss, adam, steps, counter, pw
                   + 4 * B * ((H + 15) / 16) + 2 * B * ((H + 31) / 32) + 2 * B + 2 * B;                        // qpart, dapart
    size_t mom_words = 0;
    for (int net = 0; net < 3; ++net) for (int j = 0; j < 6; ++j) mom_words += 2 * param_count(h, net, j);
    hipError_t e = hipMalloc(&h->pool, (words + mom_words) * sizeof(float));
    if (e != hipSuccess) { delete h; return td3_fail(CN_ERR_HIP, std::string("cn_td3_create: hipMalloc: ") + hipGetErrorString(e)); }
    e = hipMemset(h->pool, 0, (words + mom_words) * sizeof(float));
    if (e != hipSuccess) { (void)hipFree(h->pool); delete h; return td3_fail(CN_ERR_HIP, std::string("cn_td3_create: hipMemset: ") + hipGetErrorString(e)); }
    float* q = h->pool;
    auto take = [&](size_t n) { float* r_ = q; q += n; return r_; };
    h->counter = (unsigned long long*)take(2);       // first: 8-byte aligned
    h->pw = (double*)take(8);
    { const double one[4] = {1.0, 1.0, 1.0, 1.0}; e = hipMemcpy(h->pw, one, sizeof(one), hipMemcpyHostToDevice); }
    if (e != hipSuccess) { (void)hipFree(h->pool); delete h; return td3_fail(CN_ERR_HIP, std::string("cn_td3_create: hipMemcpy: ") + hipGetErrorString(e)); }
    h->xs = take(B * Dc); h->x2 = take(B * Dc); h->r = take(B); h->d = take(B); h->noise = take(2 * B); h->logits = take(2 * B);
    h->t_h1 = take(B * H); h->t_h2 = take(B * H);
    for (int z = 0; z < 4; ++z) { h->c_h1[z] = take(B * H); h->c_h2[z] = take(B * H); }
    h->qnt = (int)((H + 15) / 16); h->dant = (int)((H + 31) / 32);
    h->qpart = take(4 * B * h->qnt); h->dapart = take(2 * B * h->dant); h->dq[0] = take(B); h->dq[1] = take(B); h->dl = take(2 * B);
    h->a_h1 = take(B * H); h->a_h2 = take(B * H);
    for (int z = 0; z < 2; ++z) { h->dz2[z] = take(B * H); h->dz1[z] = take(B * H); }
    h->loss = take(1); h->adam = take(4); h->steps = take(2);
    for (int net = 0; net < 3; ++net) for (int j = 0; j < 6; ++j) for (int k = 0; k < 2; ++k) h->mom[net][j][k] = take(param_count(h, net, j));
    *out = h;
    return CN_OK;
}

extern "C" void cn_td3_destroy(cn_td3_handle h)
{
    if (!h) return;
    DevScope scope(h->device);
    (void)hipFree(h->pool);
    delete h;
}

extern "C" const float* cn_td3_loss_dev(cn_td3_handle h) { return h ? h->loss : nullptr; }

extern "C" int cn_td3_update(cn_td3_handle h, int do_actor, const cn_td3_batch* batch, void* stream)
{
    if (!h) return td3_fail(CN_ERR_ARG, "cn_td3_update: null handle");
    const cn_td3_config& c = h->cfg;
    if (!batch && (!c.replay_s || !c.replay_a || !c.replay_r || !c.replay_s2 || !c.replay_d || !c.replay_size_dev))
        return td3_fail(CN_ERR_ARG, "cn_td3_update: no explicit batch and no replay ring in the configuration");
    if (batch && (!batch->s || !batch->a || !batch->r || !batch->s2 || !batch->d)) return td3_fail(CN_ERR_ARG, "cn_td3_update: null batch pointer");
    DevScope scope(h->device);
    hipStream_t st = (hipStream_t)stream;
    const int B = h->B, D = h->D, Dc = h->Dc, H = h->H;
    // 0. sample / gather, noise, Adam constants
    PrepArgs pa;
    memset(&pa, 0, sizeof(pa));
    if (batch) { pa.rs = batch->s; pa.ra = batch->a; pa.rr = batch->r; pa.rs2 = batch->s2; pa.rd = batch->d; pa.noise_in = batch->target_noise; pa.size_dev = nullptr; }
    else { pa.rs = c.replay_s; pa.ra = c.replay_a; pa.rr = c.replay_r; pa.rs2 = c.replay_s2; pa.rd = c.replay_d; pa.noise_in = nullptr; pa.size_dev = c.replay_size_dev; }
    pa.xs = h->xs; pa.x2 = h->x2; pa.r = h->r; pa.d = h->d; pa.noise = h->noise; pa.counter = h->counter;
    pa.seed = c.seed; pa.B = B; pa.D = D; pa.noise_std = c.noise_std; pa.noise_clip = c.noise_clip;
    hipLaunchKernelGGL(td3_prep_kernel, dim3(B), dim3(256), 0, st, pa);

    auto fwd_job = [&](GemmJob& j, const float* X, int ldx, int K, const float* W, const float* b, float* Y) {
        memset(&j, 0, sizeof(j));
        j.A = X; j.B = W; j.C = Y; j.bias = b; j.I = B; j.J = H; j.R = K; j.lda = ldx; j.ldb = K; j.ldc = H; j.relu = 1;
    };
    auto bwd_data_job = [&](GemmJob& j, const float* dY, const float* W, const float* mask, float* dX) {   // through a hidden layer (H x H)
        memset(&j, 0, sizeof(j));
        j.A = dY; j.B = W; j.C = dX; j.mask = mask; j.I = B; j.J = H; j.R = H; j.lda = H; j.ldb = H; j.ldc = H;
    };
    auto wgrad_job = [&](GemmJob& j, const float* dY, const float* X, int ldx, int K, float* W, float* bparam, int net, int wj, const float* adam) {
        memset(&j, 0, sizeof(j));
        j.A = dY; j.B = X; j.C = W; j.I = H; j.J = K; j.R = B; j.lda = H; j.ldb = ldx; j.ldc = K;
        j.m = h->mom[net][wj][0]; j.v = h->mom[net][wj][1]; j.bparam = bparam; j.bm = h->mom[net][wj + 1][0]; j.bv = h->mom[net][wj + 1][1]; j.adam = adam;
    };
    GemmArgs ga;
    ga.beta1 = c.beta1; ga.beta2 = c.beta2; ga.eps = c.eps; ga.tau = c.tau;
    ga.tick.adam = h->adam; ga.tick.steps = h->steps; ga.tick.pw = h->pw; ga.tick.counter = h->counter; ga.tick.do_actor = do_actor ? 1 : 0;
    ga.tick.lr_critic = c.lr_critic; ga.tick.lr_actor = c.lr_actor; ga.tick.beta1 = c.beta1; ga.tick.beta2 = c.beta2;
    ga.do_tick = 0;
    // 1-2. the target actor's hidden layers on s2 (TD3:238).  On actor updates the policy's own hidden layers on s (TD3:268; they
    // read the actor, which the critic step does not touch) ride in the same two launches as a second job.
    const int na = do_actor ? 2 : 1;
    fwd_job(ga.job[0], h->x2, Dc, D, c.actor_t.w1, c.actor_t.b1, h->t_h1);
    fwd_job(ga.job[1], h->xs, Dc, D, c.actor.w1, c.actor.b1, h->a_h1);
    launch_gemm<GEMM_F>(ga, na, st);
    fwd_job(ga.job[0], h->t_h1, H, H, c.actor_t.w2, c.actor_t.b2, h->t_h2);
    fwd_job(ga.job[1], h->a_h1, H, H, c.actor.w2, c.actor.b2, h->a_h2);
    launch_gemm<GEMM_F>(ga, na, st);
    // (3, the policies' last layer and heads, runs inside the launches that consume the actions: 4 and 13)
    auto head_job = [&](GemmJob& j, const float* h2, const cn_td3_mlp& pol, const float* noise, float* logits) {
        j.R = D; j.hd_h2 = h2; j.hd_W3 = pol.w3; j.hd_b3 = pol.b3; j.hd_noise = noise; j.hd_logits = logits; j.hd_H = H; j.hd_max_v = c.max_v; j.hd_max_w = c.max_w;
    };
    // 4-6. the four critics forward: q1, q2 on (s, a); q1_t, q2_t on (s2, a2)
    const cn_td3_mlp* crit[4] = {&c.q1, &c.q2, &c.q1_t, &c.q2_t};
    for (int z = 0; z < 4; ++z) fwd_job(ga.job[z], z < 2 ? h->xs : h->x2, Dc, Dc, crit[z]->w1, crit[z]->b1, h->c_h1[z]);
    for (int z = 2; z < 4; ++z) head_job(ga.job[z], h->t_h2, c.actor_t, h->noise, nullptr);      // a2 = pi_t(s2) + clipped noise
    launch_gemm<GEMM_F>(ga, 4, st);
    // 5-6. ... their second layers, and the last (q = h2 . W3 + b3) as per-tile partial sums in the same epilogue; the tick too
    for (int z = 0; z < 4; ++z) {
        fwd_job(ga.job[z], h->c_h1[z], H, H, crit[z]->w2, crit[z]->b2, h->c_h2[z]);
        ga.job[z].qp_w3 = crit[z]->w3; ga.job[z].qp_b3 = crit[z]->b3; ga.job[z].qp_out = h->qpart + (size_t)z * B * h->qnt; ga.job[z].qp_nt = h->qnt;
    }
    ga.do_tick = 1;
    launch_gemm<GEMM_F>(ga, 4, st);
    ga.do_tick = 0;
    // 7-8. TD target, MSE gradient (per row, evaluated where it is consumed) and through the second hidden layer:
    // dz1 = (dz2 W2) (.) [h1 > 0], dz2 = dq W3 (.) [h2 > 0]   (W2, W3 are read here, stepped in 9)
    for (int z = 0; z < 2; ++z) {
        bwd_data_job(ga.job[z], h->c_h2[z], crit[z]->w2, h->c_h1[z], h->dz1[z]);
        GemmJob& j = ga.job[z];
        j.hb_h2 = h->c_h2[z]; j.hb_w3 = crit[z]->w3; j.hb_qpart = h->qpart; j.hb_qnt = h->qnt; j.hb_net = z; j.hb_r = h->r; j.hb_d = h->d;
        j.hb_gamma = c.gamma; j.hb_dq = h->dq[z]; j.hb_dz2 = h->dz2[z];
    }
    launch_gemm<GEMM_G>(ga, 2, st);
    // 9. weight gradients folded into Adam: W2, b2, W1, b1 of both critics, and linear3 (dW3 = dq^T h2, db3 = sum dq: one-row jobs)
    for (int z = 0; z < 2; ++z) {
        wgrad_job(ga.job[z], h->dz2[z], h->c_h1[z], H, H, crit[z]->w2, crit[z]->b2, 1 + z, 2, h->adam);
        wgrad_job(ga.job[2 + z], h->dz1[z], h->xs, Dc, Dc, crit[z]->w1, crit[z]->b1, 1 + z, 0, h->adam);
        wgrad_job(ga.job[4 + z], h->dq[z], h->c_h2[z], H, H, crit[z]->w3, crit[z]->b3, 1 + z, 4, h->adam);
        ga.job[4 + z].I = 1; ga.job[4 + z].lda = 1;
        if (z == 0) ga.job[4].loss_out = h->loss;             // the first critic's MSE: what Agent.learn returns
        if (do_actor) {      // the target critics follow in the same epilogue (nothing reads them again in this update)
            ga.job[z].tgt = crit[2 + z]->w2; ga.job[z].btgt = crit[2 + z]->b2;
            ga.job[2 + z].tgt = crit[2 + z]->w1; ga.job[2 + z].btgt = crit[2 + z]->b1;
            ga.job[4 + z].tgt = crit[2 + z]->w3; ga.job[4 + z].btgt = crit[2 + z]->b3;
        }
    }
    launch_gemm<GEMM_H>(ga, 6, st);
    if (do_actor) {
        // (10-11, the policy's hidden layers on s, ran inside launches 1-2; 12, its head, runs inside 13)
        // 13-14. the UPDATED first critic on (s, pi(s)) (TD3:268)
        fwd_job(ga.job[0], h->xs, Dc, Dc, c.q1.w1, c.q1.b1, h->c_h1[0]);    // (xs's own action columns are not read: head_job)
        head_job(ga.job[0], h->a_h2, c.actor, nullptr, h->logits);                                  // pi(s)
        launch_gemm<GEMM_F>(ga, 1, st);
        fwd_job(ga.job[0], h->c_h1[0], H, H, c.q1.w2, c.q1.b2, h->c_h2[0]);
        ga.job[0].dz_w3 = c.q1.w3; ga.job[0].dz_out = h->dz2[0]; ga.job[0].dz_rows = (float)B;      // 15. -mean Q's gradient at h2, in the epilogue
        launch_gemm<GEMM_F>(ga, 1, st);
        // 16-17. ... back to the action, through the heads, linear3 of the actor + Adam
        bwd_data_job(ga.job[0], h->dz2[0], c.q1.w2, h->c_h1[0], h->dz1[0]);
        ga.job[0].da_w = c.q1.w1 + D; ga.job[0].da_ld = Dc; ga.job[0].da_out = h->dapart; ga.job[0].da_nt = h->dant;     // the action columns of W1
        launch_gemm<GEMM_G>(ga, 1, st);
        // 17-19. through the heads' derivatives and the actor's hidden layers (the heads' part evaluated inside the backward GEMM,
        // linear3's gradient dl^T h2 as a two-row job of the weight-gradient launch)
        bwd_data_job(ga.job[0], h->a_h2, c.actor.w2, h->a_h1, h->dz1[1]);
        {
            GemmJob& j = ga.job[0];
            j.ab_h2 = h->a_h2; j.ab_w3 = c.actor.w3; j.ab_dapart = h->dapart; j.ab_dant = h->dant; j.ab_logits = h->logits;
            j.ab_max_v = c.max_v; j.ab_max_w = c.max_w; j.ab_dl = h->dl; j.ab_dz2 = h->dz2[1];
        }
        launch_gemm<GEMM_G>(ga, 1, st);
        wgrad_job(ga.job[0], h->dz2[1], h->a_h1, H, H, c.actor.w2, c.actor.b2, 0, 2, h->adam + 2);
        wgrad_job(ga.job[1], h->dz1[1], h->xs, Dc, D, c.actor.w1, c.actor.b1, 0, 0, h->adam + 2);
        wgrad_job(ga.job[2], h->dl, h->a_h2, H, H, c.actor.w3, c.actor.b3, 0, 4, h->adam + 2);
        ga.job[2].I = 2; ga.job[2].lda = 2;
        ga.job[0].tgt = c.actor_t.w2; ga.job[0].btgt = c.actor_t.b2; ga.job[1].tgt = c.actor_t.w1; ga.job[1].btgt = c.actor_t.b1;
        ga.job[2].tgt = c.actor_t.w3; ga.job[2].btgt = c.actor_t.b3;
        launch_gemm<GEMM_H>(ga, 3, st);
        // (20, the soft updates of the three targets, ran in the Adam epilogues of 7, 9, 17 and 19)
    }
    TD3CHK(hipGetLastError());
    return CN_OK;
}

extern "C" int cn_replay_write(const cn_replay_ring* ring, const float* s, const float* a, const float* r, const float* s2,
                               const uint8_t* done, const uint8_t* keep, int n, int32_t* slot_scratch, int device, void* stream)
{
    if (!ring || !s || !a || !r || !s2 || !done || !slot_scratch) return td3_fail(CN_ERR_ARG, "cn_replay_write: null argument");
    if (!ring->s || !ring->a || !ring->r || !ring->s2 || !ring->d || !ring->pos_dev || !ring->size_dev || ring->capacity < 1 || ring->obs_dim < 1)
        return td3_fail(CN_ERR_ARG, "cn_replay_write: incomplete ring");
    if (n < 1) return td3_fail(CN_ERR_ARG, "cn_replay_write: n < 1");
    if ((int64_t)n > ring->capacity) return td3_fail(CN_ERR_ARG, "cn_replay_write: more rows than the ring holds (two rows of one call would share a slot)");
    DevScope scope(device);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cn_replay_slot_kernel, dim3(1), dim3(1024), 0, st, keep, n, ring->capacity, ring->pos_dev, ring->size_dev, slot_scratch);
    ReplayCopyArgs ca;
    ca.ring = *ring; ca.s = s; ca.a = a; ca.r = r; ca.s2 = s2; ca.done = done; ca.slot = slot_scratch;
    hipLaunchKernelGGL(cn_replay_copy_kernel, dim3(n), dim3(256), 0, st, ca);
    TD3CHK(hipGetLastError());
    return CN_OK;
}

extern "C" int cn_episode_log_add(const cn_episode_log* log, const uint8_t* done, const int32_t* counters, int counter_cols,
                                  const float* last_return, const uint8_t* transitions, float launch, int n, int device, void* stream)
{
    if (!log || !log->rows || !log->n_dev || !log->tot_dev || !done || !counters || !last_return || !transitions)
        return td3_fail(CN_ERR_ARG, "cn_episode_log_add: null argument");
    if (n < 1 || counter_cols < 14 || log->max_rows < 0) return td3_fail(CN_ERR_ARG, "cn_episode_log_add: n < 1, fewer than 14 counter columns or max_rows < 0");
    DevScope scope(device);
    EpisodeLogArgs ea;
    ea.log = *log; ea.done = done; ea.counters = counters; ea.cols = counter_cols; ea.ret = last_return; ea.trans = transitions; ea.launch = launch; ea.n = n;
    hipLaunchKernelGGL(cn_episode_log_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ea);
    TD3CHK(hipGetLastError());
    return CN_OK;
}
