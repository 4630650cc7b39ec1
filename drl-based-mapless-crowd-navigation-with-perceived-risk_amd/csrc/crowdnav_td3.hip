// crowdnav_td3.hip -- the TD3 update (td3.py:225-285 of the reference: Agent.learn) as a short chain of HIP kernels (gfx950).
//
// The caller of the hot path.  A vectorised environment makes the learner the bottleneck: through PyTorch one
// update is ~150 small kernels (1.2 ms as a hipGraph at batch 128).  Here the same arithmetic is 8 launches for the critic
// step and 6 more when the actor and the targets move (rounds 3-4: 10 + 11), all float32 like the reference:
//   prep        sample the replay on the device (counter-based indices and target-policy noise), gather [s|a], [s2|.], r, d
//   gemm F      Y = act(X W^T + b) on the f32 matrix cores (v_mfma_f32_16x16x4_f32), up to four networks per launch;
//               optional: the policy's last layer + heads evaluated in place of X's action columns, the critic's last layer
//               as per-tile partial sums of the activation just written, the first link of the actor-loss chain
//   gemm G      dX = (dY W) (.) [H > 0]   (back-propagation through a ReLU layer); optional: the action gradient's partial sums
//   gemm H      dW = dY^T X folded into the Adam step of W (and of b) and the soft update of the target's copy: the gradient
//               never exists in memory
//   two head-backward kernels (TD target + MSE gradient + the critics' linear3 backward; heads' derivatives + the actor's)
// Launch order: prep | actor_t L1 (+ actor L1) | L2 (+ L2) | critics L1 (target actions from the head) | critics L2 (+ q partials,
// tick) | critic heads backward | G | H   and, every policy_delay-th update:  q1 L1 on (s, pi(s)) | q1 L2 (+ dz) | G (+ da
// partials) | actor head backward | G | H.
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
};
struct GemmArgs { GemmJob job[4]; float beta1, beta2, eps, tau; TickArgs tick; int do_tick; };
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
    for (int t0 = wave; t0 < nb; t0 += 4 * TD3_GKB) {
        f32x4_t av[TD3_GKB];
        f32x2_t bv[TD3_GKB][4];
#pragma unroll
        for (int u = 0; u < TD3_GKB; ++u) {
            const int t = t0 + 4 * u, k = 16 * t + 4 * lk;
            if (t < nb && inner && 16 * t + 16 <= R) {
                av[u] = *(const f32x4u_t*)(arow + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[u][e] = *(const f32x2u_t*)(B + (size_t)(k + e) * jb.ldb + jc);
            } else if (t < nb) {
                av[u] = td3_ld4(arow, k, R);
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
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { red[wave][q][lane] = acc0[q]; red[wave][4 + q][lane] = acc1[q]; }
    __syncthreads();
    const int q = wave, i = i0 + 4 * lk + q;
    float dv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int j = jc + c;
        const float d = ((red[0][4 * c + q][lane] + red[1][4 * c + q][lane]) + red[2][4 * c + q][lane]) + red[3][4 * c + q][lane];
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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lk = lane >> 4;
    const float* __restrict__ A = jb.A;
    const float* __restrict__ B = jb.B;
    const int ic = i0 + 2 * li, jc = j0 + 2 * li;
    const int ns = (R + 3) >> 2;
    // a lane's pair of rows / columns: 2 = both inside, 1 = only the first (odd extents), 0 = past the edge (a ragged tile: the lane
    // loads a pair that IS inside and its products are never stored).  Pairs that straddle the edge take the element-wise path.
    const int amode = ic + 1 < I ? 2 : ic < I ? 1 : 0, bmode = jc + 1 < J ? 2 : jc < J ? 1 : 0;
    const bool vec = __all(amode != 1 && bmode != 1) && I >= 2 && J >= 2;
    const int icv = amode == 2 ? ic : 0, jcv = bmode == 2 ? jc : 0;
    const float adam0 = jb.adam[0], adam1 = jb.adam[1];
    f32x4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float bs0 = 0.f, bs1 = 0.f;                    // sums of dY over this lane's rows (the bias gradient, first j-tile only)
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
                av[u] = *(const f32x2u_t*)(A + (size_t)k * jb.lda + icv);
                bv[u] = *(const f32x2u_t*)(B + (size_t)k * jb.ldb + jcv);
                if (amode == 0) av[u] = f32x2_t{0.f, 0.f};
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
                bs0 += av[u][0]; bs1 += av[u][1];
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
    __syncthreads();
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
// TD target, MSE gradient and linear3's backward + Adam step for the two critics; workgroup (x, z) = 16 hidden units of critic z,
// its 256 threads = 16 row groups x 16 units:
//   y = r + (1 - d) gamma min(q1_t, q2_t)  (TD3:249-252);  loss_z = mean (q_z - y)^2;  dq_z = 2 (q_z - y) / B
//   dz2_z[m][n] = dq_z[m] W3_z[n] [h2_z[m][n] > 0];  dW3_z[n] = sum_m dq_z[m] h2_z[m][n];  db3_z = sum_m dq_z[m]
struct CriticHeadBwdArgs {
    const float *r, *d;
    const float* qpart; int qnt;                   // q of the four critics as partial sums: qpart[(net * B + m) * qnt + tile] (td3_fwd_kernel)
    const float* h2[2]; float* dz2[2];
    float* W3[2]; float* b3[2]; float* m3[2]; float* v3[2]; float* mb3[2]; float* vb3[2];
    const float* adam; float* loss;
    float* W3t[2]; float* b3t[2];                  // actor updates: the target critics' last layers (soft-updated here), else null
    int B, H; float gamma, beta1, beta2, eps, tau;
};
// Shape of this kernel and of td3_actor_head_bwd_kernel: 16 hidden units x 16 row groups per workgroup (H / 16 workgroups a
// network; four workgroups of 64 units were four round trips of eight loads in series behind the dq phase: 9.8 us), a thread's
// eight rows of h2 in flight BEFORE the dq phase, the Adam moments too -- the kernel is one memory round trip, not six.
#define TD3_HC 16
__global__ void __launch_bounds__(256) td3_critic_head_bwd_kernel(CriticHeadBwdArgs a)
{
    extern __shared__ float sm[];                   // dq [B] | partial [16][16] | red [8]
    float* dq = sm; float* part = sm + a.B; float* red = part + 256;
    const int z = blockIdx.y, tid = threadIdx.x;
    const int rg = tid >> 4, c = tid & 15, n = blockIdx.x * TD3_HC + c;
    const bool on = n < a.H;
    const float* __restrict__ h2 = a.h2[z];
    float hv[8];
    auto load = [&](int m0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int m = m0 + 16 * u; hv[u] = (on && m < a.B) ? h2[(size_t)m * a.H + n] : 0.f; }
    };
    load(rg);
    const float w = on ? a.W3[z][n] : 0.f;
    float m3 = 0.f, v3 = 0.f;
    if (rg == 0 && on) { m3 = a.m3[z][n]; v3 = a.v3[z][n]; }
    float e2 = 0.f, sdq = 0.f;
    for (int m = tid; m < a.B; m += 256) {
        float qs[3];                               // this critic, the two target critics
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float* __restrict__ pp = a.qpart + ((size_t)(k == 0 ? z : 1 + k) * a.B + m) * a.qnt;
            float acc = 0.f;
            int t = 0;
#pragma unroll 4
            for (; t + 4 <= a.qnt; t += 4) { const f32x4_t v = *(const f32x4u_t*)(pp + t); acc += (v[0] + v[1]) + (v[2] + v[3]); }
            for (; t < a.qnt; ++t) acc += pp[t];
            qs[k] = acc;
        }
        const float y = a.r[m] + (1.f - a.d[m]) * a.gamma * fminf(qs[1], qs[2]);
        const float e = qs[0] - y;
        const float g = 2.f * e / (float)a.B;
        dq[m] = g; e2 += e * e; sdq += g;
    }
    if (blockIdx.x == 0) {                           // the loss (critic 1: what Agent.learn returns) and the bias of linear3
        e2 = td3_wave_sum(e2); sdq = td3_wave_sum(sdq);
        if ((tid & 63) == 0) { red[tid >> 6] = e2; red[4 + (tid >> 6)] = sdq; }
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) {
        if (z == 0) a.loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)a.B;
        const float g = (red[4] + red[5]) + (red[6] + red[7]);
        const float mm = a.beta1 * a.mb3[z][0] + (1.f - a.beta1) * g;
        const float vv = a.beta2 * a.vb3[z][0] + (1.f - a.beta2) * g * g;
        a.mb3[z][0] = mm; a.vb3[z][0] = vv;
        const float b = a.b3[z][0] - a.adam[0] * mm / (sqrtf(vv) / a.adam[1] + a.eps);
        a.b3[z][0] = b;
        if (a.b3t[z]) a.b3t[z][0] = td3_soft(a.b3t[z][0], b, a.tau);
    }
    float g = 0.f;
    for (int m0 = rg; m0 < a.B; m0 += 128) {
        float cur[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = hv[u];
        if (m0 + 128 < a.B) load(m0 + 128);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = m0 + 16 * u;
            if (on && m < a.B) { a.dz2[z][(size_t)m * a.H + n] = cur[u] > 0.f ? dq[m] * w : 0.f; g = fmaf(dq[m], cur[u], g); }
        }
    }
    part[rg * 16 + c] = g;
    __syncthreads();
    if (rg == 0 && on) {
        g = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 4) g += (part[r * 16 + c] + part[(r + 1) * 16 + c]) + (part[(r + 2) * 16 + c] + part[(r + 3) * 16 + c]);
        const float mm = a.beta1 * m3 + (1.f - a.beta1) * g;
        const float vv = a.beta2 * v3 + (1.f - a.beta2) * g * g;
        a.m3[z][n] = mm; a.v3[z][n] = vv;
        const float wn = w - a.adam[0] * mm / (sqrtf(vv) / a.adam[1] + a.eps);
        a.W3[z][n] = wn;
        if (a.W3t[z]) a.W3t[z][n] = td3_soft(a.W3t[z][n], wn, a.tau);
    }
}
// actor loss -mean Q1(s, pi(s)) (TD3:268-269): its first link (d/dh2 of the critic) is td3_fwd_kernel's optional epilogue;
// then through the critic's first layer to the action (the two action columns of W1): da[m][o] = sum_n dz1q[m][n] W1q[n][D + o],
// partial sums per tile in td3_dgrad_kernel's epilogue, added up here; through the heads' derivatives to the logits:
// dlogit = da (.) (max_v s (1 - s), max_w (1 - t^2)).  (Rounds 3-4: a kernel of its own, one wavefront per row, 4.8 us.  Every
// workgroup re-evaluating the whole product for all rows was measured then: 14 -> 30 us.)
// ... then linear3 of the ACTOR backward + its Adam step; workgroup x = 16 hidden units, 16 row groups x 16 units:
//   dz2a[m][n] = sum_o dlogit[m][o] W3a[o][n] [h2a[m][n] > 0];  dW3a[o][n] = sum_m dlogit[m][o] h2a[m][n];  db3a[o] = sum_m dlogit[m][o]
struct ActorHeadBwdArgs {
    const float* dapart; int dant;                 // the action gradient as partial sums: dapart[(2 m + o) * dant + tile] (td3_dgrad_kernel)
    const float* logits; float max_v, max_w;       // the policy's logits on the batch (td3_fwd_kernel's head)
    const float* h2a; float* dz2a;
    float *W3, *b3, *m3, *v3, *mb3, *vb3;
    float *W3t, *b3t;                              // the target actor's last layer, soft-updated here
    const float* adam;
    int B, H; float beta1, beta2, eps, tau;
};
__global__ void __launch_bounds__(256) td3_actor_head_bwd_kernel(ActorHeadBwdArgs a)
{
    extern __shared__ float sm[];                   // dl [2 B] | partial [2][16][16]
    float* dl = sm; float* part = sm + 2 * a.B;
    const int tid = threadIdx.x;
    const int rg = tid >> 4, c = tid & 15, n = blockIdx.x * TD3_HC + c;
    const bool on = n < a.H;
    float hv[8];
    auto load = [&](int m0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int m = m0 + 16 * u; hv[u] = (on && m < a.B) ? a.h2a[(size_t)m * a.H + n] : 0.f; }
    };
    load(rg);
    const float w0 = on ? a.W3[n] : 0.f, w1 = on ? a.W3[a.H + n] : 0.f;
    float m3[2] = {0.f, 0.f}, v3[2] = {0.f, 0.f};
    if (rg == 0 && on) { m3[0] = a.m3[n]; v3[0] = a.v3[n]; m3[1] = a.m3[a.H + n]; v3[1] = a.v3[a.H + n]; }
    // through the heads' derivatives to the logits: dlogit = da (.) (max_v s (1 - s), max_w (1 - t^2))
    for (int t = tid; t < 2 * a.B; t += 256) {
        const float* __restrict__ pp = a.dapart + (size_t)t * a.dant;
        float da = 0.f;
        int k = 0;
#pragma unroll 2
        for (; k + 4 <= a.dant; k += 4) { const f32x4_t v = *(const f32x4u_t*)(pp + k); da += (v[0] + v[1]) + (v[2] + v[3]); }
        for (; k < a.dant; ++k) da += pp[k];
        const float lg = a.logits[t];
        float dh;
        if ((t & 1) == 0) { const float s_ = 1.f / (1.f + expf(-lg)); dh = a.max_v * s_ * (1.f - s_); }
        else { const float th = tanhf(lg); dh = a.max_w * (1.f - th * th); }
        dl[t] = da * dh;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < 64) {               // the bias of linear3: lanes over the rows, then across the wavefront
        float s0 = 0.f, s1 = 0.f;
        for (int m = tid; m < a.B; m += 64) { s0 += dl[2 * m]; s1 += dl[2 * m + 1]; }
        s0 = td3_wave_sum(s0); s1 = td3_wave_sum(s1);
        if (tid < 2) {
            const float g = tid == 0 ? s0 : s1;
            const float mm = a.beta1 * a.mb3[tid] + (1.f - a.beta1) * g;
            const float vv = a.beta2 * a.vb3[tid] + (1.f - a.beta2) * g * g;
            a.mb3[tid] = mm; a.vb3[tid] = vv;
            const float b = a.b3[tid] - a.adam[2] * mm / (sqrtf(vv) / a.adam[3] + a.eps);
            a.b3[tid] = b;
            a.b3t[tid] = td3_soft(a.b3t[tid], b, a.tau);
        }
    }
    float g0 = 0.f, g1 = 0.f;
    for (int m0 = rg; m0 < a.B; m0 += 128) {
        float cur[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = hv[u];
        if (m0 + 128 < a.B) load(m0 + 128);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = m0 + 16 * u;
            if (on && m < a.B) {
                a.dz2a[(size_t)m * a.H + n] = cur[u] > 0.f ? fmaf(dl[2 * m + 1], w1, dl[2 * m] * w0) : 0.f;
                g0 = fmaf(dl[2 * m], cur[u], g0); g1 = fmaf(dl[2 * m + 1], cur[u], g1);
            }
        }
    }
    part[rg * 16 + c] = g0; part[256 + rg * 16 + c] = g1;
    __syncthreads();
    if (rg == 0 && on) {
        const float w[2] = {w0, w1};
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const float* pp = part + 256 * o;
            float g = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 4) g += (pp[r * 16 + c] + pp[(r + 1) * 16 + c]) + (pp[(r + 2) * 16 + c] + pp[(r + 3) * 16 + c]);
            const size_t ix = (size_t)o * a.H + n;
            const float mm = a.beta1 * m3[o] + (1.f - a.beta1) * g;
            const float vv = a.beta2 * v3[o] + (1.f - a.beta2) * g * g;
            a.m3[ix] = mm; a.v3[ix] = vv;
            const float wn = w[o] - a.adam[2] * mm / (sqrtf(vv) / a.adam[3] + a.eps);
            a.W3[ix] = wn;
            a.W3t[ix] = td3_soft(a.W3t[ix], wn, a.tau);
        }
    }
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
                   + 4 * B * ((H + 15) / 16) + 2 * B * ((H + 31) / 32);                        // qpart, dapart
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
    h->qpart = take(4 * B * h->qnt); h->dapart = take(2 * B * h->dant);
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
    // 7. TD target, MSE gradients, linear3 backward + Adam (both critics)
    CriticHeadBwdArgs ca;
    ca.r = h->r; ca.d = h->d; ca.qpart = h->qpart; ca.qnt = h->qnt;
    for (int z = 0; z < 2; ++z) {
        ca.h2[z] = h->c_h2[z]; ca.dz2[z] = h->dz2[z]; ca.W3[z] = crit[z]->w3; ca.b3[z] = crit[z]->b3;
        ca.m3[z] = h->mom[1 + z][4][0]; ca.v3[z] = h->mom[1 + z][4][1]; ca.mb3[z] = h->mom[1 + z][5][0]; ca.vb3[z] = h->mom[1 + z][5][1];
    }
    for (int z = 0; z < 2; ++z) { ca.W3t[z] = do_actor ? crit[2 + z]->w3 : nullptr; ca.b3t[z] = do_actor ? crit[2 + z]->b3 : nullptr; }
    ca.tau = c.tau;
    ca.adam = h->adam; ca.loss = h->loss; ca.B = B; ca.H = H; ca.gamma = c.gamma; ca.beta1 = c.beta1; ca.beta2 = c.beta2; ca.eps = c.eps;
    hipLaunchKernelGGL(td3_critic_head_bwd_kernel, dim3((H + TD3_HC - 1) / TD3_HC, 2), dim3(256), (B + 512) * sizeof(float), st, ca);
    // 8. through the second hidden layer: dz1 = (dz2 W2) (.) [h1 > 0]   (W2 is read here, stepped in 9)
    for (int z = 0; z < 2; ++z) bwd_data_job(ga.job[z], h->dz2[z], crit[z]->w2, h->c_h1[z], h->dz1[z]);
    launch_gemm<GEMM_G>(ga, 2, st);
    // 9. weight gradients folded into Adam: W2, b2, W1, b1 of both critics
    for (int z = 0; z < 2; ++z) {
        wgrad_job(ga.job[z], h->dz2[z], h->c_h1[z], H, H, crit[z]->w2, crit[z]->b2, 1 + z, 2, h->adam);
        wgrad_job(ga.job[2 + z], h->dz1[z], h->xs, Dc, Dc, crit[z]->w1, crit[z]->b1, 1 + z, 0, h->adam);
        if (do_actor) {      // the target critics follow in the same epilogue (nothing reads them again in this update)
            ga.job[z].tgt = crit[2 + z]->w2; ga.job[z].btgt = crit[2 + z]->b2;
            ga.job[2 + z].tgt = crit[2 + z]->w1; ga.job[2 + z].btgt = crit[2 + z]->b1;
        }
    }
    launch_gemm<GEMM_H>(ga, 4, st);
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
        ActorHeadBwdArgs aa;
        aa.dapart = h->dapart; aa.dant = h->dant; aa.logits = h->logits; aa.max_v = c.max_v; aa.max_w = c.max_w;
        aa.h2a = h->a_h2; aa.dz2a = h->dz2[1];
        aa.W3 = c.actor.w3; aa.b3 = c.actor.b3; aa.m3 = h->mom[0][4][0]; aa.v3 = h->mom[0][4][1]; aa.mb3 = h->mom[0][5][0]; aa.vb3 = h->mom[0][5][1];
        aa.W3t = c.actor_t.w3; aa.b3t = c.actor_t.b3; aa.tau = c.tau;
        aa.adam = h->adam; aa.B = B; aa.H = H; aa.beta1 = c.beta1; aa.beta2 = c.beta2; aa.eps = c.eps;
        hipLaunchKernelGGL(td3_actor_head_bwd_kernel, dim3((H + TD3_HC - 1) / TD3_HC), dim3(256), (2 * B + 512) * sizeof(float), st, aa);
        // 18-19. the actor's hidden layers
        bwd_data_job(ga.job[0], h->dz2[1], c.actor.w2, h->a_h1, h->dz1[1]); launch_gemm<GEMM_G>(ga, 1, st);
        wgrad_job(ga.job[0], h->dz2[1], h->a_h1, H, H, c.actor.w2, c.actor.b2, 0, 2, h->adam + 2);
        wgrad_job(ga.job[1], h->dz1[1], h->xs, Dc, D, c.actor.w1, c.actor.b1, 0, 0, h->adam + 2);
        ga.job[0].tgt = c.actor_t.w2; ga.job[0].btgt = c.actor_t.b2; ga.job[1].tgt = c.actor_t.w1; ga.job[1].btgt = c.actor_t.b1;
        launch_gemm<GEMM_H>(ga, 2, st);
        // (20, the soft updates of the three targets, ran in the Adam epilogues of 7, 9, 17 and 19)
    }
    TD3CHK(hipGetLastError());
    return CN_OK;
}
