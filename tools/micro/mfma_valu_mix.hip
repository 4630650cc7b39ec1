// Does a wave streaming v_mfma_f32_16x16x4_f32 slow down ANOTHER wave of the same SIMD that issues vector-ALU work, and vice versa?
// (cn_policy_kernel: can one workgroup's actor tile -- matrix cores -- run beside another's Env.step -- vector unit -- for free?)
// One workgroup per CU; waves 0..3 (one per SIMD) take role A, waves 4..7 role B, optional waves 8..11 / 12..15 repeat the roles.
//   mode 0: A = mfma, B idle     mode 1: A idle, B = valu      mode 2: A = mfma, B = valu      mode 3: A = B = mfma (2 / 4 MFMA waves per SIMD)
// VALU kinds: 0 = v_fma_f64 (8 accumulators), 1 = v_add_u32 (plain VOP2)
// Cycles: s_memtime around each wave's loop; reported per instruction of that wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/mfma_valu_mix tools/micro/mfma_valu_mix.hip && tools/micro/bin/mfma_valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND, bool BF16 = false>
__global__ void __launch_bounds__(1024) mix_kernel(int mode, int iters, int waves_per_role, long long* out, float* sink)
{
    const int wave = threadIdx.x >> 6;
    const int role = mode == 3 ? 0 : (wave / 4) & 1;          // waves 0-3: A, 4-7: B, 8-11: A, 12-15: B
    const bool act = (role == 0) ? (mode == 0 || mode >= 2) : (mode == 1 || mode == 2);
    if ((wave >> 3) >= waves_per_role) return;
    long long t0 = 0, t1 = 0;
    __syncthreads();
    if (act && role == 0) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = (float)threadIdx.x, y = 1.0f;
        bf16x8 xb, yb;
        for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)(float)(threadIdx.x + i); yb[i] = (__bf16)1.0f; }
        t0 = (long long)__builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (BF16) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, a3, 0, 0, 0);
                } else {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
                }
            }
        }
        t1 = (long long)__builtin_amdgcn_s_memtime();
        sink[blockIdx.x * 1024 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    } else if (act) {
        if constexpr (KIND == 0 || KIND == 2) {
            double v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7, m = 1.0000001, c = 1e-9;
            t0 = (long long)__builtin_amdgcn_s_memtime();
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if constexpr (KIND == 0)
                        asm volatile("v_fma_f64 %0, %0, %8, %9\nv_fma_f64 %1, %1, %8, %9\nv_fma_f64 %2, %2, %8, %9\nv_fma_f64 %3, %3, %8, %9\n"
                                     "v_fma_f64 %4, %4, %8, %9\nv_fma_f64 %5, %5, %8, %9\nv_fma_f64 %6, %6, %8, %9\nv_fma_f64 %7, %7, %8, %9\n"
                                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(m), "v"(c));
                    else
                        asm volatile("v_fma_f64 %0, %0, %8, %9\ns_add_u32 s20, s20, 1\nv_fma_f64 %1, %1, %8, %9\ns_add_u32 s20, s20, 1\n"
                                     "v_fma_f64 %2, %2, %8, %9\ns_add_u32 s20, s20, 1\nv_fma_f64 %3, %3, %8, %9\ns_add_u32 s20, s20, 1\n"
                                     "v_fma_f64 %4, %4, %8, %9\ns_add_u32 s20, s20, 1\nv_fma_f64 %5, %5, %8, %9\ns_add_u32 s20, s20, 1\n"
                                     "v_fma_f64 %6, %6, %8, %9\ns_add_u32 s20, s20, 1\nv_fma_f64 %7, %7, %8, %9\ns_add_u32 s20, s20, 1\n"
                                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(m), "v"(c) : "s20");
                }
            }
            t1 = (long long)__builtin_amdgcn_s_memtime();
            sink[blockIdx.x * 1024 + threadIdx.x] = (float)(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7);
        } else {
            unsigned v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7, m = 3;
            t0 = (long long)__builtin_amdgcn_s_memtime();
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    asm volatile("v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\n"
                                 "v_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8\n"
                                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(m));
            }
            t1 = (long long)__builtin_amdgcn_s_memtime();
            sink[blockIdx.x * 1024 + threadIdx.x] = (float)(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7);
        }
    }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + wave] = act ? t1 - t0 : 0;
}

template <int KIND, bool BF16 = false>
static int run(const char* name)
{
    const int iters = 4000, blocks = 256;
    long long* out; float* sink;
    CHK(hipMalloc(&out, blocks * 16 * 8)); CHK(hipMalloc(&sink, blocks * 1024 * 4));
    static long long h[256 * 16];
    printf("%s\n", name);
    for (int wpr = 1; wpr <= 2; ++wpr)
        for (int mode = 0; mode < (KIND == 0 ? 4 : 3); ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                CHK(hipMemset(out, 0, blocks * 16 * 8));
                hipLaunchKernelGGL((mix_kernel<KIND, BF16>), dim3(blocks), dim3(512 * wpr), 0, 0, mode, iters, wpr, out, sink);
                CHK(hipDeviceSynchronize());
            }
            CHK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
            double a = 0, b = 0; int na = 0, nb = 0;
            for (int blk = 0; blk < blocks; ++blk)
                for (int w = 0; w < 8 * wpr; ++w) {
                    if (!h[blk * 16 + w]) continue;
                    if (mode == 3 || ((w / 4) & 1) == 0) { a += (double)h[blk * 16 + w]; ++na; } else { b += (double)h[blk * 16 + w]; ++nb; }
                }
            // per wave: 32 instructions per iteration (kind 2: 32 VALU + 32 SALU, reported per VALU)
            printf("  %d wave(s) per role per SIMD, mode %d (%s): mfma wave %6.2f cycles per MFMA   valu wave %6.2f cycles per VALU instruction\n", wpr, mode,
                   mode == 0 ? "mfma alone" : mode == 1 ? "valu alone" : mode == 2 ? "both      " : "mfma x 2  ", na ? a / na / (32.0 * iters) : 0.0, nb ? b / nb / (32.0 * iters) : 0.0);
        }
    (void)hipFree(out); (void)hipFree(sink);
    return 0;
}

int main()
{
    if (run<0>("A = v_mfma_f32_16x16x4_f32 (4 accumulators)   B = v_fma_f64 (8 accumulators)")) return 1;
    if (run<1>("A = v_mfma_f32_16x16x4_f32                    B = v_add_u32 (8 accumulators)")) return 1;
    // the same beside a bf16 MFMA, for contrast (NOT the actor's arithmetic: TD3:96-106 is float32)
    if (run<0, true>("A = v_mfma_f32_16x16x32_bf16 (4 accumulators)  B = v_fma_f64 (8 accumulators)")) return 1;
    if (run<1, true>("A = v_mfma_f32_16x16x32_bf16                   B = v_add_u32 (8 accumulators)")) return 1;
    return 0;
}
