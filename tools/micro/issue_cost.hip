// What does ONE wave instruction of each class in the env kernel's ledger cost on gfx950?  ("4.1 cycles per
// VALU instruction on average says nothing about which class to cut".)
//
// For every class two kernels of the same 32-instruction block, looped:
//   thr  -- 8 independent accumulators, round robin: consecutive instructions never depend on each other -> ISSUE cost
//   lat  -- one accumulator: every instruction waits for the previous one                                -> issue + LATENCY
// each run at 1 wave per SIMD (256 workgroups x 256 threads) and at 4 waves per SIMD (256 x 1024: the env kernel's occupancy).
// Cycles come from s_memtime inside the kernel (shader clock: 2385 MHz under load, bench.py's sustained leg); the 4-wave column
// is cycles per instruction per SIMD: (latest end - earliest start over the workgroup's waves) / instructions issued per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/issue_cost tools/micro/issue_cost.hip && tools/micro/bin/issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define R8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define D8(I) I(0) I(0) I(0) I(0) I(0) I(0) I(0) I(0)
#define THR32(I) R8(I) R8(I) R8(I) R8(I)
#define LAT32(I) D8(I) D8(I) D8(I) D8(I)

// ---- instruction strings: %0..%7 accumulators, %8 / %9 loop-invariant operands ---------------------------------------------------
#define I_FMA64(i)   "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define I_MUL64(i)   "v_mul_f64 %" #i ", %" #i ", %8\n"
#define I_ADD64(i)   "v_add_f64 %" #i ", %" #i ", %8\n"
#define I_MAX64(i)   "v_max_f64 %" #i ", %" #i ", %8\n"
#define I_RCP64(i)   "v_rcp_f64 %" #i ", %" #i "\n"
#define I_SQRT64(i)  "v_sqrt_f64 %" #i ", %" #i "\n"
#define I_RNDNE64(i) "v_rndne_f64 %" #i ", %" #i "\n"
#define I_LDEXP64(i) "v_ldexp_f64 %" #i ", %" #i ", 1\n"
#define I_CMP64(i)   "v_cmp_lt_f64 vcc, %" #i ", %8\n"
#define I_CMPSEL64(i) "v_cmp_lt_f64 vcc, %8, %9\nv_cndmask_b32 %" #i ", %" #i ", %10, vcc\nv_cndmask_b32 %" #i ", %10, %" #i ", vcc\n"   /* a float64 select: compare + one v_cndmask per half */
#define I_DIVSC64(i) "v_div_scale_f64 %" #i ", vcc, %" #i ", %8, %" #i "\n"
#define I_DIVFIX64(i) "v_div_fixup_f64 %" #i ", %" #i ", %8, %9\n"
#define I_CVT_F64_I32(i) "v_cvt_f64_i32 %" #i ", %8\n"                 /* i32 -> f64 (no chain possible: type changes) */
#define I_CVT_I32_F64(i) "v_cvt_i32_f64 %" #i ", %8\n"
#define I_LSHL64(i)  "v_lshlrev_b64 %" #i ", 1, %" #i "\n"
#define I_FMA32(i)   "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_PKFMA32(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define I_ADDU32(i)  "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_MOV32(i)   "v_mov_b32 %" #i ", %8\n"
#define I_AND32(i)   "v_and_b32 %" #i ", %" #i ", %8\n"
#define I_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define I_CNDMASK_S(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[22:23]\n"          /* mask in an SGPR pair (what the compiler emits most) */
#define I_CNDMASK_EV(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"              /* VOP3 encoding, mask still vcc */
#define I_CNDMASK_SV(i) "s_mov_b64 vcc, s[22:23]\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"   /* vcc written by the SCALAR unit right before (2 instructions) */
#define I_CNDMASK_VV(i) "v_cmp_lt_i32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"   /* vcc written by a vector compare right before (2) */
#define I_MIX_VS(i)  "v_fma_f64 %" #i ", %" #i ", %8, %9\ns_add_u32 s20, s20, 1\n"                 /* one float64 VALU + one SALU, alternating (2) */
#define I_MIX_V32S(i) "v_add_u32 %" #i ", %" #i ", %8\ns_add_u32 s20, s20, 1\n"
#define I_BFI(i)     "v_bfi_b32 %" #i ", %9, %8, %" #i "\n"                          /* select by a VGPR bit mask */
#define I_ANDOR(i)   "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define I_MED3I(i)   "v_med3_i32 %" #i ", %" #i ", %8, %9\n"
#define I_MINI32(i)  "v_min_i32 %" #i ", %" #i ", %8\n"
#define I_ASHR(i)    "v_ashrrev_i32 %" #i ", 31, %" #i "\n"
#define I_CMPSW(i)   "v_cmp_lt_i32 s[22:23], %" #i ", %8\n"                         /* compare into an SGPR pair */
#define I_CMPX(i)    "v_cmpx_lt_i32 %" #i ", %8\ns_mov_b64 exec, -1\n"              /* compare into exec, restored (2 instructions) */
#define I_CMPI32(i)  "v_cmp_lt_i32 vcc, %" #i ", %8\n"
#define I_MULLO(i)   "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define I_MUL24(i)   "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define I_MAD24(i)   "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define I_ADD3(i)    "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define I_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define I_BFE(i)     "v_bfe_u32 %" #i ", %" #i ", 3, 7\n"
#define I_DPP(i)     "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPPADD(i)  "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_BCNT(i)    "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define I_MBCNT(i)   "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
#define I_RDLANE(i)  "v_readlane_b32 s20, %" #i ", 3\n"
#define I_RDLANE_USE(i) "v_readlane_b32 s20, %" #i ", 3\ns_nop 0\nv_add_u32 %" #i ", s20, %" #i "\n"   /* VALU -> SGPR -> VALU round trip, 2 + nop */
#define I_RDFIRST(i) "v_readfirstlane_b32 s20, %" #i "\n"
#define I_WRLANE(i)  "v_writelane_b32 %" #i ", s21, 5\n"
// scalar unit: accumulators are SGPRs
#define S_ADD32(i)   "s_add_u32 %" #i ", %" #i ", %8\n"
#define S_AND64(i)   "s_and_b64 %" #i ", %" #i ", %8\n"
#define S_LSHL64(i)  "s_lshl_b64 %" #i ", %" #i ", 1\n"
#define S_BCNT64(i)  "s_bcnt1_i32_b64 s20, %" #i "\n"
#define S_FF1_64(i)  "s_ff1_i32_b64 s20, %" #i "\n"
#define S_MUL32(i)   "s_mul_i32 %" #i ", %" #i ", %8\n"
#define S_CSEL(i)    "s_cmp_lt_u32 %" #i ", %8\ns_cselect_b32 %" #i ", %" #i ", %8\n"                 /* 2 instructions */

struct Res { long long cyc; };
// A SIMD's arbiter serves its oldest wave first, so one wave's own loop time says nothing about the SIMD's throughput when four
// share it: the block reports (latest end) - (earliest start) over all its waves.
#define T_BEGIN() __shared__ long long t_min, t_max; if (threadIdx.x == 0) { t_min = 0x7fffffffffffffffLL; t_max = 0; } __syncthreads(); \
                  const long long t0 = __builtin_amdgcn_s_memtime();
#define T_END()   const long long t1 = __builtin_amdgcn_s_memtime(); \
                  if ((threadIdx.x & 63) == 0) { atomicMin((unsigned long long*)&t_min, (unsigned long long)t0); atomicMax((unsigned long long*)&t_max, (unsigned long long)t1); } \
                  __syncthreads(); if (threadIdx.x == 0) out[blockIdx.x].cyc = t_max - t_min;

#define CV "v"
#define CS "s"
#define DEF_V(NAME, I, T) DEF_KERNEL(NAME##_thr, THR32(I), T, CV, "vcc", "scc", "s20", "s21", "s22", "s23") DEF_KERNEL(NAME##_lat, LAT32(I), T, CV, "vcc", "scc", "s20", "s21", "s22", "s23")
#define DEF_S(NAME, I, T) DEF_KERNEL(NAME##_thr, THR32(I), T, CS, "scc", "s20", "s21") DEF_KERNEL(NAME##_lat, LAT32(I), T, CS, "scc", "s20", "s21")
#define DEF_KERNEL(NAME, BODY32, T, CON, ...)                                                           \
    __global__ void __launch_bounds__(1024) NAME(Res* out, int iters, T x, T y)                          \
    {                                                                                                  \
        T a0 = x, a1 = x, a2 = x, a3 = x, a4 = x, a5 = x, a6 = x, a7 = x;                              \
        if (CON[0] == 'v') { a1 += (T)threadIdx.x; a2 += (T)threadIdx.x; }                              \
        T_BEGIN()                                                                                      \
        for (int it = 0; it < iters; ++it)                                                             \
            asm volatile(BODY32 : "+" CON(a0), "+" CON(a1), "+" CON(a2), "+" CON(a3), "+" CON(a4), "+" CON(a5), "+" CON(a6), "+" CON(a7) \
                         : CON(x), CON(y) : __VA_ARGS__);                                              \
        T_END()                                                                                        \
        if (a0 == (T)12345 && a1 == a2 && a3 == a4 && a5 == a6 && a7 == a0) out[blockIdx.x].cyc = 0;  \
    }

DEF_V(fma64, I_FMA64, double) DEF_V(mul64, I_MUL64, double) DEF_V(add64, I_ADD64, double) DEF_V(max64, I_MAX64, double)
DEF_V(rcp64, I_RCP64, double) DEF_V(sqrt64, I_SQRT64, double) DEF_V(rndne64, I_RNDNE64, double) DEF_V(ldexp64, I_LDEXP64, double)
DEF_V(cmp64, I_CMP64, double) DEF_V(divscale64, I_DIVSC64, double) DEF_V(divfix64, I_DIVFIX64, double)
DEF_V(lshl64, I_LSHL64, unsigned long long)
DEF_V(fma32, I_FMA32, float) DEF_V(pkfma32, I_PKFMA32, double)
DEF_V(addu32, I_ADDU32, unsigned) DEF_V(mov32, I_MOV32, unsigned) DEF_V(and32, I_AND32, unsigned) DEF_V(cndmask, I_CNDMASK, unsigned)
DEF_V(cndmask_ev, I_CNDMASK_EV, unsigned) DEF_V(cndmask_sv, I_CNDMASK_SV, unsigned) DEF_V(cndmask_vv, I_CNDMASK_VV, int)
DEF_V(mix_vs, I_MIX_VS, double) DEF_V(mix_v32s, I_MIX_V32S, unsigned)
DEF_V(cndmask_s, I_CNDMASK_S, unsigned) DEF_V(bfi, I_BFI, unsigned) DEF_V(andor, I_ANDOR, unsigned) DEF_V(med3i, I_MED3I, int)
DEF_V(mini32, I_MINI32, int) DEF_V(ashr, I_ASHR, int) DEF_V(cmpsw, I_CMPSW, int)
DEF_V(cmpi32, I_CMPI32, int) DEF_V(mullo, I_MULLO, unsigned) DEF_V(mul24, I_MUL24, unsigned) DEF_V(mad24, I_MAD24, unsigned)
DEF_V(add3, I_ADD3, unsigned) DEF_V(lshladd, I_LSHLADD, unsigned) DEF_V(bfe, I_BFE, unsigned)
DEF_V(dpp, I_DPP, unsigned) DEF_V(dppadd, I_DPPADD, unsigned) DEF_V(bcnt, I_BCNT, unsigned) DEF_V(mbcnt, I_MBCNT, unsigned)
DEF_V(rdlane, I_RDLANE, unsigned) DEF_V(rdlane_use, I_RDLANE_USE, unsigned) DEF_V(rdfirst, I_RDFIRST, unsigned) DEF_V(wrlane, I_WRLANE, unsigned)
DEF_S(s_add32, S_ADD32, unsigned) DEF_S(s_and64, S_AND64, unsigned long long) DEF_S(s_lshl64, S_LSHL64, unsigned long long)
DEF_S(s_bcnt64, S_BCNT64, unsigned long long) DEF_S(s_ff1_64, S_FF1_64, unsigned long long) DEF_S(s_mul32, S_MUL32, unsigned)
DEF_S(s_csel, S_CSEL, unsigned)

// conversions change the type: destination = accumulators (never read), source = the invariant operand
__global__ void __launch_bounds__(1024) cvt_f64_i32_thr(Res* out, int iters, int x, int)
{
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0; int xx = x + (int)threadIdx.x;
    T_BEGIN()
    for (int it = 0; it < iters; ++it)
        asm volatile(THR32(I_CVT_F64_I32) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(xx), "v"(xx));
    T_END()
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.0) out[blockIdx.x].cyc = 0;
}
__global__ void __launch_bounds__(1024) cvt_i32_f64_thr(Res* out, int iters, double x, double)
{
    int a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0; double xx = x + (double)threadIdx.x;
    T_BEGIN()
    for (int it = 0; it < iters; ++it)
        asm volatile(THR32(I_CVT_I32_F64) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(xx), "v"(xx));
    T_END()
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345) out[blockIdx.x].cyc = 0;
}
__global__ void __launch_bounds__(1024) cmpsel64_thr(Res* out, int iters, double x, double y)
{
    unsigned a0 = 1, a1 = 2, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = threadIdx.x, z = threadIdx.x;
    T_BEGIN()
    for (int it = 0; it < iters; ++it)
        asm volatile(THR32(I_CMPSEL64) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y), "v"(z) : "vcc");
    T_END()
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345u) out[blockIdx.x].cyc = 0;
}
// LDS: a dependent chain of ds_read_b32 (pointer chase) and of ds_bpermute_b32; 32 per iteration
__global__ void __launch_bounds__(1024) lds_chase_lat(Res* out, int iters, unsigned, unsigned)
{
    __shared__ unsigned tab[1024];
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)tab;     // entries hold LDS byte addresses
    tab[threadIdx.x] = base + ((threadIdx.x + 33) & 1023) * 4;
    unsigned a = base + threadIdx.x * 4;
    T_BEGIN()
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) asm volatile("ds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)\n" : "+v"(a));
    }
    T_END()
    if (a == 12345u) out[blockIdx.x].cyc = 0;
}
__global__ void __launch_bounds__(1024) bpermute_lat(Res* out, int iters, unsigned, unsigned)
{
    unsigned a = threadIdx.x, idx = ((threadIdx.x + 1) & 63) * 4;
    T_BEGIN()
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) asm volatile("ds_bpermute_b32 %0, %1, %0\ns_waitcnt lgkmcnt(0)\n" : "+v"(a) : "v"(idx));
    }
    T_END()
    if (a == 12345u) out[blockIdx.x].cyc = 0;
}
// a taken scalar branch every 4 VALU instructions (the env kernel: ~500 branches per 3 000 VALU)
__global__ void __launch_bounds__(1024) branch_thr(Res* out, int iters, unsigned x, unsigned)
{
    unsigned a0 = x + threadIdx.x, a1 = x;
    T_BEGIN()
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            asm volatile("v_add_u32 %0, %0, %1\nv_add_u32 %1, %1, %0\nv_add_u32 %0, %0, %1\nv_add_u32 %1, %1, %0\n"
                         "s_cmp_eq_u32 %2, 0x12345\ns_cbranch_scc0 1f\nv_add_u32 %0, %0, %0\n1:\n" : "+v"(a0), "+v"(a1) : "s"(iters) : "scc");
    }
    T_END()
    if (a0 + a1 == 12345u) out[blockIdx.x].cyc = 0;
}

// The env kernel's own instruction mix -- per 12 instructions: 6 VALU (4 float64 + 2 plain 32-bit), 3 SALU, 1 branch (not taken), 1 LDS
// read, 1 s_waitcnt -- issued by the four waves of a SIMD IN LOCKSTEP (one launch per step: all four run the same stage at the same
// time), and with the waves started a third of the block apart.  How busy can the vector unit get with this mix?
template <int DEPHASE>
__global__ void __launch_bounds__(1024) kernel_mix(Res* out, int iters, double x, double y)
{
    __shared__ double buf[1024];
    double a0 = x, a1 = x + threadIdx.x, a2 = x, a3 = x; unsigned b0 = threadIdx.x, b1 = 3; double l0 = 0.0;
    buf[threadIdx.x] = x;
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) double*)&buf[threadIdx.x];
    T_BEGIN()
    if (DEPHASE) {       // wave slot k of the SIMD idles k x ~1/4 of a block's issue time before it starts
        const int k = (threadIdx.x >> 6) >> 2;
        for (int w = 0; w < k * 6; ++w) asm volatile("s_nop 7\ns_nop 7\ns_nop 7\ns_nop 7\n");
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            asm volatile("v_fma_f64 %0, %0, %7, %8\n s_add_u32 s20, s20, 1\n v_mul_f64 %1, %1, %7\n v_add_u32 %4, %4, %5\n"
                         "s_and_b64 s[22:23], s[22:23], exec\n v_add_f64 %2, %2, %8\n ds_read_b64 %6, %9\n v_fma_f64 %3, %3, %7, %8\n"
                         "s_cmp_eq_u32 s20, 0x7fffffff\n s_cbranch_scc1 1f\n v_and_b32 %5, %5, %4\n s_waitcnt lgkmcnt(0)\n1:\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(l0) : "v"(x), "v"(y), "v"(la) : "scc", "s20", "s22", "s23", "memory");
    }
    T_END()
    if (a0 + a1 + a2 + a3 + l0 == 12345.0 && b0 + b1 == 7u) out[blockIdx.x].cyc = 0;
}

template <typename K, typename T>
static int run(const char* name, K kern, int per_iter, T x, T y, Res* d, double* out4)
{
    const int iters = 2000;
    Res h[256];
    double res[2];
    for (int occ = 0; occ < 2; ++occ) {
        const int threads = occ ? 1024 : 256;          // 1 or 4 waves per SIMD (one workgroup per CU)
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, iters, x, y);
            CHK(hipDeviceSynchronize());
        }
        CHK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i].cyc;
        const double per_wave = s / 256.0 / ((double)iters * per_iter);           // wave 0's cycles per instruction it issued
        res[occ] = occ ? per_wave / 4.0 : per_wave;                               // 4 waves share the SIMD: per instruction per SIMD
    }
    printf("%-22s %8.2f %12.2f\n", name, res[0], res[1]);
    if (out4) *out4 = res[1];
    return 0;
}

int main()
{
    Res* d; CHK(hipMalloc(&d, 256 * sizeof(Res)));
    printf("cycles (s_memtime) per wave instruction; 32-instruction blocks x 2000, one workgroup per CU on 256 CUs\n");
    printf("%-22s %8s %12s\n", "class", "1 wave", "4 waves/SIMD");
    printf("%-22s %8s %12s\n", "", "/SIMD", "(per instr per SIMD)");
#define RUNV(NAME, N, X, Y) do { if (run(#NAME " indep", NAME##_thr, N, X, Y, d, nullptr)) return 1; if (run(#NAME " dep", NAME##_lat, N, X, Y, d, nullptr)) return 1; } while (0)
    RUNV(fma64, 32, 1.0000001, 1e-9); RUNV(mul64, 32, 1.0000001, 0.0); RUNV(add64, 32, 1e-9, 0.0); RUNV(max64, 32, 0.5, 0.0);
    RUNV(rcp64, 32, 1.7, 0.0); RUNV(sqrt64, 32, 1.7, 0.0); RUNV(rndne64, 32, 1.7, 0.0); RUNV(ldexp64, 32, 1e-300, 0.0);
    RUNV(cmp64, 32, 1.0, 2.0); RUNV(divscale64, 32, 1.5, 2.5); RUNV(divfix64, 32, 1.5, 2.5);
    if (run("cmp64 + 2 cndmask indep", cmpsel64_thr, 96, 1.0, 2.0, d, nullptr)) return 1;
    RUNV(lshl64, 32, 1ull, 0ull);
    if (run("cvt_f64_i32 indep", cvt_f64_i32_thr, 32, 3, 0, d, nullptr)) return 1;
    if (run("cvt_i32_f64 indep", cvt_i32_f64_thr, 32, 3.0, 0.0, d, nullptr)) return 1;
    RUNV(fma32, 32, 1.0001f, 1e-6f); RUNV(pkfma32, 32, 1.0, 0.0);
    RUNV(addu32, 32, 3u, 0u); RUNV(mov32, 32, 3u, 0u); RUNV(and32, 32, 0xffffu, 0u); RUNV(cndmask, 32, 3u, 0u); RUNV(cndmask_ev, 32, 3u, 0u); RUNV(cndmask_sv, 64, 3u, 0u); RUNV(cndmask_vv, 64, 3, 0); RUNV(mix_vs, 64, 1.0000001, 1e-9); RUNV(mix_v32s, 64, 3u, 0u);
    RUNV(cndmask_s, 32, 3u, 0u); RUNV(bfi, 32, 3u, 5u); RUNV(andor, 32, 3u, 5u); RUNV(med3i, 32, 3, 5); RUNV(mini32, 32, 3, 0); RUNV(ashr, 32, 3, 0);
    RUNV(cmpsw, 32, 3, 0); RUNV(cmpi32, 32, 3, 0);
    RUNV(mullo, 32, 3u, 0u); RUNV(mul24, 32, 3u, 0u); RUNV(mad24, 32, 3u, 1u); RUNV(add3, 32, 3u, 1u); RUNV(lshladd, 32, 3u, 1u); RUNV(bfe, 32, 0xffffffu, 0u);
    RUNV(dpp, 32, 3u, 0u); RUNV(dppadd, 32, 3u, 0u); RUNV(bcnt, 32, 3u, 0u); RUNV(mbcnt, 32, 3u, 0u);
    RUNV(rdlane, 32, 3u, 0u); RUNV(rdlane_use, 64, 3u, 0u); RUNV(rdfirst, 32, 3u, 0u); RUNV(wrlane, 32, 3u, 0u);
    RUNV(s_add32, 32, 3u, 0u); RUNV(s_and64, 32, ~0ull, 0ull); RUNV(s_lshl64, 32, 1ull, 0ull); RUNV(s_bcnt64, 32, 7ull, 0ull);
    RUNV(s_ff1_64, 32, 8ull, 0ull); RUNV(s_mul32, 32, 3u, 0u); RUNV(s_csel, 64, 3u, 0u);
    if (run("ds_read_b32 chase dep", lds_chase_lat, 32, 0u, 0u, d, nullptr)) return 1;
    if (run("ds_bpermute dep", bpermute_lat, 32, 0u, 0u, d, nullptr)) return 1;
    if (run("4 v_add + taken branch", branch_thr, 8 * 6, 3u, 0u, d, nullptr)) return 1;
    {   // the mix: 48 instructions per iteration (4 x 12), of which 24 VALU (16 float64 at 4.2 cycles + 8 plain at 2.4 = 86 cycles per wave)
        double m4[2];
        if (run("kernel mix, lockstep", kernel_mix<0>, 48, 1.0000001, 1e-9, d, &m4[0])) return 1;
        if (run("kernel mix, de-phased", kernel_mix<1>, 48, 1.0000001, 1e-9, d, &m4[1])) return 1;
        for (int k = 0; k < 2; ++k)
            printf("  %s: %.1f cycles per 12-instruction group per SIMD at 4 waves -> the vector unit is busy %.0f %% of the time "
                   "(4 waves x (4 x 4.2 + 2 x 2.4) cycles of VALU issue per group)\n", k ? "de-phased" : "lockstep ", m4[k] * 12.0 * 4.0,
                   100.0 * 4.0 * (4 * 4.2 + 2 * 2.4) / (m4[k] * 12.0 * 4.0));
    }
    printf("(the cmp64 + 2 cndmask / rdlane_use / s_csel rows count each of their instructions separately; the branch row is per instruction of a "
           "{4 x v_add_u32, s_cmp, s_cbranch} group)\n");
    return 0;
}
