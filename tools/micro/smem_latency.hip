// What does a scalar load cost a wave that needs its result at once?  (The env kernels read their ~130 dwords of parameters in place from
// the kernarg segment, next to their uses: 78 s_load per env-step, most of them followed by s_waitcnt lgkmcnt(0).)
//   chain : s_load_dword -> s_waitcnt lgkmcnt(0) -> the loaded value is the next load's offset           (pure latency, scalar cache hits)
//   spaced: the same load with N independent v_fma_f64 between issue and wait                                (how much of it can be covered)
// 1 wave per SIMD and 4 waves per SIMD (the env kernel's occupancy); cycles = s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/smem_latency tools/micro/smem_latency.hip && tools/micro/bin/smem_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int FILL>
__global__ void __launch_bounds__(1024) smem_kernel(const int* __restrict__ tab, int iters, long long* out, double* sink)
{
    // tab[i] = byte offset of the next entry (a ring inside 256 bytes: always a scalar-cache hit after the first touch)
    unsigned off = 0;
    double a = threadIdx.x, b = 1.0000001, c = 1e-9;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            unsigned nxt;
            asm volatile("s_load_dword %0, %1, %2" : "=s"(nxt) : "s"(tab), "s"(off));
            if constexpr (FILL > 0) {
#pragma unroll
                for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(nxt));
            off = nxt;
        }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * 1024 + threadIdx.x] = a + (double)off;
}

template <int FILL>
static int run(const int* tab, long long* out, double* sink)
{
    const int iters = 2000, blocks = 256;
    static long long h[256 * 16];
    for (int wps = 1; wps <= 4; wps *= 4) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(smem_kernel<FILL>, dim3(blocks), dim3(256 * wps), 0, 0, tab, iters, out, sink);
            CHK(hipDeviceSynchronize());
        }
        CHK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        double s = 0; int n = 0;
        for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4 * wps; ++w) { s += (double)h[b * 16 + w]; ++n; }
        printf("  %2d dependent v_fma_f64 between issue and wait, %d wave(s) per SIMD: %7.1f cycles per load (+ fill)\n", FILL, wps, s / n / (8.0 * iters));
    }
    return 0;
}

int main()
{
    int htab[64];
    for (int i = 0; i < 64; ++i) htab[i] = ((i * 17 + 5) & 63) * 4;
    int* tab; long long* out; double* sink;
    CHK(hipMalloc(&tab, sizeof(htab))); CHK(hipMemcpy(tab, htab, sizeof(htab), hipMemcpyHostToDevice));
    CHK(hipMalloc(&out, 256 * 16 * 8)); CHK(hipMalloc(&sink, 256 * 1024 * 8));
    printf("s_load_dword -> s_waitcnt lgkmcnt(0) -> use, scalar-cache hits (a 256-byte ring)\n");
    if (run<0>(tab, out, sink)) return 1;
    if (run<4>(tab, out, sink)) return 1;
    if (run<16>(tab, out, sink)) return 1;
    if (run<32>(tab, out, sink)) return 1;
    return 0;
}
