// microbenchmark: does a wave64 f64 FMA with a partial exec mask issue faster?
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ACTIVE>
__global__ void k(double* out, int iters, double a, double b)
{
    int lane = threadIdx.x & 63;
    double x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
    if (lane < ACTIVE) {
        for (int i = 0; i < iters; ++i) {   // 4 independent chains -> issue bound, not latency bound
            x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
template <int ACTIVE> float run(double* d, int blocks, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<ACTIVE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<ACTIVE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    double* d; hipMalloc(&d, 256 * 16 * 256 * 8);
    int blocks = 256 * 16, iters = 20000;   // 16 waves per SIMD-set: issue bound
    printf("active 64: %.3f ms\n", run<64>(d, blocks, iters));
    printf("active 32: %.3f ms\n", run<32>(d, blocks, iters));
    printf("active 16: %.3f ms\n", run<16>(d, blocks, iters));
    printf("active  8: %.3f ms\n", run<8>(d, blocks, iters));
    printf("active  1: %.3f ms\n", run<1>(d, blocks, iters));
    return 0;
}
