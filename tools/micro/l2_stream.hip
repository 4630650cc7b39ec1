// How fast can every CU of an MI355X stream the SAME buffer out of L2?  (the access pattern of cn_actor_kernel's weights:
// 256 workgroups x 8 waves each read all of a 688 KB array that no other wave of the workgroup reads again)
//   hipcc --offload-arch=gfx950 -O3 -o l2_stream tools/micro/l2_stream.hip && ./l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename V>
__global__ void __launch_bounds__(512) stream_kernel(const V* __restrict__ w, size_t n_vec, int reps, float* out)
{
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        const V* p = w + ((r & 1) ? 0 : 0);
#pragma unroll 8
        for (size_t i = threadIdx.x; i < n_vec; i += 512) {
            V v = __builtin_nontemporal_load(p + i);
            if constexpr (sizeof(V) == 8) acc += v.x + v.y; else if constexpr (sizeof(V) == 16) acc += v.x + v.y + v.z + v.w; else acc += v;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
template <typename V>
__global__ void __launch_bounds__(512) stream_kernel_plain(const V* __restrict__ w, size_t n_vec, int reps, float* out)
{
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
#pragma unroll 8
        for (size_t i = threadIdx.x; i < n_vec; i += 512) {
            V v = w[i];
            if constexpr (sizeof(V) == 8) acc += v.x + v.y; else if constexpr (sizeof(V) == 16) acc += v.x + v.y + v.z + v.w; else acc += v;
        }
        asm volatile("" ::: "memory");
    }
    if (acc == 123.456f) out[0] = acc;
}
int main()
{
    const size_t bytes = 688 * 1024;
    float* w; float* out;
    hipMalloc(&w, bytes); hipMalloc(&out, 4); hipMemset(w, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    for (int grid : {32, 64, 128, 256, 512}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int it = 0; it < 5; ++it) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(stream_kernel_plain<float2>, dim3(grid), dim3(512), 0, 0, (const float2*)w, bytes / 8, reps, out);
                if (mode == 1) hipLaunchKernelGGL(stream_kernel_plain<float4>, dim3(grid), dim3(512), 0, 0, (const float4*)w, bytes / 16, reps, out);
                if (mode == 2) hipLaunchKernelGGL(stream_kernel_plain<float>, dim3(grid), dim3(512), 0, 0, (const float*)w, bytes / 4, reps, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double tot = (double)bytes * reps * grid;
            printf("grid %4d  %s  %.3f ms  %.2f TB/s aggregate  %.1f GB/s per workgroup\n", grid, mode == 0 ? "dwordx2" : mode == 1 ? "dwordx4" : "dword  ",
                   best, tot / best / 1e9, tot / grid / best / 1e6);
        }
    }
    return 0;
}
