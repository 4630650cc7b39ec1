// What does ONE launch cost in a chain of dependent launches on MI355X?  (the "launch floor".)
// A chain of N kernels that each do nothing but a store (same stream -> each waits for the previous one), timed with HIP events
// and with the host clock: as plain launches, as one hipGraph of kernel nodes, for grids of 1 / 256 / 4096 workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/launch_floor tools/micro/launch_floor.hip && tools/micro/bin/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tick(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
// a kernel that lasts `ticks` of s_memtime (100 MHz: 10 ns each) in every wave: the host runs far ahead of such a chain, so what is left
// per launch beyond the kernel's own duration is the device-side cost of a dependent launch
__global__ void spin(int* p, long long ticks)
{
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
int main()
{
    int* d; CHK(hipMalloc(&d, 4)); CHK(hipMemset(d, 0, 4));
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int N = 2000;
    const int grids[3] = {1, 256, 4096}, blocks[3] = {64, 256, 64};
    for (int g = 0; g < 3; ++g) {
        for (int rep = 0; rep < 2; ++rep) {            // rep 0 warms up
            CHK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            CHK(hipEventRecord(e0, st));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tick, dim3(grids[g]), dim3(blocks[g]), 0, st, d);
            CHK(hipEventRecord(e1, st));
            CHK(hipStreamSynchronize(st));
            double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("stream chain  grid %4d x %3d: %.2f us per launch on the device (events), %.2f us wall\n", grids[g], blocks[g], ms * 1e3 / N, wall / N);
        }
        // the same chain as a hipGraph (captured), launched once
        hipGraph_t graph; hipGraphExec_t exec;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tick, dim3(grids[g]), dim3(blocks[g]), 0, st, d);
        CHK(hipStreamEndCapture(st, &graph));
        CHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            CHK(hipStreamSynchronize(st));
            CHK(hipEventRecord(e0, st));
            CHK(hipGraphLaunch(exec, st));
            CHK(hipEventRecord(e1, st));
            CHK(hipStreamSynchronize(st));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("hipGraph      grid %4d x %3d: %.2f us per kernel node\n", grids[g], blocks[g], ms * 1e3 / N);
        }
        CHK(hipGraphExecDestroy(exec)); CHK(hipGraphDestroy(graph));
    }
    {   // kernels that last: 4096 x 64 (the env kernel's grid), chains of 500, two spin lengths -> the fixed cost per dependent launch
        // = the intercept of (time per launch) over (spin ticks), for plain stream launches and for a graph
        const int M = 500; const long long T1 = 20000, T2 = 60000;
        for (int mode = 0; mode < 2; ++mode) {
            double per[2] = {0, 0};
            for (int k = 0; k < 2; ++k) {
                const long long ticks = k ? T2 : T1;
                hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
                if (mode) {
                    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
                    for (int i = 0; i < M; ++i) hipLaunchKernelGGL(spin, dim3(4096), dim3(64), 0, st, d, ticks);
                    CHK(hipStreamEndCapture(st, &graph));
                    CHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                }
                for (int rep = 0; rep < 2; ++rep) {
                    CHK(hipStreamSynchronize(st));
                    CHK(hipEventRecord(e0, st));
                    if (mode) CHK(hipGraphLaunch(exec, st));
                    else for (int i = 0; i < M; ++i) hipLaunchKernelGGL(spin, dim3(4096), dim3(64), 0, st, d, ticks);
                    CHK(hipEventRecord(e1, st));
                    CHK(hipStreamSynchronize(st));
                    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                    per[k] = ms * 1e3 / M;
                }
                if (mode) { CHK(hipGraphExecDestroy(exec)); CHK(hipGraphDestroy(graph)); }
            }
            const double us_per_tick = (per[1] - per[0]) / (double)(T2 - T1);
            printf("%s of spinning kernels (4096 x 64): %.2f / %.2f us per launch at %lld / %lld ticks (%.4f us per tick) -> %.2f us per launch beyond the spin\n",
                   mode ? "hipGraph     " : "stream chain ", per[0], per[1], T1, T2, us_per_tick, per[0] - us_per_tick * (double)T1);
        }
    }
    int h = 0; CHK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    printf("kernels run: %d\n", h);
    return 0;
}
