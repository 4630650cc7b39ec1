// tools/profc/profc_block.h -- injected with `-include` into translation unit 1 of crowdnav_kernel.hip by tools/profc/build.sh
// (REGION-COUNTER BUILD ONLY -> lib/prof/libcrowdnav_profc.so; never part of the product or of csrc/build.sh).
//
// The device code of that unit is compiled with clang's source-region counters (-fprofile-instr-generate -fprofile-update=atomic
// -fcoverage-mapping) and every counter update in the optimised IR is rewritten (tools/profc/rewrite_ir.py) into a call of
// cn_prof_hit(), which adds 1 per WAVEFRONT (high word) and the number of active lanes (low word): a dynamic execution profile of
// the step kernel per source region.  tools/profc/report.py joins it with the product's own instruction listing (the counted build's
// instructions differ, its control flow does not).  cn_profc_kernel copies the counter section out; its own counter symbol is the
// anchor the section's base address is derived from.
#pragma once
#include <hip/hip_runtime.h>

extern "C" __device__ __attribute__((used, noinline, no_profile_instrument_function)) void cn_prof_hit(unsigned long long* ctr, unsigned long long n)
{
    const unsigned long long m = __builtin_amdgcn_ballot_w64(true);
    if ((int)__lane_id() == __builtin_ctzll(m)) atomicAdd(ctr, (n << 32) | (n * (unsigned long long)__popcll(m)));
}

extern "C" __global__ void cn_profc_kernel(unsigned long long* out, long long before_bytes, long long n, int zero)
{
    unsigned lo, hi;
    asm volatile("s_getpc_b64 s[20:21]\n\ts_add_u32 s20, s20, __profc_cn_profc_kernel@rel32@lo+4\n\t"
                 "s_addc_u32 s21, s21, __profc_cn_profc_kernel@rel32@hi+12\n\ts_mov_b32 %0, s20\n\ts_mov_b32 %1, s21"
                 : "=s"(lo), "=s"(hi) : : "s20", "s21", "scc");
    unsigned long long* const base = (unsigned long long*)(((((unsigned long long)hi) << 32) | (unsigned long long)lo) - (unsigned long long)before_bytes);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        out[i] = base[i];
        if (zero) base[i] = 0ull;
    }
}

extern "C" int cn_debug_profc(void* out_dev, long long before_bytes, long long n, int zero, void* stream)
{
    hipLaunchKernelGGL(cn_profc_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, (unsigned long long*)out_dev, before_bytes, n, zero);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
