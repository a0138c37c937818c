// Round 3: throughput of the lifting network against occupancy without inferring residency: a grid of many rounds
// (256 CUs x 96 single-wave workgroups) at LDS requests that admit 4 / 8 / 12 / 16 workgroups per CU; transforms per
// second = work / kernel wall time.  Also V4 of fdctbench3 (adds and shifts only) for contrast.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../ffmpeg_ffv2_amd/csrc/gen/fdct64_net.h"
__device__ __forceinline__ int rsh1(int a)
{
    int t;
    asm("v_sub_u32_sdwa %0, %1, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(t) : "v"(a));
    return t >> 1;
}
#define REPS 50
template <int V> __global__ __launch_bounds__(64) void k(int *p)
{
    extern __shared__ int dummy[];
    int x[64];
#pragma unroll
    for (int i = 0; i < 64; i++) x[i] = p[threadIdx.x + 64 * i];
    if (p[0] == 12345) dummy[threadIdx.x] = 1;
#pragma unroll 1
    for (int r = 0; r < REPS; r++) {
#define FFV2_RSH1(a)            (V == 4 ? ((a) >> 1) : rsh1(a))
#define FFV2_MULRS(a, K, R, S)  (V == 4 ? ((a) + (S)) : ((__mul24((a), (K)) + (R)) >> (S)))
        FDCT64_NET(x);
#undef FFV2_RSH1
#undef FFV2_MULRS
#pragma unroll
        for (int i = 0; i < 64; i++) asm volatile("" : "+v"(x[i]));
    }
    int s = 0;
#pragma unroll
    for (int i = 0; i < 64; i++) s += x[i];
    if (s == 0x7fffffff) p[4096 + threadIdx.x] = s;
}
template <int V> void run(const char *what, int *d)
{
    const int per[4] = { 4, 8, 12, 16 };
    for (int i = 0; i < 4; i++) {
        int lds = (160 * 1024 / per[i]) / 512 * 512 - 1536;
        if (lds > 64 * 1024 - 512) lds = 64 * 1024 - 512;
        (void)hipFuncSetAttribute((const void *)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int occ = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k<V>, 64, lds);
        const int blocks = 256 * 96;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        k<V><<<blocks, 64, lds>>>(d);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        k<V><<<blocks, 64, lds>>>(d);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-26s LDS %5d B -> %2d workgroups per CU (API): %8.3f ms for %d wave-transforms = %.3f us per transform and SIMD\n",
               what, lds, occ, ms, blocks * REPS, ms * 1e3 / (blocks * REPS / 1024.0));
    }
}
int main()
{
    int *d;
    (void)hipMalloc(&d, (4096 + 64) * sizeof(int)); (void)hipMemset(d, 1, 4096 * 4);
    run<0>("network as shipped", d);
    run<4>("adds and shifts only", d);
    return 0;
}
