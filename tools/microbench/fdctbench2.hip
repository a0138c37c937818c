// Round 3: does a third (fourth) wavefront per SIMD buy VALU throughput on the T-stage's instruction mix?
// The 64-point lifting network (the kernel's dominant work, same macros) back to back, occupancy set by the
// dynamic LDS request and VERIFIED with hipOccupancyMaxActiveBlocksPerMultiprocessor; the grid is exactly one
// round of resident wavefronts, so kernel wall time / repetitions is the time per transform at that occupancy.
// s_memtime is a constant 2.4 GHz counter on this chip (tools/microbench/clockcal), so ticks are time, not cycles.
// Build: hipcc --offload-arch=gfx950 -O3 -o fdctbench2 fdctbench2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include "../../ffmpeg_ffv2_amd/csrc/gen/fdct64_net.h"
__device__ __forceinline__ int rsh1(int a)
{
    int t;
    asm("v_sub_u32_sdwa %0, %1, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(t) : "v"(a));
    return t >> 1;
}
#define FFV2_RSH1(a)            rsh1(a)
#define FFV2_MULRS(a, K, R, S)  ((__mul24((a), (K)) + (R)) >> (S))
#define REPS 200
__global__ __launch_bounds__(64) void k(int *p, unsigned long long *cyc)
{
    extern __shared__ int dummy[];
    int x[64];
#pragma unroll
    for (int i = 0; i < 64; i++) x[i] = p[threadIdx.x + 64 * i];
    if (p[0] == 12345) dummy[threadIdx.x] = 1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REPS; r++) {
        FDCT64_NET(x);
#pragma unroll
        for (int i = 0; i < 64; i++) asm volatile("" : "+v"(x[i]));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 64; i++) s += x[i];
    p[blockIdx.x * 64 + threadIdx.x + 4096] = s;
    if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = t0; }
}
int main()
{
    int *d; unsigned long long *dc;
    (void)hipMalloc(&d, (4096 + 256 * 4 * 8 * 64) * sizeof(int)); (void)hipMemset(d, 1, 4096 * 4);
    (void)hipMalloc(&dc, 256 * 4 * 8 * 16);
    for (int w = 1; w <= 6; w++) {
        int lds = (160 * 1024 / (4 * w)) / 1024 * 1024 - 1024;
        if (w == 1) lds = 64 * 1024 - 1024;                      // one workgroup cannot ask for more than 64 KB: w = 1 is really 2 per CU pair
        (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int per_cu = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k, 64, lds);
        int blocks = 256 * per_cu;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        k<<<blocks, 64, lds>>>(d, dc);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(a);
        k<<<blocks, 64, lds>>>(d, dc);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h(2 * blocks);
        (void)hipMemcpy(h.data(), dc, blocks * 16, hipMemcpyDeviceToHost);
        std::vector<unsigned long long> life(blocks), start(blocks);
        for (int i = 0; i < blocks; i++) { life[i] = h[2 * i]; start[i] = h[2 * i + 1]; }
        std::sort(life.begin(), life.end()); std::sort(start.begin(), start.end());
        const double med = (double)life[blocks / 2];
        printf("LDS %6d B/wave -> %2d waves/CU (%.2f per SIMD): kernel %.3f ms; wave life median %.0f ticks (%.3f ms); starts spread %.3f ms;"
               "  per transform: %.0f ticks per wave, %.3f us per SIMD (from wall), %.3f us per SIMD (from wave life)\n",
               lds, per_cu, per_cu / 4.0, ms, med, med / 2.4e6, (double)(start[blocks - 1] - start[0]) / 2.4e6,
               med / REPS, ms * 1e3 / REPS / (per_cu / 4.0), med / 2.4e3 / REPS / (per_cu / 4.0));
    }
    return 0;
}
