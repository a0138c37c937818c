// Round 3: WHAT bounds the lifting network at ~1.87 us per wave-transform and SIMD whatever the occupancy?
// The same network with parts of its per-op cost taken away (results are wrong on purpose; only time matters):
//   V0 the kernel's form (v_mad_i32_i24 with the multiplier in an SGPR set by s_movk, shift, add; SDWA OD_RSHIFT1)
//   V1 every multiplier the same runtime SGPR (no s_movk, no fresh SGPR per multiply)
//   V2 multiply-shift replaced by one add (no mad, no shift)
//   V3 OD_RSHIFT1 as a plain shift (no SDWA)
//   V4 V2 + V3: adds and shifts only
// 8 wavefronts per CU (2 per SIMD), one round; s_memtime is a constant 2.4 GHz counter.
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
#define REPS 200
template <int V> __global__ __launch_bounds__(64) void k(int *p, unsigned long long *cyc, int kk)
{
    extern __shared__ int dummy[];
    int x[64];
#pragma unroll
    for (int i = 0; i < 64; i++) x[i] = p[threadIdx.x + 64 * i];
    if (p[0] == 12345) dummy[threadIdx.x] = 1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REPS; r++) {
#define FFV2_RSH1(a)            ((V == 3 || V == 4) ? ((a) >> 1) : rsh1(a))
#define FFV2_MULRS(a, K, R, S)  (V == 1 ? ((__mul24((a), kk) + (R)) >> (S)) : (V == 2 || V == 4) ? ((a) + (S)) : ((__mul24((a), (K)) + (R)) >> (S)))
        FDCT64_NET(x);
#undef FFV2_RSH1
#undef FFV2_MULRS
#pragma unroll
        for (int i = 0; i < 64; i++) asm volatile("" : "+v"(x[i]));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 64; i++) s += x[i];
    p[blockIdx.x * 64 + threadIdx.x + 4096] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int V> void run(const char *what, int *d, unsigned long long *dc)
{
    const int lds = 19456, blocks = 256 * 8;
    (void)hipFuncSetAttribute((const void *)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k<V><<<blocks, 64, lds>>>(d, dc, 77);
    (void)hipDeviceSynchronize();
    k<V><<<blocks, 64, lds>>>(d, dc, 77);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[blocks / 2];
    printf("%-58s %6.0f ticks per transform and wave = %.3f us per transform and SIMD (2 waves)\n", what, med / REPS, med / REPS / 2 / 2.4e3);
}
int main()
{
    int *d; unsigned long long *dc;
    (void)hipMalloc(&d, (4096 + 256 * 4 * 8 * 64) * sizeof(int)); (void)hipMemset(d, 1, 4096 * 4);
    (void)hipMalloc(&dc, 256 * 4 * 8 * 16);
    run<0>("V0 as shipped (mad_i24 + shift + add, SDWA rsh1)", d, dc);
    run<1>("V1 one runtime multiplier for all (no s_movk)", d, dc);
    run<2>("V2 multiply-shift -> one add", d, dc);
    run<3>("V3 rsh1 -> plain shift", d, dc);
    run<4>("V4 adds and shifts only", d, dc);
    return 0;
}
