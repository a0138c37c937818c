// Cost of the generated 64-point lifting network per wave at different occupancies (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include "../../ffmpeg_ffv2_amd/csrc/gen/fdct64_net.h"
#include "fdct_ktab.h"
__device__ __forceinline__ int rsh1s(int a) { int t; asm("v_sub_u32_sdwa %0, %1, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(t) : "v"(a)); return t >> 1; }
#define FFV2_RSH1(a) rsh1s(a)
typedef int v16i __attribute__((ext_vector_type(16)));
#define CAS __attribute__((address_space(4)))
struct KTab { v16i c[13]; const v16i CAS *p; };
template <int K, int R, int S, int I> __device__ __forceinline__ int mulrs(int a, KTab &t)
{
    if constexpr (I % 16 == 0 && I / 16 + 1 < 13) {
        { int first = t.c[I / 16][0]; asm volatile("; forcewait %1" :: "s"(first), "n"(I), "v"(a)); }
        const v16i CAS *q = t.p + (I / 16 + 1);
        asm volatile("; tie %2" : "+s"(q) : "v"(a), "n"(I));
        t.c[I / 16 + 1] = *q;
        __builtin_amdgcn_sched_barrier(0);
    }
    const int kt = t.c[I / 16][I % 16];
    if (K < (1 << (S - 1))) return (int)(((long long)a * kt + 0x80000000LL) >> 32);
    int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(kt), "v"(R));
    return r >> S;
}
#define FFV2_MULRS(a, K, R, S) mulrs<K, R, S, __COUNTER__ - CBASE>(a, kt_)
#define REPS 100
enum { CBASE = __COUNTER__ + 1 };
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
        KTab kt_;
        kt_.p = (const v16i CAS *)KT_COL;
        asm volatile("" : "+s"(kt_.p));
        kt_.c[0] = kt_.p[0];
        FDCT64_NET(x);
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
int main()
{
    int *d; unsigned long long *dc;
    hipMalloc(&d, (4096 + 256 * 4 * 8 * 64) * sizeof(int)); hipMemset(d, 1, 4096 * 4);
    hipMalloc(&dc, 256 * 4 * 8 * 8);
    for (int w = 1; w <= 4; w++) {
        int wavesPerCU = 4 * w;
        size_t lds = 160 * 1024 / wavesPerCU - 256;            // forces at most wavesPerCU blocks per CU
        if (w == 4) lds = 160 * 1024 / 16 - 256;
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int blocks = 256 * wavesPerCU;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        k<<<blocks, 64, lds>>>(d, dc);
        hipDeviceSynchronize();
        hipEventRecord(a);
        k<<<blocks, 64, lds>>>(d, dc);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        double med = (double)h[blocks / 2];
        printf("waves/SIMD=%d  kernel %.3f ms  median wave ticks %.0f  ticks per FDCT per wave %.0f  "
               "SIMD ticks per FDCT %.0f   wall us per FDCT per SIMD %.3f\n",
               w, ms, med, med / REPS, med / REPS / w, ms * 1e3 / REPS / w);
    }
    return 0;
}
