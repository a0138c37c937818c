// Round 3: issue cost of the packed-fp32 instructions (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32, two fp32
// results per lane) beside the integer instructions the lapping filter uses today, in s_memtime ticks per
// wave64 instruction at 1, 2 and 3 wavefronts per SIMD.  Question: can the lapping pre-filter, whose values
// fit 24 bits, run two filter instances per lane in packed fp32 (exact integer arithmetic, floor by the
// round-down mode + 1.5 * 2^23 trick) for fewer issue slots than mad_i24 + ashr + add?
// Build: hipcc --offload-arch=gfx950 -O3 -o opbench3 opbench3.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CHAINS 8
#define ITERS 20000
template <int OP> __global__ void k(int *out, unsigned long long *cyc, int seed)
{
    long long v[CHAINS];
    for (int i = 0; i < CHAINS; i++) v[i] = seed + threadIdx.x + i;
    long long kk = (seed | 13573) * 0x100000001ll, rr = (seed | 16384) * 0x100000001ll;
    int k1 = seed | 77;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            int &lo = *(int *)&v[i];
            if (OP == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(lo) : "v"(k1));
            if (OP == 1) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(lo) : "s"(k1), "v"(k1));
            if (OP == 2) asm volatile("v_ashrrev_i32 %0, 6, %0" : "+v"(lo));
            if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(kk), "v"(rr));
            if (OP == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(kk));
            if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(kk));
            if (OP == 6) asm volatile("v_pk_add_f32 %0, %0, %1 clamp" : "+v"(v[i]) : "v"(kk));
            if (OP == 7) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(lo));
            if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(lo) : "v"(k1), "v"(k1));
            if (OP == 9) asm volatile("v_add_f32 %0, %1, %0" : "+v"(lo) : "v"(k1));
            if (OP == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "s"(kk), "v"(rr));
            if (OP == 11) asm volatile("v_pk_mov_b32 %0, %0, %1" : "+v"(v[i]) : "v"(kk));
            if (OP == 12) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(k1), "v"(k1));
            if (OP == 13) asm volatile("v_floor_f32 %0, %0" : "+v"(lo));
            if (OP == 14) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(lo) : "v"(k1));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    long long s = 0;
    for (int i = 0; i < CHAINS; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)(s ^ (s >> 32));
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char *name, int *d, unsigned long long *dc, int wavesPerSimd)
{
    int blocks = 256 * 4 * wavesPerSimd;   // 64-thread blocks: one wave each
    k<OP><<<blocks, 64>>>(d, dc, 1);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    double med = (double)h[blocks / 2];
    printf("%-26s waves/SIMD=%d  median wave ticks %.0f  -> %.2f ticks per wave64 instruction (SIMD share)\n", name,
           wavesPerSimd, med, med / ((double)ITERS * CHAINS * wavesPerSimd));
}
int main()
{
    int *d; unsigned long long *dc;
    hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(int)); hipMalloc(&dc, 256 * 4 * 8 * 8);
    for (int w = 1; w <= 3; w++) {
        run<0>("v_add_u32", d, dc, w);
        run<14>("v_sub_u32", d, dc, w);
        run<1>("v_mad_i32_i24", d, dc, w);
        run<2>("v_ashrrev_i32", d, dc, w);
        run<8>("v_fma_f32", d, dc, w);
        run<9>("v_add_f32", d, dc, w);
        run<13>("v_floor_f32", d, dc, w);
        run<3>("v_pk_fma_f32", d, dc, w);
        run<10>("v_pk_fma_f32 (sgpr pair)", d, dc, w);
        run<4>("v_pk_add_f32", d, dc, w);
        run<5>("v_pk_mul_f32", d, dc, w);
        run<6>("v_pk_add_f32 clamp", d, dc, w);
        run<11>("v_pk_mov_b32", d, dc, w);
        run<7>("v_cvt_f32_i32_sdwa w1", d, dc, w);
        run<12>("v_perm_b32", d, dc, w);
        printf("\n");
    }
    return 0;
}
