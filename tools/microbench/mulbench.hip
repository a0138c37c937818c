// VALU throughput on gfx950 in SIMD cycles per wave64 instruction, measured with
// s_memtime inside the kernel (independent of the clock the chip happens to hold).
// Build: hipcc --offload-arch=gfx950 -O3 -o mulbench mulbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CHAINS 8
#define ITERS 20000
template <int OP> __global__ void k(int *out, unsigned long long *cyc, int seed)
{
    int v[CHAINS];
    for (int i = 0; i < CHAINS; i++) v[i] = seed + threadIdx.x + i;
    int kk = seed | 13573, rr = seed | 16384;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (OP == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[i]) : "v"(kk));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(kk));
            if (OP == 2) asm volatile("v_mul_i32_i24 %0, %1, %0" : "+v"(v[i]) : "v"(kk));
            if (OP == 3) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(v[i]) : "s"(kk), "v"(rr));
            if (OP == 4) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(v[i]));
            if (OP == 5) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(*(long long *)&v[i & ~1]) : "v"(kk) : "vcc");
            if (OP == 6) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[i]) : "v"(kk));
            if (OP == 7) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(kk), "v"(rr));
            if (OP == 8) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(v[i]) : "v"(kk));
            if (OP == 9) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(kk), "v"(rr));
            if (OP == 10) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(v[i]) : "v"(kk));
            if (OP == 11) asm volatile("v_mul_hi_i32_i24 %0, %1, %0" : "+v"(v[i]) : "v"(kk));
            if (OP == 12) asm volatile("v_mad_i32_i16 %0, %0, %1, %2" : "+v"(v[i]) : "v"(kk), "v"(rr));
            if (OP == 13) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(long long *)&v[i & ~1]) : "v"(*(long long *)&v[(i & ~1)]));
            if (OP == 14) asm volatile("v_bfe_i32 %0, %0, 3, 16" : "+v"(v[i]));
            if (OP == 15) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(kk), "v"(rr));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < CHAINS; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
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
    // one wave issues ITERS*CHAINS instructions in `med` cycles; wavesPerSimd waves share a SIMD
    printf("%-16s waves/SIMD=%d  median wave cycles %.0f  -> %.2f SIMD cycles per wave64 instruction\n", name,
           wavesPerSimd, med, med / ((double)ITERS * CHAINS * wavesPerSimd));
}
int main()
{
    int *d; unsigned long long *dc;
    hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(int)); hipMalloc(&dc, 256 * 4 * 8 * 8);
    for (int w = 1; w <= 8; w *= 2) {
        run<0>("v_add_u32", d, dc, w); run<6>("v_sub_u32", d, dc, w); run<4>("v_ashrrev_i32", d, dc, w);
        run<3>("v_mad_i32_i24", d, dc, w); run<2>("v_mul_i32_i24", d, dc, w); run<1>("v_mul_lo_u32", d, dc, w);
        run<5>("v_mad_u64_u32", d, dc, w); run<7>("v_add3_u32", d, dc, w); run<8>("v_lshl_add_u32", d, dc, w);
        run<9>("v_fma_f32", d, dc, w); run<10>("v_pk_add_u16", d, dc, w); run<11>("v_mul_hi_i32_i24", d, dc, w);
        run<12>("v_mad_i32_i16", d, dc, w); run<13>("v_pk_fma_f32", d, dc, w); run<14>("v_bfe_i32", d, dc, w);
        run<15>("v_and_or_b32", d, dc, w);
    }
    return 0;
}
