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
            if (OP == 1) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(v[i]) : "s"(kk));
            if (OP == 2) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(v[i]));
            if (OP == 3) asm volatile("v_sub_u32_sdwa %0, %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "+v"(v[i]));
            if (OP == 4) asm volatile("v_sub_u32_sdwa %0, sext(%0), sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(v[i]) : "v"(kk));
            if (OP == 5) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(v[i]) : "v"(kk));
            if (OP == 6) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
            if (OP == 7) asm volatile("v_cmp_lt_i32 vcc, 0, %0\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(v[i]) :: "vcc");
            if (OP == 8) asm volatile("v_cmp_lt_i32_e64 s[20:21], 0, %0\n v_addc_co_u32_e64 %0, s[22:23], 0, %0, s[20:21]" : "+v"(v[i]) :: "s20","s21","s22","s23");
            if (OP == 9) asm volatile("v_cmp_lt_i32 vcc, %1, %0" :: "v"(v[i]), "v"(kk) : "vcc");
            if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(kk));
            if (OP == 11) asm volatile("v_and_b32 %0, %1, %0" : "+v"(v[i]) : "v"(rr));
            if (OP == 12) asm volatile("v_or_b32 %0, %1, %0" : "+v"(v[i]) : "v"(rr));
            if (OP == 13) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v[i]));
            if (OP == 14) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(v[i]));
            if (OP == 15) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(kk));
            if (OP == 16) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(kk), "v"(rr));
            if (OP == 17) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(v[i]));
            if (OP == 18) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(v[i]));
            if (OP == 19) asm volatile("v_floor_f32 %0, %0" : "+v"(v[i]));
            if (OP == 20) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[i]) : "v"(kk));
            if (OP == 21) asm volatile("v_add_f32 %0, %1, %0" : "+v"(v[i]) : "v"(kk));
            if (OP == 22) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(kk), "v"(rr));
            if (OP == 23) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[i]) : "v"(kk), "v"(rr));
            if (OP == 24) asm volatile("v_max_f32_e64 %0, %0, %0 clamp" : "+v"(v[i]));
            if (OP == 25) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(kk), "v"(rr));
            if (OP == 26) asm volatile("v_max_i32 %0, %1, %0" : "+v"(v[i]) : "v"(kk));
            if (OP == 27) asm volatile("v_mad_i64_i32 %0, vcc, %1, %1, %0" : "+v"(*(long long *)&v[i & ~1]) : "v"(kk) : "vcc");
            if (OP == 28) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(v[i]) : "s"(kk), "v"(rr));
            if (OP == 29) asm volatile("v_alignbit_b32 %0, %0, %1, 1" : "+v"(v[i]) : "v"(kk));
            if (OP == 30) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(kk), "v"(rr));
            if (OP == 31) asm volatile("s_movk_i32 s20, 0x1234\n v_sub_u32 %0, %0, %1" : "+v"(v[i]) : "v"(kk) : "s20");
            if (OP == 32) asm volatile("v_ashrrev_i32 %0, 1, %0\n v_ashrrev_i32 %0, 1, %0" : "+v"(v[i]));
            if (OP == 33) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(v[i]) : "v"(kk));
            if (OP == 34) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(v[i]) : "v"(kk));
            if (OP == 35) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(v[i]) : "v"(kk), "v"(rr));
            if (OP == 36) asm volatile("v_pk_ashrrev_i16 %0, 1, %0 op_sel_hi:[0,1]" : "+v"(v[i]));
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
    printf("%-20s waves/SIMD=%d  median wave cycles %.0f  -> %.2f SIMD cycles per wave64 instruction\n", name,
           wavesPerSimd, med, med / ((double)ITERS * CHAINS * wavesPerSimd));
}
int main()
{
    int *d; unsigned long long *dc;
    hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(int)); hipMalloc(&dc, 256 * 4 * 8 * 8);
    for (int w = 2; w <= 2; w *= 2) {
        run<0>("v_add_u32", d, dc, w);
        run<1>("v_add_u32_e64_sgpr", d, dc, w);
        run<2>("v_add_u32_lit", d, dc, w);
        run<3>("v_sub_u32_sdwa", d, dc, w);
        run<4>("v_sub_sdwa_w0w1", d, dc, w);
        run<5>("v_add_sdwa_dstw1", d, dc, w);
        run<6>("v_add_u32_dpp", d, dc, w);
        run<7>("v_cmp_e32+addc_e32", d, dc, w);
        run<8>("v_cmp_e64+addc_e64", d, dc, w);
        run<9>("v_cmp_e32 only", d, dc, w);
        run<10>("v_cndmask_e32", d, dc, w);
        run<11>("v_and_b32", d, dc, w);
        run<12>("v_or_b32", d, dc, w);
        run<13>("v_lshlrev_b32", d, dc, w);
        run<14>("v_lshrrev_b32", d, dc, w);
        run<15>("v_mov_b32", d, dc, w);
        run<16>("v_perm_b32", d, dc, w);
        run<17>("v_cvt_f32_i32", d, dc, w);
        run<18>("v_cvt_i32_f32", d, dc, w);
        run<19>("v_floor_f32", d, dc, w);
        run<20>("v_mul_f32", d, dc, w);
        run<21>("v_add_f32", d, dc, w);
        run<22>("v_fma_f32", d, dc, w);
        run<23>("v_fmac_f32", d, dc, w);
        run<24>("v_max_f32_clamp", d, dc, w);
        run<25>("v_med3_i32", d, dc, w);
        run<26>("v_max_i32", d, dc, w);
        run<27>("v_mad_i64_i32", d, dc, w);
        run<28>("v_mad_i32_i24", d, dc, w);
        run<29>("v_alignbit_b32", d, dc, w);
        run<30>("v_xad_u32", d, dc, w);
        run<31>("v_sub+s_movk", d, dc, w);
        run<32>("v_ashr+v_ashr(dep)", d, dc, w);
        run<33>("v_pk_add_i16", d, dc, w);
        run<34>("v_pk_mul_lo_u16", d, dc, w);
        run<35>("v_pk_mad_i16", d, dc, w);
        run<36>("v_pk_ashrrev_i16", d, dc, w);
    }
    return 0;
}
