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
    asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333\n s_mov_b32 s22, 7" ::: "vcc", "s20", "s21", "s22");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (OP == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[i]) : "v"(kk));
            if (OP == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(kk));
            if (OP == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v[i]) : "v"(kk));
            if (OP == 3) asm volatile("v_cndmask_b32_e64 %0, 0, %1, s[20:21]" : "+v"(v[i]) : "v"(kk));
            if (OP == 4) asm volatile("v_cmp_lt_i32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(kk) : "vcc");
            if (OP == 5) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(v[i]) : "v"(kk), "v"(rr));
            if (OP == 6) asm volatile("v_xor_b32 %0, %1, %0\n v_and_b32 %0, %2, %0\n v_xor_b32 %0, %1, %0" : "+v"(v[i]) : "v"(kk), "v"(rr));
            if (OP == 7) asm volatile("v_addc_co_u32 %0, vcc, %1, %0, vcc" : "+v"(v[i]) : "v"(kk) : "vcc");
            if (OP == 8) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(v[i]) : "s20");
            if (OP == 9) asm volatile("v_writelane_b32 %0, s22, 3" : "+v"(v[i]));
            if (OP == 10) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(v[(i+1)%CHAINS]));
            if (OP == 11) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
            if (OP == 12) asm volatile("s_nop 0");
            if (OP == 13) asm volatile("s_mov_b32 s20, 0x1234" ::: "s20");
            if (OP == 14) asm volatile("s_add_u32 s20, s20, s21" ::: "s20", "scc");
            if (OP == 15) asm volatile("v_add_u32 %0, %1, %0\n s_add_u32 s20, s20, s21" : "+v"(v[i]) : "v"(kk) : "s20", "scc");
            if (OP == 16) asm volatile("ds_read_b32 %0, %1" : "=v"(v[i]) : "v"(rr & 1020));
            if (OP == 17) asm volatile("ds_write_b32 %1, %0" :: "v"(v[i]), "v"(rr & 1020));
            if (OP == 18) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v[i]));
            if (OP == 19) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(v[i]) : "v"(rr & 3));
            if (OP == 20) asm volatile("v_mul_u32_u24 %0, 0x12345, %0" : "+v"(v[i]));
            if (OP == 21) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[i]) : "v"(kk), "v"(rr));
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
    printf("%-24s waves/SIMD=%d  median wave cycles %.0f  -> %.2f SIMD cycles per wave64 instruction\n", name,
           wavesPerSimd, med, med / ((double)ITERS * CHAINS * wavesPerSimd));
}
int main()
{
    int *d; unsigned long long *dc;
    hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(int)); hipMalloc(&dc, 256 * 4 * 8 * 8);
    for (int w = 1; w <= 2; w *= 2) {
        run<0>("v_add_u32", d, dc, w);
        run<1>("v_cndmask_e32 vcc", d, dc, w);
        run<2>("v_cndmask_e64 sgpr", d, dc, w);
        run<3>("v_cndmask_e64 const", d, dc, w);
        run<4>("v_cmp+v_cndmask vcc", d, dc, w);
        run<5>("v_bfi_b32", d, dc, w);
        run<6>("xor-and-xor (3 cheap)", d, dc, w);
        run<7>("v_addc_co e32 vcc", d, dc, w);
        run<8>("v_readlane", d, dc, w);
        run<9>("v_writelane", d, dc, w);
        run<10>("v_permlane32_swap", d, dc, w);
        run<11>("v_mov_b32_dpp", d, dc, w);
        run<12>("s_nop 0", d, dc, w);
        run<13>("s_mov_b32", d, dc, w);
        run<14>("s_add_u32", d, dc, w);
        run<15>("v_add + s_add", d, dc, w);
        run<16>("ds_read_b32", d, dc, w);
        run<17>("ds_write_b32", d, dc, w);
        run<18>("v_lshlrev_b32 1", d, dc, w);
        run<19>("v_lshlrev_b32 v", d, dc, w);
        run<20>("v_mul_u32_u24 lit", d, dc, w);
        run<21>("v_mad_u32_u24", d, dc, w);
    }
    return 0;
}
