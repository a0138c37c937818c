// Throughput of the integer VALU ops the lifting network is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o mulbench mulbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHAINS 8
#define ITERS 4096
template <int OP> __global__ void k(int *out, int seed)
{
    int v[CHAINS];
    for (int i = 0; i < CHAINS; i++) v[i] = seed + threadIdx.x + i;
    int kk = seed | 13573, rr = seed | 16384;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (OP == 0) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[i]) : "v"(kk));
            if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(kk));
            if (OP == 2) asm volatile("v_mul_i32_i24 %0, %1, %0" : "+v"(v[i]) : "v"(kk));
            if (OP == 3) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(v[i]) : "s"(kk), "v"(rr));
            if (OP == 4) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(v[i]));
            if (OP == 5) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(*(long long *)&v[i & ~1]) : "v"(kk) : "vcc");
        }
    }
    int s = 0;
    for (int i = 0; i < CHAINS; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char *name, int *d, int wavesPerSimd)
{
    int blocks = 256 * 4 * wavesPerSimd;   // 64-thread blocks: one wave each
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 64>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, 64>>>(d, 1);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double winst = (double)blocks * ITERS * CHAINS;
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal
    double cyc = ms * 1e-3 * 2.4e9 / (winst / 1024.0);
    printf("%-16s waves/SIMD=%d  %.3f ms  %.2f cyc/inst/SIMD (@2.4GHz nominal)\n", name, wavesPerSimd, ms, cyc);
}
int main()
{
    int *d; hipMalloc(&d, 256 * 4 * 8 * 64 * sizeof(int));
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_add_u32", d, w); run<1>("v_mul_lo_u32", d, w); run<2>("v_mul_i32_i24", d, w);
        run<3>("v_mad_i32_i24", d, w); run<4>("v_ashrrev_i32", d, w); run<5>("v_mad_u64_u32", d, w);
    }
    return 0;
}
