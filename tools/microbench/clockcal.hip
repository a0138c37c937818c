// Calibrates s_memrealtime and s_memtime against hipEvent wall time (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(unsigned long long n, unsigned long long *out)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memrealtime() - r0 < n) { }
    out[0] = __builtin_amdgcn_s_memrealtime() - r0;
    out[1] = __builtin_amdgcn_s_memtime() - c0;
}
int main()
{
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        spin<<<1, 64>>>(1000000ull, d);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("wall %.3f ms  s_memrealtime %llu ticks (%.1f MHz)  s_memtime %llu ticks (%.1f MHz)\n", ms, h[0],
               h[0] / ms / 1e3, h[1], h[1] / ms / 1e3);
    }
    return 0;
}
