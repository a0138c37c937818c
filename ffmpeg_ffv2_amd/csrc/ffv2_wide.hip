// ffv2_wide.hip -- the T-stage in plain wrapping int32, for frames the fast kernels refuse.
//
// The fast T-stage (ffv2_kernels.hip) stages samples as int16 and multiplies with v_mul_i32_i24:
// lossless for samples inside the declared bit depth, which is all a decoder of real video ever
// hands over.  The reference, though, level-shifts ANY 16-bit sample ((v << (12 - depth)) - 2048,
// ffv2.c:26-38) and carries on in int32; a frame with samples above its depth gets a packet there,
// and band gains beyond the fast path's 32 768-entry table.  So that such a frame yields the
// reference's packet instead of FFV2AMD_ERR_RANGE, the host-synchronous entry points rerun it here:
// the reference's own structure (ffv2enc.c:469-476, 345-366) as four plain kernels over a padded
// int32 plane -- level shift, horizontal lapping of every vertical seam, vertical lapping of every
// horizontal seam, 2-D lifting DCT + scan + band energies per block -- every multiply a wrapping
// 32-bit one.  Correctness only; nothing here is tuned (garbage input is not a workload).
#include "ffv2_kernels.h"

#include "gen/fdct64_net.h"
#include "gen/scan_lut.h"

#define FFV2_RSH1(a)            (((a) + (int)((unsigned)(a) >> 31)) >> 1)
#define FFV2_MULRS(a, K, R, S)  ((int)((unsigned)(a) * (unsigned)(K) + (unsigned)(R)) >> (S))

namespace {

__device__ constexpr int WOUT[64] = { FDCT64_OUT_REG_LIST };
__device__ constexpr int WLAPP[46] = {            // ffv2.c:168-172
    91, 70, 68, 67, 67, 67, 67, 66, 66, 67, 67, 66, 67, 67, 67, 70,
    -32, -41, -42, -41, -40, -38, -36, -34, -32, -29, -24, -19, -14, -9, -5,
    58, 52, 50, 48, 45, 43, 40, 38, 35, 32, 29, 24, 18, 13, 8,
};
__device__ uint16_t g_wide_scan[4096];            // FFV2_SCAN_LUT, uploaded on first use

__device__ __forceinline__ int wmul(int a, int k) { return (int)((unsigned)a * (unsigned)k); }

// 32-tap lapping pre-filter, in place (ffv2.c:183-214), wrapping int32
__device__ __forceinline__ void wide_lap32(int (&x)[32])
{
    int t[32];
#pragma unroll
    for (int i = 0; i < 16; i++) t[31 - i] = x[i] - x[31 - i];
#pragma unroll
    for (int i = 0; i < 16; i++) t[15 - i] = x[15 - i] - (t[16 + i] >> 1);
#pragma unroll
    for (int i = 16; i < 32; i++) {
        const int v = wmul(t[i], WLAPP[i - 16]) >> 6;
        t[i] = v + (v > 0);                                   // ffv2.c:196
    }
#pragma unroll
    for (int i = 31; i > 16; i--) {
        t[i]     += (int)((unsigned)wmul(t[i - 1], WLAPP[i - 1]) + 32u) >> 6;
        t[i - 1] += (int)((unsigned)wmul(t[i], WLAPP[i + 14]) + 32u) >> 6;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] += t[31 - i] >> 1;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        x[i]      = t[i];
        x[16 + i] = t[15 - i] - t[16 + i];
    }
}

struct WideArgs {
    FFV2Geom g;
    const uint8_t *frame;      // one frame, layout of ffv2amd_info
    int32_t *plane;            // [planes][gh][gw], gw = 64 nsx, gh = 64 nsy
    int32_t *coef;             // optional [nblk][4096] coding order
    int64_t *energy;           // [nblk][13], phantom W excluded
    int32_t *c0;               // [nblk] coding index 0
};

// ref2coeff (ffv2.c:26-38) into the zeroed plane of alloc_coeff_buf (ffv2enc.c:55-75)
__global__ __launch_bounds__(256) void ffv2_wide_shift_kernel(const WideArgs a)
{
    const FFV2Geom &g = a.g;
    const int gw = g.nsx * 64, gh = g.nsy * 64;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)gw * gh * g.planes) return;
    const int x = (int)(id % gw), y = (int)((id / gw) % gh), p = (int)(id / ((long long)gw * gh));
    int v = 0;
    if (x < g.width && y < g.height) {
        const uint8_t *row = a.frame + (size_t)p * g.plane_stride + (size_t)y * g.row_pitch;
        const int s = g.bytes_per_sample == 1 ? row[x] : reinterpret_cast<const uint16_t *>(row)[x];
        v = (int)((unsigned)s << (12 - g.depth)) - 2048;
    }
    a.plane[id] = v;
}

// lap_prefilter_hor over every vertical seam x = 64 i, every row of the superblock grid (ffv2enc.c:348-355)
__global__ __launch_bounds__(256) void ffv2_wide_hlap_kernel(const WideArgs a)
{
    const FFV2Geom &g = a.g;
    const int gw = g.nsx * 64, gh = g.nsy * 64;
    const long long per_plane = (long long)(g.nsx - 1) * gh;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= per_plane * g.planes) return;
    const int y = (int)(id % gh);
    const int i = 1 + (int)((id / gh) % (g.nsx - 1));
    const int p = (int)(id / per_plane);
    int32_t *s = a.plane + ((size_t)p * gh + y) * gw + (i * 64 - 16);
    int x[32];
#pragma unroll
    for (int k = 0; k < 32; k++) x[k] = s[k];
    wide_lap32(x);
#pragma unroll
    for (int k = 0; k < 32; k++) s[k] = x[k];
}

// lap_prefilter_ver over every horizontal seam y = 64 j, every column (ffv2enc.c:357-364)
__global__ __launch_bounds__(256) void ffv2_wide_vlap_kernel(const WideArgs a)
{
    const FFV2Geom &g = a.g;
    const int gw = g.nsx * 64, gh = g.nsy * 64;
    const long long per_plane = (long long)(g.nsy - 1) * gw;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= per_plane * g.planes) return;
    const int xx = (int)(id % gw);
    const int j = 1 + (int)((id / gw) % (g.nsy - 1));
    const int p = (int)(id / per_plane);
    int32_t *s = a.plane + ((size_t)p * gh + (j * 64 - 16)) * gw + xx;
    int x[32];
#pragma unroll
    for (int k = 0; k < 32; k++) x[k] = s[(size_t)k * gw];
    wide_lap32(x);
#pragma unroll
    for (int k = 0; k < 32; k++) s[(size_t)k * gw] = x[k];
}

// tx_fwd_2d (ffv2.c:4950-4960) + raster_to_coding (ffv2.c:62-79) + the band energies of quant_block
// (ffv2enc.c:163-164): one wavefront per block-plane
__global__ __launch_bounds__(64) void ffv2_wide_tx_kernel(const WideArgs a)
{
    __shared__ int xb[64 * 65];
    const FFV2Geom &g = a.g;
    const int lane = threadIdx.x, bp = blockIdx.x;
    const int sb = bp / g.planes, p = bp % g.planes;
    const int sby = sb / g.nsx, sbx = sb % g.nsx;
    const int gw = g.nsx * 64, gh = g.nsy * 64;
    const int32_t *src = a.plane + ((size_t)p * gh + sby * 64) * gw + sbx * 64 + lane;
    int x[64];
#pragma unroll
    for (int k = 0; k < 64; k++) x[k] = src[(size_t)k * gw];           // lane = column
    FDCT64_NET(x);
#pragma unroll
    for (int v = 0; v < 64; v++) xb[lane * 65 + v] = x[WOUT[v]];       // tmp[64*col + v]
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 64; k++) x[k] = xb[k * 65 + lane];             // lane = vertical frequency v
    __syncthreads();
    FDCT64_NET(x);
#pragma unroll
    for (int u = 0; u < 64; u++) xb[lane * 65 + u] = x[WOUT[u]];       // dst[64*v + u]
    __syncthreads();
    // coding order: lane holds q = 64 k + lane
    const int BS[14] = { 0, 15, 23, 31, 63, 95, 127, 255, 383, 511, 1023, 1535, 2047, 4095 };   // ffv2.c:100-120, clipped to the block
    long long acc[13];
#pragma unroll
    for (int b = 0; b < 13; b++) acc[b] = 0;
#pragma unroll 4
    for (int k = 0; k < 64; k++) {
        const int q = 64 * k + lane;
        const int r = g_wide_scan[q];
        const int c = xb[(r >> 6) * 65 + (r & 63)];
        if (a.coef) a.coef[(size_t)bp * 4096 + q] = c;
        if (q == 0) a.c0[bp] = c;
#pragma unroll
        for (int b = 0; b < 13; b++)
            if (q >= 1 + BS[b] && q < 1 + BS[b + 1]) acc[b] += (long long)c * c;
    }
#pragma unroll
    for (int b = 0; b < 13; b++) {
        long long v = acc[b];
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0) a.energy[(size_t)bp * 13 + b] = v;
    }
}

}  // namespace

hipError_t ffv2_launch_wide_tstage(const FFV2Geom &g, const uint8_t *d_frame, int32_t *plane, int32_t *coef,
                                   int64_t *energy, int32_t *c0, hipStream_t s)
{
    static bool scan_up[16];
    int dev = 0;
    hipError_t rc = hipGetDevice(&dev);
    if (rc != hipSuccess) return rc;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    if (!scan_up[dev]) {
        rc = hipMemcpyToSymbol(HIP_SYMBOL(g_wide_scan), FFV2_SCAN_LUT, sizeof(FFV2_SCAN_LUT));
        if (rc != hipSuccess) return rc;
        scan_up[dev] = true;
    }
    WideArgs a{};
    a.g = g; a.frame = d_frame; a.plane = plane; a.coef = coef; a.energy = energy; a.c0 = c0;
    const long long gw = g.nsx * 64, gh = g.nsy * 64;
    const long long n1 = gw * gh * g.planes;
    hipLaunchKernelGGL(ffv2_wide_shift_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, a);
    if (g.nsx > 1) {
        const long long n2 = (long long)(g.nsx - 1) * gh * g.planes;
        hipLaunchKernelGGL(ffv2_wide_hlap_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, a);
    }
    if (g.nsy > 1) {
        const long long n3 = (long long)(g.nsy - 1) * gw * g.planes;
        hipLaunchKernelGGL(ffv2_wide_vlap_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(ffv2_wide_tx_kernel, dim3((unsigned)g.nblk), dim3(64), 0, s, a);
    return hipGetLastError();
}
