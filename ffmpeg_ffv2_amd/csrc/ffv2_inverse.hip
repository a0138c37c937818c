// ffv2_inverse.hip -- decoder-side inverse of the T-stage:
//   coding_to_raster   reference libavcodec/ffv2.c:81-98
//   tx_inv_2d          ffv2.c:4962-4972 over od_bin_idct64 ffv2.c:4814-4948
//   lapping post-filter ffv2.c:216-239,292-311 in the seam order of ffv2dec.c (all
//                      horizontal seams, then all vertical seams)
//   coeffs_2_ref       ffv2.c:40-52 (no clipping)
// Used as a round-trip self check of the encoder's transform stage (the lifting DCT is
// exactly invertible and, on level-shifted picture data, so is the lapping pair), not as a
// decoder: correctness first, four plain kernels, int32 intermediate plane.
// Arithmetic is the reference's wrapping int32 throughout (arbitrary coefficients allowed).
#include "ffv2_kernels.h"

#include "gen/idct64_net.h"

#define FFV2_RSH1(a)            (((a) + (int)((unsigned)(a) >> 31)) >> 1)
#define FFV2_MULRS(a, K, R, S)  ((int)((unsigned)(a) * (unsigned)(K) + (unsigned)(R)) >> (S))

namespace {

__device__ constexpr int IOUT[64] = { IDCT64_OUT_REG_LIST };

__device__ constexpr int ILAPP[46] = {            // ffv2.c:168-172
    91, 70, 68, 67, 67, 67, 67, 66, 66, 67, 67, 66, 67, 67, 67, 70,
    -32, -41, -42, -41, -40, -38, -36, -34, -32, -29, -24, -19, -14, -9, -5,
    58, 52, 50, 48, 45, 43, 40, 38, 35, 32, 29, 24, 18, 13, 8,
};

__device__ __forceinline__ int wmul_add(int a, int k, int r) { return (int)((unsigned)a * (unsigned)k + (unsigned)r); }

// 32-tap lapping post-filter, in place (ffv2.c:216-239)
__device__ __forceinline__ void inv_lap32(int (&x)[32])
{
    int t[32];
#pragma unroll
    for (int i = 0; i < 16; i++) t[31 - i] = x[i] - x[31 - i];
#pragma unroll
    for (int i = 0; i < 16; i++) t[15 - i] = x[15 - i] - (t[16 + i] >> 1);
#pragma unroll
    for (int i = 16; i < 31; i++) {
        t[i]     -= wmul_add(t[i + 1], ILAPP[i + 15], 32) >> 6;
        t[i + 1] -= wmul_add(t[i], ILAPP[i], 32) >> 6;
    }
#pragma unroll
    for (int i = 31; i >= 16; i--) t[i] = (int)((unsigned)t[i] << 6) / ILAPP[i - 16];   // truncating divide, :229-230
#pragma unroll
    for (int i = 0; i < 16; i++) { t[i] += t[31 - i] >> 1; x[i] = t[i]; }
#pragma unroll
    for (int i = 16; i < 32; i++) x[i] = t[31 - i] - t[i];
}

struct InvArgs {
    FFV2Geom g;
    int nframes;
    const int32_t *coef;       // [nframes][nblk][4096] coding order
    int32_t *plane;            // [nframes][planes][gh][gw] workspace
    uint8_t *frames;           // output pictures, layout of ffv2amd_info
    const uint16_t *lds_scan;  // forward scan table (byte offsets, raster pitch 69 dwords)
};

constexpr int RP = 69, TP = 65;

// one wavefront per block-plane: scan scatter -> row IDCTs -> transpose -> column IDCTs
__global__ __launch_bounds__(64) void ffv2_itx_kernel(const InvArgs a)
{
    __shared__ int xb[64 * RP];
    const FFV2Geom &g = a.g;
    const int lane = threadIdx.x;
    const long long id = blockIdx.x;
    const int f = (int)(id / g.nblk), bp = (int)(id % g.nblk);
    const int sb = bp / g.planes, p = bp % g.planes;
    const int sby = sb / g.nsx, sbx = sb % g.nsx;
    const int gw = g.nsx * 64, gh = g.nsy * 64;

    // coding_to_raster: lane holds coding indices q = 256j + 4*lane + k
    const int4 *cp = reinterpret_cast<const int4 *>(a.coef + ((size_t)f * g.nblk + bp) * 4096);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint4 l = reinterpret_cast<const uint4 *>(a.lds_scan)[i * 64 + lane];
        const uint32_t w[4] = { l.x, l.y, l.z, l.w };
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int4 c = cp[(2 * i + h) * 64 + lane];
            const int cv[4] = { c.x, c.y, c.z, c.w };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int e = 4 * h + k;
                const uint32_t off = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
                *reinterpret_cast<int *>(reinterpret_cast<char *>(xb) + off) = cv[k];
            }
        }
    }
    __syncthreads();
    int x[64];
#pragma unroll
    for (int u = 0; u < 64; u++) x[u] = xb[lane * RP + u];        // row v = lane of the coefficient block
    __syncthreads();
    IDCT64_NET(x);
#pragma unroll
    for (int n = 0; n < 64; n++) xb[n * TP + lane] = x[IOUT[n]];  // tmp[v + 64 n], ffv2.c:4969
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 64; v++) x[v] = xb[lane * TP + v];        // tmp + 64*x, lane = column x
    IDCT64_NET(x);
    int32_t *dst = a.plane + ((size_t)f * g.planes + p) * (size_t)gw * gh + (size_t)sby * 64 * gw + sbx * 64 + lane;
#pragma unroll
    for (int k = 0; k < 64; k++) dst[(size_t)k * gw] = x[IOUT[k]];
}

// vertical post-filter over every horizontal seam y = 64 j (ffv2dec.c DOLAP block, first loop)
__global__ __launch_bounds__(256) void ffv2_ipost_v_kernel(const InvArgs a)
{
    const FFV2Geom &g = a.g;
    const int gw = g.nsx * 64, gh = g.nsy * 64;
    const long long per_plane = (long long)(g.nsy - 1) * gw;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= per_plane * g.planes * a.nframes) return;
    const int xx = (int)(id % gw);
    const int j = 1 + (int)((id / gw) % (g.nsy - 1));
    const long long fp = id / per_plane;
    int32_t *s = a.plane + (size_t)fp * gw * gh + (size_t)(j * 64 - 16) * gw + xx;
    int x[32];
#pragma unroll
    for (int k = 0; k < 32; k++) x[k] = s[(size_t)k * gw];
    inv_lap32(x);
#pragma unroll
    for (int k = 0; k < 32; k++) s[(size_t)k * gw] = x[k];
}

// horizontal post-filter over every vertical seam x = 64 i (second loop)
__global__ __launch_bounds__(256) void ffv2_ipost_h_kernel(const InvArgs a)
{
    const FFV2Geom &g = a.g;
    const int gw = g.nsx * 64, gh = g.nsy * 64;
    const long long per_plane = (long long)(g.nsx - 1) * gh;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= per_plane * g.planes * a.nframes) return;
    const int i = 1 + (int)(id % (g.nsx - 1));
    const int yy = (int)((id / (g.nsx - 1)) % gh);
    const long long fp = id / per_plane;
    int4 *s = reinterpret_cast<int4 *>(a.plane + (size_t)fp * gw * gh + (size_t)yy * gw + i * 64 - 16);
    int x[32];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int4 w = s[q];
        x[4 * q] = w.x; x[4 * q + 1] = w.y; x[4 * q + 2] = w.z; x[4 * q + 3] = w.w;
    }
    inv_lap32(x);
#pragma unroll
    for (int q = 0; q < 8; q++) s[q] = make_int4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
}

// coeffs_2_ref (ffv2.c:40-52): (v + 2048) >> (12 - depth), written without clipping
__global__ __launch_bounds__(256) void ffv2_ipix_kernel(const InvArgs a)
{
    const FFV2Geom &g = a.g;
    const int gw = g.nsx * 64, gh = g.nsy * 64;
    const long long per_plane = (long long)g.width * g.height;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= per_plane * g.planes * a.nframes) return;
    const int xx = (int)(id % g.width);
    const int yy = (int)((id / g.width) % g.height);
    const long long fp = id / per_plane;
    const int f = (int)(fp / g.planes), p = (int)(fp % g.planes);
    const int v = (a.plane[(size_t)fp * gw * gh + (size_t)yy * gw + xx] + 2048) >> (12 - g.depth);
    uint8_t *row = a.frames + (size_t)f * g.frame_stride + (size_t)p * g.plane_stride + (size_t)yy * g.row_pitch;
    if (g.bytes_per_sample == 1) row[xx] = (uint8_t)v;
    else reinterpret_cast<uint16_t *>(row)[xx] = (uint16_t)v;
}

// dequant_block's last loop (ffv2dec.c:135-136): coefficient = pulse * mag, stored to int32 as the
// reference binary does on x86-64 (cvttss2si: truncation; NaN and out-of-range -> 0x80000000, which is
// what every coefficient of a qp == 0 packet becomes: mag = gain / sqrt(0)).  pulses: the value the
// decoder's pulses[] holds for each coding position when its band is scaled (stale slots included),
// worked out by the host's entropy parse; mag: per block-plane and band, computed on the host with its
// own pow / sqrt.  Coding index 0 is the "DC" slot, passed through.
__global__ __launch_bounds__(256) void ffv2_dequant_kernel(const int16_t *pulses, const float *mag, const int32_t *c0,
                                                           int32_t *coef, long long nbp)
{
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nbp * 4096) return;
    const long long bp = id >> 12;
    const int q = (int)(id & 4095);
    if (q == 0) { coef[id] = c0[bp]; return; }
    // band of coding index q: starts 1 + {0,15,23,31,63,95,127,255,383,511,1023,1535,2047} (ffv2.c:100-120)
    const int t = q - 1;
    const int band = (t >= 15) + (t >= 23) + (t >= 31) + (t >= 63) + (t >= 95) + (t >= 127) + (t >= 255) + (t >= 383) +
                     (t >= 511) + (t >= 1023) + (t >= 1535) + (t >= 2047);
    const float v = (float)pulses[id] * mag[bp * 13 + band];
    coef[id] = (v > -2147483904.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000;
}

}  // namespace

hipError_t ffv2_launch_dequant(const int16_t *pulses, const float *mag, const int32_t *c0, int32_t *coef, long long nbp,
                               hipStream_t s)
{
    hipLaunchKernelGGL(ffv2_dequant_kernel, dim3((unsigned)((nbp * 4096 + 255) / 256)), dim3(256), 0, s, pulses, mag, c0, coef, nbp);
    return hipGetLastError();
}

hipError_t ffv2_launch_inverse(const FFV2Geom &g, int nframes, const int32_t *coef, int32_t *plane,
                               uint8_t *frames, const uint16_t *lds_scan, hipStream_t s)
{
    InvArgs a{ g, nframes, coef, plane, frames, lds_scan };
    const long long nbp = (long long)nframes * g.nblk;
    hipLaunchKernelGGL(ffv2_itx_kernel, dim3((unsigned)nbp), dim3(64), 0, s, a);
    const int gw = g.nsx * 64, gh = g.nsy * 64;
    if (g.nsy > 1) {
        const long long n = (long long)(g.nsy - 1) * gw * g.planes * nframes;
        hipLaunchKernelGGL(ffv2_ipost_v_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    }
    if (g.nsx > 1) {
        const long long n = (long long)(g.nsx - 1) * gh * g.planes * nframes;
        hipLaunchKernelGGL(ffv2_ipost_h_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    }
    const long long n = (long long)g.width * g.height * g.planes * nframes;
    hipLaunchKernelGGL(ffv2_ipix_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
