// ffv2_upconv.hip -- 4:2:0 -> 4:4:4 in front of the T-stage.
//
// ffv2's encode2() takes 4:4:4 planar only (ffv2enc.c:596-601); handed a yuv420p* source the
// reference tool chain converts first: choose_pixel_fmt() (fftools/ffmpeg_filter.c:63-131) picks
// yuv444p* of the same depth and libavfilter inserts a scale filter with the tool's default
// flags=bicubic.  With equal luma size that is, in libswscale's generic C scaler:
//   luma    unscaled, filter size 1: the identity;
//   chroma  2x up both ways with the 4-tap bicubic (B = 0, C = 0.6) of initFilter()
//           (libswscale/utils.c:332-727) at sample positions 128/128 (get_local_pos, :303-310;
//           vf_scale.c:566-577), 14-bit horizontal and 12-bit vertical coefficients:
//             h = min((sum_k src[r][hpos[x]+k] * hf[x][k]) >> (8-bit: 7 | deeper: depth-1), 32767)
//                                                               swscale.c:96-139
//             out = clip((round + sum_j h[vpos[y]+j][x] * vf[y][j]) >> shift)
//                   8 bit: round = 64 << 12, shift 19 (yuv2planeX_8_c, no dither for 8-bit sources)
//                   deeper: round = 1 << (shift-1), shift = 27 - depth   (output.c:333-393)
// PARITY UNPINNED: no libswscale binary or vector exists here; checked against
// oracle/ffv2_swscale_oracle.c.  build_axis() below RESTATES initFilter()'s integer steps (they must
// be the same steps for the tables to be bit-exact), and so does the oracle: the coefficient tables
// are effectively compared with a second writing of the same derivation, only the device arithmetic
// of the two 4-tap passes is checked independently.
//
// Host part: the coefficient tables (integer arithmetic of initFilter, once per geometry).
// Device part: ffv2_upconv_tile_kernel -- a workgroup owns a 128 x 32 tile of one output chroma
// plane: the source patch it needs (<= 24 x 80 samples) goes to LDS, the horizontal pass runs once
// per source row into a 15-bit LDS intermediate (what swscale keeps between its two passes), the
// vertical pass reads it back as 16-byte rows and stores 8 samples per lane.  6.4 multiplies per
// output sample instead of the 20 of the one-thread-per-sample kernel, which stays as the fallback
// for geometries whose patch would not fit.  Luma is the identity: rows copied (or, in the frame
// ring, sent by the DMA engine straight into plane 0).
#include "ffv2_kernels.h"

#include <stdlib.h>
#include <vector>

namespace {

constexpr int UP_TAPS = 8;                 // table stride; a 2x bicubic needs 4

// ---- initFilter(), SWS_BICUBIC branch, default parameters, no src/dst filter vectors, C-code
// alignment (1) -- utils.c line numbers in the comments ----
struct AxisFilter {
    int taps = 0;
    std::vector<int16_t> coef;             // [n][UP_TAPS]
    std::vector<int32_t> pos;              // [n]
};

int ilog2u(unsigned v) { int n = 0; while (v >>= 1) n++; return n; }
int64_t iabs64(int64_t v) { return v < 0 ? -v : v; }

bool build_axis(AxisFilter &out, int n, int one)
{
    const int srcN = (n + 1) >> 1;                                            // :1409-1410
    const int inc = (int)((((int64_t)srcN << 16) + (n >> 1)) / n);            // :1443-1444
    const int srcPos = 128, dstPos = 128;                                     // :303-310
    const int64_t fone = 1LL << (54 - (ilog2u((unsigned)(srcN / n)) < 8 ? ilog2u((unsigned)(srcN / n)) : 8));   // :345
    std::vector<int64_t> f;
    std::vector<int32_t> pos((size_t)n);
    int size;
    if (abs(inc - 0x10000) < 10 && srcPos == dstPos) {                        // :355 (n == 1)
        size = 1;
        f.assign((size_t)n, fone);
        for (int i = 0; i < n; i++) pos[(size_t)i] = i;
    } else {
        size = inc <= 1 << 16 ? 1 + 4 : 1 + (4 * srcN + n - 1) / n;           // :418-421, size factor 4
        if (size > srcN - 2) size = srcN - 2;                                 // :423-424
        if (size < 1) size = 1;
        f.resize((size_t)n * size);
        const int64_t C = (int64_t)(0.6 * (1 << 24));                         // :447, B = 0
        int64_t at = ((dstPos * (int64_t)inc) >> 7) - ((srcPos * 0x10000LL) >> 7);   // :429
        for (int i = 0; i < n; i++) {
            int xx = (int)((at - (size - 2) * (1LL << 16)) / (1 << 17));      // :431
            pos[(size_t)i] = xx;
            for (int j = 0; j < size; j++, xx++) {
                int64_t d = iabs64((int64_t)xx * (1 << 17) - at) << 13;       // :435
                if (inc > 1 << 16) d = d * n / srcN;
                int64_t c = 0;
                if (d < 1LL << 31) {
                    const int64_t dd = (d * d) >> 30, ddd = (dd * d) >> 30;
                    c = d < 1LL << 30
                        ? (12 * (1 << 24) - 6 * C) * ddd + (-18 * (1 << 24) + 6 * C) * dd + (6 * (1 << 24)) * (1LL << 30)
                        : (-6 * C) * ddd + (30 * C) * dd + (-48 * C) * d + (24 * C) * (1LL << 30);      // :452-464
                }
                f[(size_t)i * size + j] = c / ((1LL << 54) / fone);           // :466
            }
            at += 2 * inc;
        }
    }
    // shrink: drop near-zero taps on the left (shifting the row), count them on the right  :545-584
    int need = 0;
    for (int i = n - 1; i >= 0; i--) {
        int64_t *row = &f[(size_t)i * size];
        int64_t cut = 0;
        for (int j = 0; j < size; j++) {
            cut += iabs64(row[0]);
            if ((double)cut > 0.002 * (double)fone) break;                    // SWS_MAX_REDUCE_CUTOFF
            if (i < n - 1 && pos[(size_t)i] >= pos[(size_t)i + 1]) break;     // keep positions monotone
            for (int k = 1; k < size; k++) row[k - 1] = row[k];
            row[size - 1] = 0;
            pos[(size_t)i]++;
        }
        int keep = size;
        cut = 0;
        for (int j = size - 1; j > 0; j--) {
            cut += iabs64(row[j]);
            if ((double)cut > 0.002 * (double)fone) break;
            keep--;
        }
        if (keep > need) need = keep;
    }
    if (need < 1 || need > UP_TAPS) return false;
    std::vector<int64_t> g((size_t)n * need);                                 // :617-627
    for (int i = 0; i < n; i++)
        for (int j = 0; j < need; j++) g[(size_t)i * need + j] = j < size ? f[(size_t)i * size + j] : 0;
    // borders: taps that would read outside are folded onto the edge sample  :630-671
    for (int i = 0; i < n; i++) {
        int64_t *row = &g[(size_t)i * need];
        int32_t &p = pos[(size_t)i];
        if (p < 0) {
            for (int j = 1; j < need; j++) {
                const int left = j + p > 0 ? j + p : 0;
                row[left] += row[j];
                row[j] = 0;
            }
            p = 0;
        }
        if (p + need > srcN) {
            const int shift = p + (need - srcN < 0 ? need - srcN : 0);
            int64_t acc = 0;
            for (int j = need - 1; j >= 0; j--)
                if (p + j >= srcN) { acc += row[j]; row[j] = 0; }
            for (int j = need - 1; j >= 0; j--) row[j] = j < shift ? 0 : row[j - shift];
            p -= shift;
            row[srcN - 1 - p] += acc;
        }
    }
    // normalise each row to `one` with error feedback  :679-698
    out.taps = need;
    out.pos = pos;
    out.coef.assign((size_t)n * UP_TAPS, 0);
    for (int i = 0; i < n; i++) {
        const int64_t *row = &g[(size_t)i * need];
        int64_t sum = 0, err = 0;
        for (int j = 0; j < need; j++) sum += row[j];
        sum = (sum + one / 2) / one;
        if (!sum) sum = 1;
        for (int j = 0; j < need; j++) {
            const int64_t v = row[j] + err;
            const int64_t q = (v >= 0 ? v + (sum >> 1) : v - (sum >> 1)) / sum;       // ROUNDED_DIV
            out.coef[(size_t)i * UP_TAPS + j] = (int16_t)q;
            err = v - q * sum;
        }
    }
    return true;
}

struct UpArgs {
    const uint8_t *src;        // U plane of frame 0; V at + c_plane_stride, frame f at + f * src_frame_stride
    uint8_t *dst;              // [nframes][frame_stride]: the encoder's 4:4:4 layout
    size_t src_frame_stride, c_plane_stride, c_pitch;     // bytes
    size_t frame_stride, plane_stride, row_pitch;
    int w, h, cw, ch, depth, htaps, vtaps;
    const int16_t *hf, *vf;    // [w][UP_TAPS], [h][UP_TAPS]
    const int32_t *hp, *vp;
};

template <int BPS>
__global__ __launch_bounds__(256) void ffv2_upconv_kernel(const UpArgs a)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    const int p = 1 + (int)(blockIdx.z & 1u), f = (int)(blockIdx.z >> 1);
    if (x >= a.w) return;
    const uint8_t *sp = a.src + (size_t)f * a.src_frame_stride + (size_t)(p - 1) * a.c_plane_stride;
    const int hp = a.hp[x], vp = a.vp[y];
    int hc[UP_TAPS], vc[UP_TAPS];
#pragma unroll
    for (int k = 0; k < UP_TAPS; k++) { hc[k] = a.hf[(size_t)x * UP_TAPS + k]; vc[k] = a.vf[(size_t)y * UP_TAPS + k]; }
    const int hsh = BPS == 1 ? 7 : a.depth - 1;
    const int vsh = BPS == 1 ? 19 : 27 - a.depth;
    int val = BPS == 1 ? 64 << 12 : 1 << (vsh - 1);
    for (int j = 0; j < a.vtaps; j++) {
        const uint8_t *row = sp + (size_t)(vp + j) * a.c_pitch;
        int hv = 0;
        for (int k = 0; k < a.htaps; k++) {
            const int s = BPS == 1 ? row[hp + k] : reinterpret_cast<const uint16_t *>(row)[hp + k];
            hv += s * hc[k];
        }
        hv >>= hsh;
        hv = hv < 32767 ? hv : 32767;
        val += hv * vc[j];
    }
    val >>= vsh;
    const int hi = (1 << a.depth) - 1;
    val = val < 0 ? 0 : (val > hi ? hi : val);
    uint8_t *dp = a.dst + (size_t)f * a.frame_stride + (size_t)p * a.plane_stride + (size_t)y * a.row_pitch;
    if (BPS == 1) dp[x] = (uint8_t)val;
    else reinterpret_cast<uint16_t *>(dp)[x] = (uint16_t)val;
}

// ---- tiled kernel ----
constexpr int UT_W = 128, UT_H = 32;           // output tile
constexpr int UT_SR = 24, UT_SC = 80;          // source patch bound (rows, columns); checked on the host per geometry
constexpr int UT_HP = UT_W + 8;                // int16 per row of the horizontal-pass buffer (272 B rows)

template <int BPS>
__global__ __launch_bounds__(256) void ffv2_upconv_tile_kernel(const UpArgs a)
{
    __shared__ uint16_t patch[UT_SR][UT_SC];
    __shared__ __attribute__((aligned(16))) int16_t hbuf[UT_SR][UT_HP];
    __shared__ int16_t vcs[UT_H][4];
    __shared__ int vps[UT_H];
    const int t = threadIdx.x;
    const int x0 = blockIdx.x * UT_W, y0 = blockIdx.y * UT_H;
    const int p = 1 + (int)(blockIdx.z & 1u), f = (int)(blockIdx.z >> 1);
    const uint8_t *sp = a.src + (size_t)f * a.src_frame_stride + (size_t)(p - 1) * a.c_plane_stride;
    const int xl = min(x0 + UT_W, a.w) - 1, yl = min(y0 + UT_H, a.h) - 1;     // last output column / row of the tile
    const int cs0 = a.hp[x0], cs1 = min(a.hp[xl] + a.htaps, a.cw);           // source columns [cs0, cs1)
    const int rs0 = a.vp[y0], rs1 = min(a.vp[yl] + a.vtaps, a.ch);
    const int ncols = cs1 - cs0, nrows = rs1 - rs0;
    // source patch -> LDS (rows of <= 80 samples; lanes read consecutive samples)
    for (int i = t; i < nrows * UT_SC; i += 256) {
        const int r = i / UT_SC, c = i - r * UT_SC;
        if (c < ncols) {
            const uint8_t *row = sp + (size_t)(rs0 + r) * a.c_pitch;
            patch[r][c] = BPS == 1 ? row[cs0 + c] : reinterpret_cast<const uint16_t *>(row)[cs0 + c];
        }
    }
    if (t < UT_H) {
        const int y = min(y0 + t, a.h - 1);
        vps[t] = a.vp[y] - rs0;
#pragma unroll
        for (int k = 0; k < 4; k++) vcs[t][k] = a.vf[(size_t)y * UP_TAPS + k];
    }
    // this thread's column of the horizontal pass
    const int xc = min(x0 + (t & (UT_W - 1)), a.w - 1);
    const int hp = a.hp[xc] - cs0;
    int hc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) hc[k] = a.hf[(size_t)xc * UP_TAPS + k];
    __syncthreads();
    // horizontal pass: h = min((sum s * c) >> hsh, 32767)   (hScale8To15_c / hScale16To15_c)
    const int hsh = BPS == 1 ? 7 : a.depth - 1;
    for (int r = t >> 7; r < nrows; r += 2) {
        int hv = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c = hp + k;
            hv += (k < a.htaps && c < ncols ? (int)patch[r][c] : 0) * hc[k];
        }
        hv >>= hsh;
        hbuf[r][t & (UT_W - 1)] = (int16_t)(hv < 32767 ? hv : 32767);
    }
    __syncthreads();
    // vertical pass: 8 adjacent samples of one row per item   (yuv2planeX_8_c / yuv2planeX_10_c_template)
    const int vsh = BPS == 1 ? 19 : 27 - a.depth;
    const int rnd = BPS == 1 ? 64 << 12 : 1 << (vsh - 1);
    const int hi = (1 << a.depth) - 1;
#pragma unroll
    for (int it = 0; it < UT_W * UT_H / 8 / 256; it++) {
        const int id = t + 256 * it;
        const int ty = id >> 4, xg = (id & 15) * 8;
        const int y = y0 + ty, x = x0 + xg;
        if (y >= a.h || x >= a.w) continue;
        int acc[8];
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = rnd;
        const int vp = vps[ty];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = vp + j;
            const int c = vcs[ty][j];
            if (j < a.vtaps && r < nrows) {
                const int4 w = *reinterpret_cast<const int4 *>(&hbuf[r][xg]);
                const int wv[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    acc[2 * q]     += ((wv[q] << 16) >> 16) * c;
                    acc[2 * q + 1] += (wv[q] >> 16) * c;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int v = acc[e] >> vsh;
            acc[e] = v < 0 ? 0 : (v > hi ? hi : v);
        }
        // the row pitch is a multiple of 128 bytes: a whole vector may be written where the picture
        // ends inside it (the padding behind the last sample is never read as picture)
        uint8_t *dp = a.dst + (size_t)f * a.frame_stride + (size_t)p * a.plane_stride + (size_t)y * a.row_pitch;
        if (BPS == 1) {
            const uint2 o = make_uint2((uint32_t)acc[0] | ((uint32_t)acc[1] << 8) | ((uint32_t)acc[2] << 16) | ((uint32_t)acc[3] << 24),
                                       (uint32_t)acc[4] | ((uint32_t)acc[5] << 8) | ((uint32_t)acc[6] << 16) | ((uint32_t)acc[7] << 24));
            *reinterpret_cast<uint2 *>(dp + x) = o;
        } else {
            const uint4 o = make_uint4((uint32_t)acc[0] | ((uint32_t)acc[1] << 16), (uint32_t)acc[2] | ((uint32_t)acc[3] << 16),
                                       (uint32_t)acc[4] | ((uint32_t)acc[5] << 16), (uint32_t)acc[6] | ((uint32_t)acc[7] << 16));
            *reinterpret_cast<uint4 *>(dp + (size_t)x * 2) = o;
        }
    }
}

}  // namespace

struct FFV2Upconv {
    int w = 0, h = 0, depth = 0, htaps = 0, vtaps = 0;
    bool tiled = false;        // every 128 x 32 output tile's source patch fits the tiled kernel's LDS
    int16_t *d_hf = nullptr, *d_vf = nullptr;
    int32_t *d_hp = nullptr, *d_vp = nullptr;
};

void ffv2_upconv_destroy(FFV2Upconv *u)
{
    if (!u) return;
    (void)hipFree(u->d_hf); (void)hipFree(u->d_vf); (void)hipFree(u->d_hp); (void)hipFree(u->d_vp);
    delete u;
}

// coefficient tables for a w x h picture of `depth` bits; nullptr if the geometry has no
// 4-tap-or-less... i.e. no table that fits (never for pictures of 8 x 8 and more)
FFV2Upconv *ffv2_upconv_create(int w, int h, int depth)
{
    AxisFilter hx, vy;
    try {
        if (!build_axis(hx, w, 1 << 14) || !build_axis(vy, h, 1 << 12)) return nullptr;   // :1681,:1714
    } catch (...) { return nullptr; }
    FFV2Upconv *u = new (std::nothrow) FFV2Upconv;
    if (!u) return nullptr;
    u->w = w; u->h = h; u->depth = depth; u->htaps = hx.taps; u->vtaps = vy.taps;
    {
        const int cw = (w + 1) >> 1, ch = (h + 1) >> 1;
        bool fits = hx.taps <= 4 && vy.taps <= 4;
        for (int x0 = 0; x0 < w && fits; x0 += UT_W) {
            const int xl = (x0 + UT_W < w ? x0 + UT_W : w) - 1;
            int c1 = hx.pos[(size_t)xl] + hx.taps;
            if (c1 > cw) c1 = cw;
            fits = c1 - hx.pos[(size_t)x0] <= UT_SC && hx.pos[(size_t)x0] >= 0;
            for (int x = x0; x <= xl && fits; x++) fits = hx.pos[(size_t)x] >= hx.pos[(size_t)x0] && hx.pos[(size_t)x] <= hx.pos[(size_t)xl];
        }
        for (int y0 = 0; y0 < h && fits; y0 += UT_H) {
            const int yl = (y0 + UT_H < h ? y0 + UT_H : h) - 1;
            int r1 = vy.pos[(size_t)yl] + vy.taps;
            if (r1 > ch) r1 = ch;
            fits = r1 - vy.pos[(size_t)y0] <= UT_SR && vy.pos[(size_t)y0] >= 0;
            for (int y = y0; y <= yl && fits; y++) fits = vy.pos[(size_t)y] >= vy.pos[(size_t)y0] && vy.pos[(size_t)y] <= vy.pos[(size_t)yl];
        }
        u->tiled = fits;
    }
    bool ok = hipMalloc(&u->d_hf, hx.coef.size() * 2) == hipSuccess && hipMalloc(&u->d_vf, vy.coef.size() * 2) == hipSuccess &&
              hipMalloc(&u->d_hp, hx.pos.size() * 4) == hipSuccess && hipMalloc(&u->d_vp, vy.pos.size() * 4) == hipSuccess;
    ok = ok && hipMemcpy(u->d_hf, hx.coef.data(), hx.coef.size() * 2, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(u->d_vf, vy.coef.data(), vy.coef.size() * 2, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(u->d_hp, hx.pos.data(), hx.pos.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(u->d_vp, vy.pos.data(), vy.pos.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
         hipStreamSynchronize(nullptr) == hipSuccess;      // null stream: the callers' streams do not wait for it
    if (!ok) { ffv2_upconv_destroy(u); return nullptr; }
    return u;
}

size_t ffv2_upconv_src_frame_bytes(int w, int h, int depth)
{
    const size_t bps = depth > 8 ? 2 : 1;
    return ((size_t)w * h + 2 * (size_t)((w + 1) >> 1) * ((h + 1) >> 1)) * bps;
}

// chroma planes only: src_u = U plane of frame 0 (rows c_pitch bytes apart), V at + c_plane_stride,
// the next frame at + src_frame_stride; dst = the encoder's 4:4:4 frames (planes 1 and 2 are written)
hipError_t ffv2_launch_upconv_chroma(const FFV2Upconv *u, const FFV2Geom &g, int nframes, const uint8_t *src_u,
                                     size_t c_pitch, size_t c_plane_stride, size_t src_frame_stride, uint8_t *dst,
                                     hipStream_t s)
{
    const int bps = g.bytes_per_sample;
    UpArgs a{};
    a.src = src_u; a.dst = dst; a.src_frame_stride = src_frame_stride; a.c_plane_stride = c_plane_stride; a.c_pitch = c_pitch;
    a.frame_stride = g.frame_stride; a.plane_stride = g.plane_stride; a.row_pitch = g.row_pitch;
    a.w = g.width; a.h = g.height; a.cw = (g.width + 1) >> 1; a.ch = (g.height + 1) >> 1; a.depth = g.depth;
    a.htaps = u->htaps; a.vtaps = u->vtaps; a.hf = u->d_hf; a.vf = u->d_vf; a.hp = u->d_hp; a.vp = u->d_vp;
    static const int force_naive = getenv("FFV2AMD_UPCONV_NAIVE") ? atoi(getenv("FFV2AMD_UPCONV_NAIVE")) : 0;
    if (u->tiled && !force_naive) {
        const dim3 grid((unsigned)((g.width + UT_W - 1) / UT_W), (unsigned)((g.height + UT_H - 1) / UT_H), (unsigned)(2 * nframes)), block(256);
        if (bps == 1) hipLaunchKernelGGL(ffv2_upconv_tile_kernel<1>, grid, block, 0, s, a);
        else          hipLaunchKernelGGL(ffv2_upconv_tile_kernel<2>, grid, block, 0, s, a);
    } else {
        const dim3 grid((unsigned)((g.width + 255) / 256), (unsigned)g.height, (unsigned)(2 * nframes)), block(256);
        if (bps == 1) hipLaunchKernelGGL(ffv2_upconv_kernel<1>, grid, block, 0, s, a);
        else          hipLaunchKernelGGL(ffv2_upconv_kernel<2>, grid, block, 0, s, a);
    }
    return hipGetLastError();
}

// tightly packed 4:2:0 frames (Y, U, V back to back) -> 4:4:4 frames
hipError_t ffv2_launch_upconv(const FFV2Upconv *u, const FFV2Geom &g, int nframes, const uint8_t *src,
                              size_t src_frame_stride, uint8_t *dst, hipStream_t s)
{
    const int bps = g.bytes_per_sample;
    // luma: the identity -- rows copied into the encoder's pitched layout
    for (int f = 0; f < nframes; f++) {
        const hipError_t rc = hipMemcpy2DAsync(dst + (size_t)f * g.frame_stride, g.row_pitch, src + (size_t)f * src_frame_stride,
                                               (size_t)g.width * bps, (size_t)g.width * bps, (size_t)g.height,
                                               hipMemcpyDeviceToDevice, s);
        if (rc != hipSuccess) return rc;
    }
    const size_t cw = (size_t)((g.width + 1) >> 1), ch = (size_t)((g.height + 1) >> 1);
    return ffv2_launch_upconv_chroma(u, g, nframes, src + (size_t)g.width * g.height * bps, cw * bps, cw * ch * bps,
                                     src_frame_stride, dst, s);
}
