// ffv2_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the FFV2 encode hot path.
//
// T-stage (one launch per batch of frames), fused per 64x64 superblock-plane:
//   level shift        reference libavcodec/ffv2.c:26-38   (ref_2_coeffs_*)
//   lapping pre-filter ffv2.c:183-214,285-304 driven by ffv2enc.c:345-366
//   2-D lifting DCT    ffv2.c:4950-4960 over od_bin_fdct64 ffv2.c:4678-4812
//   scan               ffv2.c:62-79 (raster_to_coding) + zigzags.h
//   band energies/gain ffv2enc.c:163-166,174
// E-stage at qp == 0 (ffv2enc.c:105-123,148-150,174,197 + daala_entropy.c:227-270,
// 698-721): every data-dependent symbol is a raw bit, so the packet tail is an
// exclusive prefix sum over code lengths followed by a scatter of the codes (one launch:
// each workgroup recomputes its prefix from the 24 KB of bit counts, assembles in LDS).
//
// Work decomposition: ONE wavefront (64 lanes) owns one 64x64 block-plane.  It
// stages the 96x96 halo tile (16 samples each side, what the two lapping passes
// reach) once, as int16, in 19.5 KB of LDS; the same LDS is then re-used as the
// int32 transposition buffer between the column and row DCT passes and as the
// raster buffer for the scan-order gather.  Eight such workgroups fit a CU
// (2 waves per SIMD).  HBM sees each source sample ~once (halo re-reads hit L2:
// block ids are dealt so that an XCD owns a contiguous run of superblocks) and
// each coefficient exactly once, written in coding order, 1 KiB per wave store (16 B per lane),
// issued last so that no load has to wait behind them (vmcnt is in-order).
// Integer lifting only; no MFMA.
//
// Exactness notes:
//  * int16 staging is lossless: |sample| <= 2048 after the level shift, the
//    lapping filter's largest output L1 gain is 3.353, so |H| <= 6.9e3 and
//    |H then V| <= 2.31e4 < 32767 (tools/derive_lifting_ir.py analysis).
//  * v_mul_i32_i24 gives the same low 32 bits as the reference's wrapping int32
//    multiply whenever |operand| < 2^23; with the bound above the largest operand
//    of any multiply in either DCT pass is < 2.1e6.
#include "ffv2_kernels.h"

#include "gen/fdct64_net.h"
#ifdef FFV2_ASM_NET        // experiment: python tools/gen_asm_net.py, then build with -DFFV2_ASM_NET
#include "gen/fdct64_asm.h"
#endif

#include <stdio.h>
#include <stdlib.h>

#define FFV2_WALK_MIN 2      // block-planes per wavefront from which the column walk pays

// OD_RSHIFT1(a) = (a + (a < 0)) >> 1 (ffv2.c:313).  Every value in the network is below
// 2^23 in magnitude, so byte 3 of a is pure sign and
// a - sext(a.byte3) = a + (a < 0) in one SDWA instruction instead of shift + add.
__device__ __forceinline__ int ffv2_rsh1(int a)
{
    int t;
    asm("v_sub_u32_sdwa %0, %1, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3"
        : "=v"(t) : "v"(a));
    return t >> 1;
}
#define FFV2_RSH1(a)            ffv2_rsh1(a)
// (a*K + R) >> S with R = 2^(S-1).  Row pass: v_mad_i32_i24 + shift, bit-identical to the
// reference's wrapping int32 even where that overflows (|a| < 2^23).  Column pass: inputs
// are bounded by 2.31e4 so a*K + R cannot overflow int32 (max over the network of
// L1(operand)*K is 59 914), and for K/2^S < 1/2 the same value is the high dword of
// a*(K << (32-S)) + 2^31: one v_mad_i64_i32 instead of multiply + shift.
#define FFV2_MULRS_WRAP(a, K, R, S)  ((__mul24((a), (K)) + (R)) >> (S))
#define FFV2_MULRS_NOOVF(a, K, R, S) (((K) < (1 << ((S) - 1))) \
    ? (int)(((long long)(a) * (int)((long long)(K) << (32 - (S))) + 0x80000000LL) >> 32) \
    : FFV2_MULRS_WRAP(a, K, R, S))

// Development aid (tools/phase_timing.py builds a separate library with this macro): shader-clock
// ticks per phase, kept in registers and written once per workgroup.  Never defined in the
// shipped build.
#ifdef FFV2_PHASE_TIMING
__device__ unsigned long long *g_phase_buf;          // [workgroup][8]
static unsigned long long *h_phase_buf;
static size_t h_phase_groups;
extern "C" int ffv2amd_debug_phase_alloc(size_t groups)
{
    if (hipMalloc(&h_phase_buf, groups * 64) != hipSuccess) return -1;
    hipMemset(h_phase_buf, 0, groups * 64);
    h_phase_groups = groups;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_buf), &h_phase_buf, sizeof(h_phase_buf)) == hipSuccess ? 0 : -1;
}
extern "C" void ffv2amd_debug_phase_ticks(unsigned long long *out)   // sums over the workgroups of the last launch
{
    hipDeviceSynchronize();
    unsigned long long *h = (unsigned long long *)malloc(h_phase_groups * 64);
    hipMemcpy(h, h_phase_buf, h_phase_groups * 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; i++) out[i] = 0;
    for (size_t g = 0; g < h_phase_groups; g++)
        for (int i = 0; i < 8; i++) out[i] += h[g * 8 + i];
    free(h);
}
extern "C" void ffv2amd_debug_phase_raw(unsigned long long *out)       // [workgroup][8] of the last launch
{
    hipDeviceSynchronize();
    hipMemcpy(out, h_phase_buf, h_phase_groups * 64, hipMemcpyDeviceToHost);
}
// slots 0..7: s_memtime (2.4 GHz, constant rate) deltas of the phases; the timing tool can also ask
// for the wave's start and life on the 100 MHz s_memrealtime clock (FFV2_PHASE_WALL: slots 0 and 2)
#define FFV2_PHASE_BEGIN  unsigned long long t_prev_ = __builtin_amdgcn_s_memtime(), pacc_[8] = {}; \
    const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime();
#define FFV2_PHASE_MARK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    pacc_[i] += t_ - t_prev_; t_prev_ = t_; } while (0)
#ifdef FFV2_PHASE_WALL
#define FFV2_PHASE_WALL_STAMP pacc_[2] = __builtin_amdgcn_s_memrealtime() - rt0_; pacc_[0] = rt0_;
#else
#define FFV2_PHASE_WALL_STAMP (void)rt0_;
#endif
#define FFV2_PHASE_END do { FFV2_PHASE_WALL_STAMP if (threadIdx.x == 0) { \
    unsigned long long *pt_ = g_phase_buf + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8; \
    for (int i_ = 0; i_ < 8; i_++) pt_[i_] = pacc_[i_]; } } while (0)
#else
#define FFV2_PHASE_BEGIN
#define FFV2_PHASE_MARK(i)
#define FFV2_PHASE_END
#endif

namespace {

// The T-stage workgroups are ONE wavefront: its DS instructions execute in issue order, so a
// cross-lane hand-off through LDS needs no s_barrier and no counter drain -- only the compiler
// must keep the LDS accesses in program order (per thread they may look independent).  Unlike
// __syncthreads() this leaves global loads in flight (vmcnt untouched).
__device__ __forceinline__ void wave_lds_fence() { asm volatile("" ::: "memory"); }

constexpr int TILE      = 96;          // 64 + 2*16 halo
constexpr int TPITCH    = 104;         // int16 per tile row: 208 B = 13 x 16 B (odd multiple -> b128 conflict free)
constexpr int XPITCH    = 65;          // dwords per row of the int32 transposition / raster buffers
constexpr int LDS_BYTES = TILE * TPITCH * 2;   // 19968 >= 64*65*4 = 16640
static_assert(64 * XPITCH * 4 <= LDS_BYTES, "transposition buffer must fit in the tile");

__device__ constexpr int OUTR[64] = { FDCT64_OUT_REG_LIST };

// reference libavcodec/ffv2.c:168-172 (lap_filt_params_32): 16 scales, 15+15 lifting taps
__device__ constexpr int LAPP[46] = {
    91, 70, 68, 67, 67, 67, 67, 66, 66, 67, 67, 66, 67, 67, 67, 70,
    -32, -41, -42, -41, -40, -38, -36, -34, -32, -29, -24, -19, -14, -9, -5,
    58, 52, 50, 48, 45, 43, 40, 38, 35, 32, 29, 24, 18, 13, 8,
};

// 32-tap lapping pre-filter, in place (ffv2.c:183-214)
__device__ __forceinline__ void lap32(int (&x)[32])
{
    int t[32];
#pragma unroll
    for (int i = 0; i < 16; i++) t[31 - i] = x[i] - x[31 - i];
#pragma unroll
    for (int i = 0; i < 16; i++) t[15 - i] = x[15 - i] - (t[16 + i] >> 1);
#pragma unroll
    for (int i = 16; i < 32; i++) {
        const int v = __mul24(t[i], LAPP[i - 16]) >> 6;
        t[i] = v + (int)((unsigned)(-v) >> 31);          // + (v > 0), ffv2.c:196
    }
#pragma unroll
    for (int i = 31; i > 16; i--) {
        t[i]     += (__mul24(t[i - 1], LAPP[i - 1]) + 32) >> 6;
        t[i - 1] += (__mul24(t[i], LAPP[i + 14]) + 32) >> 6;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] += t[31 - i] >> 1;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        x[i]      = t[i];
        x[16 + i] = t[15 - i] - t[16 + i];
    }
}

__device__ __forceinline__ int lo16(uint32_t w) { return (int)(w << 16) >> 16; }
__device__ __forceinline__ int hi16(uint32_t w) { return (int)w >> 16; }
// low halves of two registers into one dword: one v_perm_b32 (bytes 0,1 of lo; bytes 0,1 of hi)
__device__ __forceinline__ uint32_t pack16(int lo, int hi) { return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u); }

// Coded band gain, bit-identical to the host's
//   (uint32)(float)pow((double)(sqrtf((float)e) + FLT_EPSILON), (double)(1.0f/1.5f))
// (ffv2enc.c:131-138,166,174) without evaluating pow on the device: thr[n] is the
// least energy whose coded gain is >= n+1, tabulated by the host with its own libm
// at encoder creation; the float estimate below only seeds the search.
__device__ __forceinline__ uint32_t coded_gain(long long e, const int64_t *thr, int n, bool &out_of_table)
{
    // gain = #{k : thr[k] <= e}.  The float estimate is within +-1 of it; five independent
    // probes around the estimate (a fixed number of loads: the compiler can then count
    // vmcnt exactly) bracket it, anything else is reported instead of guessed.
    const float g = sqrtf((float)e);
    int v0 = (int)exp2f(log2f(g + 1e-30f) * 0.6666667f);
    v0 = v0 < 2 ? 2 : (v0 > n - 3 ? n - 3 : v0);
    const long long t0 = thr[v0 - 2], t1 = thr[v0 - 1], t2 = thr[v0], t3 = thr[v0 + 1], t4 = thr[v0 + 2];
    const int v = (v0 - 2) + (e >= t0) + (e >= t1) + (e >= t2) + (e >= t3);
    // consistent iff everything below the window is <= e and the probe above is > e
    if ((v0 > 2 && e < t0) || e >= t4) out_of_table = true;
    return (uint32_t)v;
}

__device__ __forceinline__ int golomb_len(uint32_t val)      // ffv2enc.c:105-123
{
    return 2 * (31 - __clz(val + 1)) + 1;
}

// ---------------------------------------------------------------------------
// T-stage
// ---------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));

// level shift of two packed samples in one instruction: (s << sh) - 2048 per 16-bit field
// as s * 2^sh + 0xF800 (mod 2^16), mul = (1 << sh) * 0x00010001
__device__ __forceinline__ uint32_t pk_level_shift(uint32_t w, uint32_t mul)
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(w), "s"(mul), "v"(0xF800F800u));
    return r;
}

// inclusive prefix sum over the 64 lanes, DPP only (no LDS crossbar)
__device__ __forceinline__ int wave_iscan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return v;
}

template <int CTRL> __device__ __forceinline__ int dpp_mov(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}

// 64-bit lane values a, b -> lanes 0..31: a[l] + a[l+32], lanes 32..63: b[l-32] + b[l]
__device__ __forceinline__ unsigned long long fold32(unsigned long long a, unsigned long long b)
{
    const auto l = __builtin_amdgcn_permlane32_swap((uint32_t)a, (uint32_t)b, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap((uint32_t)(a >> 32), (uint32_t)(b >> 32), false, false);
    return (((unsigned long long)h[0] << 32) | l[0]) + (((unsigned long long)h[1] << 32) | l[1]);
}
// rows of 16 lanes: (a0+a1 | b0+b1 | a2+a3 | b2+b3)
__device__ __forceinline__ unsigned long long fold16(unsigned long long a, unsigned long long b)
{
    const auto l = __builtin_amdgcn_permlane16_swap((uint32_t)a, (uint32_t)b, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap((uint32_t)(a >> 32), (uint32_t)(b >> 32), false, false);
    return (((unsigned long long)h[0] << 32) | l[0]) + (((unsigned long long)h[1] << 32) | l[1]);
}

// Inclusive prefix sums inside each row of 16 lanes.  The 64-bit values (< 2^49: up to four
// lanes' worth of at most 32 squares of |coef| < 2^21) are scanned as a 24-bit low part and
// a high part, both of which stay below 2^31 over 16 lanes.
struct RowScan {
    int lo, hi;
    __device__ __forceinline__ explicit RowScan(unsigned long long a)
    {
        lo = (int)(a & 0xffffffull);
        hi = (int)(a >> 24);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            int &v = k ? hi : lo;
            v += dpp_mov<0x111>(v);
            v += dpp_mov<0x112>(v);
            v += dpp_mov<0x114>(v);
            v += dpp_mov<0x118>(v);
        }
    }
};

// which band total a lane ends up with after the transposed reduction of phase F:
// low nibble = band, high nibble = source (0 acc0 prefix, 1 prefix - prefix[lane-2],
// 2 prefix - prefix[lane-8], 3 Y2 moved down one lane, 4 Y1 moved down two lanes); 0xff = none
__device__ const uint8_t LANE_BAND[64] = {
    0xff, 0xff, 0xff, 0x00, 0xff, 0x11, 0xff, 0x12, 0xff, 0xff, 0xff, 0xff, 0xff, 0x49, 0x37, 0x23,
    0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0x04, 0xff, 0xff, 0xff, 0xff, 0xff, 0x4b, 0xff, 0x25,
    0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0x4a, 0x38, 0xff,
    0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0x4c, 0x36, 0xff,
};

constexpr int RPITCH = 69;   // dwords per row of the raster buffer: the scan-order gather is
                             // ~2.5x conflict cycles here vs 10.8x at 65 (tools: LDS bank model)
static_assert(64 * RPITCH * 4 <= LDS_BYTES, "raster buffer must fit in the tile");

// Back half of the T-stage for one 64x64 block-plane whose lapped samples sit in x[] with
// lane = column, x[k] = row k: phases D (column DCT + transpose), E (row DCT), F (scan-order
// gather, band energies, gains, record) and the coefficient stores.  Shared by the one-block
// kernel and the column-walking kernel.  `xb` is the workgroup's LDS, free for reuse on entry
// once the caller's barrier has passed.
template <bool WRITE_COEF, bool LOAD_LUT>
__device__ __forceinline__ void tstage_back_half(int (&x)[64], uint4 (&lut)[8], int *xb,
                                                 const FFV2TStageArgs &a, const int f, const int bp,
                                                 const int lane, const int lane_info)
{
    const FFV2Geom &g = a.g;
    // ---- phase D: column transforms (lane = column), then transpose through LDS ----
#pragma unroll
    for (int k = 0; k < 64; k++) {
        // keep the 16-bit provenance from the optimiser: with known-bits it rewrites
        // __mul24 into a plain 32-bit multiply and then selects v_mul_lo_u32
        asm("" : "+v"(x[k]));
    }
    wave_lds_fence();                                         // tile is dead: LDS becomes int32 [64][65]
#ifndef FFV2_ASM_NET
#define FFV2_MULRS FFV2_MULRS_NOOVF
    FDCT64_NET(x);
#undef FFV2_MULRS
#else
    FDCT64_ASM_COL(x);      // the same network as one asm block, constants streamed by s_load (tools/gen_asm_net.py)
#endif
#pragma unroll
    for (int v = 0; v < 64; v++) xb[lane * XPITCH + v] = x[OUTR[v]];      // tmp[64*col + v], ffv2.c:4957
    if (LOAD_LUT) {
        // the walking kernel fetches the scan table per block-plane, one transform ahead of its
        // use (L1/L2 hits; held across the whole loop it would cost 32 registers)
#pragma unroll
        for (int i = 0; i < 8; i++)
            lut[i] = reinterpret_cast<const uint4 *>(a.lds_scan)[i * 64 + lane];
    }
    wave_lds_fence();

    // ---- phase E: row transforms (lane = vertical frequency v) ----
#pragma unroll
    for (int k = 0; k < 64; k++) x[k] = xb[k * XPITCH + lane];            // tmp + v, stride 64, ffv2.c:4959
    wave_lds_fence();
#ifndef FFV2_ASM_NET
#define FFV2_MULRS FFV2_MULRS_WRAP
    FDCT64_NET(x);
#undef FFV2_MULRS
#else
    FDCT64_ASM_ROW(x);
#endif
#pragma unroll
    for (int u = 0; u < 64; u++) xb[lane * RPITCH + u] = x[OUTR[u]];      // dst[64*v + u]
    wave_lds_fence();

    // ---- phase F: scan-order gather, band energies, gains, coalesced 16-byte stores ----
    // lane owns coding indices q = 256*j + 4*lane + k (j < 16, k < 4) -> x[4j + k].
    // Bands (ffv2.c:100-120): q=0 "DC" slot, then [1,16) [16,24) [24,32) [32,64) [64,96)
    // [96,128) [128,256) | [256,384) [384,512) | [512,1024) [1024,1536) [1536,2048) [2048,4096].
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t w[4] = { lut[i].x, lut[i].y, lut[i].z, lut[i].w };
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t off = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu);
            x[i * 8 + e] = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(xb) + off);
        }
    }
    auto sq = [](int c) { return (unsigned long long)((long long)c * c); };
    unsigned long long acc[6];
    acc[0] = sq(x[1]) + sq(x[2]) + sq(x[3]) + (lane == 0 ? 0ull : sq(x[0]));
    acc[1] = sq(x[4]) + sq(x[5]) + sq(x[6]) + sq(x[7]);
    {
        constexpr int XB[5] = { 8, 16, 24, 32, 64 };         // x[] ranges of bands 9..12
#pragma unroll
        for (int b = 0; b < 4; b++) {
            unsigned long long s = 0;
#pragma unroll
            for (int i = XB[b]; i < XB[b + 1]; i++) s += sq(x[i]);
            acc[2 + b] = s;
        }
    }
    // Band energies by a transposed reduction: every fold halves the number of registers
    // while it sums across lanes, and each band total ends in a lane of its own, which then
    // codes the gain - no SGPR round trip.
    //   Y1 rows (16 lanes each) = acc2 | acc4 | acc3 | acc5     -> bands 9, 11, 10, 12
    //   Y2 rows                 = acc1 rows 0+1 | - | acc1 rows 2+3 | acc0 rows 2+3 -> bands 7, -, 8, 6
    //   Y3                      = acc0 (rows 0 and 1: prefix sums give bands 0..5)
    unsigned long long my;
    {
        const unsigned long long y1 = fold16(fold32(acc[2], acc[3]), fold32(acc[4], acc[5]));
        const unsigned long long y2 = fold16(acc[1], acc[0]);
        const RowScan s1(y1), s2(y2), s3(acc[0]);
        // lane <- lane-2 / lane-8 differences of the acc0 prefix, lane <- lane+2 / lane+1 moves
        const int d2lo = s3.lo - dpp_mov<0x112>(s3.lo), d2hi = s3.hi - dpp_mov<0x112>(s3.hi);
        const int d8lo = s3.lo - dpp_mov<0x118>(s3.lo), d8hi = s3.hi - dpp_mov<0x118>(s3.hi);
        const int m1lo = dpp_mov<0x102>(s1.lo), m1hi = dpp_mov<0x102>(s1.hi);
        const int m2lo = dpp_mov<0x101>(s2.lo), m2hi = dpp_mov<0x101>(s2.hi);
        const int sel = lane_info >> 4;
        int lo = s3.lo, hi = s3.hi;
        if (sel == 1) { lo = d2lo; hi = d2hi; }
        if (sel == 2) { lo = d8lo; hi = d8hi; }
        if (sel == 3) { lo = m2lo; hi = m2hi; }
        if (sel == 4) { lo = m1lo; hi = m1hi; }
        my = (unsigned long long)(((long long)hi << 24) + lo);
    }
    const bool has_band = lane_info != 0xff;
    const int band = lane_info & 15;
    if (a.energy && has_band)
        a.energy[((size_t)f * g.nblk + bp) * FFV2_NUM_BANDS + band] = (long long)my;

    if (a.codes) {
        // band lanes: gains; lane 0: the "DC" slot (it holds coding index 0); the reference's
        // last band also squares the int32 that follows temp2[] (phantom W)
        if (band == 12 && has_band && a.W) {
            const long long w = a.W[(size_t)f * g.nblk + bp];
            my += (unsigned long long)(w * w);
        }
        bool oot = false;
        uint32_t val = 0;
        int nb = 0;
        if (has_band) {
            val = coded_gain((long long)my, a.gain_thr, a.gain_n, oot);
            nb = golomb_len(val);
        } else if (lane == 0) {
            const int c0 = x[0];
            val = (uint32_t)c0;
            const uint32_t mag = c0 < 0 ? (uint32_t)(-(long long)c0) : (uint32_t)c0;
            nb = golomb_len(mag) + (c0 != 0);
        }
        nb = wave_iscan(nb);                                   // lane 63: bits of the whole record
        uint32_t *rec = a.codes + ((size_t)f * g.nblk + bp) * FFV2_CODES_PER_BP;
        if (has_band)        rec[1 + band] = val;
        else if (lane == 0)  rec[0] = val;
        else if (lane == 63) { rec[14] = (uint32_t)nb; a.bitcnt[(size_t)f * g.nblk + bp] = (uint32_t)nb; }
        else if (lane == 1)  rec[15] = 0;
        if (__any(oot) && lane == 0)
            atomicMin(&a.status[f], -34);
    }
    // coefficients last: 16 stores of 1 KiB per wave, nothing waits on them
    if (WRITE_COEF) {
        // streamed past the caches (nontemporal): 16 KiB per block-plane written once and not read
        // again by this kernel would otherwise push the neighbours' shared halo rows out of L2
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        i32x4 *cp = reinterpret_cast<i32x4 *>(a.coef + ((size_t)f * g.nblk + bp) * 4096);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const i32x4 v = { x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3] };
            __builtin_nontemporal_store(v, &cp[j * 64 + lane]);
        }
    }
    // this block-plane's share of the frame's packet buffer, cleared for the E-stage's atomic ORs
    if (a.zero) {
        const uint32_t per = (a.zero_stride_dw + (uint32_t)g.nblk - 1u) / (uint32_t)g.nblk;
        uint32_t *z = a.zero + (size_t)f * a.zero_stride_dw;
        for (uint32_t d = (uint32_t)lane; d < per; d += 64) {
            const uint32_t i = (uint32_t)bp * per + d;
            if (i < a.zero_stride_dw) z[i] = 0;
        }
    }
}

template <int BPS, bool WRITE_COEF>
__global__ __launch_bounds__(64, 2) void ffv2_tstage_kernel(const FFV2TStageArgs a)
{
    __shared__ int4 lds_raw[LDS_BYTES / 16];
    int16_t  *tile = reinterpret_cast<int16_t *>(lds_raw);
    int      *xb   = reinterpret_cast<int *>(lds_raw);

    const FFV2Geom &g = a.g;
    const int lane = threadIdx.x;
    FFV2_PHASE_BEGIN

    // XCD-aware block id: workgroups b and b+8 share an XCD (round-robin dispatch; gridDim.x
    // is a multiple of 8 and blockIdx.y = frame), so give each XCD one contiguous run of the
    // frame's block-planes -> neighbouring tiles, which share their 32-sample halos, meet in
    // the same L2.  All of it 32-bit and scalar: no 64-bit divide on the way in.
    const uint32_t chunk = gridDim.x >> 3;
    const uint32_t ubp = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
    if (ubp >= (uint32_t)g.nblk) return;
    const int f   = (int)blockIdx.y;
    const int bp  = (int)ubp;
    // divisions by multiply-high with the host's reciprocals (exact: dividend * divisor < 2^32)
    const int sb  = g.planes > 1 ? (int)__umulhi(ubp, g.inv_planes) : bp;
    const int p   = bp - sb * g.planes;
    const int sby = g.nsx > 1 ? (int)__umulhi((uint32_t)sb, g.inv_nsx) : sb;
    const int sbx = sb - sby * g.nsx;

    const uint8_t *plane = a.frames + (size_t)f * g.frame_stride + (size_t)p * g.plane_stride;
    const int sh = 12 - g.depth;
    const int x_org = sbx * 64 - 16, y_org = sby * 64 - 16;
    const bool seamL = sbx > 0, seamR = sbx + 1 < g.nsx;
    const bool seamT = sby > 0, seamB = sby + 1 < g.nsy;
    const int grid_h = g.nsy * 64;

    // scan table for phase F, fetched now so that nothing but stores is outstanding
    // there: [i][lane] 16-byte rows of 8 byte-offsets into the raster buffer, entry e of
    // row i is coding index q = 256*(2i + e/4) + 4*lane + e%4
    uint4 lut[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
        lut[i] = reinterpret_cast<const uint4 *>(a.lds_scan)[i * 64 + lane];
    const int lane_info = LANE_BAND[lane];

    FFV2_PHASE_MARK(0);
    // ---- phase A: coalesced 16-byte reads of the 96x96 halo tile -> int16 LDS ----
    {
        constexpr int EPV = 16 / BPS;              // samples per 16-byte vector
        constexpr int VPR = TILE / EPV;            // vectors per tile row (6 | 12)
        constexpr int PER_LANE = TILE * VPR / 64;  // 9 | 18
        constexpr int RSTEP = 64 / VPR, CSTEP = 64 % VPR;   // vector index += 64  <=>  row += RSTEP, col += CSTEP
        const bool inside = (x_org >= 0) & (x_org + TILE <= g.width) & (y_org >= 0) & (y_org + TILE <= g.height);
        uint4 v[PER_LANE];
        uint32_t bad = 0;
        const uint32_t lsmul = (1u << sh) * 0x00010001u;
        // vector index of iteration `it` is it*64 + lane.  64 = RSTEP*VPR + 4 and 3*4 is a
        // multiple of VPR (6 | 12), so (row, column) repeat every three iterations, shifted
        // down by 192/VPR rows: three address phases, everything else is a constant step.
        static_assert((3 * CSTEP) % VPR == 0, "three-phase addressing");
        constexpr int RSTEP3 = 3 * 64 / VPR;
        uint32_t goff[3];
        int doff3[3], row3[3], col3[3];
        {
            int r = lane / VPR, cv = lane - (lane / VPR) * VPR;
#pragma unroll
            for (int ph = 0; ph < 3; ph++) {
                // wrapping 32-bit arithmetic: only dereferenced where the vector is inside the picture
                goff[ph]  = (uint32_t)(y_org + r) * (uint32_t)g.row_pitch + (uint32_t)(x_org + cv * EPV) * BPS;
                doff3[ph] = r * TPITCH + cv * EPV;          // int16 units
                row3[ph]  = y_org + r;
                col3[ph]  = x_org + cv * EPV;
                r += RSTEP; cv += CSTEP;
                if (cv >= VPR) { cv -= VPR; r++; }
            }
        }
        const uint32_t gstep = RSTEP3 * (uint32_t)g.row_pitch;
        if (inside) {                              // wave-uniform: no masking needed
#pragma unroll
            for (int it = 0; it < PER_LANE; it++)
                v[it] = *reinterpret_cast<const uint4 *>(plane + (goff[it % 3] + (uint32_t)(it / 3) * gstep));
        } else {
            // picture edge: everything outside is 0 after the level shift (ffv2enc.c:69-71), i.e.
            // mid-grey before it -- so vectors outside are not loaded but preset to mid-grey and
            // then take the same conversion as the interior.  x_org is a multiple of 16, so a
            // vector is inside or outside as a whole except the one the right edge may cut.
            const uint32_t grey = BPS == 1 ? 0x80808080u : (2048u >> sh) * 0x00010001u;
            bool cok[3];
#pragma unroll
            for (int ph = 0; ph < 3; ph++) cok[ph] = (uint32_t)col3[ph] < (uint32_t)g.width;
#pragma unroll
            for (int it = 0; it < PER_LANE; it++) {
                v[it] = make_uint4(grey, grey, grey, grey);
                const bool ok = cok[it % 3] & ((uint32_t)(row3[it % 3] + (it / 3) * RSTEP3) < (uint32_t)g.height);
                if (ok)
                    v[it] = *reinterpret_cast<const uint4 *>(plane + (goff[it % 3] + (uint32_t)(it / 3) * gstep));
            }
            if (g.width % EPV) {                   // the right edge cuts a vector: grey behind it
                uint32_t keep[3][4];               // bits of the vector that are picture
#pragma unroll
                for (int ph = 0; ph < 3; ph++) {
                    int nbits = (g.width - col3[ph]) * (8 * BPS);
                    nbits = nbits < 0 ? 0 : (nbits > 128 ? 128 : nbits);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int b = nbits - 32 * k;
                        keep[ph][k] = b >= 32 ? 0xffffffffu : (b <= 0 ? 0u : (1u << b) - 1u);
                    }
                }
#pragma unroll
                for (int it = 0; it < PER_LANE; it++) {
                    const uint32_t *m = keep[it % 3];
                    v[it].x = (v[it].x & m[0]) | (grey & ~m[0]);
                    v[it].y = (v[it].y & m[1]) | (grey & ~m[1]);
                    v[it].z = (v[it].z & m[2]) | (grey & ~m[2]);
                    v[it].w = (v[it].w & m[3]) | (grey & ~m[3]);
                }
            }
        }
        uint32_t seen = 0;                         // OR of every sample word: one depth test at the end
#pragma unroll
        for (int it = 0; it < PER_LANE; it++) {
            const uint32_t w[4] = { v[it].x, v[it].y, v[it].z, v[it].w };
            int4 *dst = reinterpret_cast<int4 *>(tile + doff3[it % 3] + (it / 3) * RSTEP3 * TPITCH);
            if (BPS == 1) {
                uint32_t o[8];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    o[2 * k]     = pk_level_shift(__builtin_amdgcn_perm(0, w[k], 0x0c010c00u), lsmul);
                    o[2 * k + 1] = pk_level_shift(__builtin_amdgcn_perm(0, w[k], 0x0c030c02u), lsmul);
                }
                dst[0] = make_int4(o[0], o[1], o[2], o[3]);
                dst[1] = make_int4(o[4], o[5], o[6], o[7]);
            } else {
                uint32_t o[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    seen |= w[k];
                    o[k] = pk_level_shift(w[k], lsmul);
                }
                dst[0] = make_int4(o[0], o[1], o[2], o[3]);
            }
        }
        if (BPS == 2) bad = seen & ~(((1u << g.depth) - 1u) * 0x00010001u);
        if (__any(bad != 0) && lane == 0)
            atomicMin(&a.status[f], -34);            // FFV2AMD_ERR_RANGE
    }
    wave_lds_fence();
    FFV2_PHASE_MARK(1);

    // ---- phase B: horizontal lapping on the two vertical seams (rows in parallel) ----
    // 96 rows x 2 seams = 192 filter instances = 3 rounds of 64 lanes.
#pragma unroll 1
    for (int round = 0; round < 3; round++) {
        const int inst = round * 64 + lane;
        const bool right = inst >= TILE;
        const int r = right ? inst - TILE : inst;
        const int y = y_org + r;
        const bool act = (right ? seamR : seamL) & (y >= 0) & (y < grid_h);
        if (__any(act)) {
            int x[32];
            const int4 *src = reinterpret_cast<const int4 *>(tile + r * TPITCH + (right ? 64 : 0));
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int4 w = src[q];
                x[8 * q + 0] = lo16(w.x); x[8 * q + 1] = hi16(w.x);
                x[8 * q + 2] = lo16(w.y); x[8 * q + 3] = hi16(w.y);
                x[8 * q + 4] = lo16(w.z); x[8 * q + 5] = hi16(w.z);
                x[8 * q + 6] = lo16(w.w); x[8 * q + 7] = hi16(w.w);
            }
            lap32(x);
            if (act) {
                // keep the half that lies inside this block: left seam -> taps 16..31
                // (tile cols 16..31), right seam -> taps 0..15 (tile cols 64..79)
                uint32_t o[8];
#pragma unroll
                for (int k = 0; k < 8; k++)
                    o[k] = right ? pack16(x[2 * k], x[2 * k + 1]) : pack16(x[16 + 2 * k], x[17 + 2 * k]);
                int4 *dst = reinterpret_cast<int4 *>(tile + r * TPITCH + (right ? 64 : 16));
                dst[0] = make_int4(o[0], o[1], o[2], o[3]);
                dst[1] = make_int4(o[4], o[5], o[6], o[7]);
            }
        }
    }
    wave_lds_fence();
    FFV2_PHASE_MARK(2);

    // ---- phase C: vertical lapping on the two horizontal seams, lane = column.  The lapped
    // rows stay in registers: they are this column's inputs to phase D, which has the same
    // lane = column orientation, so nothing goes back through LDS. ----
    int x[64];
    {
        const int16_t *colp = tile + 16 + lane;
        if (seamT) {                                         // wave-uniform
            int t[32];
#pragma unroll
            for (int k = 0; k < 32; k++) t[k] = colp[k * TPITCH];
            lap32(t);
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = t[16 + k];                  // tile rows 16..31
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = colp[(16 + k) * TPITCH];
        }
        if (seamB) {
            int t[32];
#pragma unroll
            for (int k = 0; k < 32; k++) t[k] = colp[(64 + k) * TPITCH];
            lap32(t);
#pragma unroll
            for (int k = 0; k < 16; k++) x[48 + k] = t[k];                  // tile rows 64..79
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) x[48 + k] = colp[(64 + k) * TPITCH];
        }
#pragma unroll
        for (int k = 16; k < 48; k++) x[k] = colp[(16 + k) * TPITCH];      // rows 32..63: untouched by the seams
    }
    FFV2_PHASE_MARK(3);

    tstage_back_half<WRITE_COEF, false>(x, lut, xb, a, f, bp, lane, lane_info);
    FFV2_PHASE_MARK(7);
    FFV2_PHASE_END;
}

// ---------------------------------------------------------------------------
// T-stage, column-walking variant
// ---------------------------------------------------------------------------
// The one-block kernel above filters every seam from both sides: per block-plane 3 rounds of
// horizontal lapping (96 halo rows x 2 seams) and 2 of vertical lapping, against the 2 rounds a
// block strictly owns.  Here ONE wavefront walks a run of block-planes DOWN a column of
// superblocks (same frame, plane and sbx; sby ascending):
//   * each step loads only the 64 new picture rows [64j+16, 64j+80) x 96 columns (the rows above
//     came with the previous step), laps them horizontally (64 rows x 2 seams = 2 rounds) and
//     filters the ONE horizontal seam below the block (1 round);
//   * the lower half of that seam's output (16 rows, lane = column) stays in registers as the
//     first 16 input rows of the next block -- the seam is never filtered twice;
//   * 3 lapping rounds per block-plane instead of 5, a third fewer tile loads, and the next
//     step's loads are issued before this step's transforms and coefficient stores (vmcnt is
//     in-order: a load issued behind the stores would wait for them).
// A run starts (and restarts at the top of the next column) with a pre-step that loads and
// laps the 32 rows around the seam above the first block.  The block-planes of a launch are
// dealt in equal runs to exactly as many wavefronts as the chip holds at once
// (ffv2_launch_tstage), so there is no dispatch tail.
constexpr int WLDS_BYTES = 64 * RPITCH * 4;        // raster buffer is the largest tenant (17 664 B)
static_assert(64 * TPITCH * 2 <= WLDS_BYTES && 64 * XPITCH * 4 <= WLDS_BYTES, "walk kernel LDS");

template <int BPS> struct WalkGeom {
    static constexpr int EPV = 16 / BPS;           // samples per 16-byte vector
    static constexpr int VPR = TILE / EPV;         // vectors per 96-sample tile row (6 | 12)
    static constexpr int RSTEP = 64 / VPR, CSTEP = 64 % VPR;
    static constexpr int RSTEP3 = 3 * 64 / VPR;    // rows per three iterations
    static_assert((3 * CSTEP) % VPR == 0, "three-phase addressing");
};

// A tile request: NROWS x 96 samples at picture position (x_org, y0), vector index it*64 + lane,
// row-major.  walk_issue sends out the 16-byte loads and returns at once; walk_fix, called when
// the data is needed, turns everything outside the picture into mid-grey (= 0 after the level
// shift, ffv2enc.c:69-71).  No load is predicated -- a vector outside the picture reads the
// plane's first 16 bytes instead and is replaced afterwards -- so that the compiler never has
// to wait between loads.  r3/c3: the lane's three (row, first column) address phases.
struct WalkReq {
    uint32_t okmask;        // bit `it`: vector `it` of this lane lies (at least partly) inside the picture
    bool inside;            // wave-uniform: the whole tile does
};

template <int BPS, int NROWS>
__device__ __forceinline__ WalkReq walk_issue(uint4 (&v)[NROWS * WalkGeom<BPS>::VPR / 64], const uint8_t *plane,
                                              const FFV2Geom &g, const int x_org, const int y0,
                                              const int (&r3)[3], const int (&c3)[3])
{
    using G = WalkGeom<BPS>;
    constexpr int PER_LANE = NROWS * G::VPR / 64;
    static_assert(PER_LANE % 3 == 0, "whole address periods");
    uint32_t goff[3];
#pragma unroll
    for (int ph = 0; ph < 3; ph++)
        goff[ph] = (uint32_t)(y0 + r3[ph]) * (uint32_t)g.row_pitch + (uint32_t)(x_org + c3[ph]) * BPS;
    const uint32_t gstep = G::RSTEP3 * (uint32_t)g.row_pitch;
    WalkReq q;
    q.inside = (x_org >= 0) & (x_org + TILE <= g.width) & (y0 >= 0) & (y0 + NROWS <= g.height);
    q.okmask = 0xffffffffu;
    uint32_t off[PER_LANE];                        // 32-bit offsets from the (uniform) plane base
#pragma unroll
    for (int it = 0; it < PER_LANE; it++) off[it] = goff[it % 3] + (uint32_t)(it / 3) * gstep;
    if (!q.inside) {
        bool cok[3];
#pragma unroll
        for (int ph = 0; ph < 3; ph++) cok[ph] = (uint32_t)(x_org + c3[ph]) < (uint32_t)g.width;
        uint32_t m = 0;
#pragma unroll
        for (int it = 0; it < PER_LANE; it++) {
            const bool ok = cok[it % 3] & ((uint32_t)(y0 + r3[it % 3] + (it / 3) * G::RSTEP3) < (uint32_t)g.height);
            off[it] = ok ? off[it] : 0u;
            m |= (ok ? 1u : 0u) << it;
        }
        q.okmask = m;
    }
#pragma unroll
    for (int it = 0; it < PER_LANE; it++)
        v[it] = *reinterpret_cast<const uint4 *>(plane + off[it]);
    return q;
}

template <int BPS, int NROWS>
__device__ __forceinline__ void walk_fix(uint4 (&v)[NROWS * WalkGeom<BPS>::VPR / 64], const WalkReq q,
                                         const FFV2Geom &g, const int x_org, const int sh, const int (&c3)[3])
{
    using G = WalkGeom<BPS>;
    constexpr int PER_LANE = NROWS * G::VPR / 64;
    if (q.inside) return;
    const uint32_t grey = BPS == 1 ? 0x80808080u : (2048u >> sh) * 0x00010001u;
#pragma unroll
    for (int it = 0; it < PER_LANE; it++) {
        const bool ok = (q.okmask >> it) & 1u;
        v[it].x = ok ? v[it].x : grey; v[it].y = ok ? v[it].y : grey;
        v[it].z = ok ? v[it].z : grey; v[it].w = ok ? v[it].w : grey;
    }
    if (g.width % G::EPV) {                        // the right edge cuts a vector: grey behind it
#pragma unroll
        for (int ph = 0; ph < 3; ph++) {
            int nbits = (g.width - (x_org + c3[ph])) * (8 * BPS);
            nbits = nbits < 0 ? 0 : (nbits > 128 ? 128 : nbits);
            uint32_t keep[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int b = nbits - 32 * k;
                keep[k] = b >= 32 ? 0xffffffffu : (b <= 0 ? 0u : (1u << b) - 1u);
            }
#pragma unroll
            for (int it = ph; it < PER_LANE; it += 3) {
                v[it].x = (v[it].x & keep[0]) | (grey & ~keep[0]);
                v[it].y = (v[it].y & keep[1]) | (grey & ~keep[1]);
                v[it].z = (v[it].z & keep[2]) | (grey & ~keep[2]);
                v[it].w = (v[it].w & keep[3]) | (grey & ~keep[3]);
            }
        }
    }
}

// level shift + int16 store of a loaded tile; returns the OR of all 16-bit sample words
template <int BPS, int NROWS>
__device__ __forceinline__ uint32_t walk_store(const uint4 (&v)[NROWS * WalkGeom<BPS>::VPR / 64], int16_t *tile,
                                               const int sh, const int (&r3)[3], const int (&c3)[3])
{
    using G = WalkGeom<BPS>;
    constexpr int PER_LANE = NROWS * G::VPR / 64;
    const uint32_t lsmul = (1u << sh) * 0x00010001u;
    uint32_t seen = 0;
#pragma unroll
    for (int it = 0; it < PER_LANE; it++) {
        const uint32_t w[4] = { v[it].x, v[it].y, v[it].z, v[it].w };
        int4 *dst = reinterpret_cast<int4 *>(tile + (r3[it % 3] + (it / 3) * G::RSTEP3) * TPITCH + c3[it % 3]);
        if (BPS == 1) {
            uint32_t o[8];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                o[2 * k]     = pk_level_shift(__builtin_amdgcn_perm(0, w[k], 0x0c010c00u), lsmul);
                o[2 * k + 1] = pk_level_shift(__builtin_amdgcn_perm(0, w[k], 0x0c030c02u), lsmul);
            }
            dst[0] = make_int4(o[0], o[1], o[2], o[3]);
            dst[1] = make_int4(o[4], o[5], o[6], o[7]);
        } else {
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                seen |= w[k];
                o[k] = pk_level_shift(w[k], lsmul);
            }
            dst[0] = make_int4(o[0], o[1], o[2], o[3]);
        }
    }
    return seen;
}

// one horizontal filter instance: tile row r, left (tile columns 0..31, keeps 16..31) or
// right seam (columns 64..95, keeps 64..79)
__device__ __forceinline__ void walk_hlap(int16_t *tile, const int r, const bool right, const bool act)
{
    int x[32];
    const int4 *src = reinterpret_cast<const int4 *>(tile + r * TPITCH + (right ? 64 : 0));
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int4 w = src[q];
        x[8 * q + 0] = lo16(w.x); x[8 * q + 1] = hi16(w.x);
        x[8 * q + 2] = lo16(w.y); x[8 * q + 3] = hi16(w.y);
        x[8 * q + 4] = lo16(w.z); x[8 * q + 5] = hi16(w.z);
        x[8 * q + 6] = lo16(w.w); x[8 * q + 7] = hi16(w.w);
    }
    lap32(x);
    if (act) {
        uint32_t o[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            o[k] = right ? pack16(x[2 * k], x[2 * k + 1]) : pack16(x[16 + 2 * k], x[17 + 2 * k]);
        int4 *dst = reinterpret_cast<int4 *>(tile + r * TPITCH + (right ? 64 : 16));
        dst[0] = make_int4(o[0], o[1], o[2], o[3]);
        dst[1] = make_int4(o[4], o[5], o[6], o[7]);
    }
}

// Runs handed to the workgroups of one launch.  Every column of superblocks (frame, plane, sbx)
// is cut into the same segments of decreasing length len[t] starting at superblock row joff[t];
// tier t = segment t of every column, and workgroups take runs tier by tier in id order: long
// runs first (few pre-steps), ever shorter ones behind them, so that the wavefront slots of the
// chip -- refilled by the dispatcher as workgroups retire -- drain at about the same time.
// Inside a tier consecutive runs are horizontally adjacent columns, and the workgroups of one
// XCD (b, b+8, ...) get a contiguous stretch of them: neighbours that share a 32-sample halo
// walk down side by side through the same L2 and the same DRAM pages.
#define FFV2_WALK_MAX_TIERS 24
struct WalkTiers {
    uint32_t ntiers, ncols, ncols8;                 // ncols8 = ncols rounded up to a multiple of 8
    uint16_t len[FFV2_WALK_MAX_TIERS], joff[FFV2_WALK_MAX_TIERS];
};

template <int BPS, bool WRITE_COEF>
__global__ __launch_bounds__(64, 2) void ffv2_tstage_walk_kernel(const FFV2TStageArgs a, const uint32_t total,
                                                                 const WalkTiers tiers)
{
    using G = WalkGeom<BPS>;
    constexpr int PL_MAIN = 64 * G::VPR / 64, PL_PRE = 32 * G::VPR / 64;
    __shared__ int4 lds_raw[WLDS_BYTES / 16];
    int16_t  *tile = reinterpret_cast<int16_t *>(lds_raw);
    int      *xb   = reinterpret_cast<int *>(lds_raw);

    const FFV2Geom &g = a.g;
    const int lane = threadIdx.x;

    uint32_t n, n_end;
    {
        uint32_t b = blockIdx.x, t = 0;
        while (b >= tiers.ncols8) { b -= tiers.ncols8; t++; }
        const uint32_t c = (b & 7u) * (tiers.ncols8 >> 3) + (b >> 3);
        if (c >= tiers.ncols || t >= tiers.ntiers) return;
        n = c * (uint32_t)g.nsy + tiers.joff[t];
        n_end = n + tiers.len[t];
    }

    const int lane_info = LANE_BAND[lane];
    const int sh = 12 - g.depth;
#ifdef FFV2_PHASE_TIMING
    // slots: 0 start (100 MHz), 1 HW_ID, 2 life (100 MHz), 3 front-half ticks, 4 back-half ticks,
    // 5 pre-step ticks, 6 block-planes done
    const unsigned long long wrt0_ = __builtin_amdgcn_s_memrealtime();
    unsigned long long wt_ = __builtin_amdgcn_s_memtime(), wacc_[3] = {};
    unsigned wdone_ = 0;
#define WALK_MARK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); wacc_[i] += t_ - wt_; wt_ = t_; } while (0)
#else
#define WALK_MARK(i)
#endif

    // the lane's three (row, column) address phases of a tile load, see walk_load
    int r3[3], c3[3];
    {
        int r = lane / G::VPR, cv = lane - (lane / G::VPR) * G::VPR;
#pragma unroll
        for (int ph = 0; ph < 3; ph++) {
            r3[ph] = r; c3[ph] = cv * G::EPV;
            r += G::RSTEP; cv += G::CSTEP;
            if (cv >= G::VPR) { cv -= G::VPR; r++; }
        }
    }

    // block-planes are numbered column by column: n = (((f*planes + p)*nsx + sbx)*nsy + sby
    uint32_t col = n / (uint32_t)g.nsy;
    int j = (int)(n - col * (uint32_t)g.nsy);
    bool fresh = true;                             // a pre-step is due (start of the run / top of a column)
    int f = 0, p = 0, sbx = 0, x_org = 0;
    bool seamL = false, seamR = false;
    const uint8_t *plane = nullptr;
    uint4 v[PL_MAIN];                              // the step's 64 new rows, requested one step ahead
    WalkReq vq{0xffffffffu, true};
    int carry[16];                                 // rows 0..15 of the current block, lane = column
    const uint32_t himask = ~(((1u << g.depth) - 1u) * 0x00010001u);

#pragma unroll 1
    for (; n < n_end; n++) {
        uint32_t seen = 0;
        if (fresh) {
            const uint32_t fp = col / (uint32_t)g.nsx;
            sbx = (int)(col - fp * (uint32_t)g.nsx);
            f = (int)(fp / (uint32_t)g.planes);
            p = (int)(fp - (uint32_t)f * (uint32_t)g.planes);
            plane = a.frames + (size_t)f * g.frame_stride + (size_t)p * g.plane_stride;
            x_org = sbx * 64 - 16;
            seamL = sbx > 0; seamR = sbx + 1 < g.nsx;
            // pre-step: the 32 rows around the seam above block j (all grey above the picture)
            uint4 pv[PL_PRE];
            const WalkReq pq = walk_issue<BPS, 32>(pv, plane, g, x_org, j * 64 - 16, r3, c3);
            vq = walk_issue<BPS, 64>(v, plane, g, x_org, j * 64 + 16, r3, c3);
            walk_fix<BPS, 32>(pv, pq, g, x_org, sh, c3);
            seen |= walk_store<BPS, 32>(pv, tile, sh, r3, c3);
            wave_lds_fence();
            {
                const bool right = lane >= 32;
                const bool act = right ? seamR : seamL;
                if (__any(act)) walk_hlap(tile, lane & 31, right, act);
            }
            wave_lds_fence();
            const int16_t *colp = tile + 16 + lane;
            if (j > 0) {
                int t[32];
#pragma unroll
                for (int k = 0; k < 32; k++) t[k] = colp[k * TPITCH];
                lap32(t);
#pragma unroll
                for (int k = 0; k < 16; k++) carry[k] = t[16 + k];
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) carry[k] = colp[(16 + k) * TPITCH];
            }
            wave_lds_fence();
            fresh = false;
            // the first block's rows have landed by now; say so here, so that the loop's common
            // path below carries no pending load (else every step would wait, through the in-order
            // vmcnt, for the previous step's coefficient stores)
#pragma unroll
            for (int it = 0; it < PL_MAIN; it++) asm volatile("" :: "v"(v[it].x), "v"(v[it].y), "v"(v[it].z), "v"(v[it].w));
            WALK_MARK(2);
        }
        // ---- the 64 new rows: picture rows [64j+16, 64j+80) = tile rows 0..63 ----
        walk_fix<BPS, 64>(v, vq, g, x_org, sh, c3);
        seen |= walk_store<BPS, 64>(v, tile, sh, r3, c3);
        if (BPS == 2 && __any((seen & himask) != 0) && lane == 0)
            atomicMin(&a.status[f], -34);            // FFV2AMD_ERR_RANGE
        wave_lds_fence();
        if (seamL) walk_hlap(tile, lane, false, true);
        if (seamR) walk_hlap(tile, lane, true, true);
        wave_lds_fence();
        int x[64];
        {
            const int16_t *colp = tile + 16 + lane;
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = carry[k];
#pragma unroll
            for (int k = 0; k < 32; k++) x[16 + k] = colp[k * TPITCH];
            if (j + 1 < g.nsy) {                             // the seam below: filtered once, shared
                int t[32];
#pragma unroll
                for (int k = 0; k < 32; k++) t[k] = colp[(32 + k) * TPITCH];
                lap32(t);
#pragma unroll
                for (int k = 0; k < 16; k++) { x[48 + k] = t[k]; carry[k] = t[16 + k]; }
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) x[48 + k] = colp[(32 + k) * TPITCH];
            }
        }
        const int bp = (j * g.nsx + sbx) * g.planes + p;
        // next step's rows: requested now, ahead of the transforms and the coefficient stores
        j++;
        if (j == g.nsy) { j = 0; col++; fresh = true; }
        if (!fresh && n + 1 < n_end) {
            vq = walk_issue<BPS, 64>(v, plane, g, x_org, j * 64 + 16, r3, c3);
        } else {
            // nothing carried into the next step: end the old tile's live range here, so that the
            // loads above can land in the very registers the loop carries (no wait-and-copy)
#pragma unroll
            for (int it = 0; it < PL_MAIN; it++)     // "defined" without an instruction
                asm("" : "=v"(v[it].x), "=v"(v[it].y), "=v"(v[it].z), "=v"(v[it].w));
        }
        uint4 lut[8];
        WALK_MARK(0);
        tstage_back_half<WRITE_COEF, true>(x, lut, xb, a, f, bp, lane, lane_info);
        WALK_MARK(1);
#ifdef FFV2_PHASE_TIMING
        wdone_++;
#endif
    }
#ifdef FFV2_PHASE_TIMING
    if (lane == 0) {
        unsigned long long *pt_ = g_phase_buf + (size_t)blockIdx.x * 8;
        pt_[0] = wrt0_; pt_[1] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
        pt_[2] = __builtin_amdgcn_s_memrealtime() - wrt0_;
        pt_[3] = wacc_[0]; pt_[4] = wacc_[1]; pt_[5] = wacc_[2]; pt_[6] = wdone_;
    }
#endif
}

// ---------------------------------------------------------------------------
// E-stage, qp == 0
// ---------------------------------------------------------------------------
// raw bit t of the frame's raw stream lives in packet byte total-1-(t>>3), bit t&7
// (daala_entropy.c:259,700: bytes are written from the buffer end backwards; the
// last, partial byte is OR-ed into the final range byte, :719-721).

constexpr int EP_THREADS = 256;                  // block-planes per workgroup
constexpr int EP_WORDS   = 3840;                 // 256 * 58 B worst case, + slack, in 32-bit words

// LSB-first bit writer into a zeroed LDS word buffer (ds_or_b32)
struct LdsBitSink {
    uint32_t *buf;
    uint32_t word;
    unsigned long long win;
    int nwin;
    __device__ void init(uint32_t *b, uint32_t bitpos)
    {
        buf = b; word = bitpos >> 5; nwin = (int)(bitpos & 31); win = 0;
    }
    __device__ void put(unsigned long long v, int n)          // n <= 32
    {
        win |= v << nwin;
        nwin += n;
        if (nwin >= 32) {
            atomicOr(&buf[word++], (uint32_t)win);
            win >>= 32;
            nwin -= 32;
        }
    }
    __device__ void flush()
    {
        if (nwin > 0 && (uint32_t)win) atomicOr(&buf[word], (uint32_t)win);
    }
};

__device__ __forceinline__ unsigned long long spread_bits(uint32_t x32)
{
    unsigned long long x = x32;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8))  & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4))  & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2))  & 0x3333333333333333ull;
    x = (x | (x << 1))  & 0x5555555555555555ull;
    return x;
}

// Exp-Golomb of ffv2enc.c:105-123 as one LSB-first bit pattern: for every bit of
// val+1 below its MSB, MSB first, the pair [0, bit]; then a single 1.
__device__ __forceinline__ void put_golomb(LdsBitSink &s, uint32_t val)
{
    const uint32_t v = val + 1;
    const int nb = 31 - __clz(v);
    unsigned long long pat = 1ull << (2 * nb);
    if (nb > 0) {
        const uint32_t m = v & ((1u << nb) - 1);
        const uint32_t r = __brev(m) >> (32 - nb);           // first-emitted bit at position 0
        pat |= spread_bits(r) << 1;
    }
    const int len = 2 * nb + 1;
    if (len > 32) {
        s.put(pat & 0xffffffffull, 32);
        s.put(pat >> 32, len - 32);
    } else {
        s.put(pat, len);
    }
}

// E-stage at qp == 0, one launch: workgroup (b, f) owns block-planes [256b, 256b+256)
// of frame f.  It (1) sums the raw-bit counts of the whole frame (24 KB, L2) to get
// its own starting bit and the packet size -- the "prefix sum" is recomputed per
// workgroup instead of being a separate serial kernel --, (2) assembles its codes
// in LDS with ds_or, (3) streams the finished words to the packet, byte-reversed
// because raw bytes run backwards from the packet end; only the first/last words,
// shared with a neighbour workgroup or the range-coded prefix, use global atomics.
__global__ __launch_bounds__(EP_THREADS) void ffv2_estage_kernel(const FFV2EStageArgs a)
{
    __shared__ uint32_t bits[EP_WORDS];
    __shared__ uint32_t scratch[8];
    const int f = blockIdx.y, tid = threadIdx.x;
    const int n = a.g.nblk, P = a.g.planes;
    const int i0 = blockIdx.x * EP_THREADS;
    const uint32_t *cnt = a.bitoff + (size_t)f * n;          // raw bits per block-plane (from the T-stage)
    // this thread's 64-byte record, requested first: its latency passes behind the prefix sums
    const int i = i0 + tid;
    const bool have = i < n;
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
    if (have) {
        const uint4 *rp = reinterpret_cast<const uint4 *>(a.codes + ((size_t)f * n + i) * FFV2_CODES_PER_BP);
        r0 = rp[0]; r1 = rp[1]; r2 = rp[2]; r3 = rp[3];
    }

    // (1) bits in front of this workgroup, and in the whole frame; every superblock is
    // preceded by its 4 transform-type bits (ffv2enc.c:197)
    // (the 4 bits per superblock are counted in closed form: ceil(i0/P) superblocks start in front
    // of this workgroup, n/P in the frame; the two sums share one pair of barriers)
    uint32_t before = 0, all = 0;
    for (int i = tid; i < n; i += EP_THREADS) {
        const uint32_t v = cnt[i];
        all += v;
        before += i < i0 ? v : 0u;
    }
    {
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane(wave_iscan((int)before), 63);
        const uint32_t t = (uint32_t)__builtin_amdgcn_readlane(wave_iscan((int)all), 63);
        if ((tid & 63) == 0) { scratch[tid >> 6] = b; scratch[4 + (tid >> 6)] = t; }
        __syncthreads();
        before = scratch[0] + scratch[1] + scratch[2] + scratch[3] + 4u * (uint32_t)((i0 + P - 1) / P);
        all = scratch[4] + scratch[5] + scratch[6] + scratch[7] + 4u * (uint32_t)(n / P);
    }

    const uint32_t tx = (have && (i % P) == 0) ? 4u : 0u;
    const uint32_t mine = have ? r3.z + tx : 0u;              // record slot 14 = this block-plane's raw bits
    // exclusive scan of `mine` over the workgroup
    const uint32_t incl = (uint32_t)wave_iscan((int)mine);
    __syncthreads();
    if ((tid & 63) == 63) scratch[tid >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < (tid >> 6); w++) wbase += scratch[w];
    const uint32_t local_total = scratch[0] + scratch[1] + scratch[2] + scratch[3];

    const uint32_t total_bits = a.header_nbits + all;
    const uint32_t slack = (uint32_t)a.slack_bits;
    const uint32_t nraw = total_bits > slack ? (total_bits - slack + 7) >> 3 : 0;
    const uint32_t total = (uint32_t)a.prefix_len + nraw;
    uint8_t *pkt = a.packets + (size_t)f * a.packet_stride;
    const bool fits = (size_t)total + 4 <= a.packet_stride;
    if (blockIdx.x == 0 && tid == 0) {
        a.sizes[f] = fits ? total : 0;
        // The frame's status word is WRITTEN here (no memset launch in front of the T-stage): the
        // T-stage's sticky error flag, or NOSPACE; the flag is handed back cleared for the next call.
        int32_t st = a.err[f];
        a.err[f] = 0;
        if (!fits && st > -28) st = -28;                      // FFV2AMD_ERR_NOSPACE
        a.status[f] = st;
    }
    if (!fits) return;                                         // workgroup-uniform

    // this workgroup's bit range [B0, B1) of the raw stream (workgroup 0 also owns the header)
    const uint32_t B0 = blockIdx.x == 0 ? 0u : a.header_nbits + before;
    const uint32_t B1 = a.header_nbits + before + local_total;
    const uint32_t lbase = B0 & ~31u;                          // LDS word 0 <-> raw bit lbase
    const uint32_t nwords = (B1 - lbase + 31) >> 5;
    for (uint32_t w = tid; w < nwords + 1 && w < EP_WORDS; w += EP_THREADS) bits[w] = 0;
    __syncthreads();

    // (2) codes -> LDS
    if (have) {
        const uint32_t rec[14] = { r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w,
                                   r2.x, r2.y, r2.z, r2.w, r3.x, r3.y };
        LdsBitSink s;
        s.init(bits, a.header_nbits + before + wbase + (incl - mine) + tx - lbase);
        const int c0 = (int)rec[0];
        const uint32_t mag = c0 < 0 ? (uint32_t)(-(long long)c0) : (uint32_t)c0;
        put_golomb(s, mag);                                   // ffv2enc.c:148-150
        if (c0) s.put(c0 < 0 ? 1u : 0u, 1);
#pragma unroll
        for (int b = 0; b < FFV2_NUM_BANDS; b++)
            put_golomb(s, rec[1 + b]);                        // ffv2enc.c:174
        s.flush();
    }
    if (blockIdx.x == 0 && tid == 0)
        atomicOr(&bits[0], a.header_bits);                    // raw header, bit 0 on (header_nbits <= 32)
    __syncthreads();

    // (3) LDS -> packet.  raw byte k sits at packet byte total-1-k.  Walk the aligned
    // destination words that contain any of our raw bytes [ka, kb).
    const uint32_t ka = B0 >> 3, kb = (B1 + 7) >> 3;           // raw bytes we contribute to
    const uint32_t ea = (B0 + 7) >> 3, eb = B1 >> 3;           // raw bytes nobody else writes
    if (kb > ka) {
        const uint32_t dlo = (total - kb) & ~3u, dhi = (total - 1 - ka) & ~3u;
        const uint8_t *lb = reinterpret_cast<const uint8_t *>(bits);
        for (uint32_t d = dlo + 4 * tid; d <= dhi; d += 4 * EP_THREADS) {
            uint32_t word = 0;
            bool exclusive = true;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int64_t k = (int64_t)total - 1 - (int64_t)(d + j);     // raw byte behind packet byte d+j
                const bool ours = k >= (int64_t)ka && k < (int64_t)kb;
                exclusive = exclusive && k >= (int64_t)ea && k < (int64_t)eb;
                const uint32_t v = ours ? lb[(uint32_t)k - (lbase >> 3)] : 0u;
                word |= v << (8 * j);
            }
            uint32_t *dst = reinterpret_cast<uint32_t *>(pkt + d);
            if (exclusive) *dst = word;
            else if (word) atomicOr(dst, word);
        }
    }
    // range-coded prefix (data independent); its last byte is shared with raw bits -> OR
    if (blockIdx.x == 0)
        for (int k = tid; k < a.prefix_len; k += EP_THREADS)
            atomicOr(reinterpret_cast<uint32_t *>(pkt + (k & ~3)), (uint32_t)a.prefix[k] << ((k & 3) * 8));
}

}  // namespace

// Wavefronts the chip holds at once for the walking kernel (occupancy x CUs), per device.
static int walk_slots(bool bps2, bool wc)
{
    static int cache[16][4];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    int &c = cache[dev][(bps2 ? 2 : 0) + (wc ? 1 : 0)];
    if (c == 0) {
        int per_cu = 0, cus = 0;
        const void *fn = bps2 ? (wc ? (const void *)ffv2_tstage_walk_kernel<2, true> : (const void *)ffv2_tstage_walk_kernel<2, false>)
                              : (wc ? (const void *)ffv2_tstage_walk_kernel<1, true> : (const void *)ffv2_tstage_walk_kernel<1, false>);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu < 1 || cus < 1)
            c = -1;
        else
            c = per_cu * cus;
    }
    return c;
}

// Column-walking kernel when every wavefront slot gets at least FFV2_WALK_MIN block-planes;
// the one-block kernel otherwise (small pictures).  Returns the slot count, 0 = one-block kernel.
static int g_tstage_force = -2;        // -2: not set (environment decides); else 0 block, 1 walk, -1 auto

void ffv2_tstage_force_variant(int mode) { g_tstage_force = mode; }

static int tstage_walk_slots(const FFV2Geom &g, int nframes, bool wc)
{
    static const int env_mode = getenv("FFV2AMD_TSTAGE") ? atoi(getenv("FFV2AMD_TSTAGE")) : -1;   // 0 block, 1 walk, -1 auto
    const int walk_mode = g_tstage_force != -2 ? g_tstage_force : env_mode;
    const uint64_t total64 = (uint64_t)g.nblk * (uint64_t)nframes;
    const int slots = walk_mode == 0 ? 0 : walk_slots(g.bytes_per_sample != 1, wc);
    if (slots > 0 && total64 < (1ull << 31) && (walk_mode == 1 || total64 >= (uint64_t)slots * FFV2_WALK_MIN)) return slots;
    return 0;
}

const char *ffv2_tstage_kernel_name(const FFV2Geom &g, int nframes, bool wc)
{
    return tstage_walk_slots(g, nframes, wc) > 0 ? "ffv2_tstage_walk_kernel" : "ffv2_tstage_kernel";
}

hipError_t ffv2_launch_tstage(const FFV2TStageArgs &a, hipStream_t s)
{
    const bool wc = a.coef != nullptr;
    const bool bps2 = a.g.bytes_per_sample != 1;
    const uint64_t total64 = (uint64_t)a.g.nblk * (uint64_t)a.nframes;
    const int slots = tstage_walk_slots(a.g, a.nframes, wc);
    if (slots > 0) {
        const uint32_t total = (uint32_t)total64;
        // Segment lengths: each takes FFV2AMD_WALK_SEG (default 0.55) of what is left of the column,
        // at least one superblock: 17, 9, 4, 2, 1, 1 for the 34 rows of eight 4K pictures.
        static const double seg_frac = getenv("FFV2AMD_WALK_SEG") ? atof(getenv("FFV2AMD_WALK_SEG")) : 0.55;
        static const double seg_cap = getenv("FFV2AMD_WALK_CAP") ? atof(getenv("FFV2AMD_WALK_CAP")) : 0.7;
        // ... but no run longer than seg_cap of a slot's even share, so that small launches
        // (fewer columns than slots) still spread over the whole chip
        int cap = (int)(seg_cap * (double)total / slots + 0.5);
        if (cap < 1) cap = 1;
        WalkTiers tiers{};
        tiers.ncols = (uint32_t)(a.g.nsx * a.g.planes * a.nframes);
        tiers.ncols8 = (tiers.ncols + 7u) / 8u * 8u;
        int left = a.g.nsy, at = 0;
        while (left > 0) {
            int l = (int)(left * seg_frac + 0.5);
            if (l > cap) l = cap;
            if (l < 1) l = 1;
            if (tiers.ntiers == FFV2_WALK_MAX_TIERS - 1) l = left;
            tiers.len[tiers.ntiers] = (uint16_t)l; tiers.joff[tiers.ntiers] = (uint16_t)at;
            tiers.ntiers++; at += l; left -= l;
        }
        const uint32_t groups = tiers.ntiers * tiers.ncols8;
                const dim3 grid(groups), block(64);
        if (bps2) {
            if (wc) hipLaunchKernelGGL((ffv2_tstage_walk_kernel<2, true>),  grid, block, 0, s, a, total, tiers);
            else    hipLaunchKernelGGL((ffv2_tstage_walk_kernel<2, false>), grid, block, 0, s, a, total, tiers);
        } else {
            if (wc) hipLaunchKernelGGL((ffv2_tstage_walk_kernel<1, true>),  grid, block, 0, s, a, total, tiers);
            else    hipLaunchKernelGGL((ffv2_tstage_walk_kernel<1, false>), grid, block, 0, s, a, total, tiers);
        }
        return hipGetLastError();
    }
    const dim3 grid((unsigned)((a.g.nblk + 7) / 8 * 8), (unsigned)a.nframes), block(64);
    if (!bps2) {
        if (wc) hipLaunchKernelGGL((ffv2_tstage_kernel<1, true>),  grid, block, 0, s, a);
        else    hipLaunchKernelGGL((ffv2_tstage_kernel<1, false>), grid, block, 0, s, a);
    } else {
        if (wc) hipLaunchKernelGGL((ffv2_tstage_kernel<2, true>),  grid, block, 0, s, a);
        else    hipLaunchKernelGGL((ffv2_tstage_kernel<2, false>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

hipError_t ffv2_launch_estage_qp0(const FFV2EStageArgs &a, hipStream_t s)
{
    hipLaunchKernelGGL(ffv2_estage_kernel, dim3((a.g.nblk + EP_THREADS - 1) / EP_THREADS, a.nframes),
                       dim3(EP_THREADS), 0, s, a);
    return hipGetLastError();
}
