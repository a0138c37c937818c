// ffv2_rangecoder.hip -- the qp > 0 entropy coder ON THE DEVICE:
// Daala's adaptive range coder (reference libavcodec/daala_entropy.c:328-379 interval update,
// :107-151 renormalisation into 16-bit pre-carry words, :428-440 adaptive CDF rows, :227-270 raw
// bits, :624-735 encode_done incl. the carry propagation :706-715) driven in ffv2enc.c's symbol
// order (:447-451 header, :222,:197 per superblock, :148-150,:174-186 per block-plane and band).
//
// One workgroup (one wavefront) per frame -- the coder is one serial chain per frame: the 13 CDF
// rows and the range are shared by every block of the frame (ffv2enc.c:461,181).
//   phase 1  the symbol loop, serial: the whole wavefront steps through it in lockstep on identical
//            state (symbols fetched 64 at a time, one byte per lane, handed round by lane reads;
//            the CDF row update is lane-parallel); range words go to a pre-carry array, raw bits
//            to a byte array, both in HBM scratch, stored by lane 0.
//   phase 2  all 64 lanes: the carry propagation as a parallel prefix ("carry look-ahead"): word i
//            contributes its low byte to byte i and its high byte to byte i-1, so byte sums are at
//            most 510 and the carry between bytes is a single bit; each lane resolves a contiguous
//            chunk for carry-in 0 and 1, a wavefront suffix scan over those 2-bit functions picks
//            the right one.
//   phase 3  all lanes: packet = [range bytes][raw bytes in reverse write order], the last range
//            byte OR-ed with the left-over raw bits (daala_entropy.c:719-721).
// This is the north_star's "device coder with wavefront prefix-sums for the renormalisation".  It is
// NOT the default: a GPU lane runs the ~20-operation dependent chain of one symbol several times
// slower than a host core, and frames are the only parallel axis, so the default
// codes on host threads behind the GPU.  ffv2amd_encoder_set_device_coder(enc, 1) selects this
// kernel; tests/ hold it to the same packets.  PARITY UNPINNED (as all of qp > 0).
#include "ffv2_kernels.h"

namespace {

struct RcState {
    unsigned long long low;
    uint32_t rng;
    int cnt;
    uint32_t npre, cap_pre;
    uint16_t *pre;
    unsigned long long win;
    int nwin;
    uint32_t nraw, cap_raw;
    uint8_t *raw;
    int status;                       // 0, FFV2AMD_ERR_ABORT (-1) or FFV2AMD_ERR_NOSPACE (-28)
};

__device__ __forceinline__ int rc_ilog(uint32_t v) { return v ? 32 - __clz(v) : 0; }

// Phase 1 is executed by the whole wavefront in lockstep on identical (wave-uniform) state, so that
// the symbols can be fetched 64 at a time, one byte per lane, and handed round with a lane read;
// only lane 0 stores.
__device__ __forceinline__ void rc_push(RcState &s, uint32_t w)
{
    if (s.npre < s.cap_pre) { if (threadIdx.x == 0) s.pre[s.npre] = (uint16_t)w; }
    else if (s.status == 0) s.status = -28;
    s.npre++;
}

__device__ void rc_encode(RcState &s, uint32_t fl, uint32_t fh, uint32_t ft)      // daala_entropy.c:362-378
{
    const int sc = (s.rng - ft) >= ft;
    fl <<= sc; fh <<= sc; ft <<= sc;
    const uint32_t d = s.rng - ft;
    const uint32_t g = 2 * d > ft ? 2 * d - ft : 0;
    const uint32_t bl = (fl > g ? fl - g : 0) >> 1, bh = (fh > g ? fh - g : 0) >> 1;
    const uint32_t u = fl + (fl < g ? fl : g) + (bl < d ? bl : d);
    const uint32_t v = fh + (fh < g ? fh : g) + (bh < d ? bh : d);
    unsigned long long l = s.low + u;                                             // :107-151
    const uint32_t r = v - u;
    const int dd = 16 - rc_ilog(r);
    int c = s.cnt, sh = c + dd;
    if (sh >= 0) {
        c += 16;
        unsigned long long m = (1ull << c) - 1;
        if (sh >= 8) { rc_push(s, (uint32_t)(l >> c)); l &= m; c -= 8; m >>= 8; }
        rc_push(s, (uint32_t)(l >> c));
        sh = c + dd - 24;
        l &= m;
    }
    s.low = l << dd; s.rng = r << dd; s.cnt = sh;
}

__device__ void rc_bits(RcState &s, uint32_t v, int n)                             // :227-270
{
    if (s.nwin + n > 64) {
        do {
            if (s.nraw < s.cap_raw) { if (threadIdx.x == 0) s.raw[s.nraw] = (uint8_t)s.win; }
            else if (s.status == 0) s.status = -28;
            s.nraw++;
            s.win >>= 8; s.nwin -= 8;
        } while (s.nwin >= 8);
    }
    s.win |= (unsigned long long)v << s.nwin;
    s.nwin += n;
}

__device__ void rc_golomb(RcState &s, uint32_t val)                                // ffv2enc.c:105-123
{
    const uint32_t v = val + 1;
    if (val == 0) { rc_bits(s, 1, 1); return; }
    const int nb = 31 - __clz(v);
    for (int i = nb - 1; i >= 0; i--) rc_bits(s, ((v >> i) & 1) << 1, 2);
    rc_bits(s, 1, 1);
}

// adaptive CDF symbol (daala_entropy.c:428-440 over :334-347)
__device__ void rc_adapt(RcState &s, uint16_t *cdf, int n, int inc, int val)
{
    if (val < 0 || val >= n) { s.status = -1; return; }
    const uint32_t fl = val ? cdf[val - 1] : 0, fh = cdf[val], ft = cdf[n - 1];
    if (!(fl < fh && fh <= ft && ft >= 2 && ft <= 32768)) { s.status = -1; return; }
    const int sc = 15 - rc_ilog(ft - 1);
    if ((ft << sc) > s.rng) { s.status = -1; return; }
    rc_encode(s, fl << sc, fh << sc, ft << sc);
    // the row update is the one data-parallel step: lane i owns entry i (n <= 64)
    const int i = (int)threadIdx.x;
    const bool halve = cdf[n - 1] + inc > 32767;
    if (i < n) {
        uint32_t c = cdf[i];
        if (halve) c = (c >> 1) + i + 1;
        if (i >= val) c += inc;
        cdf[i] = (uint16_t)c;
    }
}

struct RcShared {
    uint16_t cdf[13][64];
    uint16_t subdiv[4];
    uint32_t npre, nraw, win_lo;
    int nwin, status;
};

__global__ __launch_bounds__(64) void ffv2_rangecoder_kernel(const FFV2RangeCoderArgs a)
{
    __shared__ RcShared S;
    const int f = blockIdx.x, lane = threadIdx.x;
    const size_t nb = (size_t)a.nblk;
    uint16_t *pre = a.pre + (size_t)f * a.cap;
    uint8_t *raw = a.raw + (size_t)f * a.cap;
    uint8_t *pkt = a.packets + (size_t)f * a.packet_stride;

    {
        RcState s;
        s.low = 0; s.rng = 0x8000; s.cnt = -9;
        s.npre = 0; s.cap_pre = (uint32_t)a.cap; s.pre = pre;
        s.win = 0; s.nwin = 0; s.nraw = 0; s.cap_raw = (uint32_t)a.cap; s.raw = raw;
        s.status = a.status_in[f] < 0 ? a.status_in[f] : 0;
        const int qp = a.qp;
        if (s.status == 0) {
            // header: ff_daalaent_encode_uint(pix_fmt, 196) then Exp-Golomb(qp)   (ffv2enc.c:447-451)
            const uint32_t hi = (uint32_t)a.pix_fmt >> 4;
            rc_encode(s, hi ? (32768u * hi + 6u) / 13u : 0, (32768u * (hi + 1) + 6u) / 13u, 32768);
            rc_bits(s, (uint32_t)a.pix_fmt & 15u, 4);
            rc_golomb(s, (uint32_t)qp);
            if (lane < 4) S.subdiv[lane] = (uint16_t)(32 * (lane + 1));
            for (int r = 0; r < 13; r++) S.cdf[r][lane] = (uint16_t)(lane + 1);
            const uint32_t *codes = a.codes + (size_t)f * nb * FFV2_CODES_PER_BP;
            const FFV2SymRec *rec = a.rec + (size_t)f * nb;
            const int8_t *stream = a.stream + (size_t)f * a.stream_stride;
            for (int sb = 0; sb < a.nsb && s.status == 0; sb++) {
                rc_adapt(s, S.subdiv, 4, 128, 0);                                 // split = END (ffv2enc.c:222)
                rc_bits(s, 0, 4);                                                 // tx type (ffv2enc.c:197)
                for (int p = 0; p < a.planes && s.status == 0; p++) {
                    const size_t bp = (size_t)sb * a.planes + p;
                    const uint32_t *cr = codes + bp * FFV2_CODES_PER_BP;
                    const int8_t *sy = stream + rec[bp].offset;
                    const int c0 = (int)cr[0];
                    rc_golomb(s, c0 < 0 ? (uint32_t)(-(long long)c0) : (uint32_t)c0);
                    if (c0) rc_bits(s, c0 < 0, 1);
                    for (int b = 0; b < 13 && s.status == 0; b++) {
                        rc_golomb(s, cr[1 + b]);
                        const int len = rec[bp].count[b];
                        // the band's CDF row lives in a register for the length of the band: lane i
                        // holds entry i, the three entries a symbol needs come by lane reads
                        uint32_t row = S.cdf[b][lane];
                        for (int j0 = 0; j0 < len && s.status == 0; j0 += 64) {    // ffv2enc.c:176-186
                            const int mine = j0 + lane < len ? sy[j0 + lane] : 0;     // 64 symbols per fetch
                            const int cnt = len - j0 < 64 ? len - j0 : 64;
                            for (int k = 0; k < cnt && s.status == 0; k++) {
                                const int q = __builtin_amdgcn_readlane(mine, k), aq = q < 0 ? -q : q;
                                if (aq >= qp) { s.status = -1; break; }                  // daala_entropy.c:336
                                const uint32_t fl = aq ? (uint32_t)__builtin_amdgcn_readlane((int)row, aq - 1) : 0u;
                                const uint32_t fh = (uint32_t)__builtin_amdgcn_readlane((int)row, aq);
                                const uint32_t ft = (uint32_t)__builtin_amdgcn_readlane((int)row, qp - 1);
                                if (!(fl < fh && fh <= ft && ft >= 2 && ft <= 32768)) { s.status = -1; break; }
                                const int sc = 15 - rc_ilog(ft - 1);
                                if ((ft << sc) > s.rng) { s.status = -1; break; }        // :364
                                rc_encode(s, fl << sc, fh << sc, ft << sc);
                                if (ft + 64 > 32767) row = (row >> 1) + lane + 1;        // :434-439
                                if (lane >= aq) row += 64;
                                if (q) rc_bits(s, q < 0, 1);
                            }
                        }
                        S.cdf[b][lane] = (uint16_t)row;
                        sy += len;
                    }
                }
            }
        }
        int slack = 0;
        if (s.status == 0) {
            // ff_daalaent_encode_done, the range part (daala_entropy.c:624-674)
            unsigned long long m = 0x7FFF, e = (s.low + m) & ~m;
            int sh = 9, c = s.cnt;
            while ((e | m) >= s.low + s.rng) { sh++; m >>= 1; e = (s.low + m) & ~m; }
            sh += c;
            if (sh > 0) {
                unsigned long long n = (1ull << (c + 16)) - 1;
                do { rc_push(s, (uint32_t)(e >> (c + 16))); e &= n; sh -= 8; c -= 8; n >>= 8; } while (sh > 0);
            }
            slack = -sh;
            while (s.nwin > slack) {                                              // flush whole raw bytes (:676-700)
                if (s.nraw < s.cap_raw) { if (lane == 0) s.raw[s.nraw] = (uint8_t)s.win; }
                else if (s.status == 0) s.status = -28;
                s.nraw++;
                s.win >>= 8; s.nwin -= 8;
            }
            if (s.nwin > 0 && s.npre == 0) s.status = -1;                        // :719
            if ((size_t)s.npre + s.nraw > a.packet_stride && s.status == 0) s.status = -28;
        }
        if (lane == 0) { S.npre = s.npre; S.nraw = s.nraw; S.win_lo = (uint32_t)s.win; S.nwin = s.nwin; S.status = s.status; }
    }
    __syncthreads();
    const int status = S.status;
    if (lane == 0) { a.status[f] = status; a.sizes[f] = status < 0 ? 0 : S.npre + S.nraw; }
    if (status < 0) return;
    const uint32_t n = S.npre, nraw = S.nraw;

    // ---- phase 2: carry propagation (daala_entropy.c:706-715) as a parallel prefix ----
    // byte i = (lo(pre[i]) + hi(pre[i+1]) + k[i+1]) & 255, k[i] = that sum >> 8 (a single bit); lane L
    // owns bytes [L*m, L*m + m), resolved from the chunk's END for carry-in 0 and carry-in 1
    const uint32_t m = (n + 63) / 64;
    const uint32_t i0 = (uint32_t)lane * m, i1 = i0 + m < n ? i0 + m : n;
    uint32_t k0 = 0, k1 = 1;
    for (uint32_t i = i1; i-- > i0;) {
        const uint32_t sum = (uint32_t)(pre[i] & 0xff) + (i + 1 < n ? (uint32_t)(pre[i + 1] >> 8) : 0u);
        k0 = (sum + k0) >> 8;
        k1 = (sum + k1) >> 8;
    }
    if (i0 >= i1) { k0 = 0; k1 = 1; }                       // an empty chunk passes the carry through
    // carry into lane L = carry out of lane L+1, which depends on the carry into lane L+1, ...: each
    // chunk is a function {0,1} -> {0,1} (two bits: its value at 0 and at 1); a suffix scan over the
    // wavefront composes them (6 steps of shuffle), function composition being associative
    uint32_t fn = k0 | (k1 << 1);                           // this lane's chunk; afterwards F_L o F_{L+1} o ... o F_63
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t other = (uint32_t)__shfl_down((int)fn, o, 64);
        if (lane + o > 63) other = 2u;                      // identity: f(0) = 0, f(1) = 1
        fn = ((fn >> (other & 1u)) & 1u) | (((fn >> ((other >> 1) & 1u)) & 1u) << 1);
    }
    uint32_t k = (uint32_t)__shfl_down((int)fn, 1, 64) & 1u;      // suffix of the lanes behind, applied to carry-in 0
    if (lane == 63) k = 0;
    for (uint32_t i = i1; i-- > i0;) {
        const uint32_t sum = (uint32_t)(pre[i] & 0xff) + (i + 1 < n ? (uint32_t)(pre[i + 1] >> 8) : 0u) + k;
        pkt[i] = (uint8_t)sum;
        k = sum >> 8;
    }
    // ---- phase 3: raw bytes behind the range bytes, in reverse write order ----
    const uint32_t total = n + nraw;
    for (uint32_t i = (uint32_t)lane; i < nraw; i += 64) pkt[total - 1 - i] = raw[i];
    __syncthreads();
    if (lane == 0 && S.nwin > 0) pkt[n - 1] |= (uint8_t)S.win_lo;                 // :719-721
}

}  // namespace

hipError_t ffv2_launch_rangecoder(const FFV2RangeCoderArgs &a, int nframes, hipStream_t s)
{
    hipLaunchKernelGGL(ffv2_rangecoder_kernel, dim3((unsigned)nframes), dim3(64), 0, s, a);
    return hipGetLastError();
}
