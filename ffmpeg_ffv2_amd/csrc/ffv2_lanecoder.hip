// ffv2_lanecoder.hip -- the qp > 0 entropy coder with MANY FRAMES IN FLIGHT (SURVEY.md 8(f) rank 1,
// 8/A14): Daala's adaptive range coder (reference libavcodec/daala_entropy.c) is one dependent
// chain per frame (ffv2enc.c:461,466: one coder, one set of CDF rows per frame), so the device
// runs that chain for 64 frames per wavefront, one frame per lane, and does everything else in
// parallel.  tools/lanecoder_model.py is the arithmetic of this file in plain Python.
//
//   count    per block-plane: how many pulses of each band the coder reads (ffv2enc.c:176-186:
//            until the band's pulse count reaches qp), how many of them are non-zero, the raw
//            bits the block-plane emits.
//   scan     per frame: prefix sums over the block-planes -> where every band starts in its CDF
//            row's symbol sequence, in the frame's coding order and in the raw-bit tail.
//   scatter  per block-plane: |pulse| into the 13 per-row symbol sequences; Exp-Golomb codes and
//            sign bits into the raw-bit tail (ffv2enc.c:105-123,148-150,174,183-184;
//            daala_entropy.c:227-270).
//   cdf      per (frame, CDF row): the adaptive rows advance with the symbols alone
//            (daala_entropy.c:428-440), never with the range: between two halvings a row is its
//            value at the last halving plus 64 x a prefix count, so 64 symbols at a time get
//            their (fl, fh, ft) from ballots; written as records in coding order.
//   chain    per 64 frames, lane = frame, two wavefronts: the interval update (daala_entropy.c:362-378)
//            and the renormalisation shift (:107-151) -- the only serial part -- on one, the code
//            words on the other, a ring of tiles in LDS between them.  `low` is not carried:
//            the code is sum_k u_k << (T - D_k) (D_k = shifts before symbol k), accumulated into
//            32-bit words anchored every 16 bits of depth; a symbol shifts by <= 15 bits, so the
//            anchor advances by 0 or 1 per symbol.
//   windows  (round 3) cdf and chain do not run once over whole frames but alternate over WINDOWS of the
//            coding order -- symbols [w0, w1) of every frame, the same bounds for all frames -- so the
//            records exist only for two windows at a time (8 bytes x window x 2 per frame instead of
//            8 bytes per coefficient: 50 MB per 1080p frame before).  Both kernels keep their state
//            between windows in HBM: a CDF row as it stands (64 entries, total, position: 272 bytes
//            per row and frame), a frame's range, word position and current word.  A row's chunk of 64
//            symbols that straddles w1 is cut there -- chunk lengths are a pure function of (total,
//            position), so the next window resumes with the chunk the uncut run would have had next
//            only if nothing was cut; cutting never changes a symbol's (fl, fh, ft): they depend on
//            the symbols before it alone.  cdf of window i+1 runs beside the chain of window i.
//   finish   per frame: ff_daalaent_encode_done's rounding (:624-674) as one more addend (size),
//            a prefix sum over the packet sizes (offsets: the packets leave the device packed),
//            the words to bytes through a one-bit carry look-ahead (:706-715) and the raw bytes
//            behind them in reverse order (:676-721) (write).
// PARITY UNPINNED, as all of qp > 0; tests hold this coder to the host coder's and
// the oracle's packets.
#include "ffv2_kernels.h"

namespace {

#define LC_CDF_STATE 68                     // dwords a CDF row keeps between windows: 64 entries, total, position, block-plane, pad

__constant__ int LC_BS[FFV2_NUM_BANDS + 1] = { 0, 15, 23, 31, 63, 95, 127, 255, 383, 511, 1023, 1535, 2047, 4096 };

__device__ __forceinline__ uint32_t lane_prefix(unsigned long long m)       // set bits of m below this lane
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Records in coding order, interleaved so that the chain kernel's loads coalesce: frames are
// grouped `width` at a time (one lane each); symbol k of lane L sits in piece k/16 (a tile of the chain:
// 128 bytes per lane = one cache line, so that the cdf kernel -- one wavefront per frame and row -- writes
// whole lines; pieces of all lanes side by side), slot k%16.
__device__ __forceinline__ uint2 *lc_record(uint2 *recs, size_t group_stride, int width, int f, uint32_t k)
{
    const int g = f / width, L = f - g * width;
    return recs + (size_t)g * group_stride + ((size_t)(k >> 4) * width + L) * 16 + (k & 15u);
}

// Exp-Golomb code of ffv2enc.c:105-123 as the bits it appends (LSB first): for every bit of
// val+1 below its MSB, MSB first: a 0 then the bit; then a 1.
__device__ __forceinline__ unsigned long long golomb_code(uint32_t val, int *len)
{
    const uint32_t v = val + 1u;
    if (v == 0) { *len = 0; return 0; }                       // val = 2^32-1 cannot occur (gains < 2^16, |c0| < 2^31)
    const int nb = 31 - __clz(v);
    unsigned long long code = 1ull << (2 * nb);
    for (int i = 0; i < nb; i++) code |= (unsigned long long)((v >> i) & 1u) << (2 * (nb - 1 - i) + 1);
    *len = 2 * nb + 1;
    return code;
}

// ---------------------------------------------------------------------------------------------
// count
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void lc_count_kernel(const int16_t *y, const uint32_t *codes, int qp, int nblk,
                                                      FFV2SymRec *cnt, uint32_t *bits, int32_t *abort_)
{
    const int f = blockIdx.y, bp = blockIdx.x, lane = threadIdx.x;
    const int16_t *yy = y + ((size_t)f * nblk + bp) * FFV2_Y_STRIDE;
    FFV2SymRec *r = cnt + (size_t)f * nblk + bp;
    uint32_t total = 0, nz = 0;
    bool big = false;
#pragma unroll 1
    for (int b = 0; b < FFV2_NUM_BANDS; b++) {
        const int lo = 1 + LC_BS[b], N = LC_BS[b + 1] - LC_BS[b];
        int run = 0, stop = N;
        for (int j0 = 0; j0 < N && stop == N; j0 += 64) {
            const int j = j0 + lane;
            int a = j < N ? yy[lo + j] : 0;
            a = a < 0 ? -a : a;
            int incl = a;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            const unsigned long long hit = __ballot(run + incl >= qp);
            int end = N;
            if (hit) { stop = j0 + __ffsll((long long)hit); end = stop; }
            nz += (uint32_t)__popcll(__ballot(j < end && a > 0));
            big = big || __ballot(j < end && a >= qp) != 0;       // the reference asserts (daala_entropy.c:336)
            run += __shfl(incl, 63, 64);
        }
        if (stop > N) stop = N;
        if (lane == 0) r->count[b] = (uint16_t)stop;
        total += (uint32_t)stop;
    }
    if (lane == 0) {
        r->offset = total;
        r->pad = 0;
        bits[(size_t)f * nblk + bp] = codes[((size_t)f * nblk + bp) * FFV2_CODES_PER_BP + 14] + nz;
        if (big) atomicOr((int *)&abort_[f], 1);
    }
}

// ---------------------------------------------------------------------------------------------
// scan: one wavefront per (quantity, frame)
//   q < 13 : rowbase[q][bp] = symbols of CDF row q in front of block-plane bp; [nblk] = row length
//   q = 13 : gbase[bp] = coding-order index of bp's first band symbol (1 header symbol, one split
//            symbol per superblock in front of its first plane); [nblk] = symbols in the frame
//   q = 14 : rawbase[bp] = bit offset of bp's first raw bit (header bits, 4 tx bits per superblock
//            in front of its first plane); [nblk] = raw bits in the frame
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void lc_scan_kernel(const FFV2LaneCoderArgs a)
{
    const int q = blockIdx.x, f = a.f0 + blockIdx.y, lane = threadIdx.x;
    const int nb = a.nblk;
    const FFV2SymRec *cnt = a.cnt + (size_t)f * nb;
    const uint32_t *bits = a.bits + (size_t)f * nb;
    uint32_t *out = q < 13 ? a.rowbase + ((size_t)f * 13 + q) * (nb + 1)
                  : q == 13 ? a.gbase + (size_t)f * (nb + 1) : a.rawbase + (size_t)f * (nb + 1);
    uint32_t run = q == 13 ? 1u : q == 14 ? a.header_nbits : 0u;
    // eight rows of 64 block-planes at a time: their loads are in flight together (one row per trip, the kernel was the
    // latency of 24 loads in a row for a 1080p frame)
    for (int b0 = 0; b0 < nb; b0 += 512) {
        uint32_t vv[8], ll[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int bp = b0 + 64 * u + lane, bc = bp < nb ? bp : nb - 1;   // loaded unconditionally, masked afterwards
            const uint32_t got = q < 13 ? cnt[bc].count[q] : q == 13 ? cnt[bc].offset : bits[bc];
            vv[u] = bp < nb ? got : 0u;
            ll[u] = bp < nb && q >= 13 && bp % a.planes == 0 ? (q == 13 ? 1u : 4u) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int bp = b0 + 64 * u + lane;
            if (b0 + 64 * u >= nb) break;
            const uint32_t v = vv[u];
            uint32_t incl = v + ll[u];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, o, 64);
                if (lane >= o) incl += t;
            }
            if (bp < nb) out[bp] = run + incl - v;                     // exclusive, behind this block-plane's lead
            run += (uint32_t)__shfl((int)incl, 63, 64);
        }
    }
    if (lane == 0) out[nb] = run;
}

// ---------------------------------------------------------------------------------------------
// scatter
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void lc_scatter_kernel(const FFV2LaneCoderArgs a, const int16_t *y)
{
    // Everything the block-plane needs from memory is asked for at the top -- its pulses go to LDS in 16-byte pieces, the
    // thirteen row positions / lengths / counts and the sixteen code words into one lane each -- and waited for once:
    // from there on the kernel only computes and stores (loads and stores share a wait counter on this chip: a load in
    // the loop would wait for every store before it).
    __shared__ uint32_t sink[64];
    __shared__ __attribute__((aligned(16))) int16_t ys[FFV2_Y_STRIDE + 8];
    const int fl = blockIdx.y, f = a.f0 + fl, bp = blockIdx.x, lane = threadIdx.x;
    const int nb = a.nblk;
    const int16_t *yy = y + ((size_t)fl * nb + bp) * FFV2_Y_STRIDE;
    const FFV2SymRec *r = a.cnt + (size_t)f * nb + bp;
    const uint32_t *cr = a.codes + ((size_t)f * nb + bp) * FFV2_CODES_PER_BP;
    const uint32_t *rowbase = a.rowbase + (size_t)f * 13 * (nb + 1);
    uint8_t *rows = a.rows + (size_t)f * a.row_stride;
    uint32_t *raw = a.raw + (size_t)f * a.raw_words;

    static_assert(FFV2_Y_STRIDE % 8 == 0, "whole 16-byte pieces");
    constexpr int NV = FFV2_Y_STRIDE / 8;                           // 513
    const uint4 *yv = reinterpret_cast<const uint4 *>(yy);          // 16-byte aligned: FFV2_Y_STRIDE * 2 is a multiple of 16
    uint4 *sv = reinterpret_cast<uint4 *>(ys);
    uint4 stage[(NV + 63) / 64];
#pragma unroll
    for (int v = 0; v < (NV + 63) / 64; v++) {
        const int i = lane + 64 * v;
        stage[v] = yv[i < NV ? i : NV - 1];
    }
    const int bl = lane < FFV2_NUM_BANDS ? lane : FFV2_NUM_BANDS - 1;
    const uint32_t my_rb = rowbase[(size_t)bl * (nb + 1) + bp];     // lane b: where band b of this block-plane starts in row b
    const uint32_t my_len = rowbase[(size_t)bl * (nb + 1) + nb];    // ... the length of row b
    const uint32_t my_cnt = r->count[bl];                           // ... symbols the coder reads of the band
    const uint32_t my_cr = cr[lane < FFV2_CODES_PER_BP ? lane : 0]; // lane i: code word i (0: the DC coefficient, 1 + b: band b's gain)
    const uint32_t gb = a.gbase[(size_t)f * (nb + 1) + bp];
    const uint32_t bit0 = a.rawbase[(size_t)f * (nb + 1) + bp];
#pragma unroll
    for (int v = 0; v < (NV + 63) / 64; v++) {
        const int i = lane + 64 * v;
        if (i < NV) sv[i] = stage[v];
    }
    sink[lane] = 0;
    // the fourteen Exp-Golomb codes side by side (ffv2enc.c:105-123), one lane each
    int my_glen = 0;
    unsigned long long my_gcode = 0;
    {
        const int c0 = (int)my_cr;
        const uint32_t val = lane == 0 ? (c0 < 0 ? (uint32_t)(-(long long)c0) : (uint32_t)c0) : my_cr;
        if (lane <= FFV2_NUM_BANDS) my_gcode = golomb_code(val, &my_glen);
    }
    // rows lie back to back: row b starts at the sum of the lengths before it
    uint32_t my_rowoff = my_len;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)my_rowoff, o, 64);
        if (lane >= o) my_rowoff += t;
    }
    my_rowoff -= my_len;                                            // exclusive
    __syncthreads();
    // raw bits of this block-plane, assembled at their bit offset modulo 32 (ffv2enc.c:148-150,174,183-184)
    uint32_t pos = bit0 & 31u;
    auto put = [&](unsigned long long code, int len) {             // lane 0
        if (lane == 0 && len > 0) {
            const uint32_t w = pos >> 5, s = pos & 31u;
            atomicOr(&sink[w], (uint32_t)(code << s));
            const unsigned long long hi = s ? code >> (32 - s) : code >> 32;   // bits past the first word
            if (s) { if ((uint32_t)hi) atomicOr(&sink[w + 1], (uint32_t)hi); if (hi >> 32) atomicOr(&sink[w + 2], (uint32_t)(hi >> 32)); }
            else if ((uint32_t)hi) atomicOr(&sink[w + 1], (uint32_t)hi);
        }
        pos += (uint32_t)len;
    };
    auto code_of = [&](int i, int *len) {                          // lane i's code, for every lane
        *len = __shfl(my_glen, i, 64);
        return ((unsigned long long)(uint32_t)__shfl((int)(my_gcode >> 32), i, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)my_gcode, i, 64);
    };
    {
        const int c0 = (int)(uint32_t)__shfl((int)my_cr, 0, 64);
        int len;
        const unsigned long long code = code_of(0, &len);
        put(code, len);
        if (c0) put(c0 < 0 ? 1u : 0u, 1);
    }
    uint32_t before = 0;
#pragma unroll 1
    for (int b = 0; b < FFV2_NUM_BANDS; b++) {
        const int lo = 1 + LC_BS[b];
        const uint32_t cntb = (uint32_t)__shfl((int)my_cnt, b, 64);
        const uint32_t rb = (uint32_t)__shfl((int)my_rb, b, 64);
        const uint32_t rowoff = (uint32_t)__shfl((int)my_rowoff, b, 64);
        if (lane == 0) a.delta[((size_t)f * 13 + b) * nb + bp] = gb + before - rb;
        int len;
        const unsigned long long code = code_of(1 + b, &len);
        put(code, len);
        uint8_t *dst = rows + rowoff + rb;
        for (uint32_t j0 = 0; j0 < cntb; j0 += 64) {
            const uint32_t j = j0 + (uint32_t)lane;
            const int q = j < cntb ? ys[lo + j] : 0;
            if (j < cntb) dst[j] = (uint8_t)(q < 0 ? -q : q);
            const unsigned long long nzm = __ballot(q != 0);
            if (q < 0) { const uint32_t p = pos + lane_prefix(nzm); atomicOr(&sink[p >> 5], 1u << (p & 31u)); }
            pos += (uint32_t)__popcll(nzm);
        }
        before += cntb;
    }
    __syncthreads();
    {
        const uint32_t nwords = (pos + 31u) >> 5, w0 = bit0 >> 5;
        if ((uint32_t)lane < nwords && sink[lane] && w0 + lane < a.raw_words) atomicOr(&raw[w0 + lane], sink[lane]);
    }
    if (lane == 0 && bp == 0) atomicOr(&raw[0], a.header_bits);
}

// ---------------------------------------------------------------------------------------------
// cdf: one wavefront per (CDF row, frame)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void lc_cdf_kernel(const FFV2LaneCoderArgs a)
{
    __shared__ uint32_t flag[66];
    const int b = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    if (a.abort_[f] || a.status_in[f] < 0) return;
    const int nb = a.nblk, n = a.qp;
    const uint32_t *rowbase = a.rowbase + ((size_t)f * 13 + b) * (nb + 1);
    const uint32_t L = rowbase[nb];
    uint32_t rowoff = 0;
    for (int i = 0; i < b; i++) rowoff += a.rowbase[((size_t)f * 13 + i) * (nb + 1) + nb];
    const uint8_t *src = a.rows + (size_t)f * a.row_stride + rowoff;
    const uint32_t *delta = a.delta + ((size_t)f * 13 + b) * nb;

    uint32_t R = (uint32_t)lane + 1u;            // lane i: entry i of the row (daalaent_cdf_alloc(13, qp, 64, 0, 6, 0))
    uint32_t F0 = (uint32_t)n;                   // entry n-1: advances with the symbol count alone
    uint32_t k0 = 0, bp_cur = 0;
    // this window's records: symbol gp of the frame goes to slot gp - w0 of its lane
    uint2 *recs = lc_record(a.recs + (size_t)a.win_buf * a.buf_stride, a.group_stride, a.width, f, 0);
    const uint32_t w0 = a.win0, w1 = a.win1;
    auto put = [&](uint32_t gp, uint2 r) {
        const uint32_t k = gp - w0;
        recs[((size_t)(k >> 4) * (size_t)a.width) * 16 + (k & 15u)] = r;
    };
    auto inwin = [&](uint32_t gp) { return gp >= w0 && gp < w1; };
    // the row as the previous window left it: entries, total, position, block-plane
    uint32_t *st = a.cdfstate + ((size_t)f * 13 + b) * LC_CDF_STATE;
    if (b == 0) {
        // the symbols that carry no data: header, one "no split" per superblock (in front of its first
        // plane's symbols), the padding of the last tile with symbols of probability one
        const uint32_t *gbase = a.gbase + (size_t)f * (nb + 1);
        if (lane == 0 && w0 == 0) put(0, a.header);
        for (int sb = lane; sb * a.planes < nb; sb += 64) {
            const uint32_t gp = gbase[sb * a.planes] - 1u;
            if (inwin(gp)) put(gp, a.split[sb]);
        }
        const uint32_t nsym = gbase[nb], k = nsym + (uint32_t)lane;
        if (lane < 16 && k < ((nsym + 15u) & ~15u) && inwin(k)) put(k, make_uint2(0x80000000u, 0x8000u));   // fl 0, fh = ft = 32768
    }
    auto chunk_len = [&](uint32_t F, uint32_t k, bool *halve) {
        // symbols until (and including) the one whose update halves the row (daala_entropy.c:434)
        const uint32_t th = F + 64u > 32767u ? 0u : (32768u - 64u - F + 63u) >> 6;
        uint32_t m = L - k < 64u ? L - k : 64u;
        *halve = th + 1u <= m;
        return *halve ? th + 1u : m;
    };
    if (w0 != 0) {                               // not the first window: resume
        k0 = st[65];
        if (k0 >= L) return;                     // the row is through
        R = st[lane]; F0 = st[64]; bp_cur = st[66];
    }
    uint32_t seg_end = bp_cur < (uint32_t)nb ? rowbase[bp_cur + 1u] : 0xFFFFFFFFu;     // end of the current block-plane's band in this row
    uint32_t dl_cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)(bp_cur < (uint32_t)nb ? delta[bp_cur] : 0u));   // ... and its delta
    bool halve = false;
    uint32_t m = k0 < L ? chunk_len(F0, k0, &halve) : 0u;
    // symbols are loaded unconditionally (address clamped to the row) and masked when used: a
    // load under a lane mask makes the compiler wait for it on the spot
    const uint32_t last = L ? L - 1u : 0u;
    auto fetch = [&](uint32_t k) { const uint32_t p = k + (uint32_t)lane; return (uint32_t)src[p < last ? p : last]; };
    // Memory waits.  Loads and stores share one counter on this chip and return out of order with respect to each
    // other, so every wait the compiler inserts is for ALL of them: one chunk at a time the kernel ran at the latency
    // of each chunk's own record store (and of its symbols, asked for one chunk ahead).  Hence groups of LC_GROUP
    // chunks: where the chunks start depends on the symbol count alone, so at the top of a group the symbols of the
    // NEXT group are asked for and the records of the PREVIOUS group (kept in registers) are sent; the one wait per
    // group, where the next group's symbols change registers, then finds everything a whole group old.
    constexpr int LC_GROUP = 4;
    constexpr uint32_t NOREC = 0xFFFFFFFFu;
    auto next_chunk = [&](uint32_t Fp, uint32_t kp, uint32_t mp, bool hp, uint32_t *F, uint32_t *k, uint32_t *mm, bool *h) {
        *F = hp ? ((Fp + 64u * (mp - 1u)) >> 1) + (uint32_t)n + 64u : Fp + 64u * mp;
        *k = kp + mp;
        *h = false;
        *mm = *k < L ? chunk_len(*F, *k, h) : 0u;
    };
    uint32_t gF = F0, gk = k0, gm = m;           // the next chunk to ask for (wave-uniform, runs ahead of F0 / k0 / m)
    bool gh = halve;
    uint32_t xcur[LC_GROUP], xnext[LC_GROUP];
    // total / position / length / "halves the row" of this group's chunks and the next group's: worked out once, where
    // their symbols are asked for (the scalar unit has as much to do in this kernel as the vector units)
    uint32_t cF[LC_GROUP + 1], ck[LC_GROUP + 1], cm[LC_GROUP], nF[LC_GROUP], nk[LC_GROUP], nm[LC_GROUP];
    bool chv[LC_GROUP], nhv[LC_GROUP];
#pragma unroll
    for (int j = 0; j < LC_GROUP; j++) {
        cF[j] = gF; ck[j] = gk; cm[j] = gm; chv[j] = gh;
        xcur[j] = fetch(gk);
        next_chunk(gF, gk, gm, gh, &gF, &gk, &gm, &gh);
    }
    uint2 prec[LC_GROUP], crec[LC_GROUP];        // records of the previous / this group and where they go (byte offset, NOREC: nowhere)
    uint32_t poff[LC_GROUP], coff[LC_GROUP];
#pragma unroll
    for (int j = 0; j < LC_GROUP; j++) { poff[j] = NOREC; prec[j] = make_uint2(0u, 0u); }
    auto send = [&](uint32_t off, uint2 r) {
        if (off != NOREC) *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(recs) + off) = r;
    };
    bool done = false;
    while (!done && k0 < L) {
#pragma unroll
        for (int j = 0; j < LC_GROUP; j++) send(poff[j], prec[j]);
#pragma unroll
        for (int j = 0; j < LC_GROUP; j++) {
            nF[j] = gF; nk[j] = gk; nm[j] = gm; nhv[j] = gh;
            xnext[j] = fetch(gk);
            next_chunk(gF, gk, gm, gh, &gF, &gk, &gm, &gh);
            coff[j] = NOREC; crec[j] = make_uint2(0u, 0u);
        }
        cF[LC_GROUP] = nF[0]; ck[LC_GROUP] = nk[0];            // what follows the group's last chunk
#pragma unroll
        for (int j = 0; j < LC_GROUP; j++) {
            if (!(k0 < L)) { done = true; break; }
            m = cm[j]; halve = chv[j];                          // (F0, k0) == (cF[j], ck[j])
            uint32_t F1 = cF[j + 1];
            uint32_t k1 = ck[j + 1];

            // which block-plane a symbol belongs to.  Usually the whole chunk lies inside the current
            // block-plane's band; otherwise the band ends inside this chunk become flags in LDS and a
            // prefix count over them gives every lane its block-plane.
            uint32_t dl, rel = 0xFFFFFFFFu;
            const bool inside = seg_end - k0 >= m && seg_end - k0 > 0u;
            if (inside) {
                dl = dl_cur;
            } else {
                const uint32_t idx = bp_cur + 1u + (uint32_t)lane;
                const uint32_t rb = idx <= (uint32_t)nb ? rowbase[idx] : 0xFFFFFFFFu;
                rel = idx <= (uint32_t)nb ? rb - k0 : 0xFFFFFFFFu;         // > 0: bp_cur holds symbol k0
                flag[lane] = 0;
                if (lane < 2) flag[64 + lane] = 0;
                __syncthreads();
                if (rel <= 64u) flag[rel] = 1;
                __syncthreads();
                const uint32_t mine = flag[lane];
                const uint32_t seg = bp_cur + lane_prefix(__ballot(mine != 0)) + mine;
                dl = delta[seg < (uint32_t)nb ? seg : (uint32_t)nb - 1u];
            }
            // the window ends inside this chunk (coding-order positions grow with the lane): cut it there
            const uint32_t gp = k0 + (uint32_t)lane + dl;
            const unsigned long long over = __ballot((uint32_t)lane < m && gp >= w1);
            bool last_chunk = false;
            if (over) {
                const uint32_t mc = (uint32_t)__ffsll((long long)over) - 1u;
                if (mc == 0u) { done = true; break; }                      // nothing of this chunk belongs to the window
                m = mc; halve = false; last_chunk = true;
                F1 = F0 + 64u * m; k1 = k0 + m;
            }
            const uint32_t x = (uint32_t)lane < m ? xcur[j] : 255u;

            // prefix counts: cl / ch = symbols of this chunk in front of lane t with a value below / up
            // to lane t's own; ca (lane i as row entry i) = symbols of the chunk with value <= i.  One
            // round per DISTINCT value in the chunk (a band's pulses are mostly 0 and 1), not per value.
            uint32_t cl = 0, ch = 0, ca = 0;
            unsigned long long rem = __ballot(x < 255u);
            while (rem) {
                const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)x, __ffsll((long long)rem) - 1);
                const unsigned long long mk = __ballot(x == v);
                const uint32_t e = lane_prefix(mk);
                if (v < x) cl += e;
                if (v <= x) ch += e;
                if (v <= (uint32_t)lane) ca += (uint32_t)__popcll(mk);
                rem &= ~mk;
            }
            const uint32_t Rlo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((x - 1u) << 2), (int)R);
            const uint32_t Rhi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(x << 2), (int)R);
            uint32_t fl = x ? Rlo + 64u * cl : 0u, fh = Rhi + 64u * ch, ft = F0 + 64u * (uint32_t)lane;
            const int sc = __clz(ft - 1u) - 17;                          // 15 - ilog(ft - 1), daala_entropy.c:346
            fl <<= sc; fh <<= sc; ft <<= sc;

            // the block-plane the row stands in after this chunk
            if (inside) {
                if (seg_end - k0 == m) {                                   // the band ends with this chunk
                    bp_cur++;
                    // (wave-uniform, kept in scalar registers: the wait for them is here, once per band, not at
                    // their use in every chunk)
                    const uint32_t bq = bp_cur < (uint32_t)nb ? bp_cur : (uint32_t)nb - 1u;
                    const uint32_t t0 = rowbase[bq + 1u], t1 = delta[bq];          // both on their way, one wait
                    seg_end = bp_cur < (uint32_t)nb ? (uint32_t)__builtin_amdgcn_readfirstlane((int)t0) : 0xFFFFFFFFu;
                    dl_cur = bp_cur < (uint32_t)nb ? (uint32_t)__builtin_amdgcn_readfirstlane((int)t1) : 0u;
                }
            } else {
                bp_cur += (uint32_t)__popcll(__ballot(rel <= m));
                const uint32_t bq = bp_cur < (uint32_t)nb ? bp_cur : (uint32_t)nb - 1u;
                const uint32_t t0 = rowbase[bq + 1u], t1 = delta[bq];
                seg_end = bp_cur < (uint32_t)nb ? (uint32_t)__builtin_amdgcn_readfirstlane((int)t0) : 0xFFFFFFFFu;
                dl_cur = bp_cur < (uint32_t)nb ? (uint32_t)__builtin_amdgcn_readfirstlane((int)t1) : 0u;
            }
            if ((uint32_t)lane < m) {
                if (x >= (uint32_t)n) atomicOr((int *)&a.abort_[f], 1);       // counted out by the search already
                else {
                    const uint32_t kk = gp - w0;                           // < 2^18 * ...: the offset fits 32 bits (a window's records are < 4 GB)
                    coff[j] = (((kk >> 4) * (uint32_t)a.width) * 16u + (kk & 15u)) * 8u;
                    crec[j] = make_uint2(fl | (fh << 16), ft);
                }
            }

            // the row after this chunk (daala_entropy.c:434-439)
            const uint32_t lastx = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)m - 1);
            const uint32_t ge = lastx <= (uint32_t)lane ? 1u : 0u;
            if (halve) R = ((R + 64u * (ca - ge)) >> 1) + (uint32_t)lane + 1u + 64u * ge;
            else R += 64u * ca;
            F0 = F1; k0 = k1;
            if (last_chunk) { done = true; break; }
        }
#pragma unroll
        for (int j = 0; j < LC_GROUP; j++) {
            xcur[j] = xnext[j]; poff[j] = coff[j]; prec[j] = crec[j];
            cF[j] = nF[j]; ck[j] = nk[j]; cm[j] = nm[j]; chv[j] = nhv[j];
        }
    }
#pragma unroll
    for (int j = 0; j < LC_GROUP; j++) send(poff[j], prec[j]);
    // the row as it stands, for the next window
    st[lane] = R;
    if (lane == 0) { st[64] = F0; st[65] = k0; st[66] = bp_cur; }
}

// ---------------------------------------------------------------------------------------------
// chain: lane = frame
// ---------------------------------------------------------------------------------------------
// One workgroup = two wavefronts over the same 64 frames (lane = frame).  A lone wavefront issues a
// vector instruction only every ~8 cycles (one issue window in two), dependent or not, so the step is split by what the
// next symbol needs: wavefront 0 runs the range recurrence alone (rng -> u, d, next rng) and hands
// (u, d) of every symbol to wavefront 1 through a ring of tiles in LDS; wavefront 1 (another SIMD
// of the CU) turns them into the code words.  Neither ever waits for the other in steady state:
// the consumer's step is half the producer's.
#define LC_RING 8                          // tiles of 16 symbols x 64 lanes x 4 bytes in the ring

typedef unsigned short lc_us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t lc_recur(uint32_t &rng, uint32_t lo, uint32_t ft)
{
    // daala_entropy.c:362-378 in as few instructions as possible (a lone wavefront pays ~8 cycles
    // for each, dependent or not).  The reference scales fl, fh, ft by two when
    // rng - ft >= ft; with t = rng - ft and x = t - ft that is x >= 0 and then d = rng - 2 ft = x,
    // else d = t: d = min(t, x) as unsigned numbers.  g = sat(2 d - ft') with ft' = rng - d is
    // sat(3 d - rng).  Scaling needs ft < 32768, so fh << 1 < 65536 and both halves of `lo` shift in
    // one go; the maps of fl and fh then run as one packed 16-bit computation (every term is at
    // most rng < 65536).
    const uint32_t t = rng - ft;
    // x = t - ft and sc = 1 - borrow (the scale-by-two case) from one 64-bit subtraction: sub, sub-with-borrow
    const unsigned long long w = ((1ull << 32) | t) - ft;
    const uint32_t x = (uint32_t)w, sc = (uint32_t)(w >> 32);
    const uint32_t d = t < x ? t : x;
    const uint32_t g = __builtin_elementwise_sub_sat(3u * d, rng);
    const uint32_t los = lo << sc;
    const lc_us2 P = __builtin_bit_cast(lc_us2, los);
    const lc_us2 G = { (unsigned short)g, (unsigned short)g }, D = { (unsigned short)d, (unsigned short)d };
    const lc_us2 B = __builtin_elementwise_sub_sat(P, G) >> (lc_us2){ 1, 1 };
    // the three terms of each half add up to at most rng < 65536: no carry leaves the low half, so
    // the sums are plain 32-bit additions
    const uint32_t Uw = los + __builtin_bit_cast(uint32_t, __builtin_elementwise_min(P, G))
                            + __builtin_bit_cast(uint32_t, __builtin_elementwise_min(B, D));
    const lc_us2 U = __builtin_bit_cast(lc_us2, Uw);
    const uint32_t r = (uint32_t)U.y - (uint32_t)U.x;                 // 1 <= r < 65536
    const uint32_t dd = (uint32_t)__builtin_clz(r) - 16u;              // 16 - ilog(r), :107-151
    rng = r << dd;
    // u (low half of U, u <= rng < 65536) and the shift in one word: bytes 0, 1 of U, byte 0 of dd, zero
    return __builtin_amdgcn_perm(dd, __builtin_bit_cast(uint32_t, U), 0x0c040100u);
}

// State of one frame's code: where the next symbol's offset lands: bit `o` of word `woff` (the
// depth of the next symbol is 16 woff + 1 - o).  `acc` is that word so far.
struct LcWords {
    uint32_t o, acc, woff;
};

__device__ __forceinline__ void lc_word(LcWords &s, uint32_t ud, uint32_t *words)
{
    // the word is stored every time; its last store stands.  A symbol shifts by <= 15 bits, so the
    // position moves on by at most one word.
    const uint32_t u = ud & 0xFFFFu, dd = ud >> 16;
    s.acc += u << s.o;
    words[s.woff] = s.acc;
    const int o2 = (int)s.o - (int)dd;
    const bool adv = o2 < 0;
    s.woff += adv ? 1u : 0u;
    s.acc = adv ? 0u : s.acc;
    s.o = (uint32_t)o2 & 15u;
}

__global__ __launch_bounds__(128) void lc_chain_kernel(const FFV2LaneCoderArgs a, int nframes)
{
    __shared__ uint32_t ring[LC_RING * 16 * 64];
    __shared__ uint32_t produced, consumed, final_rng[64];
    const int g = blockIdx.x, lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const int f = g * a.width + lane;
    const bool live = lane < a.width && f < nframes;
    // this window: tiles [t0, t1) of every frame's coding order (a tile = 16 symbols)
    const uint32_t t0 = a.win0 >> 4, t1 = a.win1 >> 4;
    uint32_t nsym = 0;
    if (live && a.abort_[f] == 0 && a.status_in[f] >= 0) nsym = a.gbase[(size_t)f * (a.nblk + 1) + a.nblk];
    const uint32_t alltiles = (nsym + 15u) >> 4;
    const uint32_t ntiles = alltiles > t0 ? (alltiles < t1 ? alltiles : t1) - t0 : 0u;   // of this frame, in this window
    uint32_t maxt = ntiles;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)maxt, o, 64); maxt = t > maxt ? t : maxt; }
    if (maxt == 0) return;                                             // no frame of this group reaches into the window
    // The chain is latency, not throughput: one instruction every few cycles per wavefront.  Beside the front of the
    // next call (PVQ search: three wavefronts per SIMD) it must win every issue slot it asks for, or its windows take
    // 48 ms instead of 26; the others lose next to nothing.
    __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x == 0) { produced = 0; consumed = 0; }
    __syncthreads();
    // the two tile counters: relaxed workgroup-scope LDS accesses, ordered against the ring by hand
    // (LDS serves a wavefront's instructions in order; a fence would also wait for the global loads
    // and stores in flight, which is exactly what the prefetch and the word stores must not do)
    auto peek = [](uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto post = [](uint32_t *p, uint32_t v) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    uint32_t *words = a.words + (size_t)(live ? f : 0) * a.wcap;
    const uint32_t wlimit = a.wcap - 2u;                               // a frame that runs past its words is refused by lc_finish_kernel
    // where the previous window left the frame: the range (producer), the word position and the word so far (consumer)
    FFV2LaneState st0{ 0u, 1u, 0x8000u, 0u, 0u, 0u, 0u, 0u };
    if (t0 != 0 && live) st0 = a.state[f];
    LcWords s{ st0.o, st0.acc, st0.woff };

    if (role == 0) {
        // ---- the recurrence ----
        const uint4 *base = reinterpret_cast<const uint4 *>(a.recs + (size_t)a.win_buf * a.buf_stride + (size_t)g * a.group_stride)
                            + (size_t)(live ? lane : 0) * 8;
        const size_t piece = (size_t)a.width * 8;                      // uint4 per row of pieces (a tile: 128 bytes per lane)
        uint32_t rng = st0.rng;
        // records two tiles ahead of the one being worked on (a tile is ~2 500 cycles of recurrence)
        uint4 buf[3][8];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            if ((uint32_t)p < ntiles) {
#pragma unroll
                for (int i = 0; i < 8; i++) buf[p][i] = base[(size_t)p * piece + i];
            }
        }
        for (uint32_t t = 0; t < maxt; t += 3) {
#pragma unroll
            for (int h = 0; h < 3; h++) {
                const uint32_t tt = t + h;
                if (tt >= maxt) break;
                if (tt + 2 < ntiles) {
#pragma unroll
                    for (int i = 0; i < 8; i++) buf[(h + 2) % 3][i] = base[(size_t)(tt + 2) * piece + i];
                }
                while (tt - peek(&consumed) >= (uint32_t)LC_RING) __builtin_amdgcn_s_sleep(2);      // ring full
                asm volatile("" ::: "memory");
                uint32_t *slot = ring + (tt % LC_RING) * (16 * 64) + lane;
                if (tt < ntiles) {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        slot[(2 * i) * 64] = lc_recur(rng, buf[h][i].x, buf[h][i].y);
                        slot[(2 * i + 1) * 64] = lc_recur(rng, buf[h][i].z, buf[h][i].w);
                    }
                }
                if (lane == 0) post(&produced, tt + 1u);
            }
        }
        final_rng[lane] = rng;
    } else {
        // ---- the code words ----
        for (uint32_t tt = 0; tt < maxt; tt++) {
            while (peek(&produced) <= tt) __builtin_amdgcn_s_sleep(2);     // tile not there yet
            asm volatile("" ::: "memory");
            const uint32_t *slot = ring + (tt % LC_RING) * (16 * 64) + lane;
            if (tt < ntiles) {
                uint32_t ud[16];
#pragma unroll
                for (int i = 0; i < 16; i++) ud[i] = slot[i * 64];
#pragma unroll
                for (int i = 0; i < 16; i++) lc_word(s, ud[i], words);
                s.woff = s.woff < wlimit ? s.woff : wlimit;            // 16 symbols move on by <= 15 words: wcap has that slack
            }
            if (lane == 0) post(&consumed, tt + 1u);
        }
    }
    __syncthreads();                                                   // final_rng
    if (role == 1 && live) {
        // the word the next symbol would go to, and a clear one behind it (the next window carries on from here)
        words[s.woff] = s.acc;
        words[s.woff + 1] = 0;
        FFV2LaneState st;
        st.woff = s.woff; st.o = s.o; st.rng = final_rng[lane]; st.full = (st0.full || s.woff >= wlimit) ? 1u : 0u;
        st.acc = s.acc; st.pad0 = st.pad1 = st.pad2 = 0u;
        a.state[f] = st;
    }
}

// ---------------------------------------------------------------------------------------------
// finish: one workgroup per frame
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lc_bytesum(const uint32_t *W, uint32_t i)
{
    // word w holds the code bits of bytes 2w+1 (bits 0-7), 2w (8-15), 2w-1 (16-23), 2w-2 (24-31)
    const uint32_t w = i >> 1, a = W[w], b = W[w + 1];
    return (i & 1u) ? (a & 255u) + ((b >> 16) & 255u) : ((a >> 8) & 255u) + (b >> 24);
}

// finish, step 1 (one wavefront per frame, lane 0 works): ff_daalaent_encode_done's rounding as one
// more addend, the packet's size and status
__global__ __launch_bounds__(64) void lc_size_kernel(const FFV2LaneCoderArgs a)
{
    const int f = blockIdx.x;
    if (threadIdx.x != 0) return;
    uint32_t *W = a.words + (size_t)f * a.wcap;
    int status = a.status_in[f] < 0 ? a.status_in[f] : (a.abort_[f] || a.qp < 2) ? -1 : 0;   // qp 1: ft = 1 < 2, daala_entropy.c:342
    uint32_t nbytes = 0, slack = 0, top = 0, total = 0;
    if (status == 0) {
        const FFV2LaneState st = a.state[f];
        const uint32_t wT = st.woff, oT = st.o, rng = st.rng;
        const uint32_t T = 16u * wT + 1u - oT;                        // all shifts so far
        const uint32_t npre = T >= 1u ? (T - 1u) >> 3 : 0u;
        const int cnt = -9 + (int)(T - 8u * npre);
        // the coder's window (the low cnt + 24 bits of the code so far) read back from the last words
        unsigned long long V = W[wT];
        if (wT >= 1) V += (unsigned long long)W[wT - 1] << 16;
        if (wT >= 2) V += (unsigned long long)W[wT - 2] << 32;
        const uint32_t low = (uint32_t)(V >> oT) & ((1u << (cnt + 24)) - 1u);
        uint32_t m = 0x7FFF, e = (low + m) & ~m;                      // daala_entropy.c:624-674
        int s = 9;
        while ((e | m) >= low + rng) { s++; m >>= 1; e = (low + m) & ~m; }
        s += cnt;
        const uint32_t extra = s > 0 ? (uint32_t)(s + 7) >> 3 : 0u;
        slack = s > 0 ? 8u * extra - (uint32_t)s : (uint32_t)(-s);
        nbytes = npre + extra;
        top = 2u * wT + 2u;
        const uint32_t R = a.rawbase[(size_t)f * (a.nblk + 1) + a.nblk];
        const uint32_t nraw = R > slack ? (R - slack + 7u) >> 3 : 0u;
        total = nbytes + nraw;
        const bool leftover = 8u * nraw < R;
        if (st.full || total > a.packet_stride || (size_t)(R + 31u) / 32u > a.raw_words) status = -28;
        else if (leftover && nbytes == 0) status = -1;                // daala_entropy.c:719
        else W[wT] += (e - low) << oT;
    }
    a.status[f] = status;
    a.sizes[f] = status < 0 ? 0u : total;
    a.fin[f] = make_uint4(nbytes, slack, top, 0u);
}

// finish, step 2 (one workgroup): where every packet starts in the packed output (16-byte aligned)
__global__ __launch_bounds__(256) void lc_offsets_kernel(const FFV2LaneCoderArgs a, int nframes)
{
    __shared__ unsigned long long part[256];
    const int tid = threadIdx.x;
    const int per = (nframes + 255) / 256, f0 = tid * per, f1 = f0 + per < nframes ? f0 + per : nframes;
    unsigned long long sum = 0;
    for (int f = f0; f < f1; f++) sum += (a.sizes[f] + 15u) & ~15u;
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        for (int t = 0; t < 256; t++) { const unsigned long long v = part[t]; part[t] = run; run += v; }
        a.offs[nframes] = run;
    }
    __syncthreads();
    unsigned long long run = part[tid];
    for (int f = f0; f < f1; f++) { a.offs[f] = run; run += (a.sizes[f] + 15u) & ~15u; }
}

// finish, step 3 (one workgroup per frame): the words to bytes through a one-bit carry look-ahead
// (daala_entropy.c:706-715), the raw bytes behind them in reverse order (:676-721)
__global__ __launch_bounds__(256) void lc_write_kernel(const FFV2LaneCoderArgs a)
{
    __shared__ uint32_t fn[256];
    __shared__ uint32_t cin[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    if (a.status[f] < 0) return;
    const uint32_t *W = a.words + (size_t)f * a.wcap;
    const uint32_t *raw = a.raw + (size_t)f * a.raw_words;
    uint8_t *pkt = a.packets + a.offs[f];
    const uint4 fin = a.fin[f];
    const uint32_t nbytes = fin.x, slack = fin.y, top = fin.z, total = a.sizes[f];
    const uint32_t R = a.rawbase[(size_t)f * (a.nblk + 1) + a.nblk];
    const uint32_t nraw = R > slack ? (R - slack + 7u) >> 3 : 0u;
    const bool leftover = 8u * nraw < R;
    // each thread resolves its chunk of the `top` bytes from the end for carry-in 0 and 1, thread 0
    // composes the 256 two-bit functions
    const uint32_t per = (top + 255u) / 256u;
    const uint32_t i0 = (uint32_t)tid * per, i1 = i0 + per < top ? i0 + per : top;
    uint32_t k0 = 0, k1 = 1;
    for (uint32_t i = i1; i-- > i0;) {
        const uint32_t s = lc_bytesum(W, i);
        k0 = (s + k0) >> 8; k1 = (s + k1) >> 8;
    }
    if (i0 >= i1) { k0 = 0; k1 = 1; }
    fn[tid] = k0 | (k1 << 1);
    __syncthreads();
    if (tid == 0) {
        uint32_t k = 0;
        for (int t = 255; t >= 0; t--) { cin[t] = k; k = (fn[t] >> k) & 1u; }
    }
    __syncthreads();
    uint32_t k = cin[tid];
    for (uint32_t i = i1; i-- > i0;) {
        const uint32_t s = lc_bytesum(W, i) + k;
        if (i < nbytes) pkt[i] = (uint8_t)s;
        k = s >> 8;
    }
    for (uint32_t j = (uint32_t)tid; j < nraw; j += 256u) pkt[total - 1u - j] = (uint8_t)(raw[j >> 2] >> (8u * (j & 3u)));
    __syncthreads();
    if (tid == 0 && leftover) pkt[nbytes - 1u] |= (uint8_t)(raw[nraw >> 2] >> (8u * (nraw & 3u)));
}

}  // namespace

hipError_t ffv2_launch_lc_front(const FFV2LaneCoderArgs &a, const int16_t *y, int nframes, bool counted, hipStream_t s)
{
    // a.f0 = index of the first of these frames among the frames in flight; y holds only these frames.
    // counted: a.cnt / a.bits / a.abort_ of these frames are filled in already (ffv2_launch_pvq_counted)
    const size_t nb = (size_t)a.nblk;
    if (!counted)
        hipLaunchKernelGGL(lc_count_kernel, dim3((unsigned)a.nblk, (unsigned)nframes), dim3(64), 0, s,
                           y, a.codes + (size_t)a.f0 * nb * FFV2_CODES_PER_BP, a.qp, a.nblk,
                           a.cnt + (size_t)a.f0 * nb, a.bits + (size_t)a.f0 * nb, a.abort_ + a.f0);
    hipLaunchKernelGGL(lc_scan_kernel, dim3(15, (unsigned)nframes), dim3(64), 0, s, a);
    hipLaunchKernelGGL(lc_scatter_kernel, dim3((unsigned)a.nblk, (unsigned)nframes), dim3(64), 0, s, a, y);
    return hipGetLastError();
}

// cdf and chain of one window of the coding order: symbols [w0, w1) of every frame (w0, w1 multiples of
// 16), records in buffer `buf` (0 | 1).  cdf on stream `sc`, chain on `sk`; the caller orders them.
hipError_t ffv2_launch_lc_cdf(const FFV2LaneCoderArgs &a0, int nframes, uint32_t w0, uint32_t w1, int buf, hipStream_t sc)
{
    FFV2LaneCoderArgs a = a0;
    a.win0 = w0; a.win1 = w1; a.win_buf = buf;
    hipLaunchKernelGGL(lc_cdf_kernel, dim3(13, (unsigned)nframes), dim3(64), 0, sc, a);
    return hipGetLastError();
}

hipError_t ffv2_launch_lc_chain(const FFV2LaneCoderArgs &a0, int nframes, uint32_t w0, uint32_t w1, int buf, hipStream_t sk)
{
    FFV2LaneCoderArgs a = a0;
    a.win0 = w0; a.win1 = w1; a.win_buf = buf;
    hipLaunchKernelGGL(lc_chain_kernel, dim3((unsigned)((nframes + a.width - 1) / a.width)), dim3(128), 0, sk, a, nframes);
    return hipGetLastError();
}

hipError_t ffv2_launch_lc_finish(const FFV2LaneCoderArgs &a, int nframes, hipStream_t s)
{
    hipLaunchKernelGGL(lc_size_kernel, dim3((unsigned)nframes), dim3(64), 0, s, a);
    hipLaunchKernelGGL(lc_offsets_kernel, dim3(1), dim3(256), 0, s, a, nframes);
    hipLaunchKernelGGL(lc_write_kernel, dim3((unsigned)nframes), dim3(256), 0, s, a);
    return hipGetLastError();
}
