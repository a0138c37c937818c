// ffv2_capi.cpp -- host side of the C-ABI declared in include/ffv2_amd.h.
// Owns device workspaces and launches the gfx950 kernels; holds the two small
// pieces of host arithmetic the packet needs:
//   * the coded-gain threshold table (so that the device never evaluates pow),
//   * the data-independent range-coded prefix of a qp == 0 packet.
// There is no CPU encode path in here: without a HIP device every entry point
// that needs one returns FFV2AMD_ERR_DEVICE.
#include "../../include/ffv2_amd.h"
#include "ffv2_kernels.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "gen/scan_lut.h"

#define FFV2AMD_VERSION "ffv2-amd 0.1 (gfx950)"

namespace {

// ------------------------------------------------------------------
// coded gain (ffv2enc.c:131-138,166,174) and its threshold table
// ------------------------------------------------------------------
inline uint32_t coded_gain_host(int64_t e)
{
    volatile float fgain = sqrtf((float)e) + FLT_EPSILON;
    float g = (float)(pow((double)fgain, (double)(1.0f / 1.5f)) / (double)1);
    return (uint32_t)g;
}

constexpr int GAIN_TABLE_N = 1 << 15;   // gains 1..32768 <=> energies up to ~3.5e13
                                        // (Parseval bound of a real block: ~2.4e12)
std::once_flag g_thr_once;
std::vector<int64_t> g_thr;             // g_thr[n] = least energy e with coded_gain(e) >= n+1

void build_gain_table()
{
    g_thr.resize(GAIN_TABLE_N);
    int64_t prev = 0;
    for (int n = 0; n < GAIN_TABLE_N; n++) {
        const uint32_t t = (uint32_t)n + 1;
        // coded_gain(e) ~ e^(1/3): bracket around t^3, then bisect (the function is
        // a monotone staircase in e: int64->float, sqrtf, +eps, pow, casts all are)
        double c = (double)t * t * t;
        int64_t hi = (int64_t)(c * 1.001) + 4;
        while (coded_gain_host(hi) < t) hi = hi * 2 + 1;
        int64_t lo = (int64_t)(c * 0.999) - 4;
        if (lo < prev || coded_gain_host(lo) >= t) lo = prev;        // gain(lo) < t or lo is prev
        if (coded_gain_host(lo) >= t) { g_thr[n] = lo; prev = lo; continue; }
        while (hi - lo > 1) {                                       // gain(lo) < t <= gain(hi)
            int64_t mid = lo + (hi - lo) / 2;
            if (coded_gain_host(mid) >= t) hi = mid; else lo = mid;
        }
        g_thr[n] = hi;
        prev = hi;
    }
}

// ------------------------------------------------------------------
// Daala range encoder, the subset a qp == 0 packet needs
// (daala_entropy.c:107-151 renormalise, :362-378 interval update,
//  :399-410 uint, :428-440 adaptive CDF, :624-674 final bits)
// ------------------------------------------------------------------
struct RangeEnc {
    uint64_t low = 0;
    uint32_t rng = 0x8000;
    int cnt = -9;
    std::vector<uint16_t> pre;

    static int ilog(uint32_t v) { return v ? 32 - __builtin_clz(v) : 0; }

    void encode(uint32_t fl, uint32_t fh, uint32_t ft)      // 16384 <= ft <= 32768 <= rng
    {
        const int sc = (rng - ft) >= ft;
        fl <<= sc; fh <<= sc; ft <<= sc;
        const uint32_t d = rng - ft;
        const uint32_t g = 2 * d > ft ? 2 * d - ft : 0;
        auto map = [&](uint32_t x) {
            const uint32_t a = x < g ? x : g;
            const uint32_t b = (x > g ? x - g : 0) >> 1;
            return x + a + (b < d ? b : d);
        };
        const uint32_t u = map(fl), v = map(fh);
        renorm(low + u, v - u);
    }
    void renorm(uint64_t l, uint32_t r)
    {
        const int d = 16 - ilog(r);
        int c = cnt, s = c + d;
        if (s >= 0) {
            c += 16;
            uint64_t m = ((uint64_t)1 << c) - 1;
            if (s >= 8) { pre.push_back((uint16_t)(l >> c)); l &= m; c -= 8; m >>= 8; }
            pre.push_back((uint16_t)(l >> c));
            s = c + d - 24;
            l &= m;
        }
        low = l << d; rng = r << d; cnt = s;
    }
    // returns the number of unused low bits in the last byte
    int finish(std::vector<uint8_t> &bytes)
    {
        uint64_t m = 0x7FFF, e = (low + m) & ~m;
        int s = 9, c = cnt;
        while ((e | m) >= low + rng) { s++; m >>= 1; e = (low + m) & ~m; }
        s += c;
        if (s > 0) {
            uint64_t n = ((uint64_t)1 << (c + 16)) - 1;
            do { pre.push_back((uint16_t)(e >> (c + 16))); e &= n; s -= 8; c -= 8; n >>= 8; } while (s > 0);
        }
        bytes.resize(pre.size());
        uint32_t carry = 0;
        for (size_t i = pre.size(); i-- > 0;) { carry += pre[i]; bytes[i] = (uint8_t)carry; carry >>= 8; }
        return -s;
    }
};

int range_prefix(int pix_fmt, int num_sb, std::vector<uint8_t> &bytes, int *slack)
{
    if (pix_fmt < 0 || pix_fmt >= 196 || num_sb < 1) return FFV2AMD_ERR_INVAL;
    RangeEnc rc;
    // ff_daalaent_encode_uint(pix_fmt, 196): 13-ary uniform Q15 symbol pix_fmt>>4
    // (daalatab.c row 13: round(32768*k/13)), the low 4 bits travel as raw bits.
    const uint32_t s = (uint32_t)pix_fmt >> 4;
    auto q15 = [](uint32_t k) { return (32768u * k + 6u) / 13u; };
    rc.encode(s ? q15(s) : 0, q15(s + 1), 32768);
    // one adaptive 4-ary symbol "no split" per superblock (ffv2enc.c:222; CDF init
    // {32,64,96,128}, +128 from the coded symbol on, halved past 32767:
    // daala_entropy.h:140-161, daala_entropy.c:434-439)
    uint32_t cdf[4] = { 32, 64, 96, 128 };
    for (int i = 0; i < num_sb; i++) {
        const uint32_t ft = cdf[3];
        const int sc = 15 - RangeEnc::ilog(ft - 1);
        rc.encode(0, cdf[0] << sc, ft << sc);
        if (cdf[3] + 128 > 32767)
            for (int k = 0; k < 4; k++) cdf[k] = (cdf[k] >> 1) + k + 1;
        for (int k = 0; k < 4; k++) cdf[k] += 128;
    }
    *slack = rc.finish(bytes);
    return 0;
}

// ------------------------------------------------------------------
// Whole-packet entropy coder for qp > 0, host side (one serial chain per frame):
// range coder above + the raw-bit tail (daala_entropy.c:227-270) + packet
// assembly (daala_entropy.c:676-721).  The GPU delivers, per block-plane, the
// "DC" slot, the 13 coded gains and the PVQ pulses; the symbol order is
// ffv2enc.c:447-451 (header), :222,:197 (per superblock), :148-150,:174-186.
// ------------------------------------------------------------------
struct PacketEnc {
    RangeEnc rc;
    std::vector<uint8_t> raw;          // raw bytes in write order
    uint64_t win = 0;
    int nwin = 0;
    bool abort_ = false;               // the reference would av_assert0 here

    void bits(uint32_t v, int n)
    {
        if (nwin + n > 64) {
            do { raw.push_back((uint8_t)win); win >>= 8; nwin -= 8; } while (nwin >= 8);
        }
        win |= (uint64_t)v << nwin;
        nwin += n;
    }
    void golomb(uint32_t val)                                  // ffv2enc.c:105-123
    {
        const uint32_t v = val + 1;
        if (val == 0) { bits(1, 1); return; }
        const int nb = 31 - __builtin_clz(v);
        for (int i = nb - 1; i >= 0; i--) bits(((v >> i) & 1) << 1, 2);
        bits(1, 1);
    }
    // adaptive CDF symbol (daala_entropy.c:428-440 over :334-347)
    void adapt(uint16_t *cdf, int n, int inc, int val)
    {
        if (val < 0 || val >= n) { abort_ = true; return; }      // :336
        const uint32_t fl = val ? cdf[val - 1] : 0, fh = cdf[val], ft = cdf[n - 1];
        if (!(fl < fh && fh <= ft && ft >= 2 && ft <= 32768)) { abort_ = true; return; }   // :340-343
        const int sc = 15 - RangeEnc::ilog(ft - 1);
        if ((ft << sc) > rc.rng) { abort_ = true; return; }      // :364
        rc.encode(fl << sc, fh << sc, ft << sc);
        if (cdf[n - 1] + inc > 32767)
            for (int i = 0; i < n; i++) cdf[i] = (uint16_t)((cdf[i] >> 1) + i + 1);
        for (int i = val; i < n; i++) cdf[i] = (uint16_t)(cdf[i] + inc);
    }
    int finish(uint8_t *out, size_t cap, size_t *size)
    {
        std::vector<uint8_t> head;
        const int slack = rc.finish(head);
        while (nwin > slack) { raw.push_back((uint8_t)win); win >>= 8; nwin -= 8; }
        const size_t total = head.size() + raw.size();
        if (total > cap) return FFV2AMD_ERR_NOSPACE;
        memcpy(out, head.data(), head.size());
        for (size_t i = 0; i < raw.size(); i++) out[total - 1 - i] = raw[i];
        if (nwin > 0) {
            if (head.empty()) return FFV2AMD_ERR_ABORT;          // :719
            out[head.size() - 1] |= (uint8_t)win;
        }
        *size = total;
        return 0;
    }
};

const int BANDS_START[14] = { 0, 15, 23, 31, 63, 95, 127, 255, 383, 511, 1023, 1535, 2047, 4096 };

int encode_frame_host_qp(const ffv2amd_info &in, int qp, const uint32_t *codes, const int16_t *y,
                         uint8_t *out, size_t cap, size_t *size)
{
    PacketEnc e;
    // header: ff_daalaent_encode_uint(pix_fmt, 196) then Exp-Golomb(qp)   (ffv2enc.c:447-451)
    {
        const uint32_t s = (uint32_t)in.pix_fmt >> 4;
        auto q15 = [](uint32_t k) { return (32768u * k + 6u) / 13u; };
        e.rc.encode(s ? q15(s) : 0, q15(s + 1), 32768);
        e.bits((uint32_t)in.pix_fmt & 15u, 4);
        e.golomb((uint32_t)qp);
    }
    uint16_t subdiv[4] = { 32, 64, 96, 128 };                    // daalaent_cdf_alloc(1,4,128,0,2,0)
    std::vector<uint16_t> test((size_t)13 * qp);                  // daalaent_cdf_alloc(13,qp,64,0,6,0)
    for (int r = 0; r < 13; r++)
        for (int j = 0; j < qp; j++) test[(size_t)r * qp + j] = (uint16_t)(j + 1);
    const int nsb = in.num_sb_x * in.num_sb_y;
    for (int sb = 0; sb < nsb && !e.abort_; sb++) {
        e.adapt(subdiv, 4, 128, 0);                               // split = END (ffv2enc.c:222)
        e.bits(0, 4);                                             // tx type (ffv2enc.c:197)
        for (int p = 0; p < in.planes && !e.abort_; p++) {
            const size_t bp = (size_t)sb * in.planes + p;
            const uint32_t *rec = codes + bp * FFV2_CODES_PER_BP;
            const int16_t *yy = y + bp * FFV2_Y_STRIDE;
            const int c0 = (int)rec[0];
            e.golomb(c0 < 0 ? (uint32_t)(-(int64_t)c0) : (uint32_t)c0);
            if (c0) e.bits(c0 < 0, 1);
            for (int b = 0; b < 13 && !e.abort_; b++) {
                e.golomb(rec[1 + b]);
                const int lo = 1 + BANDS_START[b], len = BANDS_START[b + 1] - BANDS_START[b];
                int pcnt = 0;
                for (int j = 0; j < len && pcnt < qp && !e.abort_; j++) {   // ffv2enc.c:176-186
                    const int q = yy[lo + j], aq = q < 0 ? -q : q;
                    e.adapt(&test[(size_t)b * qp], qp, 64, aq);
                    if (q) e.bits(q < 0, 1);
                    pcnt += aq;
                }
            }
        }
    }
    if (e.abort_) return FFV2AMD_ERR_ABORT;
    return e.finish(out, cap, size);
}

// Adaptive CDF row of n <= 16*NV symbols as NV vectors of 16 x uint16: the update
// (daala_entropy.c:434-439: halve past 32767, then += inc from the coded symbol on) is two vector
// operations per 16 entries instead of a scalar loop.  Entries past n-1 are never read.
typedef uint16_t v16u __attribute__((vector_size(32)));
typedef int16_t v16s __attribute__((vector_size(32)));

template <int NV>
struct AdaptRow {
    alignas(32) uint16_t c[16 * NV];
    void init() { for (int i = 0; i < 16 * NV; i++) c[i] = (uint16_t)(i + 1); }   // daalaent_cdf_alloc(..., 0, 6, 0)
    inline void update(int n, int inc, int val)
    {
        v16u *v = reinterpret_cast<v16u *>(c);
        const bool halve = c[n - 1] + inc > 32767;
        for (int k = 0; k < NV; k++) {
            v16s idx;
            for (int i = 0; i < 16; i++) idx[i] = (int16_t)(16 * k + i);
            if (halve) v[k] = (v[k] >> 1) + (v16u)(idx + 1);
            const v16s ge = idx >= (int16_t)val;                 // all-ones lanes from the coded symbol on
            v[k] += (v16u)(ge & (int16_t)inc);
        }
    }
};

// Same packet as encode_frame_host_qp, fed from the device's compact symbol stream (int8 pulses,
// only those the coder reads: ffv2_compact_kernel) and with the vectorised CDF rows.
template <int NV>
int encode_frame_host_compact(const ffv2amd_info &in, int qp, const uint32_t *codes, const FFV2SymRec *rec,
                              const int8_t *stream, uint8_t *out, size_t cap, size_t *size)
{
    PacketEnc e;
    const int nsb = in.num_sb_x * in.num_sb_y;
    try {
        e.rc.pre.reserve((size_t)nsb * in.planes * 256 + 1024);
        e.raw.reserve((size_t)nsb * in.planes * 96 + 1024);
    } catch (...) { return FFV2AMD_ERR_NOMEM; }
    {
        const uint32_t s = (uint32_t)in.pix_fmt >> 4;
        auto q15 = [](uint32_t k) { return (32768u * k + 6u) / 13u; };
        e.rc.encode(s ? q15(s) : 0, q15(s + 1), 32768);
        e.bits((uint32_t)in.pix_fmt & 15u, 4);
        e.golomb((uint32_t)qp);
    }
    uint16_t subdiv[4] = { 32, 64, 96, 128 };
    AdaptRow<NV> row[13];
    for (auto &r : row) r.init();
    for (int sb = 0; sb < nsb && !e.abort_; sb++) {
        e.adapt(subdiv, 4, 128, 0);
        e.bits(0, 4);
        for (int p = 0; p < in.planes && !e.abort_; p++) {
            const size_t bp = (size_t)sb * in.planes + p;
            const uint32_t *cr = codes + bp * FFV2_CODES_PER_BP;
            const FFV2SymRec &sr = rec[bp];
            const int8_t *sy = stream + sr.offset;
            const int c0 = (int)cr[0];
            e.golomb(c0 < 0 ? (uint32_t)(-(int64_t)c0) : (uint32_t)c0);
            if (c0) e.bits(c0 < 0, 1);
            for (int b = 0; b < 13 && !e.abort_; b++) {
                e.golomb(cr[1 + b]);
                AdaptRow<NV> &r = row[b];
                const int len = sr.count[b];
                // the coder's state in locals for the length of the band: the CDF stores below
                // are uint16 writes the compiler must otherwise assume to alias it
                uint64_t low = e.rc.low;
                uint32_t rng = e.rc.rng;
                int cnt = e.rc.cnt;
                bool bad = false;
                for (int j = 0; j < len; j++) {
                    const int q = sy[j], aq = q < 0 ? -q : q;
                    if (aq >= qp) { bad = true; break; }                         // daala_entropy.c:336
                    uint32_t fl = aq ? r.c[aq - 1] : 0, fh = r.c[aq], ft = r.c[qp - 1];
                    if (!(fl < fh && fh <= ft && ft >= 2 && ft <= 32768)) { bad = true; break; }
                    int sc = 15 - RangeEnc::ilog(ft - 1);
                    if ((ft << sc) > rng) { bad = true; break; }                 // :364
                    // interval update, daala_entropy.c:362-378 (RangeEnc::encode)
                    sc += (rng - (ft << sc)) >= (ft << sc);
                    fl <<= sc; fh <<= sc; ft <<= sc;
                    const uint32_t d = rng - ft;
                    const uint32_t g = 2 * d > ft ? 2 * d - ft : 0;
                    const uint32_t bl = (fl > g ? fl - g : 0) >> 1, bh = (fh > g ? fh - g : 0) >> 1;
                    const uint32_t u = fl + (fl < g ? fl : g) + (bl < d ? bl : d);
                    const uint32_t v = fh + (fh < g ? fh : g) + (bh < d ? bh : d);
                    // renormalisation, daala_entropy.c:107-151 (RangeEnc::renorm)
                    uint64_t l = low + u;
                    const uint32_t rr = v - u;
                    const int dd = 16 - RangeEnc::ilog(rr);
                    int c = cnt, sh = c + dd;
                    if (sh >= 0) {
                        c += 16;
                        uint64_t m = ((uint64_t)1 << c) - 1;
                        if (sh >= 8) { e.rc.pre.push_back((uint16_t)(l >> c)); l &= m; c -= 8; m >>= 8; }
                        e.rc.pre.push_back((uint16_t)(l >> c));
                        sh = c + dd - 24;
                        l &= m;
                    }
                    low = l << dd; rng = rr << dd; cnt = sh;
                    r.update(qp, 64, aq);
                    if (q) e.bits(q < 0, 1);
                }
                e.rc.low = low; e.rc.rng = rng; e.rc.cnt = cnt;
                if (bad) e.abort_ = true;
                sy += len;
            }
        }
    }
    if (e.abort_) return FFV2AMD_ERR_ABORT;
    return e.finish(out, cap, size);
}

// fn(i) for i in [0, n) on up to `want` host threads (never more than the machine has);
// falls back to the calling thread alone if threads cannot be created.
template <class Fn>
void run_parallel(int n, int want, Fn fn)
{
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 4;
    int nt = want < n ? want : n;
    if (nt > (int)hw) nt = (int)hw;
    if (nt > 64) nt = 64;
    std::atomic<int> next(0);
    auto worker = [&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); };
    std::vector<std::thread> pool;
    try {
        for (int t = 1; t < nt; t++) pool.emplace_back(worker);
    } catch (...) { /* run with what we have */ }
    worker();
    for (auto &t : pool) t.join();
}

// A few persistent host threads that gather the rows of a caller's (pageable) frame into pinned
// memory, slice by slice, each slice's DMA issued by the thread that gathered it.  The calling
// thread takes part; with no workers (thread creation failed) it does everything itself.
struct GatherPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int)> *job = nullptr;
    int nslices = 0, next = 0, done = 0;
    uint64_t gen = 0;
    bool stop = false;

    void start(int n, int device)
    {
        for (int t = 0; t < n; t++) {
            try {
                th.emplace_back([this, device]() {
                    (void)hipSetDevice(device);
                    uint64_t seen = 0;
                    std::unique_lock<std::mutex> lk(mu);
                    for (;;) {
                        cv_work.wait(lk, [&]() { return stop || gen != seen; });
                        if (stop) return;
                        seen = gen;
                        while (next < nslices) {
                            const int i = next++;
                            lk.unlock();
                            (*job)(i);
                            lk.lock();
                            if (++done == nslices) cv_done.notify_all();
                        }
                    }
                });
            } catch (...) { break; }
        }
    }
    void run(int n, const std::function<void(int)> &fn)
    {
        std::unique_lock<std::mutex> lk(mu);
        job = &fn; nslices = n; next = 0; done = 0; gen++;
        cv_work.notify_all();
        while (next < nslices) {
            const int i = next++;
            lk.unlock();
            fn(i);
            lk.lock();
            ++done;
        }
        cv_done.wait(lk, [&]() { return done == nslices; });
        job = nullptr;
    }
    void shutdown()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_work.notify_all();
        for (auto &t : th) t.join();
        th.clear();
        stop = false;
    }
};

// ------------------------------------------------------------------
// Daala range DECODER, the subset an FFV2 packet needs (daala_entropy.c:79-105 fillup / renormalise,
// :200-224 raw bits from the packet's end, :273-326 decode_cdf, :382-396 uint, :413-425 adaptive CDF,
// :564-578 init) -- the entropy half of the decoder-side check (ffv2amd_decode_frame).
// ------------------------------------------------------------------
struct RangeDec {
    const uint8_t *b;
    size_t n, pos = 0, epos;
    uint64_t diff = 0, win = 0;
    uint32_t rng = 0x8000;
    int cnt = -15, nwin = 0;
    bool err = false;

    RangeDec(const uint8_t *buf, size_t size) : b(buf), n(size), epos(size) { fill(); }
    void fill()
    {
        int i = 64 - 9 - (cnt + 15);
        for (; i >= 0 && pos < n; i -= 8, pos++) { diff |= (uint64_t)b[pos] << i; cnt += 8; }
        if (pos >= n) cnt = 16384;                                   // DAALAENT_BIT_ABUNDANCE
    }
    void renorm(uint64_t d, uint32_t r)
    {
        const int i = 16 - RangeEnc::ilog(r);
        diff = d << i; rng = r << i;
        if ((cnt -= i) < 0) fill();
    }
    static uint32_t sat(uint32_t a, uint32_t c) { return a - (a < c ? a : c); }
    int decode(const uint16_t *cdf, int nsym, bool q15)
    {
        const int64_t cval = (int64_t)(diff >> 48);
        if ((uint64_t)cval >= rng) { err = true; return 0; }          // :283
        uint32_t ft, d;
        int scale;
        if (!q15) {
            ft = cdf[nsym - 1];
            if (ft < 2 || ft > 32768) { err = true; return 0; }
            scale = 15 - RangeEnc::ilog(ft - 1);
            ft <<= scale;
            if (ft > rng) { err = true; return 0; }
            if (rng - ft >= ft) { ft <<= 1; scale++; }
            d = rng - ft;
        } else {
            if (cdf[nsym - 1] != 32768 || rng < 32768) { err = true; return 0; }
            d = rng - 32768; ft = 32768; scale = 0;
        }
        const uint32_t g = sat(2 * d, ft);
        int64_t lim = cval >> 1;
        if (cval - (int64_t)d > lim) lim = cval - (int64_t)d;
        const int64_t third = (2 * cval + 1 - (int64_t)g) / 3;          // C division, toward zero
        if (third > lim) lim = third;
        lim >>= scale;
        int ret = 0;
        uint32_t u = 0, v;
        for (v = cdf[ret]; (int64_t)v <= lim; v = cdf[++ret]) {
            u = v;
            if (ret + 1 >= nsym) { err = true; return 0; }
        }
        u <<= scale; v <<= scale;
        const uint32_t bu = sat(u, g) >> 1, bv = sat(v, g) >> 1;
        u = u + (u < g ? u : g) + (bu < d ? bu : d);
        v = v + (v < g ? v : g) + (bv < d ? bv : d);
        renorm(diff - ((uint64_t)u << 48), v - u);
        return ret;
    }
    uint32_t bits(int num)
    {
        if (nwin < num) {
            do {
                if (epos == 0) { nwin = 16384; break; }
                win |= (uint64_t)b[--epos] << nwin;
                nwin += 8;
            } while (nwin <= 64 - 8);
        }
        const uint32_t r = (uint32_t)(win & (((uint64_t)1 << num) - 1));
        win >>= num; nwin -= num;
        return r;
    }
    uint32_t uint_(uint32_t num)                                       // num > 16
    {
        num--;
        const int bit = RangeEnc::ilog(num) - 4, adr = (int)(num >> bit) + 1;
        uint16_t cdf[16];
        for (int k = 0; k < adr; k++) cdf[k] = (uint16_t)((32768u * (uint32_t)(k + 1) + (uint32_t)adr / 2) / (uint32_t)adr);
        cdf[adr - 1] = 32768;
        uint32_t t = (uint32_t)decode(cdf, adr, true);
        t = (t << bit) | bits(bit);
        if (t <= num) return t;
        err = true;
        return num;
    }
    int adapt(uint16_t *cdf, int nsym, int inc)
    {
        const int r = decode(cdf, nsym, false);
        if (err) return 0;
        if (cdf[nsym - 1] + inc > 32767)
            for (int i = 0; i < nsym; i++) cdf[i] = (uint16_t)((cdf[i] >> 1) + i + 1);
        for (int i = r; i < nsym; i++) cdf[i] = (uint16_t)(cdf[i] + inc);
        return r;
    }
    uint32_t golomb()                                                  // ffv2dec.c:76-86
    {
        uint32_t c = 1;
        int guard = 0;
        while (!bits(1)) {
            c = (c << 1) | bits(1);
            if (++guard > 40) { err = true; break; }
        }
        return c - 1;
    }
};

int pixfmt_info(int pix_fmt, int *planes, int *depth)
{
    switch (pix_fmt) {                    // allowed_pix_fmts, ffv2enc.c:596-601
    case FFV2AMD_PIX_GRAY8:       *planes = 1; *depth = 8;  return 0;
    case FFV2AMD_PIX_YUV444P:
    case FFV2AMD_PIX_GBRP:        *planes = 3; *depth = 8;  return 0;
    case FFV2AMD_PIX_YUV444P10LE:
    case FFV2AMD_PIX_GBRP10LE:    *planes = 3; *depth = 10; return 0;
    case FFV2AMD_PIX_YUV444P12LE:
    case FFV2AMD_PIX_GBRP12LE:    *planes = 3; *depth = 12; return 0;
    }
    return FFV2AMD_ERR_INVAL;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct ffv2amd_encoder;
static void lanecoder_free(ffv2amd_encoder *e);

struct ffv2amd_encoder {
    ffv2amd_info info{};
    FFV2Geom geom{};
    int device = 0;
    hipStream_t stream = nullptr;
    // shared tables
    int64_t  *d_thr = nullptr;
    uint16_t *d_lds_scan = nullptr;
    uint8_t  *d_prefix = nullptr;
    int prefix_len = 0, slack = 0;
    // per-batch workspace
    uint32_t *d_codes = nullptr, *d_bitoff = nullptr;
    int32_t  *d_status = nullptr;
    int32_t  *d_err = nullptr;           // [2][max_batch] T-stage error flags of the (up to two) batch calls in flight
    // single-frame host path
    uint8_t  *d_frame = nullptr, *d_pkt = nullptr;
    uint32_t *d_meta = nullptr;          // [0] size, [1] status
    int32_t  *d_w1 = nullptr;
    uint8_t  *h_frame = nullptr, *h_pkt = nullptr;
    uint32_t *h_meta = nullptr;
    // qp > 0 workspace (allocated on first use)
    int32_t *d_coef_ws = nullptr;
    int16_t *d_y = nullptr, *h_y = nullptr;
    uint32_t *h_codes = nullptr;
    uint8_t *d_pk_ws = nullptr;
    int32_t *d_inv_plane = nullptr;
    uint32_t *d_sizes_ws = nullptr;
    // options
    int32_t *coef_sink = nullptr;
    int profiling = 0;                   // 0 off, n: HIP timing events around every n-th batch call
    unsigned prof_calls = 0;
    struct EvTriple { hipEvent_t a, b, c, d; bool d_is_b; };     // T start, T end, E end, E start (== T end when both stages share a stream)
    // pipelined mode: the E-stage of call n runs on e_stream while the caller's stream
    // already carries the T-stage of call n+1; two sets of T->E hand-off buffers
    bool pipelined = false;
    hipStream_t e_stream = nullptr;
    hipEvent_t evT[2] = { nullptr, nullptr }, evE[2] = { nullptr, nullptr };
    bool evE_valid[2] = { false, false };
    uint32_t *d_codes2 = nullptr, *d_bitoff2 = nullptr;
    int32_t *d_status2 = nullptr;
    unsigned seq = 0;
    int seq_pb = 0;                      // hand-off buffer set of the call being issued
    std::vector<EvTriple> ev_pool;       // reused
    size_t ev_used = 0;
    double prof_t = 0, prof_e = 0, prof_tmin = 0, prof_tmax = 0;   // sums (and T-stage extremes) of drained events
    int prof_n = 0;
    // qp > 0 pipeline (ffv2amd_qp_submit / _finish): two sets, so that the host coder of batch n
    // runs while the GPU works on batch n+1
    struct QpSet {
        FFV2SymRec *d_rec = nullptr, *h_rec = nullptr;
        int8_t *d_stream = nullptr, *h_stream = nullptr;
        uint32_t *d_totals = nullptr, *h_totals = nullptr;
        uint32_t *d_codes = nullptr, *h_codes = nullptr;
        int32_t *d_status = nullptr, *h_status = nullptr;
        hipEvent_t ev = nullptr;
        std::vector<hipEvent_t> ev_frame;
        int nframes = 0, qp = 0;
        bool busy = false;
        // device coder (ffv2amd_encoder_set_device_coder): scratch and packets in HBM
        uint16_t *d_pre = nullptr;
        uint8_t *d_raw = nullptr, *d_pk = nullptr;
        uint32_t *d_pk_sizes = nullptr, *h_pk_sizes = nullptr;
        int32_t *d_pk_status = nullptr, *h_pk_status = nullptr;
        bool on_device = false;          // this batch was coded by the device coder
        const uint8_t *d_frames = nullptr;   // the batch's frames and phantom words: a frame the fast T-stage refuses is rerun wide at finish
        const int32_t *d_W = nullptr;
    };
    bool device_coder = false;
    QpSet qset[2];
    size_t q_stream_stride = 0;
    hipStream_t q_copy = nullptr;
    unsigned q_sub = 0, q_fin = 0;
    // host frames through that pipeline (ffv2amd_qp_send_frame / _receive_packet): one frame per batch
    uint8_t *qh_frame[2] = { nullptr, nullptr }, *qd_frame[2] = { nullptr, nullptr };
    int32_t *qd_w[2] = { nullptr, nullptr };
    uint8_t *qd_c420[2] = { nullptr, nullptr };          // U, V of a 4:2:0 frame (ffv2amd_qp_send_frame_420)
    int64_t q_tag[2] = { 0, 0 };
    // decoder-side check (ffv2amd_decode_frame)
    int16_t *d_dec_pulses = nullptr;
    float *d_dec_mag = nullptr;
    int32_t *d_dec_c0 = nullptr, *d_dec_coef = nullptr;
    uint8_t *d_dec_frame = nullptr;
    // wide (plain int32) T-stage of one frame: the rerun of frames the fast kernels refuse (ffv2_wide.hip)
    int32_t *d_wide_plane = nullptr, *d_wide_c0 = nullptr;
    int64_t *d_wide_en = nullptr;
    // 4:2:0 -> 4:4:4 front end (ffv2amd_*_420)
    FFV2Upconv *upconv = nullptr;
    bool upconv_tried = false;
    uint8_t *d_420 = nullptr, *h_420 = nullptr;
    // asynchronous frame ring (ffv2amd_ring_*)
    struct RingSlot {
        uint8_t  *h_frame = nullptr, *d_frame = nullptr;    // pinned staging frame, device frame
        uint8_t  *d_pkt = nullptr, *h_pkt = nullptr;
        uint32_t *d_meta = nullptr, *h_meta = nullptr;      // [0] size, [1] status
        uint32_t *d_codes = nullptr, *d_bitcnt = nullptr;
        int32_t  *d_w = nullptr;
        bool has_w = false;
        uint8_t  *d_c420 = nullptr;                         // U, V of a 4:2:0 frame, rows c_pitch apart (ring_send_420)
        hipEvent_t ev_h2d = nullptr, ev_done = nullptr, ev_meta = nullptr;
        int64_t tag = 0;
    };
    std::vector<RingSlot> ring;
    int ring_head = 0, ring_count = 0;
    hipStream_t ring_h2d = nullptr, ring_comp[2] = { nullptr, nullptr }, ring_d2h = nullptr, ring_pkt = nullptr;
    GatherPool *ring_pool = nullptr;
    unsigned ring_seq = 0;
    // FFV2AMD_FRAME_REGISTER: host ranges this ring page-locked itself (frames from a pool of long-lived buffers)
    struct RegRange { const uint8_t *base; size_t bytes; };
    std::vector<RegRange> ring_reg;
    // qp > 0 coder with many frames in flight (ffv2amd_lanecoder_*, ffv2_lanecoder.hip)
    struct LaneCoder {
        int cap = 0;                     // frames in flight per call
        int group = 0;                   // frames per launch of the front (T-stage, PVQ search, scan, scatter): the coder's own
        int32_t *d_coef = nullptr;       // workspaces for that many frames -- not the encoder's max_batch, which belongs to the
        int16_t *d_y = nullptr;          // batch entry points (the codec shim creates its encoders with max_batch 1)
        uint32_t *d_bitcnt = nullptr;
        FFV2LaneCoderArgs a{};           // geometry; the scratch pointers are filled in per call from its Set and its Back
        uint2 *d_split = nullptr;
        // the front's buffers, twice: call n+1's front runs beside call n's chain
        struct Set {
            uint32_t *d_codes = nullptr, *bits = nullptr, *rowbase = nullptr, *gbase = nullptr, *rawbase = nullptr, *delta = nullptr;
            int32_t *d_status_in = nullptr, *abort_ = nullptr, *status = nullptr;
            FFV2SymRec *cnt = nullptr;
            uint8_t *rows = nullptr, *packets = nullptr;
            uint32_t *raw = nullptr, *sizes = nullptr;
            unsigned long long *offs = nullptr, *h_offs = nullptr;
            uint32_t *h_sizes = nullptr;
            int32_t *h_status = nullptr;
            hipEvent_t ev_front = nullptr, ev_done = nullptr;
            hipEvent_t ev_back0 = nullptr, ev_chain0 = nullptr, ev_chain1 = nullptr;   // timing: back begins, chain begins / ends
            int nframes = 0;
            bool busy = false;
        } set[4];
        int nsets = 2;                   // calls in flight (2 to 4): how many of the sets are in use
        // the back's scratch (records of two windows, code words, the lanes' state) and its streams.  One of them: one
        // call's chain at a time, the next call's front beside it.  Two or more: call n takes back n % nback, so that
        // many chains run side by side -- what a call costs is its chain's latency (one frame's symbols, whatever the
        // call holds), and where the memory holds too few frames to cover it with the next call's front (large pictures)
        // the second chain does.
        struct Back {
            uint2 *recs = nullptr;
            uint32_t *words = nullptr, *cdfstate = nullptr;
            FFV2LaneState *state = nullptr;
            uint4 *fin = nullptr;
            hipStream_t back = nullptr, cdfs = nullptr;                          // chain + finish | cdf windows
            hipEvent_t ev_cdf[2] = { nullptr, nullptr }, ev_chain[2] = { nullptr, nullptr };   // per record buffer: filled / read
            hipEvent_t ev_backdone = nullptr;                                    // the previous call here has let go of the scratch
            bool backdone_valid = false;
        } bk[4];
        int nback = 1;
        hipStream_t copy = nullptr;      // packets out
        uint32_t maxsym16 = 0, window = 0;                                  // symbols per frame at most; symbols per window
        unsigned sub = 0, fin = 0;
        float last_chain_ms = 0, last_back_ms = 0;   // of the call finished last (ffv2amd_lanecoder_stats)
        uint32_t last_symbols0 = 0;
        std::vector<void *> allocs;      // every device buffer above, for close
    } lc;
    // send_frame / receive_packet at qp > 0 on top of the lane coder (ffv2amd_qpring_*): frames are collected a batch
    // at a time in device memory, every full batch is one lane coder call, packets come back in send order
    struct QpRing {
        int qp = 0, cap = 0;                     // frames per batch (= per lane coder call)
        size_t pcap = 0;
        static constexpr int NBUF = 5;           // a batch being filled + up to four in flight (`calls` of them in use, + 1)
        int calls = 2;                           // lane coder calls in flight
        uint8_t *d_frames[NBUF] = {};
        uint8_t *d_c420[NBUF] = {};              // 4:2:0 chroma as it arrives: [cap][U plane, V plane]
        int32_t *d_w[NBUF] = {};
        bool any_w[NBUF] = {};
        std::vector<int64_t> tags[NBUF];
        std::vector<uint8_t> is420[NBUF];        // per frame of the batch: its chroma waits in d_c420 for the up-conversion
        int fill = 0, count = 0;                 // buffer being filled, frames in it
        int flight[4] = { -1, -1, -1, -1 };      // buffers of the calls in flight, oldest first
        int flight_n[4] = { 0, 0, 0, 0 };
        int nflight = 0;
        hipStream_t h2d = nullptr;
        hipEvent_t ev_batch = nullptr;
        // page-locked bounce frames for pageable callers
        static constexpr int NBOUNCE = 64;       // at most; nbounce of them in use: about 256 MB (a copy queued behind the
        uint8_t *bounce[NBOUNCE] = {};           // coder's kernels takes a millisecond to start: with four frames of
        hipEvent_t ev_bounce[NBOUNCE] = {};      // bounce memory the sender waited for every one of them)
        int nbounce = 4;
        unsigned nb_seq = 0;
        GatherPool *pool = nullptr;              // helper threads for the row copies of pageable frames
        // the batch finished last, waiting to be received
        uint8_t *h_buf = nullptr;
        size_t h_cap = 0;
        std::vector<uint64_t> offs;
        std::vector<uint32_t> sizes;
        std::vector<int32_t> status;
        std::vector<int64_t> done_tags;
        int done_n = 0, done_at = 0;
        bool open = false;
    } qr;
};

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "ffv2amd: %s failed: %s\n", #x, hipGetErrorString(e_)); \
    (void)hipGetLastError();      /* reported here: must not surface again in a later launch check */ \
    return FFV2AMD_ERR_DEVICE; } } while (0)

// Every entry point runs on the encoder's device and leaves the calling thread's current
// device as it found it (a host application may drive several GPUs from one thread).
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == dev) || hipSetDevice(dev) == hipSuccess;
        if (prev == dev) prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

extern "C" void ffv2amd_ring_close(ffv2amd_encoder *e);
static int encode_uploaded_frame(ffv2amd_encoder *e, int qp, const int32_t *W, uint8_t *out, size_t out_cap, size_t *out_size);

// T-stage + E-stage (qp == 0) of `nframes` frames: the one launch sequence behind
// ffv2amd_encode_batch_device and the frame ring.  `st`/`se`: streams of the two stages.
static int launch_encode_qp0(ffv2amd_encoder *e, int nframes, const void *d_frames, const int32_t *d_W,
                             void *d_packets, size_t packet_stride, uint32_t *d_sizes, int32_t *status,
                             uint32_t *codes, uint32_t *bitcnt, int32_t *coef, hipStream_t st, hipStream_t se,
                             ffv2amd_encoder::EvTriple *ev, int32_t *err)
{
    // No memset in front: `err` (zero between calls: cleared at allocation and by every E-stage)
    // collects the T-stage's errors, the E-stage writes `status` from it.  The packet buffers are
    // cleared by the T-stage itself (FFV2TStageArgs::zero).
    FFV2TStageArgs a{};
    a.g = e->geom; a.nframes = nframes; a.frames = (const uint8_t *)d_frames;
    a.coef = coef; a.energy = nullptr; a.codes = codes; a.bitcnt = bitcnt; a.W = d_W;
    a.gain_thr = e->d_thr; a.gain_n = GAIN_TABLE_N; a.lds_scan = e->d_lds_scan;
    a.status = err;
    a.zero = (uint32_t *)d_packets; a.zero_stride_dw = (uint32_t)(packet_stride / 4);
    if (ev) HIPCHK(hipEventRecord(ev->a, st));
    HIPCHK(ffv2_launch_tstage(a, st));
    if (ev) HIPCHK(hipEventRecord(ev->b, st));
    if (se != st) {
        hipEvent_t join = e->evT[e->seq_pb];
        HIPCHK(hipEventRecord(join, st));
        HIPCHK(hipStreamWaitEvent(se, join, 0));
    }
    if (ev && se != st) HIPCHK(hipEventRecord(ev->d, se));       // one stream: the T-stage's end event is the E-stage's start
    if (ev) ev->d_is_b = se == st;
    FFV2EStageArgs b{};
    b.g = e->geom; b.nframes = nframes; b.codes = codes; b.bitoff = bitcnt;
    b.packets = (uint8_t *)d_packets; b.packet_stride = packet_stride;
    b.sizes = d_sizes; b.status = status; b.err = err;
    b.prefix = e->d_prefix; b.prefix_len = e->prefix_len; b.slack_bits = e->slack;
    // raw header: pix_fmt & 15 (daala_entropy.c:406), then Exp-Golomb(qp = 0) = "1"
    b.header_bits = ((uint32_t)e->info.pix_fmt & 15u) | (1u << 4);
    b.header_nbits = 5;
    HIPCHK(ffv2_launch_estage_qp0(b, se));
    if (ev) HIPCHK(hipEventRecord(ev->c, se));
    return FFV2AMD_OK;
}

extern "C" {

const char *ffv2amd_version(void) { return FFV2AMD_VERSION; }

uint32_t ffv2amd_coded_gain(int64_t energy) { return coded_gain_host(energy); }

int ffv2amd_range_prefix(int pix_fmt, int num_sb, uint8_t *out, size_t cap, int *slack_bits)
{
    std::vector<uint8_t> b;
    int sl = 0;
    int r = range_prefix(pix_fmt, num_sb, b, &sl);
    if (r < 0) return r;
    if (b.size() > cap) return FFV2AMD_ERR_NOSPACE;
    if (out) memcpy(out, b.data(), b.size());
    if (slack_bits) *slack_bits = sl;
    return (int)b.size();
}

void ffv2amd_encoder_destroy(ffv2amd_encoder *e)
{
    if (!e) return;
    DeviceGuard guard(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    ffv2amd_qpring_close(e);
    ffv2amd_ring_close(e);
    ffv2_upconv_destroy(e->upconv);
    (void)hipFree(e->d_420);
    if (e->h_420) (void)hipHostFree(e->h_420);
    lanecoder_free(e);
    for (auto &q : e->qset) {
        (void)hipFree(q.d_rec); (void)hipFree(q.d_stream); (void)hipFree(q.d_totals); (void)hipFree(q.d_codes); (void)hipFree(q.d_status);
        if (q.h_rec) (void)hipHostFree(q.h_rec);
        if (q.h_stream) (void)hipHostFree(q.h_stream);
        if (q.h_totals) (void)hipHostFree(q.h_totals);
        if (q.h_codes) (void)hipHostFree(q.h_codes);
        if (q.h_status) (void)hipHostFree(q.h_status);
        if (q.ev) (void)hipEventDestroy(q.ev);
        for (auto ev : q.ev_frame) if (ev) (void)hipEventDestroy(ev);
        (void)hipFree(q.d_pre); (void)hipFree(q.d_raw); (void)hipFree(q.d_pk); (void)hipFree(q.d_pk_sizes); (void)hipFree(q.d_pk_status);
        if (q.h_pk_sizes) (void)hipHostFree(q.h_pk_sizes);
        if (q.h_pk_status) (void)hipHostFree(q.h_pk_status);
    }
    if (e->q_copy) { (void)hipStreamSynchronize(e->q_copy); (void)hipStreamDestroy(e->q_copy); }
    for (int k = 0; k < 2; k++) {
        (void)hipFree(e->qd_frame[k]); (void)hipFree(e->qd_w[k]); (void)hipFree(e->qd_c420[k]);
        if (e->qh_frame[k]) (void)hipHostFree(e->qh_frame[k]);
    }
    (void)hipFree(e->d_thr); (void)hipFree(e->d_lds_scan); (void)hipFree(e->d_prefix);
    (void)hipFree(e->d_codes); (void)hipFree(e->d_bitoff); (void)hipFree(e->d_status); (void)hipFree(e->d_err);
    (void)hipFree(e->d_frame); (void)hipFree(e->d_pkt); (void)hipFree(e->d_meta); (void)hipFree(e->d_w1);
    (void)hipFree(e->d_inv_plane);
    (void)hipFree(e->d_wide_plane); (void)hipFree(e->d_wide_c0); (void)hipFree(e->d_wide_en);
    (void)hipFree(e->d_dec_pulses); (void)hipFree(e->d_dec_mag); (void)hipFree(e->d_dec_c0); (void)hipFree(e->d_dec_coef);
    (void)hipFree(e->d_dec_frame);
    (void)hipFree(e->d_coef_ws); (void)hipFree(e->d_y); (void)hipFree(e->d_pk_ws); (void)hipFree(e->d_sizes_ws);
    if (e->h_y) (void)hipHostFree(e->h_y);
    if (e->h_codes) (void)hipHostFree(e->h_codes);
    if (e->h_frame) (void)hipHostFree(e->h_frame);
    if (e->h_pkt) (void)hipHostFree(e->h_pkt);
    if (e->h_meta) (void)hipHostFree(e->h_meta);
    for (auto &t : e->ev_pool) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); (void)hipEventDestroy(t.c); (void)hipEventDestroy(t.d); }
    if (e->e_stream) { (void)hipStreamSynchronize(e->e_stream); (void)hipStreamDestroy(e->e_stream); }
    for (int i = 0; i < 2; i++) { if (e->evT[i]) (void)hipEventDestroy(e->evT[i]); if (e->evE[i]) (void)hipEventDestroy(e->evE[i]); }
    (void)hipFree(e->d_codes2); (void)hipFree(e->d_bitoff2); (void)hipFree(e->d_status2);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    (void)hipGetLastError();              // nothing refused above may surface in another encoder's launch check
    delete e;
}

int ffv2amd_encoder_create(ffv2amd_encoder **out, int width, int height, int pix_fmt,
                           int device, int max_batch)
{
    if (!out) return FFV2AMD_ERR_INVAL;
    *out = nullptr;
    int planes, depth;
    if (pixfmt_info(pix_fmt, &planes, &depth) < 0) return FFV2AMD_ERR_INVAL;
    if (width < 1 || height < 1 || width > 65536 || height > 65536 || max_batch < 1) return FFV2AMD_ERR_INVAL;

    // The ring runs five HIP streams side by side and the lane coder four; the runtime multiplexes streams
    // onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams that share a queue wait for each
    // other's kernels (measured, round 3: pageable 4:2:0 frames through the ring 12.1 -> 16.1 Gpix/s, the
    // lane coder 5.7 -> 7.0 Gpix/s with 8 queues).  The variable is read when the runtime initialises, so
    // this only helps a process whose first HIP call is ours; others set it themselves.
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        fprintf(stderr, "ffv2amd: no usable HIP device %d (found %d) -- this library has no CPU path\n", device, ndev);
        return FFV2AMD_ERR_DEVICE;
    }
    DeviceGuard guard(device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;

    ffv2amd_encoder *e = new (std::nothrow) ffv2amd_encoder;
    if (!e) return FFV2AMD_ERR_NOMEM;
    e->device = device;
    ffv2amd_info &in = e->info;
    in.width = width; in.height = height; in.pix_fmt = pix_fmt;
    in.planes = planes; in.depth = depth;
    in.num_sb_x = (width + 63) / 64;                         // ffv2enc.c:503-504
    in.num_sb_y = (height + 63) / 64;
    in.block_planes = in.num_sb_x * in.num_sb_y * planes;
    in.max_batch = max_batch;
    const int bps = depth > 8 ? 2 : 1;
    in.row_pitch = align_up((size_t)width * bps, 128);
    in.plane_stride = align_up(in.row_pitch * height, 256);
    in.frame_stride = in.plane_stride * planes;
    if (in.plane_stride >= ((size_t)1 << 32)) return (delete e, FFV2AMD_ERR_INVAL);   // 32-bit in-plane offsets in the kernels
    in.tstage_bytes_per_frame = (size_t)planes * width * height * bps +
                                (size_t)planes * (64 * in.num_sb_x) * (64 * in.num_sb_y) * 4;

    std::vector<uint8_t> prefix;
    int r = range_prefix(pix_fmt, in.num_sb_x * in.num_sb_y, prefix, &e->slack);
    if (r < 0) { delete e; return r; }
    e->prefix_len = (int)prefix.size();
    // worst case raw bits per block-plane on the device path: c0 (|c0| < 2^22: 43+1 bits) + 13 gains
    // (<= 2^15: 31 bits each) + 4 tx bits per superblock  -> < 58 bytes (what the E-stage's LDS holds);
    // a frame with samples above its depth, coded on the host from the wide T-stage: c0 any int32
    // (63+1 bits), gains < 2^22 (43 bits each) -> < 80 bytes
    in.packet_cap = align_up((size_t)e->prefix_len + 16 + (size_t)in.block_planes * 80 + 64, 256);
    in.packet_cap_qp = in.packet_cap + (size_t)in.block_planes * 2100;   // generous; NOSPACE if ever exceeded

    FFV2Geom &g = e->geom;
    g.width = width; g.height = height; g.depth = depth; g.planes = planes; g.bytes_per_sample = bps;
    g.nsx = in.num_sb_x; g.nsy = in.num_sb_y; g.nblk = in.block_planes;
    g.row_pitch = in.row_pitch; g.plane_stride = in.plane_stride; g.frame_stride = in.frame_stride;
    g.inv_planes = (uint32_t)(((uint64_t)1 << 32) / (uint32_t)planes + 1);
    g.inv_nsx = (uint32_t)(((uint64_t)1 << 32) / (uint32_t)g.nsx + 1);

    std::call_once(g_thr_once, build_gain_table);
    // phase-F scan table: [i][lane][e] -> byte offset of coding index
    // q = 256*(2i + e/4) + 4*lane + e%4 in the raster buffer (row pitch 69 dwords)
    uint16_t lds_scan[4096];                                  // per call: create() may run concurrently (INIT_THREADSAFE)
    for (int i = 0; i < 8; i++)
        for (int l = 0; l < 64; l++)
            for (int e = 0; e < 8; e++) {
                const int q = 256 * (2 * i + e / 4) + 4 * l + (e & 3);
                lds_scan[(i * 64 + l) * 8 + e] =
                    (uint16_t)(((FFV2_SCAN_LUT[q] >> 6) * 69 + (FFV2_SCAN_LUT[q] & 63)) * 4);
            }

#define CK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "ffv2amd: %s failed: %s\n", #x, hipGetErrorString(hipGetLastError())); \
    ffv2amd_encoder_destroy(e); return FFV2AMD_ERR_DEVICE; } } while (0)
    CK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    CK(hipMalloc(&e->d_thr, sizeof(int64_t) * GAIN_TABLE_N));
    CK(hipMemcpy(e->d_thr, g_thr.data(), sizeof(int64_t) * GAIN_TABLE_N, hipMemcpyHostToDevice));
    CK(hipMalloc(&e->d_lds_scan, sizeof(lds_scan)));
    CK(hipMemcpy(e->d_lds_scan, lds_scan, sizeof(lds_scan), hipMemcpyHostToDevice));
    CK(hipMalloc(&e->d_prefix, align_up(prefix.size(), 16)));
    CK(hipMemcpy(e->d_prefix, prefix.data(), prefix.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&e->d_codes, sizeof(uint32_t) * FFV2_CODES_PER_BP * g.nblk * (size_t)max_batch));
    CK(hipMalloc(&e->d_bitoff, sizeof(uint32_t) * g.nblk * (size_t)max_batch));
    CK(hipMalloc(&e->d_status, sizeof(int32_t) * (size_t)max_batch));
    CK(hipMalloc(&e->d_err, sizeof(int32_t) * 2 * (size_t)max_batch));
    // hipMemset() of device memory does not wait for the fill (it runs on the null stream, which none of this library's
    // non-blocking streams synchronise with): a fill still in flight when the first E-stage has cleared or read the
    // flags would land on top of them.  On the encoder's stream, and waited for.
    CK(hipMemsetAsync(e->d_err, 0, sizeof(int32_t) * 2 * (size_t)max_batch, e->stream));
    CK(hipStreamSynchronize(e->stream));
    CK(hipStreamSynchronize(nullptr));                           // the tables above went over the null stream: through, whatever a plain copy promises
    CK(hipMalloc(&e->d_frame, in.frame_stride));
    CK(hipMalloc(&e->d_pkt, in.packet_cap));
    CK(hipMalloc(&e->d_meta, 16));
    CK(hipMalloc(&e->d_w1, sizeof(int32_t) * g.nblk));
    CK(hipHostMalloc(&e->h_frame, in.frame_stride, hipHostMallocDefault));
    CK(hipHostMalloc(&e->h_pkt, in.packet_cap, hipHostMallocDefault));
    CK(hipHostMalloc(&e->h_meta, 16, hipHostMallocDefault));
#undef CK
    memset(e->h_frame, 0, in.frame_stride);
    *out = e;
    return FFV2AMD_OK;
}

int ffv2amd_encoder_info(const ffv2amd_encoder *e, ffv2amd_info *info)
{
    if (!e || !info) return FFV2AMD_ERR_INVAL;
    *info = e->info;
    return FFV2AMD_OK;
}

int ffv2amd_tstage_device(ffv2amd_encoder *e, int nframes, const void *d_frames,
                          int32_t *d_coef, int64_t *d_energy, void *stream)
{
    if (!e || !d_frames || nframes < 1 || nframes > e->info.max_batch) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    hipStream_t s = (hipStream_t)stream;           // NULL = the HIP default stream
    HIPCHK(hipMemsetAsync(e->d_status, 0, sizeof(int32_t) * nframes, s));
    FFV2TStageArgs a{};
    a.g = e->geom; a.nframes = nframes; a.frames = (const uint8_t *)d_frames;
    a.coef = d_coef; a.energy = d_energy; a.codes = nullptr; a.W = nullptr;
    a.gain_thr = e->d_thr; a.gain_n = GAIN_TABLE_N; a.lds_scan = e->d_lds_scan;
    a.status = e->d_status;
    HIPCHK(ffv2_launch_tstage(a, s));
    return FFV2AMD_OK;
}

int ffv2amd_encode_batch_device(ffv2amd_encoder *e, int nframes, const void *d_frames,
                                int qp, const int32_t *d_W,
                                void *d_packets, size_t packet_stride,
                                uint32_t *d_sizes, int32_t *d_status, void *stream)
{
    if (!e || !d_frames || !d_packets || !d_sizes || nframes < 1 || nframes > e->info.max_batch)
        return FFV2AMD_ERR_INVAL;
    if (packet_stride & 3) return FFV2AMD_ERR_INVAL;
    if (qp < 0) return FFV2AMD_ERR_INVAL;
    if (qp != 0) return FFV2AMD_ERR_UNSUPPORTED;              // packets of qp > 0 are finished on the host:
                                                              // use ffv2amd_encode_batch_to_host
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    hipStream_t s = (hipStream_t)stream;           // NULL = the HIP default stream
    const bool pipe = e->pipelined;
    const int pb = pipe ? (int)(e->seq++ & 1u) : 0;
    // pipelined mode keeps two calls in flight: each has its own hand-off buffers AND its own
    // default status words (the memset / T-stage of call n+1 must not touch what the E-stage of
    // call n still updates)
    int32_t *status = d_status ? d_status : (pb ? e->d_status2 : e->d_status);
    uint32_t *codes = pb ? e->d_codes2 : e->d_codes, *bitcnt = pb ? e->d_bitoff2 : e->d_bitoff;
    hipStream_t se = pipe ? e->e_stream : s;       // stream of the E-stage
    if (pipe && e->evE_valid[pb])                  // the E-stage two calls ago read this hand-off buffer
        HIPCHK(hipStreamWaitEvent(s, e->evE[pb], 0));
    ffv2amd_encoder::EvTriple *ev = nullptr;
    if (e->profiling > 0 && (e->prof_calls++ % (unsigned)e->profiling) == 0) {
        if (e->ev_used == e->ev_pool.size()) {
            if (e->ev_pool.size() >= 8192) {                 // drain before growing without bound
                double t, x, lo, hi; int n;
                ffv2amd_profile_read_ex(e, &t, &x, &n, &lo, &hi);
                e->prof_t = t; e->prof_e = x; e->prof_n = n; e->prof_tmin = lo; e->prof_tmax = hi;
            } else {
                ffv2amd_encoder::EvTriple t{};
                HIPCHK(hipEventCreate(&t.a)); HIPCHK(hipEventCreate(&t.b)); HIPCHK(hipEventCreate(&t.c));
                HIPCHK(hipEventCreate(&t.d));
                try { e->ev_pool.push_back(t); } catch (...) { return FFV2AMD_ERR_NOMEM; }
            }
        }
        ev = &e->ev_pool[e->ev_used++];
    }
    e->seq_pb = pb;
    int r = launch_encode_qp0(e, nframes, d_frames, d_W, d_packets, packet_stride, d_sizes, status, codes, bitcnt,
                              e->coef_sink, s, se, ev, e->d_err + (size_t)pb * e->info.max_batch);
    if (r < 0) return r;
    if (pipe) {
        HIPCHK(hipEventRecord(e->evE[pb], se));
        e->evE_valid[pb] = true;
    }
    return FFV2AMD_OK;
}

int ffv2amd_encoder_set_pipelined(ffv2amd_encoder *e, int on)
{
    if (!e) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    if (on && !e->e_stream) {
        const size_t nb = (size_t)e->info.block_planes, B = (size_t)e->info.max_batch;
        HIPCHK(hipStreamCreateWithFlags(&e->e_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            HIPCHK(hipEventCreateWithFlags(&e->evT[i], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&e->evE[i], hipEventDisableTiming));
        }
        HIPCHK(hipMalloc(&e->d_codes2, sizeof(uint32_t) * FFV2_CODES_PER_BP * nb * B));
        HIPCHK(hipMalloc(&e->d_bitoff2, sizeof(uint32_t) * nb * B));
        HIPCHK(hipMalloc(&e->d_status2, sizeof(int32_t) * B));
    }
    if (!on && e->pipelined) HIPCHK(hipStreamSynchronize(e->e_stream));
    e->pipelined = on != 0;
    return FFV2AMD_OK;
}

int ffv2amd_encoder_flush(ffv2amd_encoder *e, void *stream)
{
    if (!e) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    for (int i = 0; i < 2; i++)
        if (e->evE_valid[i]) HIPCHK(hipStreamWaitEvent((hipStream_t)stream, e->evE[i], 0));
    return FFV2AMD_OK;
}

void ffv2amd_debug_force_tstage(int mode) { ffv2_tstage_force_variant(mode); }

const char *ffv2amd_tstage_kernel_name(ffv2amd_encoder *e, int nframes)
{
    if (!e || nframes < 1) return "";
    DeviceGuard guard(e->device);
    if (!guard.ok) return "";
    return ffv2_tstage_kernel_name(e->geom, nframes, e->coef_sink != nullptr);
}

int ffv2amd_encoder_set_coef_sink(ffv2amd_encoder *e, int32_t *d_coef)
{
    if (!e) return FFV2AMD_ERR_INVAL;
    e->coef_sink = d_coef;
    return FFV2AMD_OK;
}

int ffv2amd_profile_enable(ffv2amd_encoder *e, int on)
{
    if (!e) return FFV2AMD_ERR_INVAL;
    e->profiling = on < 0 ? 0 : on;
    e->prof_calls = 0;
    return FFV2AMD_OK;
}

int ffv2amd_profile_read_ex(ffv2amd_encoder *e, double *tstage_ms, double *estage_ms, int *launches,
                            double *tstage_min_ms, double *tstage_max_ms)
{
    if (!e) return FFV2AMD_ERR_INVAL;
    double t = e->prof_t, x = e->prof_e, lo = e->prof_tmin, hi = e->prof_tmax;
    int n = e->prof_n;
    for (size_t i = 0; i < e->ev_used; i++) {
        float ms = 0;
        HIPCHK(hipEventSynchronize(e->ev_pool[i].c));
        HIPCHK(hipEventElapsedTime(&ms, e->ev_pool[i].a, e->ev_pool[i].b)); t += ms;
        if (n == 0 || ms < lo) lo = ms;
        if (n == 0 || ms > hi) hi = ms;
        HIPCHK(hipEventElapsedTime(&ms, e->ev_pool[i].d_is_b ? e->ev_pool[i].b : e->ev_pool[i].d, e->ev_pool[i].c)); x += ms;
        n++;
    }
    e->ev_used = 0;
    e->prof_t = e->prof_e = e->prof_tmin = e->prof_tmax = 0; e->prof_n = 0;
    if (tstage_ms) *tstage_ms = t;
    if (estage_ms) *estage_ms = x;
    if (launches) *launches = n;
    if (tstage_min_ms) *tstage_min_ms = lo;
    if (tstage_max_ms) *tstage_max_ms = hi;
    return FFV2AMD_OK;
}

int ffv2amd_profile_read(ffv2amd_encoder *e, double *tstage_ms, double *estage_ms, int *launches)
{
    return ffv2amd_profile_read_ex(e, tstage_ms, estage_ms, launches, nullptr, nullptr);
}

int ffv2amd_encode_frame(ffv2amd_encoder *e,
                         const uint8_t *const data[4], const ptrdiff_t linesize[4],
                         int qp, const int32_t *W,
                         uint8_t *out, size_t out_cap, size_t *out_size)
{
    if (!e || !data || !linesize || !out || !out_size) return FFV2AMD_ERR_INVAL;
    const ffv2amd_info &in = e->info;
    const size_t row_bytes = (size_t)in.width * (in.depth > 8 ? 2 : 1);
    for (int p = 0; p < in.planes; p++)
        if (!data[p]) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    hipStream_t s = e->stream;
    // The caller's rows go through the pinned staging frame in slices: while slice n crosses
    // PCIe, slice n+1 is being gathered, so the upload costs max(gather, DMA) rather than their sum.
    const int slice_rows = in.height > 64 ? (in.height + 7) / 8 : in.height;
    auto upload_plane = [&](int p) -> hipError_t {
        for (int y0 = 0; y0 < in.height; y0 += slice_rows) {
            const int y1 = y0 + slice_rows < in.height ? y0 + slice_rows : in.height;
            const size_t off = (size_t)p * in.plane_stride + (size_t)y0 * in.row_pitch;
            if (linesize[p] == (ptrdiff_t)in.row_pitch) {
                memcpy(e->h_frame + off, data[p] + (ptrdiff_t)y0 * linesize[p],
                       (size_t)(y1 - y0) * in.row_pitch - (in.row_pitch - row_bytes));
            } else {
                for (int y = y0; y < y1; y++)
                    memcpy(e->h_frame + (size_t)p * in.plane_stride + (size_t)y * in.row_pitch,
                           data[p] + (ptrdiff_t)y * linesize[p], row_bytes);
            }
            const hipError_t r = hipMemcpyAsync(e->d_frame + off, e->h_frame + off, (size_t)(y1 - y0) * in.row_pitch,
                                                hipMemcpyHostToDevice, s);
            if (r != hipSuccess) return r;
        }
        return hipSuccess;
    };
    // one gathering thread per plane once a plane is worth a thread start (a 4K plane: 0.5 ms of memcpy)
    hipError_t up[4] = { hipSuccess, hipSuccess, hipSuccess, hipSuccess };
    if (in.planes > 1 && in.plane_stride >= (size_t)4 << 20) {
        std::vector<std::thread> helpers;
        for (int p = 1; p < in.planes; p++)
            helpers.emplace_back([&, p]() { up[p] = hipSetDevice(e->device) == hipSuccess ? upload_plane(p) : hipErrorInvalidDevice; });
        up[0] = upload_plane(0);
        for (auto &t : helpers) t.join();
    } else {
        for (int p = 0; p < in.planes; p++) up[p] = upload_plane(p);
    }
    for (int p = 0; p < in.planes; p++) HIPCHK(up[p]);
    return encode_uploaded_frame(e, qp, W, out, out_cap, out_size);
}

// One frame through the wide T-stage (ffv2_wide.hip) and a host-assembled qp == 0 packet: what the
// reference makes of a frame whose samples exceed the declared depth (ffv2.c:26-38 shifts whatever it
// is given) or whose band gains leave the device's threshold table.  Synchronous, on the encoder's
// stream; d_frame must be complete.  Gains with the host's own pow (ffv2enc.c:131-138,174).
static int wide_tstage(ffv2amd_encoder *e, const uint8_t *d_frame, int32_t *d_coef, hipStream_t s)
{
    const FFV2Geom &g = e->geom;
    if (!e->d_wide_plane) {
        HIPCHK(hipMalloc(&e->d_wide_plane, sizeof(int32_t) * (size_t)g.nsx * 64 * g.nsy * 64 * g.planes));
        HIPCHK(hipMalloc(&e->d_wide_c0, sizeof(int32_t) * g.nblk));
        HIPCHK(hipMalloc(&e->d_wide_en, sizeof(int64_t) * 13 * g.nblk));
    }
    HIPCHK(ffv2_launch_wide_tstage(g, d_frame, e->d_wide_plane, d_coef, e->d_wide_en, e->d_wide_c0, s));
    return FFV2AMD_OK;
}

static int wide_encode_frame(ffv2amd_encoder *e, const uint8_t *d_frame, const int32_t *d_W,
                             uint8_t *out, size_t cap, size_t *size)
{
    const ffv2amd_info &in = e->info;
    hipStream_t s = e->stream;
    int r = wide_tstage(e, d_frame, nullptr, s);
    if (r < 0) return r;
    std::vector<int64_t> en;
    std::vector<int32_t> c0, W;
    try {
        en.resize((size_t)13 * in.block_planes); c0.resize((size_t)in.block_planes);
        if (d_W) W.resize((size_t)in.block_planes);
    } catch (...) { return FFV2AMD_ERR_NOMEM; }
    HIPCHK(hipMemcpyAsync(en.data(), e->d_wide_en, sizeof(int64_t) * en.size(), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(c0.data(), e->d_wide_c0, sizeof(int32_t) * c0.size(), hipMemcpyDeviceToHost, s));
    if (d_W) HIPCHK(hipMemcpyAsync(W.data(), d_W, sizeof(int32_t) * W.size(), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    try {
        PacketEnc pe;
        const uint32_t hs = (uint32_t)in.pix_fmt >> 4;
        auto q15 = [](uint32_t k) { return (32768u * k + 6u) / 13u; };
        pe.rc.encode(hs ? q15(hs) : 0, q15(hs + 1), 32768);                 // ffv2enc.c:449
        pe.bits((uint32_t)in.pix_fmt & 15u, 4);
        pe.golomb(0);                                                       // qp
        uint16_t subdiv[4] = { 32, 64, 96, 128 };
        const int nsb = in.num_sb_x * in.num_sb_y;
        for (int sb = 0; sb < nsb; sb++) {
            pe.adapt(subdiv, 4, 128, 0);                                    // ffv2enc.c:222
            pe.bits(0, 4);                                                  // :197
            for (int p = 0; p < in.planes; p++) {
                const size_t bp = (size_t)sb * in.planes + p;
                const int32_t c = c0[bp];
                pe.golomb(c < 0 ? (uint32_t)(-(int64_t)c) : (uint32_t)c);   // :148-150
                if (c) pe.bits(c < 0, 1);
                for (int b = 0; b < 13; b++) {
                    int64_t eb = en[bp * 13 + b];
                    if (b == 12 && d_W) eb = (int64_t)((uint64_t)eb + (uint64_t)((int64_t)W[bp] * W[bp]));   // phantom coefficient
                    pe.golomb(coded_gain_host(eb));                         // :166,174
                }
            }
        }
        if (pe.abort_) return FFV2AMD_ERR_ABORT;
        return pe.finish(out, cap, size);
    } catch (...) { return FFV2AMD_ERR_NOMEM; }
}

// e->d_frame holds (or will hold, in order on e->stream) one 4:4:4 frame: encode it to `out`
static int encode_uploaded_frame(ffv2amd_encoder *e, int qp, const int32_t *W, uint8_t *out, size_t out_cap, size_t *out_size)
{
    const ffv2amd_info &in = e->info;
    hipStream_t s = e->stream;
    const int32_t *dW = nullptr;
    if (W) {
        HIPCHK(hipMemcpyAsync(e->d_w1, W, sizeof(int32_t) * in.block_planes, hipMemcpyHostToDevice, s));
        dW = e->d_w1;
    }
    if (qp > 0) {
        HIPCHK(hipStreamSynchronize(s));
        uint32_t sz = 0;
        int32_t st = 0;
        int r2 = ffv2amd_encode_batch_to_host(e, 1, e->d_frame, qp, dW, out, out_cap, &sz, &st);
        if (r2 < 0) return r2;
        if (st < 0) return st;
        *out_size = sz;
        return FFV2AMD_OK;
    }
    int r = ffv2amd_encode_batch_device(e, 1, e->d_frame, qp, dW, e->d_pkt, in.packet_cap,
                                        e->d_meta, (int32_t *)(e->d_meta + 1), s);
    if (r < 0) return r;
    HIPCHK(ffv2amd_encoder_flush(e, s) < 0 ? hipErrorUnknown : hipSuccess);
    HIPCHK(hipMemcpyAsync(e->h_meta, e->d_meta, 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(e->h_pkt, e->d_pkt, in.packet_cap, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int32_t st = (int32_t)e->h_meta[1];
    if (st == FFV2AMD_ERR_RANGE)       // samples above the declared depth / a gain beyond the table: the reference codes them
        return wide_encode_frame(e, e->d_frame, dW, out, out_cap, out_size);
    if (st < 0) return st;
    const size_t n = e->h_meta[0];
    if (n == 0) return FFV2AMD_ERR_DEVICE;
    if (n > out_cap) return FFV2AMD_ERR_NOSPACE;
    memcpy(out, e->h_pkt, n);
    *out_size = n;
    return FFV2AMD_OK;
}

// ------------------------------------------------------------------
// 4:2:0 front end: what the reference tool chain does before encode2() when handed
// yuv420p / yuv420p10le / yuv420p12le (fftools/ffmpeg_filter.c:63-131 + the auto-inserted
// bicubic scale filter, libswscale/utils.c:332-727) -- see ffv2_upconv.hip.  The encoder must
// have been created for the yuv444p format of the same depth.  PARITY UNPINNED.
// ------------------------------------------------------------------
static int upconv_ready(ffv2amd_encoder *e)
{
    const ffv2amd_info &in = e->info;
    if (in.planes != 3 || (in.pix_fmt != FFV2AMD_PIX_YUV444P && in.pix_fmt != FFV2AMD_PIX_YUV444P10LE &&
                           in.pix_fmt != FFV2AMD_PIX_YUV444P12LE))
        return FFV2AMD_ERR_INVAL;
    if (!e->upconv && !e->upconv_tried) {
        e->upconv_tried = true;
        e->upconv = ffv2_upconv_create(in.width, in.height, in.depth);
    }
    return e->upconv ? FFV2AMD_OK : FFV2AMD_ERR_UNSUPPORTED;
}

size_t ffv2amd_frame_bytes_420(const ffv2amd_encoder *e)
{
    return e ? ffv2_upconv_src_frame_bytes(e->info.width, e->info.height, e->info.depth) : 0;
}

int ffv2amd_upconvert_420_device(ffv2amd_encoder *e, int nframes, const void *d_src420, void *d_frames444, void *stream)
{
    if (!e || !d_src420 || !d_frames444 || nframes < 1) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    int r = upconv_ready(e);
    if (r < 0) return r;
    HIPCHK(ffv2_launch_upconv(e->upconv, e->geom, nframes, (const uint8_t *)d_src420, ffv2amd_frame_bytes_420(e),
                              (uint8_t *)d_frames444, (hipStream_t)stream));
    return FFV2AMD_OK;
}

int ffv2amd_encode_frame_420(ffv2amd_encoder *e, const uint8_t *const data[3], const ptrdiff_t linesize[3],
                             int qp, uint8_t *out, size_t out_cap, size_t *out_size)
{
    if (!e || !data || !linesize || !out || !out_size || !data[0] || !data[1] || !data[2]) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    int r = upconv_ready(e);
    if (r < 0) return r;
    const ffv2amd_info &in = e->info;
    const size_t bps = in.depth > 8 ? 2 : 1, total = ffv2amd_frame_bytes_420(e);
    if (!e->d_420) {
        HIPCHK(hipMalloc(&e->d_420, total));
        HIPCHK(hipHostMalloc(&e->h_420, total, hipHostMallocDefault));
    }
    const int cw = (in.width + 1) >> 1, ch = (in.height + 1) >> 1;
    uint8_t *dst = e->h_420;
    for (int p = 0; p < 3; p++) {                                  // tight rows: Y, U, V
        const int w = p ? cw : in.width, h = p ? ch : in.height;
        for (int y = 0; y < h; y++, dst += (size_t)w * bps)
            memcpy(dst, data[p] + (ptrdiff_t)y * linesize[p], (size_t)w * bps);
    }
    hipStream_t s = e->stream;
    HIPCHK(hipMemcpyAsync(e->d_420, e->h_420, total, hipMemcpyHostToDevice, s));
    HIPCHK(ffv2_launch_upconv(e->upconv, e->geom, 1, e->d_420, total, e->d_frame, s));
    return encode_uploaded_frame(e, qp, nullptr, out, out_cap, out_size);
}

int ffv2amd_tstage_wide_device(ffv2amd_encoder *e, const void *d_frame, int32_t *d_coef, int64_t *d_energy, void *stream)
{
    if (!e || !d_frame) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    const int r = wide_tstage(e, (const uint8_t *)d_frame, d_coef, (hipStream_t)stream);
    if (r < 0) return r;
    if (d_energy)
        HIPCHK(hipMemcpyAsync(d_energy, e->d_wide_en, sizeof(int64_t) * 13 * e->geom.nblk, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return FFV2AMD_OK;
}

int ffv2amd_inverse_tstage_device(ffv2amd_encoder *e, int nframes, const int32_t *d_coef,
                                  void *d_frames_out, void *stream)
{
    if (!e || !d_coef || !d_frames_out || nframes < 1 || nframes > e->info.max_batch) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    const FFV2Geom &g = e->geom;
    if (!e->d_inv_plane)
        HIPCHK(hipMalloc(&e->d_inv_plane, sizeof(int32_t) * (size_t)g.nsx * 64 * g.nsy * 64 * g.planes * e->info.max_batch));
    HIPCHK(ffv2_launch_inverse(g, nframes, d_coef, e->d_inv_plane, (uint8_t *)d_frames_out, e->d_lds_scan,
                               (hipStream_t)stream));
    return FFV2AMD_OK;
}

// ------------------------------------------------------------------
// Decoder-side check of a finished packet, the shape of FATE's enc_dec (tests/fate-run.sh:188-210):
// ffv2_decode_frame (ffv2dec.c:315-377).  The entropy layer is one serial chain per packet and is
// parsed on the host (RangeDec above; symbol order of dequant_block, ffv2dec.c:100-136, with its one
// pulses[] array per block-plane whose unread slots keep earlier bands' values); the scaling of the
// pulses, the inverse T-stage (ffv2_inverse.hip) and coeffs_2_ref run on the device.  The band
// magnitudes (pow, sqrt: ffv2dec.c:91-98,134) are the host libm's.  Not a product decoder: a
// self check for the encoder and the PSNR line of the report.  PARITY UNPINNED.
// ------------------------------------------------------------------
// The host half of ffv2amd_decode_frame, callable without a GPU: the packet's entropy layer in the symbol order of
// dequant_block.  pulses[bp][q]: what the decoder's pulses[] holds for coding position q when its band is scaled
// (slots a band does not read keep earlier bands' values); mag[bp][b]: the band's scale (float)pow(gain, 1.5f) /
// sqrt(sum of the squares of the pulses read: inf or NaN at qp 0); c0[bp]: the "DC" slot.
int ffv2amd_parse_packet(const uint8_t *pkt, size_t size, int width, int height, int *pix_fmt_out, int *qp_out,
                         int16_t *pulses, float *mag, int32_t *c0)
{
    if (!pkt || !pulses || !mag || !c0 || width < 1 || height < 1) return FFV2AMD_ERR_INVAL;
    std::vector<uint16_t> test;
    int slot[4097];
    try {
        RangeDec d(pkt, size);
        const int pix_fmt = (int)d.uint_(196);                           // ffv2dec.c:276
        const int qp = (int)d.golomb();                                  // :277
        int planes, depth;
        if (d.err || pixfmt_info(pix_fmt, &planes, &depth) < 0 || qp < 0 || qp > 16384) return FFV2AMD_ERR_INVAL;
        if (pix_fmt_out) *pix_fmt_out = pix_fmt;
        if (qp_out) *qp_out = qp;
        test.resize((size_t)13 * (qp > 0 ? qp : 1));
        for (int r = 0; r < 13; r++)
            for (int j = 0; j < qp; j++) test[(size_t)r * qp + j] = (uint16_t)(j + 1);
        uint16_t subdiv[4] = { 32, 64, 96, 128 };
        const int nsb = ((width + 63) / 64) * ((height + 63) / 64);
        for (int sb = 0; sb < nsb; sb++) {
            if (d.adapt(subdiv, 4, 128) != 0 || d.err) return FFV2AMD_ERR_INVAL;   // the encoder never splits
            (void)d.bits(4);                                             // tx type
            for (int p = 0; p < planes; p++) {
                const size_t bp = (size_t)sb * planes + p;
                memset(slot, 0, sizeof(slot));                           // int pulses[4096] = { 0 }
                memset(pulses + bp * 4096, 0, sizeof(int16_t) * 4096);
                int32_t v = (int32_t)d.golomb();
                if (v) v = (int32_t)((uint32_t)v * (uint32_t)(1 - 2 * (int)d.bits(1)));
                c0[bp] = v;
                for (int b = 0; b < 13; b++) {
                    const int lo = 1 + BANDS_START[b], len = BANDS_START[b + 1] - BANDS_START[b];
                    const float cg = (float)d.golomb();
                    const float m = (float)pow((double)(cg * 1), (double)1.5f);   // gain_expand(cg, 1, 1.5f)
                    int cnt = 0, pcnt = 0;
                    for (int j = 0; j < len; j++) {
                        if (pcnt >= qp) break;
                        int q = d.adapt(&test[(size_t)b * qp], qp, 64);
                        if (q) q *= 1 - 2 * (int)d.bits(1);
                        slot[j] = q;
                        pcnt += q < 0 ? -q : q;
                        cnt += q * q;
                    }
                    if (d.err) return FFV2AMD_ERR_INVAL;
                    mag[bp * 13 + b] = (float)((double)m / sqrt((double)cnt));    // mag /= sqrt(cnt)
                    for (int j = 0; j < len && lo + j < 4096; j++)
                        pulses[bp * 4096 + lo + j] = (int16_t)slot[j];
                }
            }
        }
    } catch (...) { return FFV2AMD_ERR_NOMEM; }
    return FFV2AMD_OK;
}

int ffv2amd_decode_frame(ffv2amd_encoder *e, const uint8_t *pkt, size_t size, uint8_t *const data[4],
                         const ptrdiff_t linesize[4], unsigned flags, int *qp_out)
{
    if (!e || !pkt || !data || !linesize) return FFV2AMD_ERR_INVAL;
    const ffv2amd_info &in = e->info;
    for (int p = 0; p < in.planes; p++)
        if (!data[p]) return FFV2AMD_ERR_INVAL;
    const size_t nb = (size_t)in.block_planes;
    std::vector<int16_t> pulses;
    std::vector<float> mag;
    std::vector<int32_t> c0;
    int qp = 0, pix_fmt = -1;
    try { pulses.resize(nb * 4096); mag.assign(nb * 13, 0.f); c0.assign(nb, 0); } catch (...) { return FFV2AMD_ERR_NOMEM; }
    {
        // the header decides how many planes the packet carries: it must be this encoder's format before the
        // symbols are parsed into arrays sized for it
        RangeDec h(pkt, size);
        if ((int)h.uint_(196) != in.pix_fmt || h.err) return FFV2AMD_ERR_INVAL;
    }
    const int pr = ffv2amd_parse_packet(pkt, size, in.width, in.height, &pix_fmt, &qp, pulses.data(), mag.data(), c0.data());
    if (pr < 0) return pr;
    if (qp_out) *qp_out = qp;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    const FFV2Geom &g = e->geom;
    if (!e->d_dec_pulses) {
        HIPCHK(hipMalloc(&e->d_dec_pulses, sizeof(int16_t) * 4096 * nb));
        HIPCHK(hipMalloc(&e->d_dec_mag, sizeof(float) * 13 * nb));
        HIPCHK(hipMalloc(&e->d_dec_c0, sizeof(int32_t) * nb));
        HIPCHK(hipMalloc(&e->d_dec_coef, sizeof(int32_t) * 4096 * nb));
        HIPCHK(hipMalloc(&e->d_dec_frame, in.frame_stride));
    }
    if (!e->d_inv_plane)
        HIPCHK(hipMalloc(&e->d_inv_plane, sizeof(int32_t) * (size_t)g.nsx * 64 * g.nsy * 64 * g.planes * e->info.max_batch));
    hipStream_t s = e->stream;
    HIPCHK(hipMemcpyAsync(e->d_dec_pulses, pulses.data(), sizeof(int16_t) * pulses.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->d_dec_mag, mag.data(), sizeof(float) * mag.size(), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->d_dec_c0, c0.data(), sizeof(int32_t) * c0.size(), hipMemcpyHostToDevice, s));
    HIPCHK(ffv2_launch_dequant(e->d_dec_pulses, e->d_dec_mag, e->d_dec_c0, e->d_dec_coef, (long long)nb, s));
    HIPCHK(ffv2_launch_inverse(g, 1, e->d_dec_coef, e->d_inv_plane, e->d_dec_frame, e->d_lds_scan, s));
    const size_t bps = in.depth > 8 ? 2 : 1;
    for (int p = 0; p < in.planes; p++)
        HIPCHK(hipMemcpy2DAsync(data[p], (size_t)linesize[p], e->d_dec_frame + (size_t)p * in.plane_stride, in.row_pitch,
                                (size_t)in.width * bps, (size_t)in.height, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (flags & FFV2AMD_DECODE_GRID) {
        // the reference decoder's `#define DEBUGGING` (ffv2dec.c:88,258-273): first row and column of every superblock
        for (int p = 0; p < in.planes; p++) {
            const int v = ((p ? 0 : -2048) + 2048) >> (12 - in.depth);
            for (int y = 0; y < in.height; y++) {
                uint8_t *row = data[p] + (ptrdiff_t)y * linesize[p];
                for (int x = 0; x < in.width; x++) {
                    if ((x & 63) && (y & 63)) continue;
                    if (bps == 1) row[x] = (uint8_t)v;
                    else { const uint16_t w = (uint16_t)v; memcpy(row + 2 * x, &w, 2); }
                }
            }
        }
    }
    return FFV2AMD_OK;
}

int ffv2amd_pvq_search_device(ffv2amd_encoder *e, const float *d_X, int stride, int N, int K,
                              int count, int16_t *d_y, void *stream)
{
    if (!e || !d_X || !d_y || count < 1 || K < 0) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    HIPCHK(ffv2_launch_pvq_vectors(d_X, stride, N, K, count, d_y, (hipStream_t)stream));
    return FFV2AMD_OK;
}

// ------------------------------------------------------------------
// qp > 0 as a two-stage pipeline.  submit: T-stage (coefficients kept), PVQ search, symbol
// compaction and the small D2H copies, all asynchronous on the encoder's stream.  finish: the
// oldest submitted batch -- waits for it, pulls each frame's compact symbol stream (its own
// size, int8) and runs the adaptive range coder (daala_entropy.c:328-379,428-440; one serial
// chain per frame, ffv2enc.c:461) on host threads, one frame per thread, each starting as soon
// as its stream has arrived.  Two batches may be in flight: submit(n+1) before finish(n)
// overlaps the host coder with the GPU.  1 <= qp <= 64.
// ------------------------------------------------------------------
// ------------------------------------------------------------------
// qp > 0 with many frames in flight (ffv2_lanecoder.hip).  The range coder is one dependent chain
// per frame (ffv2enc.c:461,466), so throughput on the device comes from coding many frames side by
// side, one per lane: open(F) sizes the coder's HBM scratch for F frames (about 40 bytes per
// coefficient: 84 MB per 1080p frame).  submit() takes up to F device-resident frames through
// T-stage, PVQ search and the symbol bookkeeping max_batch frames at a time on the encoder's
// stream (the "front"), then queues the CDF, chain and packet kernels over all of them on a second
// stream (the "back").  The back's scratch (records, code words) exists once, the front's twice:
// submit(n+1) before finish(n) runs the front of call n+1 beside the chain of call n, which keeps
// only 1/16 of the chip's SIMDs busy.  With a third set (calls_in_flight = 3) the fronts follow each other
// without waiting for a finish: the period of back-to-back calls falls from (back + front + copy) / 2 to
// max(back, front).
// ------------------------------------------------------------------
// Symbols of every frame's coding order that cdf and chain work on at a time (ffv2_lanecoder.hip,
// "windows"): the records exist for two windows only.  Default 2^18 symbols = 4 MB of records per
// frame; FFV2AMD_LC_WINDOW or ffv2amd_debug_lanecoder_window() override it (tests use tiny windows).
static uint32_t g_lc_window = 0;
static uint32_t lanecoder_window(size_t maxsym16)
{
    static const uint32_t env = getenv("FFV2AMD_LC_WINDOW") ? (uint32_t)strtoul(getenv("FFV2AMD_LC_WINDOW"), nullptr, 10) : 0u;
    uint64_t w = g_lc_window ? g_lc_window : env ? env : (1u << 18);
    w = (w + 15) / 16 * 16;
    if (w < 16) w = 16;
    const uint64_t least = ((uint64_t)maxsym16 / 4096 + 15) / 16 * 16;     // never more than ~4 096 windows (two launches each)
    if (w < least) w = least;
    if (w > maxsym16) w = maxsym16;
    if (w > (1u << 22)) w = 1u << 22;                 // lc_cdf_kernel addresses a window's records of 64 frames with 32 bits (8 bytes each)
    return (uint32_t)w;
}

static int lanecoder_alloc(ffv2amd_encoder *e, int frames, size_t pcap, int nsets, int nback)
{
    auto &lc = e->lc;
    const ffv2amd_info &in = e->info;
    const size_t nb = (size_t)in.block_planes, F = (size_t)frames, nsb = (size_t)in.num_sb_x * in.num_sb_y;
    auto dev = [&](auto **p, size_t bytes) {
        if (hipMalloc((void **)p, bytes) != hipSuccess) { *p = nullptr; return false; }
        try { lc.allocs.push_back((void *)*p); } catch (...) { (void)hipFree((void *)*p); *p = nullptr; return false; }
        return true;
    };
    FFV2LaneCoderArgs &a = lc.a;
    lc.nsets = nsets; lc.nback = nback;
    a.nblk = (int)nb; a.planes = in.planes;
    // Frames (lanes) per chain workgroup: as few as give about 256 workgroups (at most 333), 4 to 64.  A chain step is
    // shorter with few active lanes (20.2 -> 16.8 ms per window of 1 024 1080p frames from 64 lanes to 4), the chip has
    // SIMDs to spare for a few hundred two-wavefront workgroups, and more than that get in the way of the next call's
    // front.  Measured at 1080p / qp 16 against 64 lanes: 1 024 frames +17 %, 2 048 +38 %, 3 328 +25 %, 4 096 +22 %,
    // 6 656 +7 to +12 %.  FFV2AMD_LC_WIDTH overrides.
    static const int wenv = getenv("FFV2AMD_LC_WIDTH") ? atoi(getenv("FFV2AMD_LC_WIDTH")) : 0;
    int wmax = 4;
    if (wenv >= 1) wmax = wenv;
    else while (wmax < 64 && (frames + wmax - 1) / wmax > 333) wmax <<= 1;
    int width = 1;
    while (width < frames && width < 64 && width < wmax) width <<= 1;
    a.width = width;
    const size_t groups = (F + width - 1) / width;
    const size_t maxsym = ((1 + nsb + nb * 4097) + 15) / 16 * 16;
    if (maxsym >= ((size_t)1 << 32)) return FFV2AMD_ERR_INVAL;
    lc.maxsym16 = (uint32_t)maxsym;
    lc.window = lanecoder_window(maxsym);
    a.group_stride = (size_t)lc.window * (size_t)width;            // uint2 per group and window
    a.buf_stride = a.group_stride * groups;
    a.row_stride = (nb * 4097 + 255) / 256 * 256;
    a.raw_words = (uint32_t)(pcap / 4 + 4);
    a.wcap = (uint32_t)(pcap / 2 + 32);
    a.packet_stride = pcap;
    {   // frames per launch of the front: up to 64, within about 2.5 GB of coefficient / pulse workspace
        const size_t per = nb * (sizeof(int32_t) * 4096 + sizeof(int16_t) * FFV2_Y_STRIDE + sizeof(uint32_t));
        size_t g = (((size_t)5 << 29)) / per;
        if (g > 64) g = 64;
        if (g < 1) g = 1;
        if (g > F) g = F;
        lc.group = (int)g;
    }
    bool ok = dev(&lc.d_coef, sizeof(int32_t) * 4096 * nb * (size_t)lc.group)
           && dev(&lc.d_y, sizeof(int16_t) * FFV2_Y_STRIDE * nb * (size_t)lc.group)
           && dev(&lc.d_bitcnt, sizeof(uint32_t) * nb * (size_t)lc.group) && dev(&lc.d_split, sizeof(uint2) * nsb);
    for (int k = 0; k < nback; k++) {
        auto &b = lc.bk[k];
        ok = ok && dev(&b.recs, sizeof(uint2) * a.buf_stride * 2) && dev(&b.words, sizeof(uint32_t) * a.wcap * F)
           && dev(&b.cdfstate, sizeof(uint32_t) * 68 * 13 * F)
           && dev(&b.state, sizeof(FFV2LaneState) * F) && dev(&b.fin, sizeof(uint4) * F);
    }
    for (int k = 0; k < nsets; k++) {
        auto &q = lc.set[k];
        ok = ok && dev(&q.d_codes, sizeof(uint32_t) * FFV2_CODES_PER_BP * nb * F)
           && dev(&q.d_status_in, sizeof(int32_t) * F) && dev(&q.abort_, sizeof(int32_t) * F)
           && dev(&q.cnt, sizeof(FFV2SymRec) * nb * F) && dev(&q.bits, sizeof(uint32_t) * nb * F)
           && dev(&q.rowbase, sizeof(uint32_t) * 13 * (nb + 1) * F) && dev(&q.gbase, sizeof(uint32_t) * (nb + 1) * F)
           && dev(&q.rawbase, sizeof(uint32_t) * (nb + 1) * F) && dev(&q.delta, sizeof(uint32_t) * 13 * nb * F)
           && dev(&q.rows, a.row_stride * F) && dev(&q.raw, sizeof(uint32_t) * a.raw_words * F)
           && dev(&q.packets, a.packet_stride * F) && dev(&q.sizes, sizeof(uint32_t) * F) && dev(&q.status, sizeof(int32_t) * F)
           && dev(&q.offs, sizeof(unsigned long long) * (F + 1));
    }
    if (!ok) return FFV2AMD_ERR_NOMEM;
    a.split = lc.d_split;
    for (int k = 0; k < nsets; k++) {
        auto &q = lc.set[k];
        HIPCHK(hipHostMalloc(&q.h_sizes, sizeof(uint32_t) * F, hipHostMallocDefault));
        HIPCHK(hipHostMalloc(&q.h_status, sizeof(int32_t) * F, hipHostMallocDefault));
        HIPCHK(hipHostMalloc(&q.h_offs, sizeof(unsigned long long) * (F + 2), hipHostMallocDefault));   // [F + 1]: symbols of frame 0
        HIPCHK(hipEventCreateWithFlags(&q.ev_front, hipEventDisableTiming));
        HIPCHK(hipEventCreate(&q.ev_done));
        HIPCHK(hipEventCreate(&q.ev_back0));
        HIPCHK(hipEventCreate(&q.ev_chain0));
        HIPCHK(hipEventCreate(&q.ev_chain1));
    }
    HIPCHK(hipStreamCreateWithFlags(&lc.copy, hipStreamNonBlocking));
    for (int j = 0; j < nback; j++) {
        auto &b = lc.bk[j];
        HIPCHK(hipStreamCreateWithFlags(&b.back, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&b.cdfs, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            HIPCHK(hipEventCreateWithFlags(&b.ev_cdf[k], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&b.ev_chain[k], hipEventDisableTiming));
        }
        HIPCHK(hipEventCreateWithFlags(&b.ev_backdone, hipEventDisableTiming));
    }
    // the symbols that carry no data: "no split" of every superblock (ffv2enc.c:222; the CDF of
    // daala_entropy.h:140-161 advances by itself), the range-coded part of the header (ffv2enc.c:449)
    std::vector<uint2> split;
    try { split.resize(nsb); } catch (...) { return FFV2AMD_ERR_NOMEM; }
    uint32_t cdf[4] = { 32, 64, 96, 128 };
    for (size_t i = 0; i < nsb; i++) {
        const uint32_t ft = cdf[3];
        const int sc = 15 - RangeEnc::ilog(ft - 1);
        split[i] = make_uint2((cdf[0] << sc) << 16, ft << sc);
        if (cdf[3] + 128 > 32767)
            for (int k = 0; k < 4; k++) cdf[k] = (cdf[k] >> 1) + k + 1;
        for (int k = 0; k < 4; k++) cdf[k] += 128;
    }
    HIPCHK(hipMemcpy(lc.d_split, split.data(), sizeof(uint2) * nsb, hipMemcpyHostToDevice));
    HIPCHK(hipStreamSynchronize(nullptr));                       // null stream: the coder's streams do not wait for it
    const uint32_t hs = (uint32_t)in.pix_fmt >> 4;
    auto q15 = [](uint32_t k) { return (32768u * k + 6u) / 13u; };
    a.header = make_uint2((hs ? q15(hs) : 0u) | (q15(hs + 1) << 16), 32768u);
    lc.cap = frames;
    return FFV2AMD_OK;
}

static void lanecoder_free(ffv2amd_encoder *e)
{
    auto &lc = e->lc;
    for (auto &b : lc.bk) {
        if (b.back) { (void)hipStreamSynchronize(b.back); (void)hipStreamDestroy(b.back); }
        if (b.cdfs) { (void)hipStreamSynchronize(b.cdfs); (void)hipStreamDestroy(b.cdfs); }
        for (int k = 0; k < 2; k++) {
            if (b.ev_cdf[k]) (void)hipEventDestroy(b.ev_cdf[k]);
            if (b.ev_chain[k]) (void)hipEventDestroy(b.ev_chain[k]);
        }
        if (b.ev_backdone) (void)hipEventDestroy(b.ev_backdone);
    }
    if (lc.copy) { (void)hipStreamSynchronize(lc.copy); (void)hipStreamDestroy(lc.copy); }
    for (void *p : lc.allocs) (void)hipFree(p);
    lc.allocs.clear();
    for (auto &q : lc.set) {
        if (q.h_sizes) (void)hipHostFree(q.h_sizes);
        if (q.h_status) (void)hipHostFree(q.h_status);
        if (q.h_offs) (void)hipHostFree(q.h_offs);
        if (q.ev_front) (void)hipEventDestroy(q.ev_front);
        if (q.ev_done) (void)hipEventDestroy(q.ev_done);
        if (q.ev_back0) (void)hipEventDestroy(q.ev_back0);
        if (q.ev_chain0) (void)hipEventDestroy(q.ev_chain0);
        if (q.ev_chain1) (void)hipEventDestroy(q.ev_chain1);
    }
    lc = ffv2amd_encoder::LaneCoder{};
}

static int lanecoder_default_backs(int calls_in_flight)
{
    const int env = getenv("FFV2AMD_LC_BACKS") ? atoi(getenv("FFV2AMD_LC_BACKS")) : 0;
    const int b = env >= 1 ? env : 1;
    return b > calls_in_flight ? calls_in_flight : b;
}

int ffv2amd_lanecoder_open(ffv2amd_encoder *e, int frames_in_flight, size_t packet_cap, int calls_in_flight)
{
    if (calls_in_flight == 0) calls_in_flight = 2;
    return ffv2amd_lanecoder_open_ex(e, frames_in_flight, packet_cap, calls_in_flight, lanecoder_default_backs(calls_in_flight));
}

int ffv2amd_lanecoder_open_ex(ffv2amd_encoder *e, int frames_in_flight, size_t packet_cap, int calls_in_flight, int backs)
{
    if (!e || frames_in_flight < 1 || frames_in_flight > (1 << 20)) return FFV2AMD_ERR_INVAL;
    if (calls_in_flight == 0) calls_in_flight = 2;
    if (calls_in_flight < 2 || calls_in_flight > 4) return FFV2AMD_ERR_INVAL;
    if (backs == 0) backs = 1;
    if (backs < 1 || backs > calls_in_flight) return FFV2AMD_ERR_INVAL;
    if (packet_cap == 0 || packet_cap > e->info.packet_cap_qp) packet_cap = e->info.packet_cap_qp;
    if (packet_cap < 64) return FFV2AMD_ERR_INVAL;
    packet_cap = (packet_cap + 15) / 16 * 16;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    if (e->lc.cap) {
        HIPCHK(hipStreamSynchronize(e->stream));
        lanecoder_free(e);
    }
    const int r = lanecoder_alloc(e, frames_in_flight, packet_cap, calls_in_flight, backs);
    if (r < 0) {
        lanecoder_free(e);
        (void)hipGetLastError();                                  // a refused hipMalloc must not surface in the next launch check
    }
    return r;
}

void ffv2amd_debug_lanecoder_window(uint32_t symbols) { g_lc_window = symbols; }

// Benchmark aid: the Q-stage kernel alone.  T-stage of `nframes` (<= max_batch) device-resident frames once,
// then `reps` launches of ffv2_pvq_kernel over their coefficients between two events -> ms per launch.
int ffv2amd_debug_pvq_time(ffv2amd_encoder *e, int nframes, const void *d_frames, int qp, int reps, float *ms_per_launch)
{
    if (!e || !d_frames || !ms_per_launch || nframes < 1 || nframes > e->info.max_batch || qp < 1 || reps < 1) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    const size_t nb = (size_t)e->info.block_planes, B = (size_t)e->info.max_batch;
    if (!e->d_coef_ws) {
        HIPCHK(hipMalloc(&e->d_coef_ws, sizeof(int32_t) * 4096 * nb * B));
        HIPCHK(hipMalloc(&e->d_y, sizeof(int16_t) * FFV2_Y_STRIDE * nb * B));
    }
    hipStream_t s = e->stream;
    FFV2TStageArgs t{};
    t.g = e->geom; t.nframes = nframes; t.frames = (const uint8_t *)d_frames;
    t.coef = e->d_coef_ws; t.codes = e->d_codes; t.bitcnt = e->d_bitoff;
    t.gain_thr = e->d_thr; t.gain_n = GAIN_TABLE_N; t.lds_scan = e->d_lds_scan; t.status = e->d_status;
    HIPCHK(hipMemsetAsync(e->d_status, 0, sizeof(int32_t) * nframes, s));
    HIPCHK(ffv2_launch_tstage(t, s));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    HIPCHK(ffv2_launch_pvq(e->d_coef_ws, nullptr, e->d_y, qp, (long long)nb * nframes, s));      // warm
    HIPCHK(hipEventRecord(a, s));
    for (int r = 0; r < reps; r++) HIPCHK(ffv2_launch_pvq(e->d_coef_ws, nullptr, e->d_y, qp, (long long)nb * nframes, s));
    HIPCHK(hipEventRecord(b, s));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *ms_per_launch = ms / reps;
    return FFV2AMD_OK;
}

int ffv2amd_lanecoder_close(ffv2amd_encoder *e)
{
    if (!e) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    if (e->lc.cap) (void)hipStreamSynchronize(e->stream);
    lanecoder_free(e);
    return FFV2AMD_OK;
}

size_t ffv2amd_lanecoder_bytes_per_frame(const ffv2amd_encoder *e, size_t packet_cap, int calls_in_flight)
{
    if (calls_in_flight < 2) calls_in_flight = 2;
    return ffv2amd_lanecoder_bytes_per_frame_ex(e, packet_cap, calls_in_flight, lanecoder_default_backs(calls_in_flight));
}

size_t ffv2amd_lanecoder_bytes_per_frame_ex(const ffv2amd_encoder *e, size_t packet_cap, int calls_in_flight, int backs)
{
    if (!e) return 0;
    if (calls_in_flight < 2) calls_in_flight = 2;
    if (calls_in_flight > 4) calls_in_flight = 4;
    if (backs < 1) backs = 1;
    if (backs > calls_in_flight) backs = calls_in_flight;
    const ffv2amd_info &in = e->info;
    if (packet_cap == 0 || packet_cap > in.packet_cap_qp) packet_cap = in.packet_cap_qp;
    const size_t nb = (size_t)in.block_planes, nsb = (size_t)in.num_sb_x * in.num_sb_y;
    const size_t maxsym = ((1 + nsb + nb * 4097) + 15) / 16 * 16;
    // records of two windows, code words, the rows' and the frame's state between windows
    const size_t shared = (size_t)lanecoder_window(maxsym) * 2 * sizeof(uint2) + packet_cap * 2 + 13 * 68 * 4 + 256;
    const size_t per_set = (nb * 4097 + 255) / 256 * 256 + packet_cap * 2
                         + nb * (sizeof(uint32_t) * FFV2_CODES_PER_BP + sizeof(FFV2SymRec) + sizeof(uint32_t) * 30) + 256;
    return (size_t)backs * shared + (size_t)calls_in_flight * per_set;
}

// timing of the call that has just completed (events recorded on the back stream)
static void lanecoder_note_times(ffv2amd_encoder::LaneCoder &lc, ffv2amd_encoder::LaneCoder::Set &q)
{
    float c = 0, b = 0;
    if (hipEventElapsedTime(&c, q.ev_chain0, q.ev_chain1) != hipSuccess) c = 0;
    if (hipEventElapsedTime(&b, q.ev_back0, q.ev_done) != hipSuccess) b = 0;
    lc.last_chain_ms = c; lc.last_back_ms = b;
    lc.last_symbols0 = *reinterpret_cast<const uint32_t *>(q.h_offs + (size_t)lc.cap + 1);
}

int ffv2amd_lanecoder_stats(const ffv2amd_encoder *e, float *chain_ms, float *back_ms, uint32_t *symbols_frame0)
{
    if (!e || !e->lc.cap) return FFV2AMD_ERR_INVAL;
    if (chain_ms) *chain_ms = e->lc.last_chain_ms;
    if (back_ms) *back_ms = e->lc.last_back_ms;
    if (symbols_frame0) *symbols_frame0 = e->lc.last_symbols0;
    return FFV2AMD_OK;
}

int ffv2amd_lanecoder_submit(ffv2amd_encoder *e, int nframes, const void *d_frames, int qp, const int32_t *d_W)
{
    if (!e || !d_frames || nframes < 1) return FFV2AMD_ERR_INVAL;
    if (qp < 1 || qp > 64) return FFV2AMD_ERR_UNSUPPORTED;
    auto &lc = e->lc;
    if (nframes > lc.cap) return FFV2AMD_ERR_INVAL;
    auto &q = lc.set[lc.sub % (unsigned)lc.nsets];
    if (q.busy) return FFV2AMD_ERR_AGAIN;                        // every set is in flight
    auto &bk = lc.bk[lc.sub % (unsigned)lc.nback];
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    const ffv2amd_info &in = e->info;
    const size_t nb = (size_t)in.block_planes, B = (size_t)lc.group;
    hipStream_t s = e->stream;
    FFV2LaneCoderArgs a = lc.a;
    a.qp = qp;
    a.codes = q.d_codes; a.status_in = q.d_status_in; a.abort_ = q.abort_; a.cnt = q.cnt; a.bits = q.bits;
    a.rowbase = q.rowbase; a.gbase = q.gbase; a.rawbase = q.rawbase; a.delta = q.delta; a.rows = q.rows; a.raw = q.raw;
    a.packets = q.packets; a.sizes = q.sizes; a.status = q.status; a.offs = q.offs;
    a.recs = bk.recs; a.words = bk.words; a.cdfstate = bk.cdfstate; a.state = bk.state; a.fin = bk.fin;
    {   // raw header: pix_fmt & 15, then Exp-Golomb(qp) (ffv2enc.c:449-450)
        const uint32_t v = (uint32_t)qp + 1u;
        const int nbits = 31 - __builtin_clz(v);
        uint32_t code = 1u << (2 * nbits);
        for (int i = 0; i < nbits; i++) code |= ((v >> i) & 1u) << (2 * (nbits - 1 - i) + 1);
        a.header_bits = ((uint32_t)in.pix_fmt & 15u) | (code << 4);
        a.header_nbits = 4u + 2u * (uint32_t)nbits + 1u;
    }
    // front: the encoder's stream
    HIPCHK(hipMemsetAsync(q.d_status_in, 0, sizeof(int32_t) * nframes, s));
    HIPCHK(hipMemsetAsync(q.abort_, 0, sizeof(int32_t) * nframes, s));
    HIPCHK(hipMemsetAsync(q.raw, 0, sizeof(uint32_t) * a.raw_words * (size_t)nframes, s));
    for (int f0 = 0; f0 < nframes; f0 += (int)B) {
        const int n = nframes - f0 < (int)B ? nframes - f0 : (int)B;
        FFV2TStageArgs t{};
        t.g = e->geom; t.nframes = n; t.frames = (const uint8_t *)d_frames + (size_t)f0 * in.frame_stride;
        t.coef = lc.d_coef; t.energy = nullptr; t.codes = q.d_codes + (size_t)f0 * nb * FFV2_CODES_PER_BP;
        t.bitcnt = lc.d_bitcnt; t.W = d_W ? d_W + (size_t)f0 * nb : nullptr;
        t.gain_thr = e->d_thr; t.gain_n = GAIN_TABLE_N; t.lds_scan = e->d_lds_scan; t.status = q.d_status_in + f0;
        HIPCHK(ffv2_launch_tstage(t, s));
        // the search also notes what the coder will read of every band (FFV2AMD_LC_COUNT_KERNEL=1: a pass of its own
        // over the pulses finds out, as before round 3; same numbers)
        const bool count_pass = getenv("FFV2AMD_LC_COUNT_KERNEL") && atoi(getenv("FFV2AMD_LC_COUNT_KERNEL")) != 0;   // read per call: tests flip it
        if (count_pass)
            HIPCHK(ffv2_launch_pvq(lc.d_coef, t.W, lc.d_y, qp, (long long)nb * n, s));
        else
            HIPCHK(ffv2_launch_pvq_counted(lc.d_coef, t.W, lc.d_y, qp, (long long)nb * n, (int)nb, t.codes,
                                           a.cnt + (size_t)f0 * nb, a.bits + (size_t)f0 * nb, a.abort_ + f0, s));
        a.f0 = f0;
        HIPCHK(ffv2_launch_lc_front(a, lc.d_y, n, !count_pass, s));
    }
    HIPCHK(hipEventRecord(q.ev_front, s));
    // back: cdf and chain window by window, then the packets.  One call runs on a back at a time (its
    // scratch exists once per back).  cdf of window i+1 (its own stream) beside the chain of window i; a record
    // buffer is refilled once the chain of two windows ago has read it.
    // (FFV2AMD_LC_SERIAL=1: everything on the back stream, cdf and chain of a window one after the other)
    static const bool serial = getenv("FFV2AMD_LC_SERIAL") && atoi(getenv("FFV2AMD_LC_SERIAL")) != 0;
    hipStream_t sc = serial ? bk.back : bk.cdfs;
    HIPCHK(hipStreamWaitEvent(bk.back, q.ev_front, 0));
    if (!serial) {
        HIPCHK(hipStreamWaitEvent(sc, q.ev_front, 0));
        if (bk.backdone_valid) HIPCHK(hipStreamWaitEvent(sc, bk.ev_backdone, 0));
    }
    HIPCHK(hipEventRecord(q.ev_back0, bk.back));
    {
        int i = 0;
        for (uint32_t w0 = 0; w0 < lc.maxsym16; w0 += lc.window, i++) {
            const uint32_t w1 = lc.maxsym16 - w0 < lc.window ? lc.maxsym16 : w0 + lc.window;
            const int buf = serial ? 0 : i & 1;
            if (!serial && i >= 2) HIPCHK(hipStreamWaitEvent(sc, bk.ev_chain[buf], 0));
            HIPCHK(ffv2_launch_lc_cdf(a, nframes, w0, w1, buf, sc));
            if (!serial) {
                HIPCHK(hipEventRecord(bk.ev_cdf[buf], sc));
                HIPCHK(hipStreamWaitEvent(bk.back, bk.ev_cdf[buf], 0));
            }
            if (i == 0) HIPCHK(hipEventRecord(q.ev_chain0, bk.back));
            HIPCHK(ffv2_launch_lc_chain(a, nframes, w0, w1, buf, bk.back));
            if (!serial) HIPCHK(hipEventRecord(bk.ev_chain[buf], bk.back));
        }
    }
    HIPCHK(hipEventRecord(q.ev_chain1, bk.back));
    HIPCHK(ffv2_launch_lc_finish(a, nframes, bk.back));
    HIPCHK(hipEventRecord(bk.ev_backdone, bk.back));
    bk.backdone_valid = true;
    HIPCHK(hipMemcpyAsync(q.h_sizes, q.sizes, sizeof(uint32_t) * nframes, hipMemcpyDeviceToHost, bk.back));
    HIPCHK(hipMemcpyAsync(q.h_status, q.status, sizeof(int32_t) * nframes, hipMemcpyDeviceToHost, bk.back));
    HIPCHK(hipMemcpyAsync(q.h_offs, q.offs, sizeof(unsigned long long) * ((size_t)nframes + 1), hipMemcpyDeviceToHost, bk.back));
    HIPCHK(hipMemcpyAsync(q.h_offs + (size_t)lc.cap + 1, q.gbase + nb, sizeof(uint32_t), hipMemcpyDeviceToHost, bk.back));
    HIPCHK(hipEventRecord(q.ev_done, bk.back));
    q.nframes = nframes; q.busy = true;
    lc.sub++;
    return FFV2AMD_OK;
}

int ffv2amd_lanecoder_finish(ffv2amd_encoder *e, uint8_t *h_packets, size_t packet_stride, uint32_t *h_sizes, int32_t *h_status)
{
    if (!e || !h_packets || !h_sizes || !h_status) return FFV2AMD_ERR_INVAL;
    auto &lc = e->lc;
    if (lc.fin == lc.sub) return FFV2AMD_ERR_AGAIN;              // nothing submitted
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    auto &q = lc.set[lc.fin % (unsigned)lc.nsets];
    HIPCHK(hipEventSynchronize(q.ev_done));
    lanecoder_note_times(lc, q);
    // packets on the copy stream (the back stream may already hold the next call): they lie packed
    // on the device, each goes to its own row of the caller's array
    for (int f = 0; f < q.nframes; f++) {
        h_status[f] = q.h_status[f];
        h_sizes[f] = 0;
        if (h_status[f] < 0) continue;
        if (q.h_sizes[f] > packet_stride) { h_status[f] = FFV2AMD_ERR_NOSPACE; continue; }
        h_sizes[f] = q.h_sizes[f];
        HIPCHK(hipMemcpyAsync(h_packets + (size_t)f * packet_stride, q.packets + q.h_offs[f], h_sizes[f],
                              hipMemcpyDeviceToHost, lc.copy));
    }
    HIPCHK(hipStreamSynchronize(lc.copy));
    q.busy = false;
    lc.fin++;
    return FFV2AMD_OK;
}

int ffv2amd_lanecoder_finish_packed(ffv2amd_encoder *e, uint8_t *h_buf, size_t h_cap, uint64_t *h_offsets,
                                    uint32_t *h_sizes, int32_t *h_status)
{
    if (!e || !h_buf || !h_offsets || !h_sizes || !h_status) return FFV2AMD_ERR_INVAL;
    auto &lc = e->lc;
    if (lc.fin == lc.sub) return FFV2AMD_ERR_AGAIN;              // nothing submitted
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    auto &q = lc.set[lc.fin % (unsigned)lc.nsets];
    HIPCHK(hipEventSynchronize(q.ev_done));
    lanecoder_note_times(lc, q);
    const size_t total = (size_t)q.h_offs[q.nframes];
    if (total > h_cap) return FFV2AMD_ERR_NOSPACE;               // the call stays queued: finish it with a larger buffer
    // the packets lie packed on the device: one copy
    static const size_t piece = getenv("FFV2AMD_LC_PIECE") ? (size_t)atol(getenv("FFV2AMD_LC_PIECE")) : ((size_t)4 << 20);
    for (size_t from = 0; from < total; from += piece)            // in pieces: a single large copy crawls while kernels run
        HIPCHK(hipMemcpyAsync(h_buf + from, q.packets + from, total - from < piece ? total - from : piece, hipMemcpyDeviceToHost, lc.copy));
    for (int f = 0; f < q.nframes; f++) {
        h_status[f] = q.h_status[f];
        h_sizes[f] = h_status[f] < 0 ? 0u : q.h_sizes[f];
        h_offsets[f] = q.h_offs[f];
    }
    HIPCHK(hipStreamSynchronize(lc.copy));
    q.busy = false;
    lc.fin++;
    return FFV2AMD_OK;
}

int ffv2amd_lanecoder_encode(ffv2amd_encoder *e, int nframes, const void *d_frames, int qp, const int32_t *d_W,
                             uint8_t *h_packets, size_t packet_stride, uint32_t *h_sizes, int32_t *h_status)
{
    if (!e || !h_packets || !h_sizes || !h_status) return FFV2AMD_ERR_INVAL;
    if (e->lc.fin != e->lc.sub) return FFV2AMD_ERR_INVAL;        // a submitted call is still waiting for its finish
    const int r = ffv2amd_lanecoder_submit(e, nframes, d_frames, qp, d_W);
    if (r < 0) return r;
    return ffv2amd_lanecoder_finish(e, h_packets, packet_stride, h_sizes, h_status);
}

static int qp_alloc(ffv2amd_encoder *e)
{
    const ffv2amd_info &in = e->info;
    const size_t nb = (size_t)in.block_planes, B = (size_t)in.max_batch;
    if (!e->d_coef_ws) {
        HIPCHK(hipMalloc(&e->d_coef_ws, sizeof(int32_t) * 4096 * nb * B));
        HIPCHK(hipMalloc(&e->d_y, sizeof(int16_t) * FFV2_Y_STRIDE * nb * B));
    }
    if (e->qset[0].d_rec) return FFV2AMD_OK;
    e->q_stream_stride = (nb * 4097 + 255) / 256 * 256;          // every coefficient + the phantom slot, 1 byte each
    HIPCHK(hipStreamCreateWithFlags(&e->q_copy, hipStreamNonBlocking));
    for (auto &q : e->qset) {
        HIPCHK(hipMalloc(&q.d_rec, sizeof(FFV2SymRec) * nb * B));
        HIPCHK(hipMalloc(&q.d_stream, e->q_stream_stride * B));
        HIPCHK(hipMalloc(&q.d_totals, sizeof(uint32_t) * B));
        HIPCHK(hipMalloc(&q.d_codes, sizeof(uint32_t) * FFV2_CODES_PER_BP * nb * B));
        HIPCHK(hipMalloc(&q.d_status, sizeof(int32_t) * B));
        HIPCHK(hipHostMalloc(&q.h_rec, sizeof(FFV2SymRec) * nb * B, hipHostMallocDefault));
        HIPCHK(hipHostMalloc(&q.h_stream, e->q_stream_stride * B, hipHostMallocDefault));
        HIPCHK(hipHostMalloc(&q.h_totals, sizeof(uint32_t) * B, hipHostMallocDefault));
        HIPCHK(hipHostMalloc(&q.h_codes, sizeof(uint32_t) * FFV2_CODES_PER_BP * nb * B, hipHostMallocDefault));
        HIPCHK(hipHostMalloc(&q.h_status, sizeof(int32_t) * B, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&q.ev, hipEventDisableTiming));
        try { q.ev_frame.resize(B); } catch (...) { return FFV2AMD_ERR_NOMEM; }
        for (auto &ev : q.ev_frame) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    return FFV2AMD_OK;
}

int ffv2amd_encoder_set_device_coder(ffv2amd_encoder *e, int on)
{
    if (!e) return FFV2AMD_ERR_INVAL;
    if (e->q_fin != e->q_sub) return FFV2AMD_ERR_INVAL;          // not while batches are in flight
    e->device_coder = on != 0;
    return FFV2AMD_OK;
}

int ffv2amd_qp_submit(ffv2amd_encoder *e, int nframes, const void *d_frames, int qp, const int32_t *d_W)
{
    if (!e || !d_frames || nframes < 1 || nframes > e->info.max_batch) return FFV2AMD_ERR_INVAL;
    if (qp < 1 || qp > 64) return FFV2AMD_ERR_UNSUPPORTED;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    int r = qp_alloc(e);
    if (r < 0) return r;
    auto &q = e->qset[e->q_sub & 1u];
    if (q.busy) return FFV2AMD_ERR_AGAIN;                        // two batches already in flight
    const ffv2amd_info &in = e->info;
    const size_t nb = (size_t)in.block_planes;
    hipStream_t s = e->stream;
    HIPCHK(hipMemsetAsync(q.d_status, 0, sizeof(int32_t) * nframes, s));
    HIPCHK(hipMemsetAsync(q.d_totals, 0, sizeof(uint32_t) * nframes, s));
    FFV2TStageArgs a{};
    a.g = e->geom; a.nframes = nframes; a.frames = (const uint8_t *)d_frames;
    a.coef = e->d_coef_ws; a.energy = nullptr; a.codes = q.d_codes; a.bitcnt = e->d_bitoff; a.W = d_W;
    a.gain_thr = e->d_thr; a.gain_n = GAIN_TABLE_N; a.lds_scan = e->d_lds_scan; a.status = q.d_status;
    HIPCHK(ffv2_launch_tstage(a, s));
    HIPCHK(ffv2_launch_pvq(e->d_coef_ws, d_W, e->d_y, qp, (long long)nb * nframes, s));
    HIPCHK(ffv2_launch_compact(e->d_y, qp, (int)nb, nframes, q.d_rec, q.d_stream, e->q_stream_stride, q.d_totals, s));
    q.d_frames = (const uint8_t *)d_frames; q.d_W = d_W;
    q.on_device = e->device_coder;
    if (q.on_device) {
        // the whole entropy coder on the device: one wavefront per frame (ffv2_rangecoder.hip)
        const size_t cap = in.packet_cap_qp, B = (size_t)in.max_batch;
        if (!q.d_pre) {
            HIPCHK(hipMalloc(&q.d_pre, sizeof(uint16_t) * cap * B));
            HIPCHK(hipMalloc(&q.d_raw, cap * B));
            HIPCHK(hipMalloc(&q.d_pk, cap * B));
            HIPCHK(hipMalloc(&q.d_pk_sizes, sizeof(uint32_t) * B));
            HIPCHK(hipMalloc(&q.d_pk_status, sizeof(int32_t) * B));
            HIPCHK(hipHostMalloc(&q.h_pk_sizes, sizeof(uint32_t) * B, hipHostMallocDefault));
            HIPCHK(hipHostMalloc(&q.h_pk_status, sizeof(int32_t) * B, hipHostMallocDefault));
        }
        FFV2RangeCoderArgs rc{};
        rc.codes = q.d_codes; rc.rec = q.d_rec; rc.stream = q.d_stream; rc.stream_stride = e->q_stream_stride;
        rc.status_in = q.d_status; rc.pre = q.d_pre; rc.raw = q.d_raw; rc.cap = cap;
        rc.packets = q.d_pk; rc.packet_stride = cap; rc.sizes = q.d_pk_sizes; rc.status = q.d_pk_status;
        rc.nblk = (int)nb; rc.nsb = in.num_sb_x * in.num_sb_y; rc.planes = in.planes; rc.pix_fmt = in.pix_fmt; rc.qp = qp;
        HIPCHK(ffv2_launch_rangecoder(rc, nframes, s));
        HIPCHK(hipMemcpyAsync(q.h_pk_sizes, q.d_pk_sizes, sizeof(uint32_t) * nframes, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(q.h_pk_status, q.d_pk_status, sizeof(int32_t) * nframes, hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(q.ev, s));
        q.nframes = nframes; q.qp = qp; q.busy = true;
        e->q_sub++;
        return FFV2AMD_OK;
    }
    HIPCHK(hipMemcpyAsync(q.h_totals, q.d_totals, sizeof(uint32_t) * nframes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(q.h_status, q.d_status, sizeof(int32_t) * nframes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(q.h_rec, q.d_rec, sizeof(FFV2SymRec) * nb * nframes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(q.h_codes, q.d_codes, sizeof(uint32_t) * FFV2_CODES_PER_BP * nb * nframes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(q.ev, s));
    q.nframes = nframes; q.qp = qp; q.busy = true;
    e->q_sub++;
    return FFV2AMD_OK;
}

int ffv2amd_qp_finish(ffv2amd_encoder *e, uint8_t *h_packets, size_t packet_stride, uint32_t *h_sizes, int32_t *h_status)
{
    if (!e || !h_packets || !h_sizes || !h_status) return FFV2AMD_ERR_INVAL;
    if (e->q_fin == e->q_sub) return FFV2AMD_ERR_AGAIN;          // nothing submitted
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    auto &q = e->qset[e->q_fin & 1u];
    const ffv2amd_info &in = e->info;
    const size_t nb = (size_t)in.block_planes;
    const int nframes = q.nframes, qp = q.qp;
    HIPCHK(hipEventSynchronize(q.ev));
    if (q.on_device) {
        // packets were finished on the device: bring each one back with its own size
        for (int f = 0; f < nframes; f++) {
            h_status[f] = q.h_pk_status[f];
            h_sizes[f] = 0;
            if (h_status[f] < 0) continue;
            if (q.h_pk_sizes[f] > packet_stride) { h_status[f] = FFV2AMD_ERR_NOSPACE; continue; }
            h_sizes[f] = q.h_pk_sizes[f];
            HIPCHK(hipMemcpyAsync(h_packets + (size_t)f * packet_stride, q.d_pk + (size_t)f * in.packet_cap_qp, q.h_pk_sizes[f],
                                  hipMemcpyDeviceToHost, e->q_copy));
        }
        HIPCHK(hipStreamSynchronize(e->q_copy));
        q.busy = false;
        e->q_fin++;
        return FFV2AMD_OK;
    }
    // A frame the fast T-stage refused (samples above the declared depth, a gain beyond the device table) is coded by
    // the reference all the same (ffv2.c:26-38, ffv2enc.c:163-174): rerun it through the wide T-stage, the PVQ search and
    // the compaction (all take any int32 coefficient), take its gains with the host's pow, and let the host coder below
    // treat it like the others.  Synchronous, on the encoder's stream: the frames are still the caller's to keep alive.
    for (int f = 0; f < nframes; f++) {
        if (q.h_status[f] != FFV2AMD_ERR_RANGE) continue;
        hipStream_t s = e->stream;
        int32_t *coef = e->d_coef_ws + (size_t)f * nb * 4096;
        int16_t *y = e->d_y + (size_t)f * nb * FFV2_Y_STRIDE;
        const int32_t *dW = q.d_W ? q.d_W + (size_t)f * nb : nullptr;
        int r = wide_tstage(e, q.d_frames + (size_t)f * in.frame_stride, coef, s);
        if (r < 0) return r;
        HIPCHK(ffv2_launch_pvq(coef, dW, y, qp, (long long)nb, s));
        HIPCHK(hipMemsetAsync(q.d_totals + f, 0, sizeof(uint32_t), s));
        HIPCHK(ffv2_launch_compact(y, qp, (int)nb, 1, q.d_rec + (size_t)f * nb, q.d_stream + (size_t)f * e->q_stream_stride,
                                   e->q_stream_stride, q.d_totals + f, s));
        std::vector<int64_t> en;
        std::vector<int32_t> c0, W;
        try { en.resize(nb * 13); c0.resize(nb); if (dW) W.resize(nb); } catch (...) { return FFV2AMD_ERR_NOMEM; }
        HIPCHK(hipMemcpyAsync(en.data(), e->d_wide_en, sizeof(int64_t) * en.size(), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(c0.data(), e->d_wide_c0, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, s));
        if (dW) HIPCHK(hipMemcpyAsync(W.data(), dW, sizeof(int32_t) * nb, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(q.h_rec + (size_t)f * nb, q.d_rec + (size_t)f * nb, sizeof(FFV2SymRec) * nb, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(q.h_totals + f, q.d_totals + f, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        uint32_t *codes = q.h_codes + (size_t)f * nb * FFV2_CODES_PER_BP;
        for (size_t bp = 0; bp < nb; bp++) {
            codes[bp * FFV2_CODES_PER_BP] = (uint32_t)c0[bp];
            for (int b = 0; b < 13; b++) {
                int64_t eb = en[bp * 13 + b];
                if (b == 12 && dW) eb = (int64_t)((uint64_t)eb + (uint64_t)((int64_t)W[bp] * W[bp]));
                codes[bp * FFV2_CODES_PER_BP + 1 + b] = coded_gain_host(eb);
            }
        }
        q.h_status[f] = 0;
    }
    // each frame's symbol stream: its own size, on the copy stream, one event per frame
    for (int f = 0; f < nframes; f++) {
        h_status[f] = q.h_status[f];
        h_sizes[f] = 0;
        if (h_status[f] < 0) continue;
        if (q.h_totals[f] > e->q_stream_stride) { h_status[f] = FFV2AMD_ERR_DEVICE; continue; }
        HIPCHK(hipMemcpyAsync(q.h_stream + (size_t)f * e->q_stream_stride, q.d_stream + (size_t)f * e->q_stream_stride,
                              q.h_totals[f], hipMemcpyDeviceToHost, e->q_copy));
        HIPCHK(hipEventRecord(q.ev_frame[(size_t)f], e->q_copy));
    }
    const int device = e->device;
    run_parallel(nframes, nframes, [&](int f) {
        if (h_status[f] < 0) return;
        if (hipSetDevice(device) != hipSuccess || hipEventSynchronize(q.ev_frame[(size_t)f]) != hipSuccess) {
            h_status[f] = FFV2AMD_ERR_DEVICE;
            return;
        }
        size_t n = 0;
        const uint32_t *codes = q.h_codes + (size_t)f * nb * FFV2_CODES_PER_BP;
        const FFV2SymRec *rec = q.h_rec + (size_t)f * nb;
        const int8_t *st = q.h_stream + (size_t)f * e->q_stream_stride;
        uint8_t *out = h_packets + (size_t)f * packet_stride;
        int r;
        try {
            r = qp <= 16 ? encode_frame_host_compact<1>(in, qp, codes, rec, st, out, packet_stride, &n)
              : qp <= 32 ? encode_frame_host_compact<2>(in, qp, codes, rec, st, out, packet_stride, &n)
                         : encode_frame_host_compact<4>(in, qp, codes, rec, st, out, packet_stride, &n);
        } catch (...) { r = FFV2AMD_ERR_NOMEM; }
        h_status[f] = r;
        h_sizes[f] = r < 0 ? 0 : (uint32_t)n;
    });
    q.busy = false;
    e->q_fin++;
    return FFV2AMD_OK;
}

// Host frames through the qp > 0 pipeline, one frame per batch: the shape of avcodec_send_frame /
// avcodec_receive_packet (encode.c:420,449) for global_quality > 0.  send: the caller's rows are
// gathered into a page-locked frame, copied over on the encoder's stream and followed there by
// T-stage, PVQ search and symbol compaction (ffv2amd_qp_submit); at most two frames in flight.
// receive: the oldest frame's range coder on the calling thread (ffv2amd_qp_finish) -- the GPU
// meanwhile works on the frame sent after it.
int ffv2amd_qp_send_frame(ffv2amd_encoder *e, const uint8_t *const data[4], const ptrdiff_t linesize[4],
                          int qp, const int32_t *W, int64_t tag)
{
    if (!e || !data || !linesize) return FFV2AMD_ERR_INVAL;
    if (qp < 1 || qp > 64) return FFV2AMD_ERR_UNSUPPORTED;
    const ffv2amd_info &in = e->info;
    for (int p = 0; p < in.planes; p++)
        if (!data[p]) return FFV2AMD_ERR_INVAL;
    if (e->q_sub - e->q_fin >= 2) return FFV2AMD_ERR_AGAIN;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    const int k = (int)(e->q_sub & 1u);
    if (!e->qd_frame[k]) {
        HIPCHK(hipMalloc(&e->qd_frame[k], in.frame_stride));
        HIPCHK(hipMalloc(&e->qd_w[k], sizeof(int32_t) * in.block_planes));
        HIPCHK(hipHostMalloc(&e->qh_frame[k], in.frame_stride, hipHostMallocDefault));
        memset(e->qh_frame[k], 0, in.frame_stride);
    }
    const size_t row_bytes = (size_t)in.width * (in.depth > 8 ? 2 : 1);
    for (int p = 0; p < in.planes; p++) {
        uint8_t *dst = e->qh_frame[k] + (size_t)p * in.plane_stride;
        for (int y = 0; y < in.height; y++)
            memcpy(dst + (size_t)y * in.row_pitch, data[p] + (ptrdiff_t)y * linesize[p], row_bytes);
        HIPCHK(hipMemcpyAsync(e->qd_frame[k] + (size_t)p * in.plane_stride, dst, in.row_pitch * (size_t)in.height,
                              hipMemcpyHostToDevice, e->stream));
    }
    const int32_t *dW = nullptr;
    if (W) {
        // W may be pageable and reused by the caller: this copy is complete when the call returns
        HIPCHK(hipMemcpyAsync(e->qd_w[k], W, sizeof(int32_t) * in.block_planes, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        dW = e->qd_w[k];
    }
    const int r = ffv2amd_qp_submit(e, 1, e->qd_frame[k], qp, dW);
    if (r < 0) return r;
    e->q_tag[k] = tag;
    return FFV2AMD_OK;
}

// The same for a yuv420p* frame (data[0..2] = Y, U, V): the ffmpeg tool's format step (ffv2_upconv.hip) on the encoder's
// stream in front of the T-stage, as ffv2amd_ring_send_420 does for qp 0.  Parity unpinned twice over.
int ffv2amd_qp_send_frame_420(ffv2amd_encoder *e, const uint8_t *const data[3], const ptrdiff_t linesize[3], int qp, int64_t tag)
{
    if (!e || !data || !linesize || !data[0] || !data[1] || !data[2]) return FFV2AMD_ERR_INVAL;
    if (qp < 1 || qp > 64) return FFV2AMD_ERR_UNSUPPORTED;
    if (e->q_sub - e->q_fin >= 2) return FFV2AMD_ERR_AGAIN;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    int r = upconv_ready(e);
    if (r < 0) return r;
    const ffv2amd_info &in = e->info;
    const int k = (int)(e->q_sub & 1u);
    const size_t bps = in.depth > 8 ? 2 : 1;
    const int cw = (in.width + 1) >> 1, ch = (in.height + 1) >> 1;
    const size_t c_pitch = align_up((size_t)cw * bps, 128);
    if (!e->qd_frame[k]) {
        HIPCHK(hipMalloc(&e->qd_frame[k], in.frame_stride));
        HIPCHK(hipMalloc(&e->qd_w[k], sizeof(int32_t) * in.block_planes));
        HIPCHK(hipHostMalloc(&e->qh_frame[k], in.frame_stride, hipHostMallocDefault));
        memset(e->qh_frame[k], 0, in.frame_stride);
    }
    if (!e->qd_c420[k]) HIPCHK(hipMalloc(&e->qd_c420[k], 2 * c_pitch * (size_t)ch));
    // staging: luma in plane 0 of the page-locked frame, U and V behind it (they fit planes 1 and 2, see ring_send_420)
    uint8_t *hy = e->qh_frame[k], *hc = e->qh_frame[k] + in.plane_stride;
    for (int y = 0; y < in.height; y++)
        memcpy(hy + (size_t)y * in.row_pitch, data[0] + (ptrdiff_t)y * linesize[0], (size_t)in.width * bps);
    for (int p = 0; p < 2; p++)
        for (int y = 0; y < ch; y++)
            memcpy(hc + ((size_t)p * ch + y) * c_pitch, data[1 + p] + (ptrdiff_t)y * linesize[1 + p], (size_t)cw * bps);
    hipStream_t s = e->stream;
    HIPCHK(hipMemcpyAsync(e->qd_frame[k], hy, in.row_pitch * (size_t)in.height, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(e->qd_c420[k], hc, 2 * c_pitch * (size_t)ch, hipMemcpyHostToDevice, s));
    HIPCHK(ffv2_launch_upconv_chroma(e->upconv, e->geom, 1, e->qd_c420[k], c_pitch, c_pitch * (size_t)ch, 0, e->qd_frame[k], s));
    r = ffv2amd_qp_submit(e, 1, e->qd_frame[k], qp, nullptr);
    if (r < 0) return r;
    e->q_tag[k] = tag;
    return FFV2AMD_OK;
}

int ffv2amd_qp_receive_packet(ffv2amd_encoder *e, uint8_t *out, size_t out_cap, size_t *out_size, int64_t *tag)
{
    if (!e || !out || !out_size) return FFV2AMD_ERR_INVAL;
    if (e->q_fin == e->q_sub) return FFV2AMD_ERR_AGAIN;
    const int k = (int)(e->q_fin & 1u);
    if (e->qset[k].nframes != 1) return FFV2AMD_ERR_INVAL;      // the oldest batch came from ffv2amd_qp_submit
    uint32_t sz = 0;
    int32_t st = 0;
    if (tag) *tag = e->q_tag[k];
    const int r = ffv2amd_qp_finish(e, out, out_cap, &sz, &st);
    if (r < 0) return r;
    if (st < 0) return st;                                       // the frame has left the pipeline all the same
    *out_size = sz;
    return FFV2AMD_OK;
}


// ---------------------------------------------------------------------------------------------
// send_frame / receive_packet at qp > 0 at the device coder's rate (encode.c:420,449 around ffv2enc.c:453).
// The adaptive range coder is one chain per frame, so frames are coded many at a time (ffv2_lanecoder.hip):
// this ring collects them `frames_per_call` at a time in device memory as they arrive (H2D on its own stream),
// hands every full batch to the lane coder (two calls in flight) and gives the packets back in send order.
// ---------------------------------------------------------------------------------------------
static bool host_range_registered(ffv2amd_encoder *e, const uint8_t *src, size_t need);

int ffv2amd_qpring_close(ffv2amd_encoder *e)
{
    if (!e) return FFV2AMD_ERR_INVAL;
    auto &r = e->qr;
    if (!r.open && !r.h2d) return FFV2AMD_OK;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    if (r.h2d) (void)hipStreamSynchronize(r.h2d);
    (void)hipStreamSynchronize(e->stream);
    (void)ffv2amd_lanecoder_close(e);
    for (int i = 0; i < ffv2amd_encoder::QpRing::NBUF; i++) {
        (void)hipFree(r.d_frames[i]); (void)hipFree(r.d_c420[i]); (void)hipFree(r.d_w[i]);
        r.d_frames[i] = nullptr; r.d_c420[i] = nullptr; r.d_w[i] = nullptr; r.any_w[i] = false; r.tags[i].clear();
    }
    for (int i = 0; i < ffv2amd_encoder::QpRing::NBOUNCE; i++) {
        if (r.bounce[i]) (void)hipHostFree(r.bounce[i]);
        if (r.ev_bounce[i]) (void)hipEventDestroy(r.ev_bounce[i]);
        r.bounce[i] = nullptr; r.ev_bounce[i] = nullptr;
    }
    if (r.h_buf) (void)hipHostFree(r.h_buf);
    r.h_buf = nullptr; r.h_cap = 0;
    if (r.pool) { r.pool->shutdown(); delete r.pool; r.pool = nullptr; }
    if (e->ring.empty()) {                                        // FFV2AMD_FRAME_REGISTER: what this encoder page-locked (the frame ring is not using it)
        for (auto &g : e->ring_reg) (void)hipHostUnregister((void *)g.base);
        e->ring_reg.clear();
    }
    if (r.ev_batch) (void)hipEventDestroy(r.ev_batch);
    if (r.h2d) (void)hipStreamDestroy(r.h2d);
    r.ev_batch = nullptr; r.h2d = nullptr;
    r.fill = r.count = r.nflight = r.done_n = r.done_at = 0;
    r.open = false;
    (void)hipGetLastError();
    return FFV2AMD_OK;
}

static int qpring_open_with(ffv2amd_encoder *e, int qp, int frames_per_call, size_t packet_cap, int calls, int backs);

int ffv2amd_qpring_open(ffv2amd_encoder *e, int qp, int frames_per_call, size_t packet_cap)
{
    if (!e || frames_per_call < 1) return FFV2AMD_ERR_INVAL;
    if (qp < 1 || qp > 64) return FFV2AMD_ERR_UNSUPPORTED;
    if (e->qr.open) return FFV2AMD_ERR_INVAL;
    if (e->lc.cap) return FFV2AMD_ERR_INVAL;                     // the lane coder is in use by its own entry points
    // Calls in flight, and range chains side by side.  A call lasts one frame's chain whatever it holds (265 ns per
    // pixel at qp 16: 0.55 s at 1080p, 1.9 s at 4K), the rest of its work about 67 ps per pixel and frame: below some
    // 4 000 frames per call -- of any size -- the chain is what a call waits for, and further calls' chains beside it
    // are nearly free (1080p / qp 16, page-locked frames, same box: 512 frames per call 2.5 -> 6.9 Gpix/s, 1 024:
    // 4.6 -> 8.8, 2 048: 7.1 -> 10.1 with four calls and four chains; 4 096 per call is at the PCIe rate with two
    // calls and one chain).  Four calls hold five batches of frames and four sets of coder scratch: where the device
    // cannot, fewer are tried.  FFV2AMD_QPRING_CALLS (2..4) and FFV2AMD_LC_BACKS (1..calls) override.
    const int cenv = getenv("FFV2AMD_QPRING_CALLS") ? atoi(getenv("FFV2AMD_QPRING_CALLS")) : 0;      // read per open: tests flip them
    const int benv = getenv("FFV2AMD_LC_BACKS") ? atoi(getenv("FFV2AMD_LC_BACKS")) : 0;
    const bool many = frames_per_call >= 3800;
    int calls = many ? 2 : 4;
    if (cenv >= 2 && cenv <= 4) calls = cenv;
    for (;;) {
        int backs = many ? 1 : calls;
        if (benv >= 1) backs = benv > calls ? calls : benv;
        const int rc = qpring_open_with(e, qp, frames_per_call, packet_cap, calls, backs);
        if (calls == 2 || (rc < 0 && rc != FFV2AMD_ERR_NOMEM)) return rc;
        if (rc == FFV2AMD_OK) {
            // what send allocates on first use has to fit as well: 4:2:0 chroma and W of every batch
            const ffv2amd_info &in = e->info;
            const size_t bps = in.depth > 8 ? 2 : 1;
            const size_t c420 = in.planes == 3 ? 2 * align_up((size_t)((in.width + 1) >> 1) * bps, 128) * (size_t)((in.height + 1) >> 1) : 0;
            const size_t later = (size_t)(calls + 1) * (size_t)frames_per_call * (c420 + sizeof(int32_t) * (size_t)in.block_planes)
                               + ((size_t)512 << 20);
            DeviceGuard guard(e->device);
            size_t free_b = 0, total_b = 0;
            if (!guard.ok || hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b >= later) return rc;
            (void)ffv2amd_qpring_close(e);
        }
        calls--;
    }
}

static int qpring_open_with(ffv2amd_encoder *e, int qp, int frames_per_call, size_t packet_cap, int calls, int backs)
{
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    const ffv2amd_info &in = e->info;
    auto &r = e->qr;
    int rc = ffv2amd_lanecoder_open_ex(e, frames_per_call, packet_cap, calls, backs);
    if (rc < 0) return rc;
    r.qp = qp; r.cap = frames_per_call; r.calls = calls;
    r.pcap = packet_cap ? packet_cap : in.packet_cap_qp;
#define QK(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); r.open = true; ffv2amd_qpring_close(e); return FFV2AMD_ERR_NOMEM; } } while (0)
    QK(hipStreamCreateWithFlags(&r.h2d, hipStreamNonBlocking));
    QK(hipEventCreateWithFlags(&r.ev_batch, hipEventDisableTiming));
    for (int i = 0; i <= calls; i++) {
        QK(hipMalloc(&r.d_frames[i], in.frame_stride * (size_t)frames_per_call));
        try { r.tags[i].assign((size_t)frames_per_call, 0); r.is420[i].assign((size_t)frames_per_call, 0); }
        catch (...) { r.open = true; ffv2amd_qpring_close(e); return FFV2AMD_ERR_NOMEM; }
    }
    r.h_cap = (size_t)frames_per_call * (r.pcap + 16);
    QK(hipHostMalloc(&r.h_buf, r.h_cap, hipHostMallocDefault));
#undef QK
    try {
        r.offs.assign((size_t)frames_per_call + 1, 0); r.sizes.assign((size_t)frames_per_call, 0);
        r.status.assign((size_t)frames_per_call, 0); r.done_tags.assign((size_t)frames_per_call, 0);
    } catch (...) { r.open = true; ffv2amd_qpring_close(e); return FFV2AMD_ERR_NOMEM; }
    r.fill = 0; r.count = 0; r.nflight = 0; r.done_n = r.done_at = 0;
    r.open = true;
    {
        static const size_t mb = getenv("FFV2AMD_QPRING_BOUNCE_MB") ? (size_t)atol(getenv("FFV2AMD_QPRING_BOUNCE_MB")) : 256;
        const size_t n = (mb << 20) / in.frame_stride;
        r.nbounce = n < 4 ? 4 : n > (size_t)ffv2amd_encoder::QpRing::NBOUNCE ? ffv2amd_encoder::QpRing::NBOUNCE : (int)n;
        if (r.nbounce > frames_per_call + 1) r.nbounce = frames_per_call + 1 < 4 ? 4 : frames_per_call + 1;
    }
    if (in.frame_stride >= ((size_t)2 << 20)) {                  // helpers for the row copies of pageable frames
        const unsigned hw = std::thread::hardware_concurrency();
        const int want = getenv("FFV2AMD_GATHER_THREADS") ? atoi(getenv("FFV2AMD_GATHER_THREADS")) : 6;   // caller included
        const int helpers = (int)hw >= want ? want - 1 : (hw > 1 ? (int)hw - 1 : 0);
        if (helpers > 0) {
            r.pool = new (std::nothrow) GatherPool;
            if (r.pool) r.pool->start(helpers, e->device);
        }
    }
    return FFV2AMD_OK;
}

// the oldest call in flight -> the packets waiting to be received (blocks until that call is through)
static int qpring_collect(ffv2amd_encoder *e)
{
    auto &r = e->qr;
    if (r.nflight == 0) return FFV2AMD_ERR_AGAIN;
    if (r.done_at < r.done_n) return FFV2AMD_ERR_AGAIN;          // the previous batch has not been received yet
    const int b = r.flight[0], n = r.flight_n[0];
    const int rc = ffv2amd_lanecoder_finish_packed(e, r.h_buf, r.h_cap, r.offs.data(), r.sizes.data(), r.status.data());
    if (rc < 0) return rc;
    for (int i = 0; i < n; i++) r.done_tags[(size_t)i] = r.tags[b][(size_t)i];
    r.done_n = n; r.done_at = 0;
    for (int k = 1; k < r.nflight; k++) { r.flight[k - 1] = r.flight[k]; r.flight_n[k - 1] = r.flight_n[k]; }
    r.nflight--;
    return FFV2AMD_OK;
}

// the batch being filled -> the lane coder.  FFV2AMD_ERR_AGAIN: `calls` calls are in flight and the one before them has
// not been received yet (the caller has to take packets first).
static int qpring_submit(ffv2amd_encoder *e)
{
    auto &r = e->qr;
    if (r.count == 0) return FFV2AMD_OK;
    if (r.nflight == r.calls) {
        const int rc = qpring_collect(e);
        if (rc < 0) return rc;
    }
    HIPCHK(hipEventRecord(r.ev_batch, r.h2d));
    HIPCHK(hipStreamWaitEvent(e->stream, r.ev_batch, 0));
    const int b = r.fill;
    {   // 4:2:0 frames of the batch: their chroma is up-converted now, one launch per run of such frames (a launch per
        // frame on the copy stream stood in the queue behind the coder's long kernels every other run)
        const ffv2amd_info &in = e->info;
        const size_t bps = in.depth > 8 ? 2 : 1;
        const int cw = (in.width + 1) >> 1, ch = (in.height + 1) >> 1;
        const size_t c_pitch = align_up((size_t)cw * bps, 128), c_frame = 2 * c_pitch * (size_t)ch;
        for (int i0 = 0; i0 < r.count; ) {
            if (!r.is420[b][(size_t)i0]) { i0++; continue; }
            int i1 = i0;
            while (i1 < r.count && r.is420[b][(size_t)i1]) i1++;
            HIPCHK(ffv2_launch_upconv_chroma(e->upconv, e->geom, i1 - i0, r.d_c420[b] + (size_t)i0 * c_frame, c_pitch,
                                             c_pitch * (size_t)ch, c_frame, r.d_frames[b] + (size_t)i0 * in.frame_stride, e->stream));
            i0 = i1;
        }
    }
    const int rc = ffv2amd_lanecoder_submit(e, r.count, r.d_frames[b], r.qp, r.any_w[b] ? r.d_w[b] : nullptr);
    if (rc < 0) return rc;
    r.flight[r.nflight] = b; r.flight_n[r.nflight] = r.count; r.nflight++;
    // the next batch goes into the buffer no call in flight reads
    for (int i = 0; i <= r.calls; i++) {
        bool used = false;
        for (int k = 0; k < r.nflight; k++) used = used || r.flight[k] == i;
        if (!used) { r.fill = i; break; }
    }
    r.count = 0;
    r.any_w[r.fill] = false;
    return FFV2AMD_OK;
}

int ffv2amd_qpring_send(ffv2amd_encoder *e, const uint8_t *const data[4], const ptrdiff_t linesize[4],
                        const int32_t *W, int64_t tag, unsigned flags)
{
    if (!e || !data || !linesize) return FFV2AMD_ERR_INVAL;
    auto &r = e->qr;
    if (!r.open) return FFV2AMD_ERR_INVAL;
    const ffv2amd_info &in = e->info;
    const bool is420 = (flags & FFV2AMD_FRAME_YUV420) != 0;
    const int npl = is420 ? 3 : in.planes;
    if (is420 && in.planes != 3) return FFV2AMD_ERR_INVAL;
    for (int p = 0; p < npl; p++)
        if (!data[p]) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    if (r.count == r.cap) {                                      // a full batch that could not leave yet
        const int rc = qpring_submit(e);
        if (rc < 0) return rc;
    }
    const int b = r.fill;
    const size_t bps = in.depth > 8 ? 2 : 1, nb = (size_t)in.block_planes;
    const int cw = (in.width + 1) >> 1, ch = (in.height + 1) >> 1;
    const size_t c_pitch = align_up((size_t)cw * bps, 128);
    if (is420) {
        int rc = upconv_ready(e);
        if (rc < 0) return rc;
        if (!r.d_c420[b]) HIPCHK(hipMalloc(&r.d_c420[b], 2 * c_pitch * (size_t)ch * (size_t)r.cap));
    }
    uint8_t *d_frame = r.d_frames[b] + (size_t)r.count * in.frame_stride;
    uint8_t *d_c = is420 ? r.d_c420[b] + (size_t)r.count * 2 * c_pitch * (size_t)ch : nullptr;
    // where each plane goes: rows `pitch` apart on the device
    struct Pl { const uint8_t *src; ptrdiff_t ls; size_t row_bytes, pitch; int rows; uint8_t *dst; } pl[4];
    for (int p = 0; p < npl; p++) {
        const bool chroma = is420 && p > 0;
        pl[p].src = data[p]; pl[p].ls = linesize[p];
        pl[p].row_bytes = (size_t)(chroma ? cw : in.width) * bps;
        pl[p].pitch = chroma ? c_pitch : in.row_pitch;
        pl[p].rows = chroma ? ch : in.height;
        pl[p].dst = chroma ? d_c + (size_t)(p - 1) * c_pitch * (size_t)ch : d_frame + (size_t)(is420 ? 0 : p) * in.plane_stride;
    }
    hipStream_t sh = r.h2d;
    bool in_place = (flags & FFV2AMD_FRAME_PINNED) != 0;
    if (!in_place && (flags & FFV2AMD_FRAME_REGISTER)) {         // pageable memory from a pool: page-locked on first sight
        in_place = true;
        for (int p = 0; p < npl; p++)
            in_place = in_place && pl[p].ls > 0 &&
                       host_range_registered(e, pl[p].src, (size_t)pl[p].ls * (size_t)(pl[p].rows - 1) + pl[p].row_bytes);
    }
    if (in_place) {
        for (int p = 0; p < npl; p++) {
            if (pl[p].ls == (ptrdiff_t)pl[p].pitch)
                HIPCHK(hipMemcpyAsync(pl[p].dst, pl[p].src, pl[p].pitch * (size_t)(pl[p].rows - 1) + pl[p].row_bytes, hipMemcpyHostToDevice, sh));
            else
                HIPCHK(hipMemcpy2DAsync(pl[p].dst, pl[p].pitch, pl[p].src, (size_t)pl[p].ls, pl[p].row_bytes, (size_t)pl[p].rows, hipMemcpyHostToDevice, sh));
        }
    } else {
        // pageable memory: through one of a few page-locked frames (complete when the call returns as far as the caller
        // is concerned: its rows are copied here)
        const int k = (int)(r.nb_seq++ % (unsigned)r.nbounce);
        if (!r.bounce[k]) {
            HIPCHK(hipHostMalloc(&r.bounce[k], in.frame_stride, hipHostMallocDefault));
            HIPCHK(hipEventCreateWithFlags(&r.ev_bounce[k], hipEventDisableTiming));
        } else {
            HIPCHK(hipEventSynchronize(r.ev_bounce[k]));
        }
        // rows are copied in slices, by the ring's helper threads where the picture is large enough to pay for the
        // hand-over (FFV2AMD_GATHER_THREADS, caller included); then one DMA per run of planes that lie back to back on
        // both sides (a DMA per slice, queued by whichever thread had copied it, cost more in the runtime's stream lock
        // than its early start bought: twelve calls per 6 MB frame.  The next frame's rows are copied while this
        // frame's DMA runs, which is all the overlap the rate needs)
        uint8_t *at[4];
        at[0] = r.bounce[k];
        for (int p = 1; p < npl; p++) at[p] = at[p - 1] + pl[p - 1].pitch * (size_t)pl[p - 1].rows;
        const int per = r.pool ? 4 : 1, nsl = npl * per;
        const std::function<void(int)> slice = [&](int i) {
            const int p = i / per, q = i % per;
            const int y0 = (int)((long long)pl[p].rows * q / per), y1 = (int)((long long)pl[p].rows * (q + 1) / per);
            if (pl[p].ls == (ptrdiff_t)pl[p].pitch && y1 > y0) {
                memcpy(at[p] + (size_t)y0 * pl[p].pitch, pl[p].src + (ptrdiff_t)y0 * pl[p].ls,
                       pl[p].pitch * (size_t)(y1 - y0 - 1) + pl[p].row_bytes);
                return;
            }
            for (int y = y0; y < y1; y++)
                memcpy(at[p] + (size_t)y * pl[p].pitch, pl[p].src + (ptrdiff_t)y * pl[p].ls, pl[p].row_bytes);
        };
        if (r.pool) {
            try { r.pool->run(nsl, slice); } catch (...) { return FFV2AMD_ERR_NOMEM; }
        } else {
            for (int i = 0; i < nsl; i++) slice(i);
        }
        for (int p = 0; p < npl; ) {
            int p1 = p + 1;
            while (p1 < npl && pl[p1].dst == pl[p1 - 1].dst + pl[p1 - 1].pitch * (size_t)pl[p1 - 1].rows) p1++;   // at[] is back to back by construction
            const size_t bytes = (size_t)(at[p1 - 1] - at[p]) + pl[p1 - 1].pitch * (size_t)(pl[p1 - 1].rows - 1) + pl[p1 - 1].row_bytes;
            HIPCHK(hipMemcpyAsync(pl[p].dst, at[p], bytes, hipMemcpyHostToDevice, sh));
            p = p1;
        }
        HIPCHK(hipEventRecord(r.ev_bounce[k], sh));
    }
    r.is420[b][(size_t)r.count] = is420 ? 1 : 0;                 // up-converted when the batch leaves (qpring_submit)
    if (W) {
        if (!r.d_w[b]) HIPCHK(hipMalloc(&r.d_w[b], sizeof(int32_t) * nb * (size_t)r.cap));
        if (!r.any_w[b]) {
            HIPCHK(hipMemsetAsync(r.d_w[b], 0, sizeof(int32_t) * nb * (size_t)r.cap, sh));
            r.any_w[b] = true;
        }
        // W may be pageable and reused by the caller: the copy is complete when the call returns
        HIPCHK(hipMemcpyAsync(r.d_w[b] + (size_t)r.count * nb, W, sizeof(int32_t) * nb, hipMemcpyHostToDevice, sh));
        HIPCHK(hipStreamSynchronize(sh));
    }
    r.tags[b][(size_t)r.count] = tag;
    r.count++;
    if (r.count == r.cap) {
        const int rc = qpring_submit(e);                         // AGAIN here is not the caller's: the frame is in
        if (rc < 0 && rc != FFV2AMD_ERR_AGAIN) return rc;
    }
    return FFV2AMD_OK;
}

int ffv2amd_qpring_flush(ffv2amd_encoder *e)
{
    if (!e || !e->qr.open) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    return qpring_submit(e);
}

int ffv2amd_qpring_pending(const ffv2amd_encoder *e)
{
    if (!e || !e->qr.open) return 0;
    const auto &r = e->qr;
    int n = r.count + (r.done_n - r.done_at);
    for (int k = 0; k < r.nflight; k++) n += r.flight_n[k];
    return n;
}

int ffv2amd_qpring_receive(ffv2amd_encoder *e, uint8_t *out, size_t out_cap, size_t *out_size, int64_t *tag, int wait)
{
    if (!e || !out || !out_size) return FFV2AMD_ERR_INVAL;
    auto &r = e->qr;
    if (!r.open) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    if (r.done_at == r.done_n) {
        if (r.nflight == 0) return FFV2AMD_ERR_AGAIN;            // nothing submitted: send more frames, or flush
        if (!wait) {
            auto &q = e->lc.set[e->lc.fin % (unsigned)e->lc.nsets];
            const hipError_t st = hipEventQuery(q.ev_done);
            if (st == hipErrorNotReady) { (void)hipGetLastError(); return FFV2AMD_ERR_AGAIN; }
            HIPCHK(st);
        }
        const int rc = qpring_collect(e);
        if (rc < 0) return rc;
        if (r.count == r.cap) (void)qpring_submit(e);            // a full batch was waiting for this call's place
    }
    const int i = r.done_at;
    if (tag) *tag = r.done_tags[(size_t)i];
    const int32_t st = r.status[(size_t)i];
    if (st < 0) { r.done_at++; return st; }                      // the frame has left the ring all the same
    const size_t n = r.sizes[(size_t)i];
    if (n > out_cap) return FFV2AMD_ERR_NOSPACE;
    memcpy(out, r.h_buf + r.offs[(size_t)i], n);
    *out_size = n;
    r.done_at++;
    return FFV2AMD_OK;
}

int ffv2amd_qp_pending(const ffv2amd_encoder *e) { return e ? (int)(e->q_sub - e->q_fin) : 0; }

int ffv2amd_encode_batch_to_host(ffv2amd_encoder *e, int nframes, const void *d_frames,
                                 int qp, const int32_t *d_W,
                                 uint8_t *h_packets, size_t packet_stride,
                                 uint32_t *h_sizes, int32_t *h_status)
{
    if (!e || !d_frames || !h_packets || !h_sizes || !h_status || nframes < 1 || nframes > e->info.max_batch || qp < 0)
        return FFV2AMD_ERR_INVAL;
    const ffv2amd_info &in = e->info;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    hipStream_t s = e->stream;
    const size_t nb = (size_t)in.block_planes, B = (size_t)in.max_batch;
    if (!e->d_pk_ws) {
        HIPCHK(hipMalloc(&e->d_pk_ws, in.packet_cap * B));
        HIPCHK(hipMalloc(&e->d_sizes_ws, sizeof(uint32_t) * B));
    }
    if (qp == 0) {
        int r = ffv2amd_encode_batch_device(e, nframes, d_frames, 0, d_W, e->d_pk_ws, in.packet_cap,
                                            e->d_sizes_ws, e->d_status, s);
        if (r < 0) return r;
        HIPCHK(ffv2amd_encoder_flush(e, s) < 0 ? hipErrorUnknown : hipSuccess);
        std::vector<uint8_t> tmp(in.packet_cap * (size_t)nframes);
        HIPCHK(hipMemcpyAsync(h_sizes, e->d_sizes_ws, sizeof(uint32_t) * nframes, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(h_status, e->d_status, sizeof(int32_t) * nframes, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(tmp.data(), e->d_pk_ws, tmp.size(), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (int f = 0; f < nframes; f++) {
            if (h_status[f] == FFV2AMD_ERR_RANGE) {                  // the reference codes such a frame: rerun it wide
                size_t n = 0;
                const int rw = wide_encode_frame(e, (const uint8_t *)d_frames + (size_t)f * in.frame_stride,
                                                 d_W ? d_W + (size_t)f * nb : nullptr, h_packets + (size_t)f * packet_stride,
                                                 packet_stride, &n);
                h_status[f] = rw;
                h_sizes[f] = rw < 0 ? 0u : (uint32_t)n;
                continue;
            }
            if (h_status[f] < 0) continue;
            if (h_sizes[f] > packet_stride) { h_status[f] = FFV2AMD_ERR_NOSPACE; continue; }
            memcpy(h_packets + (size_t)f * packet_stride, tmp.data() + (size_t)f * in.packet_cap, h_sizes[f]);
        }
        return FFV2AMD_OK;
    }
    if (qp <= 64) {
        if (e->q_fin != e->q_sub) return FFV2AMD_ERR_INVAL;      // a submitted batch is still waiting for its finish
        int r = ffv2amd_qp_submit(e, nframes, d_frames, qp, d_W);
        if (r < 0) return r;
        return ffv2amd_qp_finish(e, h_packets, packet_stride, h_sizes, h_status);
    }
    const auto t_begin = std::chrono::steady_clock::now();
    // qp > 64: pulses no longer fit the compact int8 stream / the vectorised CDF rows: the plain
    // path -- T-stage with coefficients kept, PVQ search, every int16 pulse to the host
    if (!e->d_coef_ws) {
        HIPCHK(hipMalloc(&e->d_coef_ws, sizeof(int32_t) * 4096 * nb * B));
        HIPCHK(hipMalloc(&e->d_y, sizeof(int16_t) * FFV2_Y_STRIDE * nb * B));
    }
    if (!e->h_y) {
        HIPCHK(hipHostMalloc(&e->h_y, sizeof(int16_t) * FFV2_Y_STRIDE * nb * B, hipHostMallocDefault));
        HIPCHK(hipHostMalloc(&e->h_codes, sizeof(uint32_t) * FFV2_CODES_PER_BP * nb * B, hipHostMallocDefault));
    }
    HIPCHK(hipMemsetAsync(e->d_status, 0, sizeof(int32_t) * nframes, s));
    FFV2TStageArgs a{};
    a.g = e->geom; a.nframes = nframes; a.frames = (const uint8_t *)d_frames;
    a.coef = e->d_coef_ws; a.energy = nullptr; a.codes = e->d_codes; a.bitcnt = e->d_bitoff; a.W = d_W;
    a.gain_thr = e->d_thr; a.gain_n = GAIN_TABLE_N; a.lds_scan = e->d_lds_scan; a.status = e->d_status;
    HIPCHK(ffv2_launch_tstage(a, s));
    HIPCHK(ffv2_launch_pvq(e->d_coef_ws, d_W, e->d_y, qp, (long long)nb * nframes, s));
    HIPCHK(hipMemcpyAsync(e->h_y, e->d_y, sizeof(int16_t) * FFV2_Y_STRIDE * nb * nframes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(e->h_codes, e->d_codes, sizeof(uint32_t) * FFV2_CODES_PER_BP * nb * nframes,
                          hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(h_status, e->d_status, sizeof(int32_t) * nframes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const bool trace = getenv("FFV2AMD_TRACE") != nullptr;
    const auto t_dev = std::chrono::steady_clock::now();
    run_parallel(nframes, nframes, [&](int f) {
        if (h_status[f] < 0) { h_sizes[f] = 0; return; }
        size_t n = 0;
        int r;
        try {
            r = encode_frame_host_qp(in, qp, e->h_codes + (size_t)f * nb * FFV2_CODES_PER_BP,
                                     e->h_y + (size_t)f * nb * FFV2_Y_STRIDE,
                                     h_packets + (size_t)f * packet_stride, packet_stride, &n);
        } catch (...) { r = FFV2AMD_ERR_NOMEM; }
        h_status[f] = r;
        h_sizes[f] = r < 0 ? 0 : (uint32_t)n;
    });
    if (trace) {
        const auto t_end = std::chrono::steady_clock::now();
        fprintf(stderr, "ffv2amd: qp=%d batch of %d: device (T-stage + PVQ + copies) %.3f ms, host range coder %.3f ms\n",
                qp, nframes, std::chrono::duration<double, std::milli>(t_dev - t_begin).count(),
                std::chrono::duration<double, std::milli>(t_end - t_dev).count());
    }
    return FFV2AMD_OK;
}


// ------------------------------------------------------------------
// Asynchronous frame ring: avcodec_send_frame / avcodec_receive_packet shape
// (reference encode.c:420,449 over ffv2enc.c:453).  `depth` frames in flight:
//   stream ring_h2d   : host frame -> device frame (DMA straight from the caller's planes when
//                       they are page-locked, else through the slot's pinned staging frame,
//                       gathered in slices by one thread per plane)
//   streams ring_comp : T-stage + E-stage of one frame, frames alternate between two streams
//                       (the tail of one frame's kernels overlaps the head of the next)
//   stream ring_d2h   : the 8 bytes {size, status}; the packet itself (size bytes, not the
//                       capacity) comes back in receive() on ring_pkt
// so that H2D(n+1) || T/E(n) || D2H(n-1).  Packets are delivered in send order.
// ------------------------------------------------------------------
void ffv2amd_ring_close(ffv2amd_encoder *e)
{
    if (!e || e->ring.empty()) return;
    (void)hipSetDevice(e->device);
    for (hipStream_t st : { e->ring_h2d, e->ring_comp[0], e->ring_comp[1], e->ring_d2h, e->ring_pkt })
        if (st) (void)hipStreamSynchronize(st);
    for (auto &r : e->ring) {
        (void)hipFree(r.d_frame); (void)hipFree(r.d_pkt); (void)hipFree(r.d_meta);
        (void)hipFree(r.d_codes); (void)hipFree(r.d_bitcnt); (void)hipFree(r.d_w); (void)hipFree(r.d_c420);
        if (r.h_frame) (void)hipHostFree(r.h_frame);
        if (r.h_pkt) (void)hipHostFree(r.h_pkt);
        if (r.h_meta) (void)hipHostFree(r.h_meta);
        if (r.ev_h2d) (void)hipEventDestroy(r.ev_h2d);
        if (r.ev_done) (void)hipEventDestroy(r.ev_done);
        if (r.ev_meta) (void)hipEventDestroy(r.ev_meta);
    }
    e->ring.clear();
    for (auto &g : e->ring_reg) (void)hipHostUnregister((void *)g.base);
    e->ring_reg.clear();
    if (e->ring_pool) { e->ring_pool->shutdown(); delete e->ring_pool; e->ring_pool = nullptr; }
    for (hipStream_t *st : { &e->ring_h2d, &e->ring_comp[0], &e->ring_comp[1], &e->ring_d2h, &e->ring_pkt })
        if (*st) { (void)hipStreamDestroy(*st); *st = nullptr; }
    e->ring_head = e->ring_count = 0;
}

int ffv2amd_ring_open(ffv2amd_encoder *e, int depth)
{
    if (!e || depth < 1 || depth > 64) return FFV2AMD_ERR_INVAL;
    if (!e->ring.empty()) return FFV2AMD_ERR_INVAL;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    const ffv2amd_info &in = e->info;
    const size_t nb = (size_t)in.block_planes;
#define RK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "ffv2amd: %s failed: %s\n", #x, hipGetErrorString(hipGetLastError())); \
    ffv2amd_ring_close(e); return FFV2AMD_ERR_DEVICE; } } while (0)
    try { e->ring.resize((size_t)depth); } catch (...) { return FFV2AMD_ERR_NOMEM; }
    RK(hipStreamCreateWithFlags(&e->ring_h2d, hipStreamNonBlocking));
    RK(hipStreamCreateWithFlags(&e->ring_comp[0], hipStreamNonBlocking));
    RK(hipStreamCreateWithFlags(&e->ring_comp[1], hipStreamNonBlocking));
    RK(hipStreamCreateWithFlags(&e->ring_d2h, hipStreamNonBlocking));
    RK(hipStreamCreateWithFlags(&e->ring_pkt, hipStreamNonBlocking));
    for (auto &r : e->ring) {
        RK(hipMalloc(&r.d_frame, in.frame_stride));
        RK(hipMalloc(&r.d_pkt, in.packet_cap));
        RK(hipMalloc(&r.d_meta, 16));
        RK(hipMemsetAsync(r.d_meta, 0, 16, e->ring_d2h));        // [2]: the T-stage's error flag, zero between calls (waited for below)
        RK(hipMalloc(&r.d_codes, sizeof(uint32_t) * FFV2_CODES_PER_BP * nb));
        RK(hipMalloc(&r.d_bitcnt, sizeof(uint32_t) * nb));
        RK(hipMalloc(&r.d_w, sizeof(int32_t) * nb));
        RK(hipHostMalloc(&r.h_frame, in.frame_stride, hipHostMallocDefault));
        RK(hipHostMalloc(&r.h_pkt, in.packet_cap, hipHostMallocDefault));
        RK(hipHostMalloc(&r.h_meta, 16, hipHostMallocDefault));
        RK(hipEventCreateWithFlags(&r.ev_h2d, hipEventDisableTiming));
        RK(hipEventCreateWithFlags(&r.ev_done, hipEventDisableTiming));
        RK(hipEventCreateWithFlags(&r.ev_meta, hipEventDisableTiming));
    }
    // the fills above have to be through before the first frame: a plain hipMemset() of device memory returns before
    // the fill has run (null stream; the ring's streams are non-blocking), and one that landed after the first
    // E-stage had written {size, status} there made ring_receive see a packet of 0 bytes (once in some 5 000 first
    // frames of a ring, and only once 16 hardware queues had taken the accidental ordering away)
    RK(hipStreamSynchronize(e->ring_d2h));
#undef RK
    e->ring_head = e->ring_count = 0;
    // helpers for pageable frames: a 4K plane is 16 MB of row copies, PCIe moves 55 GB/s, one host
    // thread ~12 GB/s -- five besides the caller, fewer on a small machine, none for small pictures
    if (e->info.frame_stride >= ((size_t)8 << 20)) {
        unsigned hw = std::thread::hardware_concurrency();
        static const int want = getenv("FFV2AMD_GATHER_THREADS") ? atoi(getenv("FFV2AMD_GATHER_THREADS")) : 6;   // caller included
        const int helpers = (int)hw >= want ? want - 1 : (hw > 1 ? (int)hw - 1 : 0);
        e->ring_pool = new (std::nothrow) GatherPool;
        if (e->ring_pool && helpers > 0) e->ring_pool->start(helpers, e->device);
    }
    return FFV2AMD_OK;
}

int ffv2amd_ring_pending(const ffv2amd_encoder *e) { return e ? e->ring_count : 0; }

// FFV2AMD_FRAME_REGISTER: is [src, src + need) page-locked by this encoder -- on first sight of a buffer it becomes so
// (hipHostRegister: milliseconds, once per buffer of the caller's pool), afterwards the cache answers.
static bool host_range_registered(ffv2amd_encoder *e, const uint8_t *src, size_t need)
{
    for (const auto &g : e->ring_reg)
        if (g.base <= src && src + need <= g.base + g.bytes) return true;
    if (e->ring_reg.size() >= 256) return false;
    // Small planes are not worth it: they are copied in microseconds, and page-locking them means page-locking bits of
    // the C library's heap (allocations below its mmap threshold), pages they share with whatever else lives there.
    if (need < ((size_t)256 << 10)) return false;
    if (hipHostRegister((void *)src, need, hipHostRegisterDefault) == hipSuccess) {
        try { e->ring_reg.push_back({ src, need }); return true; }
        catch (...) { (void)hipHostUnregister((void *)src); return false; }
    }
    (void)hipGetLastError();                  // not registrable (already part of another registration, ...)
    return false;
}

// One plane of a frame on its way into a ring slot: `rows` rows of `row_bytes` bytes from the caller's
// (src, linesize) to device memory (d_dst, rows pitch bytes apart); `stage` is the slot's page-locked
// copy of it, used when the caller's memory is pageable.
struct RingPlane {
    const uint8_t *src;
    ptrdiff_t linesize;
    size_t row_bytes, pitch;
    int rows, slices;
    uint8_t *d_dst, *stage;
};

// H2D of the planes, then T-stage + E-stage on a compute stream (4:2:0: the chroma up-conversion in
// front of them), then the {size, status} D2H: the body of ring_send / ring_send_420.
static int ring_submit(ffv2amd_encoder *e, ffv2amd_encoder::RingSlot &r, const RingPlane *pl, int npl, bool chroma420,
                       const int32_t *W, int64_t tag, unsigned flags)
{
    const ffv2amd_info &in = e->info;
    hipStream_t sh = e->ring_h2d;
    // Which planes the DMA engine may read in place: all of them when the caller says they are page-locked; with
    // FFV2AMD_FRAME_REGISTER those whose memory this ring has page-locked itself -- on first sight of a buffer
    // (hipHostRegister: milliseconds, once per buffer of the caller's pool), from the cache afterwards.
    bool direct[4] = { false, false, false, false };
    for (int i = 0; i < npl; i++) {
        const RingPlane &q = pl[i];
        if (flags & FFV2AMD_FRAME_PINNED) { direct[i] = true; continue; }
        if (!(flags & FFV2AMD_FRAME_REGISTER) || q.linesize <= 0) continue;
        const size_t need = (size_t)q.linesize * (size_t)(q.rows - 1) + q.row_bytes;
        direct[i] = host_range_registered(e, q.src, need);       // false: gather it
    }
    int ngather = 0;
    for (int i = 0; i < npl; i++) {
        const RingPlane &q = pl[i];
        if (!direct[i]) { ngather++; continue; }
        if (q.linesize == (ptrdiff_t)q.pitch) {
            // planes of a page-locked frame that follow each other without a gap on both sides travel as one copy:
            // an API call and a DMA descriptor less per plane (1080p: 14.0 -> 16.0 Gpix/s 4:4:4, 21.5 -> 23.5 4:2:0).
            // Not planes the ring registered itself: a copy may not cross two registrations.
            size_t bytes = q.pitch * (size_t)(q.rows - 1) + q.row_bytes;
            int j = i;
            while ((flags & FFV2AMD_FRAME_PINNED) && j + 1 < npl &&
                   pl[j + 1].linesize == (ptrdiff_t)pl[j + 1].pitch &&
                   pl[j + 1].src == pl[j].src + pl[j].pitch * (size_t)pl[j].rows &&
                   pl[j + 1].d_dst == pl[j].d_dst + pl[j].pitch * (size_t)pl[j].rows) {
                j++;
                bytes = (size_t)(pl[j].src - q.src) + pl[j].pitch * (size_t)(pl[j].rows - 1) + pl[j].row_bytes;
            }
            HIPCHK(hipMemcpyAsync(q.d_dst, q.src, bytes, hipMemcpyHostToDevice, sh));
            i = j;
        } else
            HIPCHK(hipMemcpy2DAsync(q.d_dst, q.pitch, q.src, (size_t)q.linesize, q.row_bytes, (size_t)q.rows,
                                    hipMemcpyHostToDevice, sh));
    }
    if (ngather) {
        // pageable planes: gather into the slot's pinned frame in slices of rows; every slice's DMA
        // is issued as soon as it is gathered (by the pool's threads when the picture is large)
        int first[5] = { 0, 0, 0, 0, 0 };
        for (int i = 0; i < npl; i++) first[i + 1] = first[i] + (direct[i] ? 0 : e->ring_pool ? pl[i].slices : 1);
        const int nsl = first[npl];
        hipError_t up[32];
        for (int i = 0; i < nsl; i++) up[i] = hipSuccess;
        const std::function<void(int)> slice = [&](int i) {
            int p = 0;
            while (i >= first[p + 1]) p++;
            const RingPlane &q = pl[p];
            const int k = i - first[p], per = first[p + 1] - first[p];
            const int y0 = (int)((long long)q.rows * k / per), y1 = (int)((long long)q.rows * (k + 1) / per);
            if (y1 <= y0) return;
            for (int y = y0; y < y1; y++)
                memcpy(q.stage + (size_t)y * q.pitch, q.src + (ptrdiff_t)y * q.linesize, q.row_bytes);
            up[i] = hipMemcpyAsync(q.d_dst + (size_t)y0 * q.pitch, q.stage + (size_t)y0 * q.pitch, (size_t)(y1 - y0) * q.pitch,
                                   hipMemcpyHostToDevice, sh);
        };
        if (e->ring_pool) {
            try { e->ring_pool->run(nsl, slice); } catch (...) { return FFV2AMD_ERR_NOMEM; }
        } else {
            for (int i = 0; i < nsl; i++) slice(i);
        }
        for (int i = 0; i < nsl; i++) HIPCHK(up[i]);
    }
    const int32_t *dW = nullptr;
    r.has_w = W != nullptr;
    if (W) {
        HIPCHK(hipMemcpyAsync(r.d_w, W, sizeof(int32_t) * in.block_planes, hipMemcpyHostToDevice, sh));
        dW = r.d_w;
    }
    HIPCHK(hipEventRecord(r.ev_h2d, sh));
    hipStream_t sc = e->ring_comp[e->ring_seq++ & 1u];
    HIPCHK(hipStreamWaitEvent(sc, r.ev_h2d, 0));
    if (chroma420)
        HIPCHK(ffv2_launch_upconv_chroma(e->upconv, e->geom, 1, r.d_c420, pl[1].pitch, pl[1].pitch * (size_t)pl[1].rows, 0,
                                         r.d_frame, sc));
    int rc = launch_encode_qp0(e, 1, r.d_frame, dW, r.d_pkt, in.packet_cap, r.d_meta, (int32_t *)(r.d_meta + 1),
                               r.d_codes, r.d_bitcnt, nullptr, sc, sc, nullptr, (int32_t *)(r.d_meta + 2));
    if (rc < 0) return rc;
    HIPCHK(hipEventRecord(r.ev_done, sc));
    HIPCHK(hipStreamWaitEvent(e->ring_d2h, r.ev_done, 0));
    HIPCHK(hipMemcpyAsync(r.h_meta, r.d_meta, 8, hipMemcpyDeviceToHost, e->ring_d2h));
    HIPCHK(hipEventRecord(r.ev_meta, e->ring_d2h));
    r.tag = tag;
    e->ring_count++;
    return FFV2AMD_OK;
}

int ffv2amd_ring_send(ffv2amd_encoder *e, const uint8_t *const data[4], const ptrdiff_t linesize[4],
                      const int32_t *W, int64_t tag, unsigned flags)
{
    if (!e || !data || !linesize || e->ring.empty()) return FFV2AMD_ERR_INVAL;
    const ffv2amd_info &in = e->info;
    for (int p = 0; p < in.planes; p++)
        if (!data[p]) return FFV2AMD_ERR_INVAL;
    if (e->ring_count == (int)e->ring.size()) return FFV2AMD_ERR_AGAIN;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    auto &r = e->ring[(size_t)((e->ring_head + e->ring_count) % (int)e->ring.size())];
    RingPlane pl[4];
    for (int p = 0; p < in.planes; p++)
        pl[p] = RingPlane{ data[p], linesize[p], (size_t)in.width * (in.depth > 8 ? 2 : 1), in.row_pitch, in.height, 4,
                           r.d_frame + (size_t)p * in.plane_stride, r.h_frame + (size_t)p * in.plane_stride };
    return ring_submit(e, r, pl, in.planes, false, W, tag, flags);
}

// 4:2:0 frames through the ring (SURVEY.md 8(f) rank 4 at the asynchronous boundary): half the PCIe
// bytes of the 4:4:4 form.  Luma goes straight into plane 0 of the slot's device frame, the two
// chroma planes into a small staging area, and the up-conversion (ffv2_upconv.hip) runs on the
// frame's compute stream in front of its T-stage.
int ffv2amd_ring_send_420(ffv2amd_encoder *e, const uint8_t *const data[3], const ptrdiff_t linesize[3],
                          const int32_t *W, int64_t tag, unsigned flags)
{
    if (!e || !data || !linesize || e->ring.empty() || !data[0] || !data[1] || !data[2]) return FFV2AMD_ERR_INVAL;
    if (e->ring_count == (int)e->ring.size()) return FFV2AMD_ERR_AGAIN;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    int rc = upconv_ready(e);
    if (rc < 0) return rc;
    const ffv2amd_info &in = e->info;
    const size_t bps = in.depth > 8 ? 2 : 1;
    const int cw = (in.width + 1) >> 1, ch = (in.height + 1) >> 1;
    const size_t c_pitch = align_up((size_t)cw * bps, 128);
    auto &r = e->ring[(size_t)((e->ring_head + e->ring_count) % (int)e->ring.size())];
    if (!r.d_c420) HIPCHK(hipMalloc(&r.d_c420, 2 * c_pitch * (size_t)ch));
    // pinned staging of a pageable frame: luma in plane 0 of the slot's host frame, U and V behind it
    // ((h + 1) * c_pitch <= 2 * h * row_pitch: they fit planes 1 and 2)
    uint8_t *cst = r.h_frame + in.plane_stride;
    RingPlane pl[3] = {
        RingPlane{ data[0], linesize[0], (size_t)in.width * bps, in.row_pitch, in.height, 4, r.d_frame, r.h_frame },
        RingPlane{ data[1], linesize[1], (size_t)cw * bps, c_pitch, ch, 1, r.d_c420, cst },
        RingPlane{ data[2], linesize[2], (size_t)cw * bps, c_pitch, ch, 1, r.d_c420 + c_pitch * (size_t)ch, cst + c_pitch * (size_t)ch },
    };
    return ring_submit(e, r, pl, 3, true, W, tag, flags);
}

int ffv2amd_ring_receive(ffv2amd_encoder *e, uint8_t *out, size_t out_cap, size_t *out_size, int64_t *tag, int wait)
{
    if (!e || !out || !out_size || e->ring.empty()) return FFV2AMD_ERR_INVAL;
    if (e->ring_count == 0) return FFV2AMD_ERR_AGAIN;
    DeviceGuard guard(e->device);
    if (!guard.ok) return FFV2AMD_ERR_DEVICE;
    auto &r = e->ring[(size_t)e->ring_head];
    if (!wait) {
        const hipError_t q = hipEventQuery(r.ev_meta);
        if (q == hipErrorNotReady) return FFV2AMD_ERR_AGAIN;
        HIPCHK(q);
    } else {
        HIPCHK(hipEventSynchronize(r.ev_meta));
    }
    // the oldest frame is finished: from here on it leaves the ring whatever happens
    e->ring_head = (e->ring_head + 1) % (int)e->ring.size();
    e->ring_count--;
    if (tag) *tag = r.tag;
    const int32_t st = (int32_t)r.h_meta[1];
    if (st == FFV2AMD_ERR_RANGE)       // the slot's device frame is intact until the next send: rerun it wide
        return wide_encode_frame(e, r.d_frame, r.has_w ? r.d_w : nullptr, out, out_cap, out_size);
    if (st < 0) return st;
    const size_t n = r.h_meta[0];
    if (n == 0) return FFV2AMD_ERR_DEVICE;
    if (n > out_cap) return FFV2AMD_ERR_NOSPACE;
    HIPCHK(hipMemcpyAsync(r.h_pkt, r.d_pkt, n, hipMemcpyDeviceToHost, e->ring_pkt));
    HIPCHK(hipStreamSynchronize(e->ring_pkt));
    memcpy(out, r.h_pkt, n);
    *out_size = n;
    return FFV2AMD_OK;
}

void *ffv2amd_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void ffv2amd_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

}  // extern "C"
