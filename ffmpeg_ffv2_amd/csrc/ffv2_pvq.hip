// ffv2_pvq.hip -- Q-stage for qp > 0: per-band gain normalisation and the greedy
// pyramid-VQ pulse search (reference libavcodec/ffv2enc.c:163-171 calling
// ff_pvq_search_exact_avx, libavcodec/x86/celt_pvq_search.asm:85-191,214-368,
// INIT_XMM avx, USE_APPROXIMATION 0; horizontal sums libavutil/x86/x86util.asm:968-977).
//
// One wavefront per 64x64 block-plane walks its 13 bands.  Element i of a band
// lives in lane i & 63; the asm's four XMM lanes are the classes i & 3.
// What has to be reproduced for bit-exact pulses:
//   * the three float sums (|x|, |x|*y, y*y) are accumulated per class from the LAST
//     4-vector down to the first, then combined as (c0+c2)+(c1+c3): float addition is
//     not associative, so those chains run sequentially (one lane per chain, LDS staging);
//   * no fused multiply-add anywhere (built with -ffp-contract=off), IEEE divide, RNE rint;
//   * per pulse: p = (|x|+Sxy)^2 / (y+Syy) per element; within a class the FIRST index
//     with the strictly largest p wins; classes 2,3 beat 0,1 only when strictly larger;
//     class 1 beats class 0 unless p1 < p0; the padding lanes of the last 4-vector take part.
// PARITY UNPINNED with respect to the reference binary (no assembler here, no reference
// vectors): pinned only against oracle/ffv2_oracle.c::ffv2o_pvq_search, which restates
// the same asm independently.
#include "ffv2_kernels.h"
#include <cstdlib>

namespace {

constexpr int PVQ_MAXN4 = 2052;               // band 12: 2049 coefficients -> 513 4-vectors

struct PvqLds {
    float s[PVQ_MAXN4];                       // staging for the sequential sums (8 KB -> 4 wavefronts per SIMD)
};

// sum of class `cls` from the last 4-vector down to vector 0 (celt_pvq_search.asm:236-251)
__device__ __forceinline__ float chain_desc(const float *a, int nv, int cls)
{
    float s = a[(nv - 1) * 4 + cls];
    for (int v = nv - 2; v >= 0; v--) s = __fadd_rn(s, a[v * 4 + cls]);
    return s;
}

__device__ __forceinline__ float hsum4(float c0, float c1, float c2, float c3)      // x86util.asm:968-977
{
    return __fadd_rn(__fadd_rn(c0, c2), __fadd_rn(c1, c3));
}

template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}

// over the 16 lanes of a class (lane & 3): every lane ends up with its class's value
__device__ __forceinline__ int class_max_i32(int v)
{
    int t = dpp_mov<0x124>(v); v = t > v ? t : v;              // row_ror:4
    t = dpp_mov<0x128>(v); v = t > v ? t : v;                  // row_ror:8
    t = __shfl_xor(v, 16, 64); v = t > v ? t : v;
    t = __shfl_xor(v, 32, 64); v = t > v ? t : v;
    return v;
}

__device__ __forceinline__ int class_min_i32(int v)
{
    int t = dpp_mov<0x124>(v); v = t < v ? t : v;
    t = dpp_mov<0x128>(v); v = t < v ? t : v;
    t = __shfl_xor(v, 16, 64); v = t < v ? t : v;
    t = __shfl_xor(v, 32, 64); v = t < v ? t : v;
    return v;
}

// What the range coder will do with a band (ffv2enc.c:175-186): it reads pulses until their magnitudes add
// up to qp (`stop` symbols, all N if they never do), appends a sign bit for each non-zero one (`nz`), and
// asserts when a magnitude reaches the alphabet size (daala_entropy.c:336; `big`).  Wave-uniform.
struct PvqBandCount { int stop, nz; bool big; };

// The same from the pulses in memory, 64 at a time: for the case the search cannot produce (more than K
// pulses in a band), kept so that whatever a band holds is counted as the coder would read it.
__device__ __noinline__ PvqBandCount pvq_count_from_memory(const int16_t *yy, int N, int K, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    PvqBandCount r{ N, 0, false };
    int run = 0;
    for (int j0 = 0; j0 < N && r.stop == N; j0 += 64) {
        const int j = j0 + lane;
        int a = j < N ? yy[j] : 0;
        a = a < 0 ? -a : a;
        int incl = a;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        const unsigned long long hit = __ballot(run + incl >= K);
        int end = N;
        if (hit) { r.stop = j0 + __ffsll((long long)hit); end = r.stop; }
        r.nz += (int)__popcll(__ballot(j < end && a > 0));
        r.big = r.big || __ballot(j < end && a >= K) != 0;
        run += __shfl(incl, 63, 64);
    }
    if (r.stop > N) r.stop = N;
    return r;
}

// M = elements per lane (ceil(N4 / 64)).  x[m] = normalised coefficient of element
// i = lane + 64 m (0 beyond N).  Writes y[i] for i < N.
template <int M>
__device__ PvqBandCount pvq_search_wave(const float (&x)[M], int N, int K, PvqLds &L, int lane, int16_t *yout)
{
    const int nv = (N + 3) >> 2, N4 = nv * 4;
    float ax[M], fy[M];
    unsigned long long neg = 0;                               // sign bits of x: all that is needed of it at the end
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int i = lane + 64 * m;
        ax[m] = i < N ? fabsf(x[m]) : 0.0f;
        neg |= (unsigned long long)(signbit(x[m]) ? 1 : 0) << m;
        if (i < N4) L.s[i] = ax[m];
    }
    __syncthreads();
    float c = lane < 4 ? chain_desc(L.s, nv, lane) : 0.0f;
    const float Sx = hsum4(__shfl(c, 0, 64), __shfl(c, 1, 64), __shfl(c, 2, 64), __shfl(c, 3, 64));
    __syncthreads();
    if (Sx == 0.0f || Sx != Sx) {                             // comiss + jz: zero (or unordered) input
#pragma unroll
        for (int m = 0; m < M; m++) {
            const int i = lane + 64 * m;
            if (i < N) yout[i] = 0;
        }
        return PvqBandCount{ N, 0, false };
    }
    const float b = __fdiv_rn((float)K, Sx);
    int sy = 0;
    bool anyy = false;
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int yt = __float2int_rn(__fmul_rn(b, ax[m]));   // cvtps2dq: round to nearest even
        fy[m] = (float)yt;
        sy += yt;
        anyy = anyy || yt != 0;
    }
    float Sxy = 0.0f, Syy = 0.0f;
    // Sxy and Syy are sums in the asm's order; when the projection leaves every y at zero (flat
    // spectra: K pulses spread over N >> K coefficients) every term is +0 and so are the sums
    if (__ballot(anyy)) {
#pragma unroll
        for (int m = 0; m < M; m++) {
            const int i = lane + 64 * m;
            if (i < N4) L.s[i] = __fmul_rn(ax[m], fy[m]);
        }
        __syncthreads();
        c = lane < 4 ? chain_desc(L.s, nv, lane) : 0.0f;
        Sxy = hsum4(__shfl(c, 0, 64), __shfl(c, 1, 64), __shfl(c, 2, 64), __shfl(c, 3, 64));
        __syncthreads();
#pragma unroll
        for (int m = 0; m < M; m++) {
            const int i = lane + 64 * m;
            if (i < N4) L.s[i] = __fmul_rn(fy[m], fy[m]);
        }
        __syncthreads();
        c = lane < 4 ? chain_desc(L.s, nv, lane) : 0.0f;
        Syy = hsum4(__shfl(c, 0, 64), __shfl(c, 1, 64), __shfl(c, 2, 64), __shfl(c, 3, 64));
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sy += __shfl_xor(sy, o, 64);   // integer: any order
    int Kr = K - sy;
    if (Kr != 0) {
        const bool add = Kr > 0;
        // every non-zero |x| within [2^-40, 2^40] (normalised coefficients are: 1 / (45 * 2^22) <= |x| <= 1):
        // the shared-denominator division below then never meets a scaled or special operand
        bool wild = false;
#pragma unroll
        for (int m = 0; m < M; m++) wild = wild || (ax[m] != 0.0f && !(ax[m] >= 0x1p-40f && ax[m] <= 0x1p40f));
        const bool tame = __ballot(wild) == 0 && K <= 4096;
        Syy = __fmul_rn(Syy, 0.5f);
        for (int it = add ? Kr : -Kr; it > 0; it--) {
            Syy = __fadd_rn(Syy, 0.5f);
            // lanes start from their own first element with p = 0: if nothing is strictly
            // better the class winner is its lowest index (the asm's initial max_idx)
            float bp = lane < N4 ? 0.0f : -1.0f;
            int bi = lane < N4 ? lane : 0x7fffffff;
            if (add && tame) {
                // Elements without a pulse share the denominator 0 + Syy = Syy.  The IEEE division is
                // the sequence rcp, two FMAs refining it, q0 = n r, and three FMA corrections of q;
                // v_div_scale / v_div_fmas / v_div_fixup only act on operands near the ends of the
                // exponent range, which `tame` excludes (0.5 <= Syy <= 2 K^2; the numerator
                // (|x| + Sxy)^2 is 0 or within [2^-80, 2^90]).  The first three steps depend on the
                // denominator alone and are done once per pulse; a slot (64 elements) in which some
                // lane's element carries a pulse takes the full division.
                const float r0 = __builtin_amdgcn_rcpf(Syy);
                const float r1 = __fmaf_rn(__fmaf_rn(-Syy, r0, 1.0f), r0, r0);
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const int i = lane + 64 * m;
                    const float num = __fadd_rn(ax[m], Sxy);
                    const float n2 = __fmul_rn(num, num);
                    float pp;
                    if (__ballot(fy[m] != 0.0f) == 0) {
                        const float q0 = __fmul_rn(n2, r1);
                        const float q1 = __fmaf_rn(__fmaf_rn(-Syy, q0, n2), r1, q0);
                        pp = __fmaf_rn(__fmaf_rn(-Syy, q1, n2), r1, q1);
                    } else {
                        pp = __fdiv_rn(n2, __fadd_rn(fy[m], Syy));
                    }
                    if (i < N4 && bp < pp) { bp = pp; bi = i; }
                }
            } else {
#pragma unroll
                for (int m = 0; m < M; m++) {
                    const int i = lane + 64 * m;
                    float num, den;
                    if (add) {
                        den = __fadd_rn(fy[m], Syy);
                        num = __fadd_rn(ax[m], Sxy);
                    } else {
                        den = __fsub_rn(Syy, fy[m]);
                        num = (0.0f < fy[m]) ? __fsub_rn(Sxy, ax[m]) : 0.0f;
                    }
                    const float pp = __fdiv_rn(__fmul_rn(num, num), den);
                    if (i < N4 && bp < pp) { bp = pp; bi = i; }
                }
            }
            int best;
            if (add && tame) {
                // p >= 0 here (-1 beyond N4), ordered like its bit pattern: the class's largest p by an integer maximum,
                // the lowest index that has it by a minimum, the four classes merged on scalars
                const int pm = class_max_i32(__float_as_int(bp));
                const int im = class_min_i32(__float_as_int(bp) == pm && bp >= 0.0f ? bi : 0x7fffffff);
                float q0 = __int_as_float(__builtin_amdgcn_readlane(pm, 0)), q1 = __int_as_float(__builtin_amdgcn_readlane(pm, 1));
                const float q2 = __int_as_float(__builtin_amdgcn_readlane(pm, 2)), q3 = __int_as_float(__builtin_amdgcn_readlane(pm, 3));
                int i0 = __builtin_amdgcn_readlane(im, 0), i1 = __builtin_amdgcn_readlane(im, 1);
                const int i2 = __builtin_amdgcn_readlane(im, 2), i3 = __builtin_amdgcn_readlane(im, 3);
                if (q0 < q2) { q0 = q2; i0 = i2; }            // classes (3,2) replace (1,0) only when strictly greater
                if (q1 < q3) { q1 = q3; i1 = i3; }
                best = !(q1 < q0) ? i1 : i0;                  // class 1 replaces class 0 unless p1 < p0 (cmpss predicate 5 = NLT)
            } else {
                // same class across lanes (lane ^ 4, 8, 16, 32): larger p, then lower index
#pragma unroll
                for (int o = 4; o <= 32; o <<= 1) {
                    const float op = __shfl_xor(bp, o, 64);
                    const int oi = __shfl_xor(bi, o, 64);
                    if (op > bp || (op == bp && oi < bi)) { bp = op; bi = oi; }
                }
                {   // classes (3,2) replace (1,0) only when strictly greater
                    const float op = __shfl_xor(bp, 2, 64);
                    const int oi = __shfl_xor(bi, 2, 64);
                    if ((lane & 2) == 0 && bp < op) { bp = op; bi = oi; }
                }
                // class 1 replaces class 0 unless p1 < p0 (cmpss predicate 5 = NLT)
                const float p1 = __shfl(bp, 1, 64), p0 = __shfl(bp, 0, 64);
                const int i1 = __shfl(bi, 1, 64), i0 = __shfl(bi, 0, 64);
                best = !(p1 < p0) ? i1 : i0;
            }
            // the winner's |x| and pulse count live in its owner's registers: the owner (lane
            // best & 63) picks them out, one lane read hands them round -- no LDS, no barrier
            float axb = 0.0f, fyb = 0.0f;
#pragma unroll
            for (int m = 0; m < M; m++)
                if (lane + 64 * m == best) { axb = ax[m]; fyb = fy[m]; }
            axb = __shfl(axb, best & 63, 64);
            fyb = __shfl(fyb, best & 63, 64);
            if (add) { Sxy = __fadd_rn(Sxy, axb); Syy = __fadd_rn(Syy, fyb); }
            else     { Sxy = __fsub_rn(Sxy, axb); Syy = __fsub_rn(Syy, fyb); }
            const float nf = add ? __fadd_rn(fyb, 1.0f) : __fsub_rn(fyb, 1.0f);
#pragma unroll
            for (int m = 0; m < M; m++)
                if (lane + 64 * m == best) fy[m] = nf;
        }
    }
    // per lane only the sum; non-zero pulses, the last of them and oversized ones through ballots (scalar work)
    int tot = 0, nzc = 0, last = 0;
    unsigned long long bigm = 0;
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int i = lane + 64 * m;
        const bool in = i < N;
        const int iv = __float2int_rn(fy[m]);
        if (in) yout[i] = (int16_t)(((neg >> m) & 1ull) ? -iv : iv);    // orps sign, cvtps2dq
        const int a = in ? (iv < 0 ? -iv : iv) : 0;
        tot += a;
        const unsigned long long nzm = __ballot(a != 0);
        nzc += (int)__popcll(nzm);
        if (nzm) last = 64 * m + 64 - __clzll((long long)nzm);     // m ascends: the highest index so far
        bigm |= __ballot(a >= K);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o, 64);
    // The search leaves exactly K pulses (or none at all), so the coder stops behind the last non-zero one
    // or never; more than K can only come out of the removal branch picking a pulse-free element, which it
    // does not do as long as any p is positive -- counted from memory then, as the coder would.
    if (tot > K) return pvq_count_from_memory(yout, N, K, lane);
    return PvqBandCount{ tot == K ? last : N, nzc, bigm != 0 };
}

// ---------------------------------------------------------------------------------------------
// The same search for the large bands, by lists (round 3).  Result identical to pvq_search_wave; what
// differs is the work per pulse.  The elements of a band are of two kinds:
//   * carriers (a pulse already): at most 2 K of them ever (the projection gives a pulse only where
//     b |x| > 1/2 and the b |x| add up to K).  They live in a list, one entry per lane, each in a lane of
//     its own class (i & 3) so that the class-wise reduction network applies to them as it stands; their
//     p = (|x| + Sxy)^2 / (y + Syy) costs one division per lane and pulse whatever the band's size.
//   * pulse-free elements: they share the denominator Syy, and p = fl(fl(n n) / Syy) with n = fl(|x| + Sxy)
//     is a monotone function of |x| (every step is a correctly rounded monotone operation on
//     non-negative numbers).  So a class's largest p among them is p(A), A its largest |x|, and the
//     asm's winner -- the FIRST index whose p equals it -- is the lowest index with n >= n_lo, where
//     n_lo is the lowest float whose p still equals p(A).  n_lo is found by evaluating p on the floats
//     just below n(A), one per lane (two consecutive floats almost never share a p: squaring doubles
//     their relative distance), and the elements are then only compared with it: an add, a compare and
//     two selects per element instead of the division.
// Removal (the projection overshot K): only carriers can lose a pulse; every other element has p = +0
// and loses to the class's first element, the asm's initial maximum.
// Exact for what ffv2_pvq_kernel feeds it (0 or 2^-40 <= |x| <= 2^40, K <= 64); the test hook routes
// anything else to pvq_search_wave.
// ---------------------------------------------------------------------------------------------
// the list: up to PVQ_LIST_E entries per lane, in the staging buffer of the sums once they are done
constexpr int PVQ_LIST_E = 8;                  // 16 entries per class and row: 128 per class >= 2 K for K <= 64
struct PvqList {
    int idx[PVQ_LIST_E][64];                   // -1: empty
    float x[PVQ_LIST_E][64];                   // with its sign
    float fy[PVQ_LIST_E][64];                  // pulses so far
};
static_assert(sizeof(PvqList) <= sizeof(PvqLds), "the list lives in the staging buffer");

#define PVQ_SLOTS(C) C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) \
    C(17) C(18) C(19) C(20) C(21) C(22) C(23) C(24) C(25) C(26) C(27) C(28) C(29) C(30) C(31) C(32)

// af[slot] of lane `owner` := -inf / its value, slot and owner wave-uniform: one select / lane read behind a
// jump on the slot, the array stays in registers (an indexed access would move it to scratch, a chain of
// selects costs one per slot)
template <int M>
__device__ __forceinline__ void take_out(float (&af)[M], int slot, int owner, int lane)
{
    switch (slot) {
#define C(m) case m: if constexpr (m < M) af[m < M ? m : 0] = lane == owner ? -__builtin_inff() : af[m < M ? m : 0]; break;
        PVQ_SLOTS(C)
#undef C
    default: break;
    }
}

template <int M>
__device__ __forceinline__ float look_up(const float (&af)[M], int slot, int owner)
{
    int v = 0;
    switch (slot) {
#define C(m) case m: if constexpr (m < M) v = __builtin_amdgcn_readlane(__float_as_int(af[m < M ? m : 0]), owner); break;
        PVQ_SLOTS(C)
#undef C
    default: break;
    }
    return __int_as_float(v);
}

template <int M>
__device__ PvqBandCount pvq_search_lists(const float (&x)[M], int N, int K, PvqLds &L, int lane, int16_t *yout)
{
    static_assert(M <= 33, "slots");
    const int nv = (N + 3) >> 2, N4 = nv * 4;
    const int cls = lane & 3;
    const float NONE = -__builtin_inff();
    PvqList &Q = *reinterpret_cast<PvqList *>(&L);
    float af[M];                                              // |x| of an element without a pulse; NONE: it carries one, or lies beyond N4
    unsigned long long neg = 0;
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int i = lane + 64 * m;
        const float a = i < N ? fabsf(x[m]) : 0.0f;
        neg |= (unsigned long long)(signbit(x[m]) ? 1 : 0) << m;
        if (i < N4) L.s[i] = a;
        af[m] = i < N4 ? a : NONE;
    }
    __syncthreads();
    float c = lane < 4 ? chain_desc(L.s, nv, lane) : 0.0f;
    const float Sx = hsum4(__shfl(c, 0, 64), __shfl(c, 1, 64), __shfl(c, 2, 64), __shfl(c, 3, 64));
    __syncthreads();
    if (Sx == 0.0f || Sx != Sx) {
#pragma unroll
        for (int m = 0; m < M; m++) {
            const int i = lane + 64 * m;
            if (i < N) yout[i] = 0;
        }
        return PvqBandCount{ N, 0, false };
    }
    const float b = __fdiv_rn((float)K, Sx);

    // projection (celt_pvq_search.asm:253-285): y = rint(b |x|)
    int yt[M];
    int sy = 0;
    bool anyy = false;
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int i = lane + 64 * m;
        const float a = i < N ? fabsf(x[m]) : 0.0f;
        yt[m] = __float2int_rn(__fmul_rn(b, a));
        sy += yt[m];
        anyy = anyy || yt[m] != 0;
    }
    float Sxy = 0.0f, Syy = 0.0f;
    const bool projected = __ballot(anyy) != 0;
    if (projected) {
#pragma unroll
        for (int m = 0; m < M; m++) {
            const int i = lane + 64 * m;
            const float a = i < N ? fabsf(x[m]) : 0.0f;
            if (i < N4) L.s[i] = __fmul_rn(a, (float)yt[m]);
        }
        __syncthreads();
        c = lane < 4 ? chain_desc(L.s, nv, lane) : 0.0f;
        Sxy = hsum4(__shfl(c, 0, 64), __shfl(c, 1, 64), __shfl(c, 2, 64), __shfl(c, 3, 64));
        __syncthreads();
#pragma unroll
        for (int m = 0; m < M; m++) {
            const int i = lane + 64 * m;
            if (i < N4) L.s[i] = __fmul_rn((float)yt[m], (float)yt[m]);
        }
        __syncthreads();
        c = lane < 4 ? chain_desc(L.s, nv, lane) : 0.0f;
        Syy = hsum4(__shfl(c, 0, 64), __shfl(c, 1, 64), __shfl(c, 2, 64), __shfl(c, 3, 64));
        __syncthreads();
    }
    // the staging buffer becomes the list
#pragma unroll
    for (int k = 0; k < PVQ_LIST_E; k++) Q.idx[k][lane] = -1;
    uint32_t cnt = 0;                                         // entries in use per class, a byte each
    bool overflow = false;
    auto insert = [&](int idx, float xs, float fyv) {         // wave-uniform arguments
        const int cl = idx & 3;
        const uint32_t pos = (cnt >> (8 * cl)) & 0xffu;
        if (pos >= 16u * PVQ_LIST_E) { overflow = true; return; }
        cnt += 1u << (8 * cl);
        const int e = (int)(pos >> 4), tl = 4 * (int)(pos & 15u) + cl;
        if (lane == tl) { Q.idx[e][tl] = idx; Q.x[e][tl] = xs; Q.fy[e][tl] = fyv; }
    };
    if (projected) {                                          // whoever got a pulse goes on the list
#pragma unroll
        for (int m = 0; m < M; m++) {
            unsigned long long nzm = __ballot(yt[m] != 0);
            if (yt[m] != 0) af[m] = NONE;
            while (nzm) {
                const int l = __ffsll((long long)nzm) - 1;
                nzm &= nzm - 1;
                insert(l + 64 * m, __shfl(x[m], l, 64), (float)__shfl(yt[m], l, 64));
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sy += __shfl_xor(sy, o, 64);
    const int Kr = __builtin_amdgcn_readfirstlane(K - sy);
    auto rows_in_use = [&]() {
        const uint32_t u01 = (cnt & 0xffu) > ((cnt >> 8) & 0xffu) ? (cnt & 0xffu) : ((cnt >> 8) & 0xffu);
        const uint32_t u23 = ((cnt >> 16) & 0xffu) > (cnt >> 24) ? ((cnt >> 16) & 0xffu) : (cnt >> 24);
        return (int)(((u01 > u23 ? u01 : u23) + 15u) >> 4);
    };
    if (Kr != 0) {
        const bool add = Kr > 0;
        Syy = __fmul_rn(Syy, 0.5f);
        bool dirty = true;                                    // the classes' largest pulse-free |x| have to be (re)computed
        float Acls = NONE;
        for (int it = add ? Kr : -Kr; it > 0; it--) {
            Syy = __fadd_rn(Syy, 0.5f);
            const int used_e = rows_in_use();
            int best;
            float axb = 0.0f, fyb = 0.0f;
            if (add) {
                if (dirty) {
                    int lm = __float_as_int(NONE);
#pragma unroll
                    for (int m = 0; m < M; m++) {
                        const int v = __float_as_int(af[m]);  // |x| >= 0 or -inf: ordered like their bit patterns
                        lm = v > lm ? v : lm;
                    }
                    Acls = __int_as_float(class_max_i32(lm));
                    dirty = false;
                }
                // p of the class's best pulse-free element, and the lowest numerator that still reaches it
                const bool hasF = Acls >= 0.0f;
                const float nA = __fadd_rn(Acls, Sxy);
                const int nAb = __float_as_int(nA);
                float pF = -1.0f;
                int tstar = -1;
                for (int ext = 0; ; ext += 16) {
                    const int t = ext + (lane >> 2);
                    const int tb = nAb - t;
                    const bool ok = hasF && (t == 0 || tb > 0);
                    const float n = __int_as_float(ok ? tb : 0);
                    const float q = __fdiv_rn(__fmul_rn(n, n), Syy);
                    if (ext == 0) pF = hasF ? __shfl(q, cls, 64) : -1.0f;
                    const unsigned long long tm = __ballot(ok && q == pF);
                    if (ext == 0 && (tm >> 4) == 0) { tstar = 1; break; }       // the rule: no float below n(A) reaches p(A)
                    const unsigned long long cm = ~(tm >> cls) & 0x1111111111111111ull;
                    const int nt = cm ? (__ffsll((long long)cm) - 1) >> 2 : 16;
                    if (tstar < 0 && nt < 16) tstar = ext + nt;
                    if (__ballot(tstar < 0) == 0) break;
                }
                const float nlo = hasF ? __int_as_float(nAb - (tstar - 1)) : __builtin_inff();
                int mF = -1;
#pragma unroll
                for (int m = M - 1; m >= 0; m--) mF = __fadd_rn(af[m], Sxy) >= nlo ? m : mF;
                float p = mF >= 0 ? pF : -1.0f;
                int idx = mF >= 0 ? lane + 64 * mF : 0x7fffffff;
                for (int k = 0; k < used_e; k++) {
                    const int ci = Q.idx[k][lane];
                    const float num = __fadd_rn(fabsf(Q.x[k][lane]), Sxy);
                    const float pc = __fdiv_rn(__fmul_rn(num, num), __fadd_rn(Q.fy[k][lane], Syy));
                    const bool better = (ci >= 0) & ((pc > p) | ((pc == p) & (ci < idx)));     // selects, no branches
                    p = better ? pc : p;
                    idx = better ? ci : idx;
                }
                // the class's largest p (p >= 0, or -1 for "nothing": ordered like the bit patterns), lowest index
                const int pm = class_max_i32(__float_as_int(p));
                const int im = class_min_i32(__float_as_int(p) == pm && p >= 0.0f ? idx : 0x7fffffff);
                float p0 = __int_as_float(__builtin_amdgcn_readlane(pm, 0)), p1 = __int_as_float(__builtin_amdgcn_readlane(pm, 1));
                const float p2 = __int_as_float(__builtin_amdgcn_readlane(pm, 2)), p3 = __int_as_float(__builtin_amdgcn_readlane(pm, 3));
                int i0 = __builtin_amdgcn_readlane(im, 0), i1 = __builtin_amdgcn_readlane(im, 1);
                const int i2 = __builtin_amdgcn_readlane(im, 2), i3 = __builtin_amdgcn_readlane(im, 3);
                if (p0 < p2) { p0 = p2; i0 = i2; }            // classes (3,2) replace (1,0) only when strictly greater
                if (p1 < p3) { p1 = p3; i1 = i3; }
                best = !(p1 < p0) ? i1 : i0;                  // class 1 replaces class 0 unless p1 < p0
            } else {
                // every lane's first element at p = 0 (the asm's initial maximum), then the carriers
                float p = lane < N4 ? 0.0f : -1.0f;
                int idx = lane < N4 ? lane : 0x7fffffff;
                for (int k = 0; k < used_e; k++) {
                    const int ci = Q.idx[k][lane];
                    const float fk = Q.fy[k][lane];
                    const float num = 0.0f < fk ? __fsub_rn(Sxy, fabsf(Q.x[k][lane])) : 0.0f;
                    const float pc = __fdiv_rn(__fmul_rn(num, num), __fsub_rn(Syy, fk));
                    if (ci >= 0 && (pc > p || (pc == p && ci < idx))) { p = pc; idx = ci; }
                }
#pragma unroll
                for (int o = 4; o <= 32; o <<= 1) {
                    const float op = __shfl_xor(p, o, 64);
                    const int oi = __shfl_xor(idx, o, 64);
                    if (op > p || (op == p && oi < idx)) { p = op; idx = oi; }
                }
                {
                    const float op = __shfl_xor(p, 2, 64);
                    const int oi = __shfl_xor(idx, 2, 64);
                    if ((lane & 2) == 0 && p < op) { p = op; idx = oi; }
                }
                const float p1 = __shfl(p, 1, 64), p0 = __shfl(p, 0, 64);
                const int i1 = __shfl(idx, 1, 64), i0 = __shfl(idx, 0, 64);
                best = __builtin_amdgcn_readfirstlane(!(p1 < p0) ? i1 : i0);
            }
            // the winner: on the list already, or a pulse-free element that joins it
            bool listed = false;
            const float one = add ? 1.0f : -1.0f;
            for (int k = 0; k < used_e; k++) {
                const unsigned long long hit = __ballot(Q.idx[k][lane] == best);
                if (hit) {
                    const int l = __ffsll((long long)hit) - 1;
                    axb = fabsf(Q.x[k][l]);
                    fyb = Q.fy[k][l];
                    if (lane == l) Q.fy[k][l] = __fadd_rn(fyb, one);
                    listed = true;
                }
            }
            if (!listed) {
                const int owner = best & 63, slot = best >> 6;         // removal: best < 4, the class's first element
                axb = look_up<M>(af, slot, owner);
                const int sg = __shfl((int)((neg >> slot) & 1ull), owner, 64);
                take_out<M>(af, slot, owner, lane);
                insert(best, sg ? -axb : axb, one);
                dirty = true;
            }
            if (add) { Sxy = __fadd_rn(Sxy, axb); Syy = __fadd_rn(Syy, fyb); }
            else     { Sxy = __fsub_rn(Sxy, axb); Syy = __fsub_rn(Syy, fyb); }
        }
    }
    // zeros everywhere, then the carriers' pulses on top (the zeros have arrived before they leave)
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int i = lane + 64 * m;
        if (i < N) yout[i] = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int tot = 0, nzc = 0, last = 0;
    bool bigl = overflow;                                     // a full list cannot happen (2 K entries per class): fail the frame if it does
    const int rows = rows_in_use();
    for (int k = 0; k < rows; k++) {
        const int ci = Q.idx[k][lane];
        const float xs = Q.x[k][lane];
        const int iv = __float2int_rn(Q.fy[k][lane]);
        const bool on = ci >= 0 && ci < N;
        if (on) yout[ci] = (int16_t)(signbit(xs) ? -iv : iv);              // orps sign, cvtps2dq
        const int a = on ? (iv < 0 ? -iv : iv) : 0;
        tot += a;
        nzc += a != 0 ? 1 : 0;
        last = a != 0 && ci + 1 > last ? ci + 1 : last;
        bigl = bigl || a >= K;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        tot += __shfl_xor(tot, o, 64);
        nzc += __shfl_xor(nzc, o, 64);
        const int t = __shfl_xor(last, o, 64);
        last = t > last ? t : last;
    }
    __syncthreads();                                          // the next band's sums reuse the buffer
    if (tot > K) return pvq_count_from_memory(yout, N, K, lane);
    return PvqBandCount{ tot == K ? last : N, nzc, __ballot(bigl) != 0 };
}

// bands in coding order (ffv2.c:100-120): band b = coefficients [1+BS[b], 1+BS[b+1])
__device__ constexpr int PVQ_BS[14] = { 0, 15, 23, 31, 63, 95, 127, 255, 383, 511, 1023, 1535, 2047, 4096 };

template <int M>
__device__ __forceinline__ PvqBandCount quant_band(const int32_t *coef, int b, int32_t W, int K, PvqLds &L, int lane, int16_t *y)
{
    const int lo = 1 + PVQ_BS[b];
    const int N = PVQ_BS[b + 1] - PVQ_BS[b];                   // 2049 for the last band: it also
    int cv[M];                                                 // reads temp2[4096]
    long long e = 0;
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int i = lane + 64 * m;
        cv[m] = i < N ? (lo + i < 4096 ? coef[lo + i] : W) : 0;
        e += (long long)cv[m] * cv[m];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) e += __shfl_xor(e, o, 64);
    const float fgain = __fadd_rn(sqrtf((float)e), 1.1920929e-7f);     // ffv2enc.c:166
    float x[M];
#pragma unroll
    for (int m = 0; m < M; m++) x[m] = __fdiv_rn((float)cv[m], fgain);  // ffv2enc.c:169
    return pvq_search_wave<M>(x, N, K, L, lane, y + lo);
}

struct FFV2PvqArgs {
    const int32_t *coef;      // [nbp][4096] coding order
    const int32_t *W;         // [nbp] or null
    int16_t *y;               // [nbp][FFV2_Y_STRIDE]; element 1+j of the coding order at y[1+j], W slot at y[4096]
    int qp;
    long long nbp;
    // for the lane coder (null otherwise): what it will read of every band, so that no kernel has to walk
    // the pulses again to find out
    FFV2SymRec *cnt;          // [nbp] count[13], offset = their sum
    uint32_t *bits;           // [nbp] raw bits of the block-plane: codes[14] (Exp-Golomb bits, T-stage) + sign bits
    const uint32_t *codes;    // [nbp][FFV2_CODES_PER_BP]
    int32_t *abort_;          // [nbp / nblk] |= 1: a pulse as large as the alphabet
    int nblk;
};

__global__ __launch_bounds__(64, 3) void ffv2_pvq_kernel(const FFV2PvqArgs a)
{
    __shared__ PvqLds L;
    const long long bp = blockIdx.x;
    const int lane = threadIdx.x;
    const int32_t *coef = a.coef + bp * 4096;
    const int32_t W = a.W ? a.W[bp] : 0;
    int16_t *y = a.y + bp * FFV2_Y_STRIDE;
    FFV2SymRec *r = a.cnt ? a.cnt + bp : nullptr;
    uint32_t total = 0, nz = 0;
    bool big = false;
    auto note = [&](int b, const PvqBandCount c) {
        if (r && lane == 0) r->count[b] = (uint16_t)c.stop;
        total += (uint32_t)c.stop; nz += (uint32_t)c.nz; big = big || c.big;
    };
    for (int b = 0; b < 6; b++)  note(b, quant_band<1>(coef, b, W, a.qp, L, lane, y));
    for (int b = 6; b < 9; b++)  note(b, quant_band<2>(coef, b, W, a.qp, L, lane, y));
    for (int b = 9; b < 12; b++) note(b, quant_band<8>(coef, b, W, a.qp, L, lane, y));
    note(12, quant_band<33>(coef, 12, W, a.qp, L, lane, y));
    if (r && lane == 0) {
        r->offset = total;
        r->pad = 0;
        a.bits[bp] = a.codes[bp * FFV2_CODES_PER_BP + 14] + nz;
        if (big) atomicOr((int *)&a.abort_[bp / a.nblk], 1);
    }
}

// The same with the list search in the bands of 512 and 2049 coefficients (qp <= 64 = 8 PVQ_LIST_E): without the
// per-element pulse counts in registers four wavefronts fit a SIMD.
template <int M>
__device__ __forceinline__ PvqBandCount quant_band_lists(const int32_t *coef, int b, int32_t W, int K, PvqLds &L, int lane, int16_t *y)
{
    const int lo = 1 + PVQ_BS[b];
    const int N = PVQ_BS[b + 1] - PVQ_BS[b];
    float x[M];
    long long e = 0;
#pragma unroll
    for (int m = 0; m < M; m++) {
        const int i = lane + 64 * m;
        const int cv = i < N ? (lo + i < 4096 ? coef[lo + i] : W) : 0;
        x[m] = (float)cv;
        e += (long long)cv * cv;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) e += __shfl_xor(e, o, 64);
    const float fgain = __fadd_rn(sqrtf((float)e), 1.1920929e-7f);     // ffv2enc.c:166
    // ffv2enc.c:169, the IEEE division by a denominator all elements share: its reciprocal and the reciprocal's
    // refinement once, the two quotient corrections per element -- the sequence the hardware division runs, without its
    // scaling and fix-up instructions, which do nothing here: numerators are 0 or 1 ... 2^31 in magnitude, the
    // denominator lies in [2^-23, 2^37] and is at least as large as any numerator (same argument as in pvq_search_wave)
    const float r0 = __builtin_amdgcn_rcpf(fgain);
    const float r1 = __fmaf_rn(__fmaf_rn(-fgain, r0, 1.0f), r0, r0);
#pragma unroll
    for (int m = 0; m < M; m++) {
        const float n = x[m];
        const float q0 = __fmul_rn(n, r1);
        const float q1 = __fmaf_rn(__fmaf_rn(-fgain, q0, n), r1, q0);
        x[m] = __fmaf_rn(__fmaf_rn(-fgain, q1, n), r1, q1);
    }
    return pvq_search_lists<M>(x, N, K, L, lane, y + lo);
}

__global__ __launch_bounds__(64, 4) void ffv2_pvq_lists_kernel(const FFV2PvqArgs a)
{
    __shared__ PvqLds L;
    const long long bp = blockIdx.x;
    const int lane = threadIdx.x;
    const int32_t *coef = a.coef + bp * 4096;
    const int32_t W = a.W ? a.W[bp] : 0;
    int16_t *y = a.y + bp * FFV2_Y_STRIDE;
    FFV2SymRec *r = a.cnt ? a.cnt + bp : nullptr;
    uint32_t total = 0, nz = 0;
    bool big = false;
    auto note = [&](int b, const PvqBandCount c) {
        if (r && lane == 0) r->count[b] = (uint16_t)c.stop;
        total += (uint32_t)c.stop; nz += (uint32_t)c.nz; big = big || c.big;
    };
    for (int b = 0; b < 6; b++)  note(b, quant_band<1>(coef, b, W, a.qp, L, lane, y));
    for (int b = 6; b < 9; b++)  note(b, quant_band<2>(coef, b, W, a.qp, L, lane, y));
    for (int b = 9; b < 12; b++) note(b, quant_band_lists<8>(coef, b, W, a.qp, L, lane, y));
    note(12, quant_band_lists<33>(coef, 12, W, a.qp, L, lane, y));
    if (r && lane == 0) {
        r->offset = total;
        r->pad = 0;
        a.bits[bp] = a.codes[bp * FFV2_CODES_PER_BP + 14] + nz;
        if (big) atomicOr((int *)&a.abort_[bp / a.nblk], 1);
    }
}

// test hook: the bare search on caller-provided float vectors
__global__ __launch_bounds__(64) void ffv2_pvq_vectors_kernel(const float *X, int stride, int N, int K, int16_t *y, int general)
{
    __shared__ PvqLds L;
    const int lane = threadIdx.x;
    const float *xv = X + (size_t)blockIdx.x * stride;
    int16_t *yv = y + (size_t)blockIdx.x * stride;
    float x[33];
#pragma unroll
    for (int m = 0; m < 33; m++) {
        const int i = lane + 64 * m;
        x[m] = i < N ? xv[i] : 0.0f;
    }
    // the list search where it is exact (see there); `general` != 0: the element-by-element search whatever the input
    bool wild = false;
#pragma unroll
    for (int m = 0; m < 33; m++) {
        const float a = fabsf(x[m]);
        wild = wild || (a != 0.0f && !(a >= 0x1p-40f && a <= 0x1p40f));
    }
    if (general || K > 64 || K < 1 || __ballot(wild)) pvq_search_wave<33>(x, N, K, L, lane, yv);
    else pvq_search_lists<33>(x, N, K, L, lane, yv);
}

// ---------------------------------------------------------------------------
// Symbol compaction for the host range coder (qp > 0).  The coder reads a band's pulses only
// until their magnitudes add up to qp (ffv2enc.c:176-186), so everything behind that point --
// and the upper byte of every int16 -- never has to cross PCIe.  One wavefront per block-plane:
// per band the number of symbols the coder will read, then the symbols themselves as int8 into
// a per-frame stream at an offset reserved with one atomic add.  The stream's order is whatever
// order the workgroups arrive in; the per-block-plane record {offset, 13 counts} indexes it.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ffv2_compact_kernel(const int16_t *y, int qp, int nblk, FFV2SymRec *rec,
                                                          int8_t *stream, size_t stream_stride, uint32_t *totals)
{
    const int f = blockIdx.y, bp = blockIdx.x, lane = threadIdx.x;
    const int16_t *yy = y + ((size_t)f * nblk + bp) * FFV2_Y_STRIDE;
    uint32_t cnt[FFV2_NUM_BANDS];
    uint32_t total = 0;
#pragma unroll 1
    for (int b = 0; b < FFV2_NUM_BANDS; b++) {
        const int lo = 1 + PVQ_BS[b], N = PVQ_BS[b + 1] - PVQ_BS[b];
        // first j with |y_0| + ... + |y_j| >= qp, walking the band in rows of 64
        int run = 0, stop = N;
        for (int j0 = 0; j0 < N && stop == N; j0 += 64) {
            const int j = j0 + lane;
            int a = j < N ? yy[lo + j] : 0;
            a = a < 0 ? -a : a;
            int incl = a;                                      // inclusive prefix over the row
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            const unsigned long long hit = __ballot(run + incl >= qp);
            if (hit) stop = j0 + __ffsll((long long)hit);      // 1-based position -> count
            run += __shfl(incl, 63, 64);
        }
        if (stop > N) stop = N;
        cnt[b] = (uint32_t)stop;
        total += (uint32_t)stop;
    }
    uint32_t off = 0;
    if (lane == 0) off = atomicAdd(&totals[f], total);
    off = (uint32_t)__shfl((int)off, 0, 64);
    FFV2SymRec *r = rec + (size_t)f * nblk + bp;
    if (lane == 0) r->offset = off;
    int8_t *dst = stream + (size_t)f * stream_stride + off;
#pragma unroll 1
    for (int b = 0; b < FFV2_NUM_BANDS; b++) {
        const int lo = 1 + PVQ_BS[b];
        if (lane == 0) r->count[b] = (uint16_t)cnt[b];
        for (uint32_t j = (uint32_t)lane; j < cnt[b]; j += 64) dst[j] = (int8_t)yy[lo + j];
        dst += cnt[b];
    }
}

}  // namespace

hipError_t ffv2_launch_compact(const int16_t *y, int qp, int nblk, int nframes, FFV2SymRec *rec, int8_t *stream,
                               size_t stream_stride, uint32_t *totals, hipStream_t s)
{
    hipLaunchKernelGGL(ffv2_compact_kernel, dim3((unsigned)nblk, (unsigned)nframes), dim3(64), 0, s,
                       y, qp, nblk, rec, stream, stream_stride, totals);
    return hipGetLastError();
}

// Which search: the lists for qp <= 64 (FFV2AMD_PVQ_GENERAL=1: the element-by-element kernel whatever qp; read per
// launch so that tests can compare the two)
static bool pvq_general(int qp)
{
    const char *g = getenv("FFV2AMD_PVQ_GENERAL");
    return qp > 64 || qp < 1 || (g && atoi(g) != 0);
}

static void pvq_launch(const FFV2PvqArgs &a, hipStream_t s)
{
    const dim3 grid((unsigned)a.nbp), block(64);
    if (pvq_general(a.qp)) hipLaunchKernelGGL(ffv2_pvq_kernel, grid, block, 0, s, a);
    else                   hipLaunchKernelGGL(ffv2_pvq_lists_kernel, grid, block, 0, s, a);
}

hipError_t ffv2_launch_pvq(const int32_t *coef, const int32_t *W, int16_t *y, int qp, long long nbp, hipStream_t s)
{
    FFV2PvqArgs a{ coef, W, y, qp, nbp, nullptr, nullptr, nullptr, nullptr, 1 };
    pvq_launch(a, s);
    return hipGetLastError();
}

// The same, and for each block-plane what the lane coder's count pass would find (ffv2_lanecoder.hip):
// cnt / bits / codes indexed like y (block-plane 0 = the launch's first), abort_ by frame (nblk block-planes each).
hipError_t ffv2_launch_pvq_counted(const int32_t *coef, const int32_t *W, int16_t *y, int qp, long long nbp, int nblk,
                                   const uint32_t *codes, FFV2SymRec *cnt, uint32_t *bits, int32_t *abort_, hipStream_t s)
{
    if (!codes || !cnt || !bits || !abort_ || nblk < 1) return hipErrorInvalidValue;
    FFV2PvqArgs a{ coef, W, y, qp, nbp, cnt, bits, codes, abort_, nblk };
    pvq_launch(a, s);
    return hipGetLastError();
}

hipError_t ffv2_launch_pvq_vectors(const float *X, int stride, int N, int K, int count, int16_t *y, hipStream_t s)
{
    if (N < 1 || N > 2049 || stride < N) return hipErrorInvalidValue;
    const char *g = getenv("FFV2AMD_PVQ_GENERAL");
    hipLaunchKernelGGL(ffv2_pvq_vectors_kernel, dim3(count), dim3(64), 0, s, X, stride, N, K, y, g && atoi(g) != 0 ? 1 : 0);
    return hipGetLastError();
}
