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
            const int best = !(p1 < p0) ? i1 : i0;
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

// test hook: the bare search on caller-provided float vectors
__global__ __launch_bounds__(64) void ffv2_pvq_vectors_kernel(const float *X, int stride, int N, int K, int16_t *y)
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
    pvq_search_wave<33>(x, N, K, L, lane, yv);
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

hipError_t ffv2_launch_pvq(const int32_t *coef, const int32_t *W, int16_t *y, int qp, long long nbp, hipStream_t s)
{
    FFV2PvqArgs a{ coef, W, y, qp, nbp, nullptr, nullptr, nullptr, nullptr, 1 };
    hipLaunchKernelGGL(ffv2_pvq_kernel, dim3((unsigned)nbp), dim3(64), 0, s, a);
    return hipGetLastError();
}

// The same, and for each block-plane what the lane coder's count pass would find (ffv2_lanecoder.hip):
// cnt / bits / codes indexed like y (block-plane 0 = the launch's first), abort_ by frame (nblk block-planes each).
hipError_t ffv2_launch_pvq_counted(const int32_t *coef, const int32_t *W, int16_t *y, int qp, long long nbp, int nblk,
                                   const uint32_t *codes, FFV2SymRec *cnt, uint32_t *bits, int32_t *abort_, hipStream_t s)
{
    if (!codes || !cnt || !bits || !abort_ || nblk < 1) return hipErrorInvalidValue;
    FFV2PvqArgs a{ coef, W, y, qp, nbp, cnt, bits, codes, abort_, nblk };
    hipLaunchKernelGGL(ffv2_pvq_kernel, dim3((unsigned)nbp), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t ffv2_launch_pvq_vectors(const float *X, int stride, int N, int K, int count, int16_t *y, hipStream_t s)
{
    if (N < 1 || N > 2049 || stride < N) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ffv2_pvq_vectors_kernel, dim3(count), dim3(64), 0, s, X, stride, N, K, y);
    return hipGetLastError();
}
