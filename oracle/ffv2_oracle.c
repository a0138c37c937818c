/*
 * ffv2_oracle.c -- plain-C restatement of the FFV2 encoder hot path of
 * cyanreg/ffmpeg_ffv2.  TEST INFRASTRUCTURE ONLY (see ffv2_oracle.h).
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference checkout).  Nothing here is copied from the reference: the 64-point
 * transform is an interpreter over our own IR table, the scan is one
 * permutation table, everything else is written from the behaviour described
 * in SURVEY.md section 8 and checked against the reference's known answers.
 *
 * Build:  make -C oracle      (gcc -O2 -ffp-contract=off: float steps must not fuse)
 */
#include "ffv2_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "gen/fdct64_ir_table.h"
#include "gen/idct64_ir_table.h"
#include "gen/scan_lut.h"

#define SB 64

/* ------------------------------------------------------------------ */
/* pix_fmt table  (libavcodec/ffv2enc.c:596-601, :500-501)             */
/* ------------------------------------------------------------------ */
int ffv2o_pixfmt_info(int pix_fmt, int *planes, int *depth)
{
    int p, d;
    switch (pix_fmt) {
    case FFV2O_PIX_GRAY8:       p = 1; d = 8;  break;
    case FFV2O_PIX_YUV444P:
    case FFV2O_PIX_GBRP:        p = 3; d = 8;  break;
    case FFV2O_PIX_YUV444P10LE:
    case FFV2O_PIX_GBRP10LE:    p = 3; d = 10; break;
    case FFV2O_PIX_YUV444P12LE:
    case FFV2O_PIX_GBRP12LE:    p = 3; d = 12; break;
    default: return FFV2O_ERR_PIXFMT;
    }
    if (planes) *planes = p;
    if (depth)  *depth  = d;
    return 0;
}

/* int32 arithmetic with two's-complement wrap, as the compiled reference behaves */
static inline int32_t wmul_add(int32_t a, int32_t k, int32_t r)
{
    return (int32_t)((uint32_t)a * (uint32_t)k + (uint32_t)r);
}

/* ------------------------------------------------------------------ */
/* 1-D 64-point forward lifting DCT: IR interpreter                    */
/* (libavcodec/ffv2.c:4678-4812 od_bin_fdct64 and the OD_F* macro tree */
/*  :313-4001; op semantics in oracle/gen/fdct64_ir_table.h)           */
/* ------------------------------------------------------------------ */
static void run_ir(const int32_t (*ops)[6], int nops, int32_t *r)
{
    for (int i = 0; i < nops; i++) {
        const int32_t *op = ops[i];
        int32_t a = r[op[2]];
        switch (op[0]) {
        case 0: r[op[1]] = (int32_t)((uint32_t)a - (uint32_t)r[op[3]]); break;
        case 1: r[op[1]] = (int32_t)((uint32_t)a + (uint32_t)r[op[3]]); break;
        case 2: r[op[1]] = (a + (a < 0)) >> 1; break;
        case 3: r[op[1]] = (int32_t)((uint32_t)r[op[1]] + (uint32_t)(wmul_add(a, op[3], op[4]) >> op[5])); break;
        case 4: r[op[1]] = (int32_t)((uint32_t)r[op[1]] - (uint32_t)(wmul_add(a, op[3], op[4]) >> op[5])); break;
        case 5: r[op[1]] = (int32_t)(0u - (uint32_t)a); break;
        }
    }
}

void ffv2o_fdct64(int32_t y[64], const int32_t *x, int xstride)
{
    int32_t r[FDCT64_IR_NREGS];
    for (int k = 0; k < 64; k++)
        r[k] = x[k * xstride];
    run_ir(FDCT64_IR_OPS, FDCT64_IR_NOPS, r);
    for (int k = 0; k < 64; k++)
        y[k] = r[FDCT64_IR_OUT[k]];
}

/* 1-D 64-point inverse (libavcodec/ffv2.c:4814-4948 od_bin_idct64 over the OD_I* macros):
 * decoder side, SURVEY.md section 8(f) rank 2. */
void ffv2o_idct64(int32_t *x, int xstride, const int32_t y[64])
{
    int32_t r[IDCT64_IR_NREGS];
    for (int k = 0; k < 64; k++)
        r[k] = y[k];
    run_ir(IDCT64_IR_OPS, IDCT64_IR_NOPS, r);
    for (int k = 0; k < 64; k++)
        x[k * xstride] = r[IDCT64_IR_OUT[k]];
}

/* ------------------------------------------------------------------ */
/* lapping pre-filter, 32 taps (libavcodec/ffv2.c:183-214, params      */
/* :168-172).  P[0..15] scales, P[16..30] / P[31..45] the two lifting   */
/* ladders.                                                             */
/* ------------------------------------------------------------------ */
static const int32_t LAP32_P[46] = {
    91, 70, 68, 67, 67, 67, 67, 66, 66, 67, 67, 66, 67, 67, 67, 70,
    -32, -41, -42, -41, -40, -38, -36, -34, -32, -29, -24, -19, -14, -9, -5,
    58, 52, 50, 48, 45, 43, 40, 38, 35, 32, 29, 24, 18, 13, 8,
};

void ffv2o_lap_filter32(int32_t y[32], const int32_t x[32])
{
    int32_t t[32];
    for (int i = 0; i < 16; i++)
        t[31 - i] = x[i] - x[31 - i];
    for (int i = 0; i < 16; i++)
        t[15 - i] = x[15 - i] - (t[16 + i] >> 1);
    for (int i = 16; i < 32; i++) {
        t[i] = wmul_add(t[i], LAP32_P[i - 16], 0) >> 6;
        if (t[i] > 0)
            t[i]++;
    }
    for (int i = 31; i > 16; i--) {
        t[i]     += wmul_add(t[i - 1], LAP32_P[i - 1], 32) >> 6;
        t[i - 1] += wmul_add(t[i], LAP32_P[i + 14], 32) >> 6;
    }
    for (int i = 0; i < 16; i++)
        t[i] += t[31 - i] >> 1;
    for (int i = 0; i < 16; i++) {
        y[i]      = t[i];
        y[16 + i] = t[15 - i] - t[16 + i];
    }
}

/* ------------------------------------------------------------------ */
/* T-stage on a whole frame                                            */
/* ------------------------------------------------------------------ */
typedef struct {
    int w, h, planes, depth, nsx, nsy;
    int gw, gh;            /* 64*nsx, 64*nsy : the part of the padded plane ever touched */
    int32_t *pix[4];
} OFrame;

static void oframe_free(OFrame *f)
{
    for (int p = 0; p < 4; p++)
        free(f->pix[p]);
}

/* ffv2enc.c:55-75 (zeroed int32 planes; only the 64-aligned grid matters),
 * ffv2.c:26-38 (level shift), ffv2enc.c:345-366 (H seams of all SBs, then V seams) */
static int oframe_build(OFrame *f, const uint8_t *const data[4], const ptrdiff_t linesize[4],
                        int w, int h, int pix_fmt)
{
    memset(f, 0, sizeof(*f));
    if (ffv2o_pixfmt_info(pix_fmt, &f->planes, &f->depth) < 0 || w <= 0 || h <= 0)
        return FFV2O_ERR_PIXFMT;
    f->w = w; f->h = h;
    f->nsx = (w + SB - 1) / SB;
    f->nsy = (h + SB - 1) / SB;
    f->gw = f->nsx * SB;
    f->gh = f->nsy * SB;
    for (int p = 0; p < f->planes; p++) {
        int32_t *pl = calloc((size_t)f->gw * f->gh, sizeof(int32_t));
        if (!pl) { oframe_free(f); return FFV2O_ERR_NOMEM; }
        f->pix[p] = pl;
        for (int yy = 0; yy < h; yy++) {
            const uint8_t *row = data[p] + (ptrdiff_t)yy * linesize[p];
            int32_t *dst = pl + (size_t)yy * f->gw;
            if (f->depth == 8) {
                for (int xx = 0; xx < w; xx++)
                    dst[xx] = ((int32_t)row[xx] << 4) - 2048;
            } else {
                for (int xx = 0; xx < w; xx++) {
                    uint16_t v;
                    memcpy(&v, row + 2 * xx, 2);            /* AV_RN16: native endian */
                    dst[xx] = ((int32_t)v << (12 - f->depth)) - 2048;
                }
            }
        }
        /* pass 1: vertical seams x = 64*i, i >= 1, every row of the grid */
        for (int yy = 0; yy < f->gh; yy++)
            for (int i = 1; i < f->nsx; i++) {
                int32_t *s = pl + (size_t)yy * f->gw + i * SB - 16;
                ffv2o_lap_filter32(s, s);
            }
        /* pass 2: horizontal seams y = 64*j, j >= 1, every column of the grid */
        for (int j = 1; j < f->nsy; j++)
            for (int xx = 0; xx < f->gw; xx++) {
                int32_t col[32];
                int32_t *s = pl + (size_t)(j * SB - 16) * f->gw + xx;
                for (int k = 0; k < 32; k++) col[k] = s[(size_t)k * f->gw];
                ffv2o_lap_filter32(col, col);
                for (int k = 0; k < 32; k++) s[(size_t)k * f->gw] = col[k];
            }
    }
    return 0;
}

/* ffv2.c:4950-4960 (columns first, into rows of tmp; then rows reading tmp
 * columns) followed by ffv2.c:62-79 via the scan permutation. */
static void block_coeffs(const OFrame *f, int p, int sbx, int sby, int32_t out[4096])
{
    int32_t tmp[4096], dst[4096];
    const int32_t *src = f->pix[p] + (size_t)sby * SB * f->gw + sbx * SB;
    for (int i = 0; i < 64; i++)
        ffv2o_fdct64(tmp + 64 * i, src + i, f->gw);
    for (int i = 0; i < 64; i++)
        ffv2o_fdct64(dst + 64 * i, tmp + i, 64);
    for (int q = 0; q < 4096; q++)
        out[q] = dst[FFV2_SCAN_LUT[q]];
}

/* band boundaries in coding order (ffv2.c:100-120 over the five layouts'
 * bands_start fields, zigzags.h): band b = coding indices [1+BS[b], 1+BS[b+1]) */
static const int BANDS_START[14] = { 0, 15, 23, 31, 63, 95, 127, 255, 383, 511, 1023, 1535, 2047, 4096 };
#define NUM_BANDS 13

int ffv2o_tstage(const uint8_t *const data[4], const ptrdiff_t linesize[4],
                 int width, int height, int pix_fmt,
                 int32_t *coef, int64_t *energy)
{
    OFrame f;
    int ret = oframe_build(&f, data, linesize, width, height, pix_fmt);
    if (ret < 0)
        return ret;
    for (int sby = 0; sby < f.nsy; sby++)
        for (int sbx = 0; sbx < f.nsx; sbx++)
            for (int p = 0; p < f.planes; p++) {
                int32_t c[4096];
                size_t bp = ((size_t)sby * f.nsx + sbx) * f.planes + p;
                block_coeffs(&f, p, sbx, sby, c);
                if (coef)
                    memcpy(coef + bp * 4096, c, sizeof(c));
                if (energy)
                    for (int b = 0; b < NUM_BANDS; b++) {
                        int64_t e = 0;
                        int lo = 1 + BANDS_START[b], hi = 1 + BANDS_START[b + 1];
                        if (hi > 4096) hi = 4096;
                        for (int q = lo; q < hi; q++)
                            e += (int64_t)c[q] * c[q];
                        energy[bp * NUM_BANDS + b] = e;
                    }
            }
    oframe_free(&f);
    return 0;
}

/* ------------------------------------------------------------------ */
/* decoder-side inverse of the T-stage (round-trip self check)         */
/* ------------------------------------------------------------------ */
/* lapping post-filter, 32 taps, in place (libavcodec/ffv2.c:216-239; note the
 * truncating per-sample divide of :229-230, which makes it only approximately the
 * inverse of the pre-filter) */
void ffv2o_inv_lap_filter32(int32_t x[32])
{
    int32_t t[32];
    for (int i = 0; i < 16; i++)
        t[31 - i] = x[i] - x[31 - i];
    for (int i = 0; i < 16; i++)
        t[15 - i] = x[15 - i] - (t[16 + i] >> 1);
    for (int i = 16; i < 31; i++) {
        t[i]     -= wmul_add(t[i + 1], LAP32_P[i + 15], 32) >> 6;
        t[i + 1] -= wmul_add(t[i], LAP32_P[i], 32) >> 6;
    }
    for (int i = 31; i >= 16; i--)
        t[i] = (int32_t)((uint32_t)t[i] << 6) / LAP32_P[i - 16];
    for (int i = 0; i < 16; i++) {
        t[i] += t[31 - i] >> 1;
        x[i] = t[i];
    }
    for (int i = 16; i < 32; i++)
        x[i] = t[31 - i] - t[i];
}

/* coefficients in coding order [nsb*planes][4096] -> picture planes:
 * coding_to_raster (ffv2.c:81-98), tx_inv_2d (ffv2.c:4962-4972: rows first, into columns
 * of tmp, then columns), post-filters on all horizontal seams then on all vertical seams
 * (ffv2dec.c, DOLAP block), coeffs_2_ref (ffv2.c:40-52; no clipping). The debugging
 * overlay of ffv2dec.c:258-273 is not part of the transform and is left out. */
int ffv2o_inverse_tstage(const int32_t *coef, int width, int height, int pix_fmt,
                         uint8_t *const data[4], const ptrdiff_t linesize[4])
{
    int planes, depth;
    if (ffv2o_pixfmt_info(pix_fmt, &planes, &depth) < 0 || width <= 0 || height <= 0)
        return FFV2O_ERR_PIXFMT;
    const int nsx = (width + SB - 1) / SB, nsy = (height + SB - 1) / SB;
    const int gw = nsx * SB, gh = nsy * SB;
    int32_t *pl = malloc((size_t)gw * gh * sizeof(int32_t));
    if (!pl)
        return FFV2O_ERR_NOMEM;
    for (int p = 0; p < planes; p++) {
        for (int sby = 0; sby < nsy; sby++)
            for (int sbx = 0; sbx < nsx; sbx++) {
                const int32_t *c = coef + (((size_t)sby * nsx + sbx) * planes + p) * 4096;
                int32_t src[4096], tmp[4096];
                int32_t *dst = pl + (size_t)sby * SB * gw + sbx * SB;
                for (int q = 0; q < 4096; q++)
                    src[FFV2_SCAN_LUT[q]] = c[q];
                for (int i = 0; i < 64; i++)
                    ffv2o_idct64(tmp + i, 64, src + 64 * i);
                for (int i = 0; i < 64; i++)
                    ffv2o_idct64(dst + i, gw, tmp + 64 * i);
            }
        for (int j = 1; j < nsy; j++)
            for (int xx = 0; xx < gw; xx++) {
                int32_t col[32];
                int32_t *s = pl + (size_t)(j * SB - 16) * gw + xx;
                for (int k = 0; k < 32; k++) col[k] = s[(size_t)k * gw];
                ffv2o_inv_lap_filter32(col);
                for (int k = 0; k < 32; k++) s[(size_t)k * gw] = col[k];
            }
        for (int yy = 0; yy < gh; yy++)
            for (int i = 1; i < nsx; i++)
                ffv2o_inv_lap_filter32(pl + (size_t)yy * gw + i * SB - 16);
        for (int yy = 0; yy < height; yy++) {
            uint8_t *row = data[p] + (ptrdiff_t)yy * linesize[p];
            const int32_t *srow = pl + (size_t)yy * gw;
            for (int xx = 0; xx < width; xx++) {
                int32_t v = (srow[xx] + 2048) >> (12 - depth);
                if (depth == 8) {
                    row[xx] = (uint8_t)v;
                } else {
                    uint16_t w = (uint16_t)v;
                    memcpy(row + 2 * xx, &w, 2);
                }
            }
        }
    }
    free(pl);
    return 0;
}

/* ------------------------------------------------------------------ */
/* Daala entropy encoder (libavcodec/daala_entropy.c)                  */
/* ------------------------------------------------------------------ */
typedef struct {
    uint64_t low;          /* :362 window                                  */
    uint32_t rng;          /* 16-bit range, 0x8000 at reset (:588)         */
    int      cnt;          /* :589 starts at -9                            */
    uint16_t *pre;         /* pre-carry words (:107-151)                   */
    size_t   npre, cappre;
    uint8_t *raw;          /* raw bytes in write order (:227-270 writes them
                              from the end of the buffer backwards)        */
    size_t   nraw, capraw;
    uint64_t win;          /* raw-bit window, LSB first                    */
    int      nwin;
    int      err;
} OEnt;

static int ilog(uint32_t v)              /* daalaent_log2 = 1 + floor(log2 v), v > 0 */
{
    int n = 0;
    while (v) { n++; v >>= 1; }
    return n;
}

static void oent_init(OEnt *e)
{
    memset(e, 0, sizeof(*e));
    e->rng = 0x8000;
    e->cnt = -9;
}

static void oent_free(OEnt *e)
{
    free(e->pre);
    free(e->raw);
}

static void push_pre(OEnt *e, uint16_t w)
{
    if (e->npre == e->cappre) {
        e->cappre = e->cappre ? 2 * e->cappre : 1024;
        e->pre = realloc(e->pre, e->cappre * sizeof(uint16_t));
        if (!e->pre) { e->err = 1; e->npre = e->cappre = 0; return; }
    }
    e->pre[e->npre++] = w;
}

static void push_raw(OEnt *e, uint8_t b)
{
    if (e->nraw == e->capraw) {
        e->capraw = e->capraw ? 2 * e->capraw : 4096;
        e->raw = realloc(e->raw, e->capraw);
        if (!e->raw) { e->err = 1; e->nraw = e->capraw = 0; return; }
    }
    e->raw[e->nraw++] = b;
}

/* daala_entropy.c:107-151 */
static void oent_renorm(OEnt *e, uint64_t low, uint32_t rng)
{
    int c = e->cnt;
    int d = 16 - ilog(rng);
    int s = c + d;
    if (s >= 0) {
        uint64_t m;
        c += 16;
        m = ((uint64_t)1 << c) - 1;
        if (s >= 8) {
            push_pre(e, (uint16_t)(low >> c));
            low &= m;
            c -= 8;
            m >>= 8;
        }
        push_pre(e, (uint16_t)(low >> c));
        s = c + d - 24;
        low &= m;
    }
    e->low = low << d;
    e->rng = rng << d;
    e->cnt = s;
}

/* daala_entropy.c:362-378 with fl/fh/ft already scaled so that 16384 <= ft <= 32768 */
static int oent_code(OEnt *e, uint32_t fl, uint32_t fh, uint32_t ft)
{
    uint32_t r = e->rng, d, g, u, v;
    int sc;
    if (!(fl < fh && fh <= ft && ft >= 16384 && ft <= 32768 && ft <= r))
        return FFV2O_ERR_ABORT;
    sc = (r - ft) >= ft;
    ft <<= sc; fl <<= sc; fh <<= sc;
    d = r - ft;
    g = 2 * d > ft ? 2 * d - ft : 0;                       /* SAT(2d, ft) */
#define STEP(f) ((f) + ((f) < g ? (f) : g) + ((((f) > g ? (f) - g : 0) >> 1) < d ? (((f) > g ? (f) - g : 0) >> 1) : d))
    u = STEP(fl);
    v = STEP(fh);
#undef STEP
    oent_renorm(e, e->low + u, v - u);
    return 0;
}

/* daala_entropy.c:348-354: Q15 uniform CDF with n symbols; the table row
 * (daalatab.c:50-64) is round(32768*k/n) for the rows this encoder touches */
static int oent_uniform(OEnt *e, int s, int n)
{
    uint32_t fl, fh;
    if (s < 0 || s >= n)
        return FFV2O_ERR_ABORT;
    fl = s ? (uint32_t)((32768u * (uint32_t)s + n / 2) / n) : 0;
    fh = (uint32_t)((32768u * (uint32_t)(s + 1) + n / 2) / n);
    return oent_code(e, fl, fh, 32768);
}

/* daala_entropy.c:399-410 */
static int oent_uint(OEnt *e, uint32_t val, uint32_t num);
static void oent_bits(OEnt *e, uint32_t val, int n);

/* adaptive CDF (daala_entropy.h:140-161, daala_entropy.c:428-440, :334-347) */
typedef struct { uint16_t *cdf; int x, y, inc; } OCdf;

static int ocdf_init(OCdf *c, int x, int y, int inc, int inc_shift)
{
    int inc_g = inc >> inc_shift;
    c->x = x; c->y = y; c->inc = inc;
    c->cdf = malloc((size_t)((x * y) > 0 ? x * y : 1) * sizeof(uint16_t));
    if (!c->cdf)
        return FFV2O_ERR_NOMEM;
    for (int i = 0; i < x; i++)
        for (int j = 0; j < y; j++)
            c->cdf[i * y + j] = (uint16_t)(inc_g * j + inc_g);   /* fir == inc_g here */
    return 0;
}

static int oent_adapt(OEnt *e, OCdf *c, int val, int row, int n)
{
    uint16_t *cdf = c->cdf + row * c->y;
    uint32_t fl, fh, ft;
    int sc, ret;
    if (val < 0 || val >= n)
        return FFV2O_ERR_ABORT;                            /* av_assert0(s < nsyms) :336 */
    fl = val ? cdf[val - 1] : 0;
    fh = cdf[val];
    ft = cdf[n - 1];
    if (!(fl < fh && fh <= ft && ft >= 2 && ft <= 32768))
        return FFV2O_ERR_ABORT;                            /* :340-343 */
    sc = 15 - ilog(ft - 1);
    if ((ret = oent_code(e, fl << sc, fh << sc, ft << sc)) < 0)
        return ret;
    if (cdf[n - 1] + c->inc > 32767)
        for (int i = 0; i < n; i++)
            cdf[i] = (uint16_t)((cdf[i] >> 1) + i + 1);
    for (int i = val; i < n; i++)
        cdf[i] = (uint16_t)(cdf[i] + c->inc);
    return 0;
}

/* daala_entropy.c:227-270 (n <= 25): bits are appended LSB-first; whole bytes
 * leave the window oldest-first. */
static void oent_bits(OEnt *e, uint32_t val, int n)
{
    if (e->nwin + n > 64) {
        do {
            push_raw(e, (uint8_t)e->win);
            e->win >>= 8;
            e->nwin -= 8;
        } while (e->nwin >= 8);
    }
    e->win |= (uint64_t)val << e->nwin;
    e->nwin += n;
}

static int oent_uint(OEnt *e, uint32_t val, uint32_t num)
{
    if (num > 16) {
        int bit, adr, ret;
        num--;
        bit = ilog(num) - 4;
        adr = (int)(num >> bit) + 1;
        if ((ret = oent_uniform(e, (int)(val >> bit), adr)) < 0)
            return ret;
        oent_bits(e, val & (((uint32_t)1 << bit) - 1), bit);
        return 0;
    }
    return oent_uniform(e, (int)val, (int)num);
}

/* daala_entropy.c:624-735.  Writes the finished packet to out. */
static int oent_done(OEnt *e, uint8_t *out, size_t cap, size_t *size)
{
    uint64_t l = e->low, m = 0x7FFF, end;
    uint32_t r = e->rng;
    int c = e->cnt, s = 9, nbits;
    size_t nrange, total;
    uint64_t win;

    if (e->err)
        return FFV2O_ERR_NOMEM;
    end = (l + m) & ~m;
    while ((end | m) >= l + r) {
        s++;
        m >>= 1;
        end = (l + m) & ~m;
    }
    s += c;
    if (s > 0) {
        uint64_t n = ((uint64_t)1 << (c + 16)) - 1;
        do {
            push_pre(e, (uint16_t)(end >> (c + 16)));
            end &= n;
            s -= 8;
            c -= 8;
            n >>= 8;
        } while (s > 0);
    }
    if (e->err)
        return FFV2O_ERR_NOMEM;
    /* -s = free bits left in the last range byte; raw bits fill bytes from the
     * packet end backwards until what is left fits in that slack */
    s = -s;
    win = e->win;
    nbits = e->nwin;
    while (nbits > s) {
        push_raw(e, (uint8_t)win);
        win >>= 8;
        nbits -= 8;
    }
    if (e->err)
        return FFV2O_ERR_NOMEM;
    nrange = e->npre;
    total = nrange + e->nraw;
    if (total > cap)
        return FFV2O_ERR_NOSPACE;
    {   /* carry propagation, last word first (:706-715) */
        uint32_t carry = 0;
        for (size_t i = nrange; i-- > 0;) {
            carry += e->pre[i];
            out[i] = (uint8_t)carry;
            carry >>= 8;
        }
    }
    for (size_t i = 0; i < e->nraw; i++)
        out[total - 1 - i] = e->raw[i];
    if (nbits > 0) {
        if (!nrange)
            return FFV2O_ERR_ABORT;                        /* :719 */
        out[nrange - 1] |= (uint8_t)win;
    }
    *size = total;
    return 0;
}

/* ------------------------------------------------------------------ */
/* Exp-Golomb (ffv2enc.c:105-123)                                      */
/* ------------------------------------------------------------------ */
int ffv2o_golomb(uint32_t val, uint64_t *pattern)
{
    uint64_t v = (uint64_t)val + 1, pat = 0;
    int nb = 0, len = 0;
    while ((v >> (nb + 1)) != 0) nb++;                     /* nb = floor(log2(v)) */
    for (int i = nb - 1; i >= 0; i--) {
        pat |= (uint64_t)(((v >> i) & 1) << 1) << len;     /* two bits: [0, bit] */
        len += 2;
    }
    pat |= (uint64_t)1 << len;
    len += 1;
    if (pattern) *pattern = pat;
    return len;
}

static void put_golomb(OEnt *e, uint32_t val)
{
    /* note ffv2enc.c:109 `if (!val++)`: val+1 is taken in uint32, so
     * val = 0xFFFFFFFF wraps to 0; gains/coefficients never get there. */
    uint32_t v = val + 1;
    int nb = 0;
    if (val == 0) { oent_bits(e, 1, 1); return; }
    while ((v >> (nb + 1)) != 0) nb++;
    for (int i = nb - 1; i >= 0; i--)
        oent_bits(e, ((v >> i) & 1) << 1, 2);
    oent_bits(e, 1, 1);
}

/* ffv2enc.c:131-138,166,174 */
uint32_t ffv2o_coded_gain(int64_t igain)
{
    float fgain = sqrtf((float)igain) + FLT_EPSILON;
    float g = (float)(pow((double)fgain, (double)(1.0f / 1.5f)) / (double)1);
    return (uint32_t)g;
}

/* ------------------------------------------------------------------ */
/* PVQ search (libavcodec/x86/celt_pvq_search.asm:85-191,214-368,      */
/* INIT_XMM avx, USE_APPROXIMATION 0; libavutil/x86/x86util.asm:422,968)*/
/* PARITY UNPINNED: see ffv2_oracle.h.                                  */
/* ------------------------------------------------------------------ */
static float hsum4(const float v[4])
{
    float a = v[0] + v[2], b = v[1] + v[3];
    return a + b;
}

float ffv2o_pvq_search(float *X, int *y, int K, int N)
{
    int nv = (N + 3) / 4, n4 = nv * 4;
    float *ax = malloc(sizeof(float) * n4);
    float *fy = malloc(sizeof(float) * n4);
    float lane[4], Sx, Syy, Sxy, b;
    int sy[4] = { 0, 0, 0, 0 }, pulses;
    float lxy[4] = { 0, 0, 0, 0 }, lyy[4] = { 0, 0, 0, 0 };

    for (int i = 0; i < n4; i++)
        ax[i] = i < N ? fabsf(X[i]) : 0.0f;
    for (int l = 0; l < 4; l++)
        lane[l] = ax[(nv - 1) * 4 + l];
    for (int v = nv - 2; v >= 0; v--)
        for (int l = 0; l < 4; l++)
            lane[l] = lane[l] + ax[v * 4 + l];
    Sx = hsum4(lane);
    if (Sx == 0.0f || Sx != Sx) {                 /* comiss + jz: equal or unordered */
        for (int i = 0; i < n4; i++)
            if (i < N) y[i] = 0;
        free(ax); free(fy);
        return 1.0f;
    }
    b = (float)K / Sx;
    for (int v = nv - 1; v >= 0; v--)
        for (int l = 0; l < 4; l++) {
            int i = v * 4 + l;
            float t = b * ax[i];
            int yt = (int)lrintf(t);              /* cvtps2dq, round to nearest even */
            float fyt = (float)yt;
            float xy = ax[i] * fyt;
            float yy = fyt * fyt;
            sy[l] += yt;
            fy[i] = fyt;
            lxy[l] = lxy[l] + xy;
            lyy[l] = lyy[l] + yy;
        }
    Syy = hsum4(lyy);
    pulses = (sy[0] + sy[2]) + (sy[1] + sy[3]);
    K -= pulses;
    if (K != 0) {
        int add = K > 0;
        Sxy = hsum4(lxy);
        Syy = Syy * 0.5f;
        for (int it = add ? K : -K; it > 0; it--) {
            float pmax[4] = { 0, 0, 0, 0 };
            int   imax[4] = { 0, 1, 2, 3 };       /* byte offset 0 | lane offset */
            int best;
            Syy = Syy + 0.5f;
            for (int v = 0; v < nv; v++)
                for (int l = 0; l < 4; l++) {
                    int i = v * 4 + l;
                    float num, den, p;
                    if (add) {
                        den = fy[i] + Syy;
                        num = ax[i] + Sxy;
                    } else {
                        den = Syy - fy[i];
                        num = Sxy - ax[i];
                        if (!(0.0f < fy[i])) num = 0.0f;
                    }
                    num = num * num;
                    p = num / den;
                    if (pmax[l] < p) imax[l] = i; /* pand + pmaxsw: last strict improvement */
                    pmax[l] = pmax[l] > p ? pmax[l] : p;   /* maxps: second operand on tie/NaN */
                }
            /* lanes (3,2) replace (1,0) only when strictly greater */
            for (int l = 0; l < 2; l++)
                if (pmax[l] < pmax[l + 2]) { pmax[l] = pmax[l + 2]; imax[l] = imax[l + 2]; }
            /* lane 1 replaces lane 0 unless p1 < p0 (cmpss predicate 5 = NLT) */
            best = !(pmax[1] < pmax[0]) ? imax[1] : imax[0];
            if (add) {
                Sxy = Sxy + ax[best];
                Syy = Syy + fy[best];
                fy[best] = fy[best] + 1.0f;
            } else {
                Sxy = Sxy - ax[best];
                Syy = Syy - fy[best];
                fy[best] = fy[best] - 1.0f;
            }
        }
        Syy = Syy + Syy;
    }
    for (int i = 0; i < N; i++) {
        float v = fy[i];
        int iv = (int)lrintf(v);
        y[i] = signbit(X[i]) ? -iv : iv;          /* orps sign then cvtps2dq */
    }
    free(ax); free(fy);
    return Syy;
}

/* ------------------------------------------------------------------ */
/* quantiser + packet (ffv2enc.c:140-188,190-206,437-451,453-493)      */
/* ------------------------------------------------------------------ */
static int quant_block(OEnt *e, OCdf *test_cdf, const int32_t c[4096], int32_t W, int qp)
{
    int ret;
    put_golomb(e, (uint32_t)(c[0] < 0 ? -(int64_t)c[0] : c[0]));
    if (c[0])
        oent_bits(e, c[0] < 0, 1);
    for (int b = 0; b < NUM_BANDS; b++) {
        int lo = 1 + BANDS_START[b];
        int len = BANDS_START[b + 1] - BANDS_START[b];      /* 2049 for the last band */
        int64_t igain = 0;
        float fgain;
        for (int j = 0; j < len; j++) {
            /* temp2[4096]: SURVEY.md 8/A9.  The reference reads that word TWICE -- here for the energy
             * (ffv2enc.c:163-164) and again below for the normalised vector (:168-169).  One W serves both
             * reads in this model: it covers the stack layouts in which nothing writes the slot in between
             * (the survey's gcc build).  In a layout where the slot IS one of quant_block's own scratch
             * arrays (AMD clang: norm_coeffs[0], stored by the loop at :168 before its last iteration reads
             * src_c[2048]) the second read sees float bits instead, and that build aborts on every qp > 0
             * input -- not modelled, there is no output to match. */
            int32_t v = lo + j < 4096 ? c[lo + j] : W;
            igain += (int64_t)v * v;
        }
        fgain = sqrtf((float)igain) + FLT_EPSILON;
        put_golomb(e, (uint32_t)(float)(pow((double)fgain, (double)(1.0f / 1.5f)) / (double)1));
        if (qp > 0) {
            float norm[2049 + 7];
            int   yq[2049 + 7];
            int pcnt = 0;
            memset(norm, 0, sizeof(norm));
            memset(yq, 0, sizeof(yq));
            for (int j = 0; j < len; j++) {
                int32_t v = lo + j < 4096 ? c[lo + j] : W;
                norm[j] = v / fgain;
            }
            ffv2o_pvq_search(norm, yq, qp, len);
            for (int j = 0; j < len && pcnt < qp; j++) {
                int q = yq[j], a = q < 0 ? -q : q;
                if ((ret = oent_adapt(e, test_cdf, a, b, qp)) < 0)
                    return ret;
                if (q)
                    oent_bits(e, q < 0, 1);
                pcnt += a;
            }
        }
    }
    return 0;
}

int ffv2o_encode_frame(const uint8_t *const data[4], const ptrdiff_t linesize[4],
                       int width, int height, int pix_fmt, int qp,
                       const int32_t *W,
                       uint8_t *out, size_t out_cap, size_t *out_size)
{
    OFrame f;
    OEnt e;
    OCdf subdiv = { 0 }, test = { 0 };
    int ret = oframe_build(&f, data, linesize, width, height, pix_fmt);
    if (ret < 0)
        return ret;
    oent_init(&e);
    if ((ret = ocdf_init(&subdiv, 1, 4, 128, 2)) < 0 ||      /* ffv2enc.c:506 */
        (ret = ocdf_init(&test, 13, qp > 0 ? qp : 0, 64, 6)) < 0)  /* :461 */
        goto end;

    /* frame header (ffv2enc.c:447-451; AV_PIX_FMT_NB = 196) */
    if ((ret = oent_uint(&e, (uint32_t)pix_fmt, 196)) < 0)
        goto end;
    put_golomb(&e, (uint32_t)qp);

    for (int sby = 0; sby < f.nsy; sby++)
        for (int sbx = 0; sbx < f.nsx; sbx++) {
            if ((ret = oent_adapt(&e, &subdiv, 0, 0, 4)) < 0)   /* split = END, :222 */
                goto end;
            oent_bits(&e, 0, 4);                                 /* tx type DCT, :197 */
            for (int p = 0; p < f.planes; p++) {
                int32_t c[4096];
                size_t bp = ((size_t)sby * f.nsx + sbx) * f.planes + p;
                block_coeffs(&f, p, sbx, sby, c);
                if ((ret = quant_block(&e, &test, c, W ? W[bp] : 0, qp)) < 0)
                    goto end;
            }
        }
    ret = oent_done(&e, out, out_cap, out_size);
end:
    free(subdiv.cdf);
    free(test.cdf);
    oent_free(&e);
    oframe_free(&f);
    return ret;
}

/* ------------------------------------------------------------------ */
/* Decoder side (libavcodec/ffv2dec.c over daala_entropy.c) -- what the FATE-style     */
/* enc/dec report (tests/fate-run.sh:188-210) decodes the encoder's packets with.       */
/* Restated with the decoder's own quirks, because they are its observable behaviour:   */
/*  * dequant_block keeps ONE pulses[4096] per block-plane, zeroed once and indexed from */
/*    0 by every band (ffv2dec.c:103,118-136): slots a band does not read (its loop      */
/*    stops once qp pulses are in) still hold what EARLIER bands put there, and are      */
/*    scaled into this band's coefficients all the same.                                 */
/*  * mag /= sqrt(cnt) (:134): at qp == 0 no pulse is ever read, cnt = 0, mag = inf or   */
/*    NaN, every coefficient 0 * mag = NaN, and the float -> int32 store of NaN is what  */
/*    x86's cvttss2si makes of it, 0x80000000 (UB in C; spelt out below).                */
/*  * the last band is 2049 long: its last write lands one int32 past temp[] (:136);     */
/*    dropped here.                                                                      */
/*  * #define DEBUGGING (:88) is on in the reference: decode_sbs overwrites row 0 / column */
/*    0 of every superblock with -2048 (plane 0) or 0 (:258-273), and for 8-bit pictures */
/*    ffv2_decode_frame draws a text overlay that contains the decoding time (:361-372). */
/*    The grid is reproduced on request (FFV2O_DEC_GRID); the text is not reproducible.  */
/* ------------------------------------------------------------------ */
typedef struct {
    const uint8_t *b;
    size_t n, pos, epos;      /* range bytes read forward from pos, raw bytes backward from epos */
    uint64_t diff;
    uint32_t rng;
    int cnt;
    uint64_t win;
    int nwin;
    int err;
} ODec;

#define ODEC_ABUNDANCE 16384                                    /* DAALAENT_BIT_ABUNDANCE */

static void odec_fill(ODec *d)                                  /* daala_entropy.c:79-95 */
{
    int i = 64 - 9 - (d->cnt + 15);
    for (; i >= 0 && d->pos < d->n; i -= 8, d->pos++) {
        d->diff |= (uint64_t)d->b[d->pos] << i;
        d->cnt += 8;
    }
    if (d->pos >= d->n)
        d->cnt = ODEC_ABUNDANCE;
}

static void odec_init(ODec *d, const uint8_t *buf, size_t size) /* :564-578 */
{
    memset(d, 0, sizeof(*d));
    d->b = buf; d->n = size; d->epos = size;
    d->rng = 0x8000; d->cnt = -15;
    odec_fill(d);
}

static void odec_renorm(ODec *d, uint64_t diff, uint32_t rng)   /* :97-105 */
{
    const int i = 16 - ilog(rng);
    d->diff = diff << i;
    d->rng = rng << i;
    if ((d->cnt -= i) < 0)
        odec_fill(d);
}

static uint32_t osat(uint32_t a, uint32_t b) { return a - (a < b ? a : b); }   /* DAALAENT_SAT */

/* daalaent_decode_cdf (:273-326), CDF_UNSCALED (q15 == 0) or CDF_Q15 */
static int odec_cdf(ODec *d, const uint16_t *cdf, int n, int q15)
{
    uint32_t rng = d->rng, ft, dd, g;
    int scale, ret = 0;
    const uint64_t diff = d->diff;
    const int64_t cval = (int64_t)(diff >> 48);
    if ((uint64_t)cval >= rng) { d->err = 1; return 0; }         /* :283 av_assert0 */
    if (!q15) {
        ft = cdf[n - 1];
        if (ft < 2 || ft > 32768) { d->err = 1; return 0; }
        scale = 15 - ilog(ft - 1);
        ft <<= scale;
        if (ft > rng) { d->err = 1; return 0; }
        if (rng - ft >= ft) { ft <<= 1; scale++; }
        dd = rng - ft;
    } else {
        if (cdf[n - 1] != 32768 || rng < 32768) { d->err = 1; return 0; }
        dd = rng - 32768; ft = 32768; scale = 0;
    }
    g = osat(2 * dd, ft);
    {
        int64_t a = cval >> 1, b = cval - (int64_t)dd, c = (2 * cval + 1 - (int64_t)g) / 3;   /* C division: toward zero */
        int64_t lim = a > b ? a : b;
        uint32_t u = 0, v;
        if (c > lim) lim = c;
        lim >>= scale;
        for (v = cdf[ret]; (int64_t)v <= lim; v = cdf[++ret]) {
            u = v;
            if (ret + 1 >= n) { d->err = 1; return 0; }          /* the reference would run off the row */
        }
        u <<= scale; v <<= scale;
        {
            const uint32_t bu = osat(u, g) >> 1, bv = osat(v, g) >> 1;
            u = u + (u < g ? u : g) + (bu < dd ? bu : dd);
            v = v + (v < g ? v : g) + (bv < dd ? bv : dd);
        }
        odec_renorm(d, diff - ((uint64_t)u << 48), v - u);
    }
    return ret;
}

static uint32_t odec_bits(ODec *d, int num)                     /* :200-224 */
{
    int avail = d->nwin;
    uint64_t win = d->win;
    uint32_t ret;
    if (avail < num) {
        do {
            if (d->epos == 0) { avail = ODEC_ABUNDANCE; break; }
            win |= (uint64_t)d->b[--d->epos] << avail;
            avail += 8;
        } while (avail <= 64 - 8);
    }
    ret = (uint32_t)(win & (((uint64_t)1 << num) - 1));
    d->win = win >> num;
    d->nwin = avail - num;
    return ret;
}

static uint32_t odec_uint(ODec *d, uint32_t num)                /* :382-396, num > 16 */
{
    uint16_t cdf[16];
    int bit, adr, t;
    num--;
    bit = ilog(num) - 4;
    adr = (int)(num >> bit) + 1;
    for (int k = 0; k < adr; k++)                               /* daalatab.c:50-64, uniform Q15 rows */
        cdf[k] = (uint16_t)((32768u * (uint32_t)(k + 1) + (uint32_t)adr / 2) / (uint32_t)adr);
    cdf[adr - 1] = 32768;
    t = odec_cdf(d, cdf, adr, 1);
    t = (int)(((uint32_t)t << bit) | odec_bits(d, bit));
    if ((uint32_t)t <= num) return (uint32_t)t;
    d->err = 1;
    return num;
}

static int odec_adapt(ODec *d, uint16_t *cdf, int n, int inc)   /* :413-425 */
{
    const int r = odec_cdf(d, cdf, n, 0);
    if (d->err) return 0;
    if (cdf[n - 1] + inc > 32767)
        for (int i = 0; i < n; i++) cdf[i] = (uint16_t)((cdf[i] >> 1) + i + 1);
    for (int i = r; i < n; i++) cdf[i] = (uint16_t)(cdf[i] + inc);
    return r;
}

static uint32_t odec_golomb(ODec *d)                            /* ffv2dec.c:76-86 */
{
    uint32_t c = 1;
    int guard = 0;
    while (!odec_bits(d, 1)) {
        c = (c << 1) | odec_bits(d, 1);
        if (++guard > 40) { d->err = 1; break; }                 /* a truncated packet reads zeros for ever */
    }
    return c - 1;
}

/* float -> int32 as the reference binary stores it on x86-64 (cvttss2si): truncation, and the
 * "integer indefinite" 0x80000000 for NaN and for everything outside int32 */
static int32_t f2i_x86(float v)
{
    if (!(v > -2147483904.0f && v < 2147483648.0f))
        return INT32_MIN;
    return (int32_t)v;
}

/* Entropy layer + dequant_block of one packet: coding-order coefficients [nblk][4096]. */
int ffv2o_decode_coefficients(const uint8_t *pkt, size_t size, int width, int height, int expect_pix_fmt,
                              int *pix_fmt_out, int *qp_out, int32_t *coef)
{
    ODec d;
    int planes, depth;
    if (!pkt || !coef || width <= 0 || height <= 0) return FFV2O_ERR_PIXFMT;
    odec_init(&d, pkt, size);
    const int pix_fmt = (int)odec_uint(&d, 196);                 /* ffv2dec.c:276 */
    const int qp = (int)odec_golomb(&d);                         /* :277 */
    if (d.err || ffv2o_pixfmt_info(pix_fmt, &planes, &depth) < 0 || qp < 0 || qp > 32767)
        return FFV2O_ERR_ABORT;
    if (pix_fmt_out) *pix_fmt_out = pix_fmt;
    if (qp_out) *qp_out = qp;
    if (pix_fmt != expect_pix_fmt) return FFV2O_ERR_PIXFMT;      /* coef[] was sized for expect_pix_fmt's planes */
    const int nsx = (width + SB - 1) / SB, nsy = (height + SB - 1) / SB;
    uint16_t subdiv[4] = { 32, 64, 96, 128 };                    /* daalaent_cdf_alloc(1,4,128,0,2,0), reset :332 */
    uint16_t *test = malloc(sizeof(uint16_t) * 13 * (size_t)(qp > 0 ? qp : 1));
    int *pulses = malloc(sizeof(int) * 4097);
    if (!test || !pulses) { free(test); free(pulses); return FFV2O_ERR_NOMEM; }
    for (int r = 0; r < 13; r++)
        for (int j = 0; j < qp; j++) test[(size_t)r * qp + j] = (uint16_t)(j + 1);   /* daalaent_cdf_alloc(13,qp,64,0,6,0) */
    int rc = 0;
    for (int sb = 0; sb < nsx * nsy && !rc; sb++) {
        const int split = odec_adapt(&d, subdiv, 4, 128);        /* decode_block_rec :215 */
        if (d.err || split != 0) { rc = FFV2O_ERR_ABORT; break; } /* the encoder never splits (ffv2enc.c:272) */
        (void)odec_bits(&d, 4);                                  /* tx type, :162 */
        for (int p = 0; p < planes && !rc; p++) {
            int32_t *dst = coef + ((size_t)sb * planes + p) * 4096;
            memset(dst, 0, sizeof(int32_t) * 4096);
            memset(pulses, 0, sizeof(int) * 4097);               /* int pulses[4096] = { 0 }, once per block-plane */
            {
                const uint32_t c0 = odec_golomb(&d);             /* :109-111: dctcoef = uint32, then *= +-1 */
                int32_t v = (int32_t)c0;
                if (v) v = (int32_t)((uint32_t)v * (uint32_t)(1 - 2 * (int)odec_bits(&d, 1)));
                dst[0] = v;
            }
            for (int b = 0; b < NUM_BANDS; b++) {
                const int lo = 1 + BANDS_START[b], len = BANDS_START[b + 1] - BANDS_START[b];
                /* gain_expand(cg, 1, 1.5f): (float)pow((double)(cg * 1), (double)1.5f), cg a float */
                const float cg = (float)odec_golomb(&d);
                float mag = (float)pow((double)(cg * 1), (double)1.5f);
                int cnt = 0, pcnt = 0;
                for (int j = 0; j < len; j++) {
                    if (pcnt >= qp) break;
                    int q = odec_adapt(&d, test + (size_t)b * qp, qp, 64);
                    if (q) q *= 1 - 2 * (int)odec_bits(&d, 1);
                    pulses[j] = q;
                    pcnt += q < 0 ? -q : q;
                    cnt += q * q;
                }
                if (d.err) { rc = FFV2O_ERR_ABORT; break; }
                mag = (float)((double)mag / sqrt((double)cnt));  /* mag /= sqrt(cnt): float /= double */
                for (int j = 0; j < len && lo + j < 4096; j++)
                    dst[lo + j] = f2i_x86((float)pulses[j] * mag);
            }
        }
    }
    free(test);
    free(pulses);
    return rc;
}

/* ffv2_decode_frame (ffv2dec.c:315-377): packet -> picture planes of the packet's own pix_fmt
 * (which must be expect_pix_fmt: the caller sized the planes for it).  flags & FFV2O_DEC_GRID:
 * the DEBUGGING overwrite of every superblock's first row and column (:258-273). */
int ffv2o_decode_frame(const uint8_t *pkt, size_t size, int width, int height, int expect_pix_fmt, int flags,
                       uint8_t *const data[4], const ptrdiff_t linesize[4], int *qp_out)
{
    int planes, depth, pix_fmt = -1;
    if (ffv2o_pixfmt_info(expect_pix_fmt, &planes, &depth) < 0) return FFV2O_ERR_PIXFMT;
    const int nsx = (width + SB - 1) / SB, nsy = (height + SB - 1) / SB;
    int32_t *coef = malloc(sizeof(int32_t) * 4096 * (size_t)nsx * nsy * planes);
    if (!coef) return FFV2O_ERR_NOMEM;
    int rc = ffv2o_decode_coefficients(pkt, size, width, height, expect_pix_fmt, &pix_fmt, qp_out, coef);
    if (!rc) rc = ffv2o_inverse_tstage(coef, width, height, pix_fmt, data, linesize);
    free(coef);
    if (!rc && (flags & FFV2O_DEC_GRID)) {
        for (int p = 0; p < planes; p++) {
            const int v = ((p ? 0 : -2048) + 2048) >> (12 - depth);
            for (int y = 0; y < height; y++) {
                uint8_t *row = data[p] + (ptrdiff_t)y * linesize[p];
                for (int x = 0; x < width; x++) {
                    if ((x & 63) && (y & 63)) continue;
                    if (depth == 8) row[x] = (uint8_t)v;
                    else { const uint16_t w = (uint16_t)v; memcpy(row + 2 * x, &w, 2); }
                }
            }
        }
    }
    return rc;
}
