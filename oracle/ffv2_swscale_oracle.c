/*
 * ffv2_swscale_oracle.c -- CPU restatement of what the reference tool chain does to a
 * yuv420p / yuv420p10le / yuv420p12le frame before ffv2's encode2() sees it.
 *
 * TEST INFRASTRUCTURE ONLY (see ffv2_oracle.h): never linked or called by the product.
 *
 * PARITY UNPINNED: the reference holds no fixture for this conversion and no libswscale
 * binary can be run here; the restatement follows the C code paths read below.
 *
 * The path (reference files, FFmpeg 4.2 tree):
 *   fftools/ffmpeg_filter.c:63-131   choose_pixel_fmt(): the encoder's pix_fmts do not hold
 *                                    4:2:0, avcodec_find_best_pix_fmt_of_2 picks yuv444p* of
 *                                    the same depth; libavfilter inserts a scale filter with
 *                                    the tool's default "flags=bicubic".
 *   libavfilter/vf_scale.c:534-577   chroma positions: -513 (default) horizontally; 128
 *                                    vertically for yuv420p, the default for the deeper ones
 *                                    (which get_local_pos turns into the same 128).
 *   libswscale/utils.c:303-310       get_local_pos(): srcPos = dstPos = 128 both ways.
 *   libswscale/utils.c:1409-1444     chrSrcW/H = ceil(W/2), ceil(H/2); chrXInc, chrYInc.
 *   libswscale/utils.c:332-727       initFilter(): the SWS_BICUBIC branch (B = 0, C = 0.6),
 *                                    filter-size reduction, border folding, normalisation
 *                                    with error feedback.  filterAlign 1 (C code; MMX's 4/2
 *                                    give the same 4 taps), no src/dst filter vectors.
 *   libswscale/swscale.c:96-139      hScale8To15_c / hScale16To15_c: 14-bit filter, >> 7 resp.
 *                                    >> (depth-1), clipped above to 32767 only.
 *   libswscale/output.c:333-393      yuv2planeX_8_c (dither = the constant 64: sources of 8
 *                                    bits are not dithered, swscale.c:263,346) and
 *                                    yuv2planeX_10_c_template (also used for 12 bits).
 *   luma                             lumXInc = lumYInc = 1 << 16 with equal positions: filter
 *                                    size 1, (pix << 7 + 64) >> 7 resp. (pix << (15-d) +
 *                                    (1 << (14-d))) >> (15-d): the identity.
 *   libswscale/swscale_unscaled.c:2122-2137  no special converter applies (the subsampling
 *                                    differs), so the generic scaler above is what runs.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SWS_MAX_REDUCE_CUTOFF 0.002            /* swscale.h:87 */

static int av_log2_i(unsigned v) { int n = 0; while (v >>= 1) n++; return n; }
static int64_t i64abs(int64_t v) { return v < 0 ? -v : v; }
/* libavutil/common.h ROUNDED_DIV */
static int64_t rounded_div(int64_t a, int64_t b) { return (a >= 0 ? a + (b >> 1) : a - (b >> 1)) / b; }

/* initFilter (utils.c:332-727) for flags = SWS_BICUBIC, default params, filterAlign 1,
 * no src/dst filter.  Returns the filter size; *out_filter is [dstW][size] int16,
 * *out_pos [dstW] int32 (both malloc'ed). */
static int init_filter_bicubic(int16_t **out_filter, int32_t **out_pos, int xInc, int srcW, int dstW,
                               int one, int srcPos, int dstPos)
{
    int shiftv = av_log2_i((unsigned)(srcW / dstW));
    if (shiftv > 8) shiftv = 8;
    const int64_t fone = 1LL << (54 - shiftv);
    int filterSize, i, j;
    int64_t *filter, *filter2;
    int32_t *pos = malloc(sizeof(int32_t) * (size_t)(dstW + 3));
    const int d_unscaled = xInc - 0x10000;
    if ((d_unscaled < 0 ? -d_unscaled : d_unscaled) < 10 && srcPos == dstPos) {      /* :355 unscaled */
        filterSize = 1;
        filter = calloc((size_t)dstW, sizeof(int64_t));
        for (i = 0; i < dstW; i++) { filter[i] = fone; pos[i] = i; }
    } else {
        const int sizeFactor = 4;                                   /* scale_algorithms[]: bicubic */
        int64_t xDstInSrc;
        if (xInc <= 1 << 16) filterSize = 1 + sizeFactor;           /* upscale */
        else                 filterSize = 1 + (sizeFactor * srcW + dstW - 1) / dstW;
        if (filterSize > srcW - 2) filterSize = srcW - 2;
        if (filterSize < 1) filterSize = 1;
        filter = malloc(sizeof(int64_t) * (size_t)dstW * (size_t)filterSize);
        xDstInSrc = ((dstPos * (int64_t)xInc) >> 7) - ((srcPos * 0x10000LL) >> 7);
        for (i = 0; i < dstW; i++) {
            int xx = (int)((xDstInSrc - (filterSize - 2) * (1LL << 16)) / (1 << 17));
            pos[i] = xx;
            for (j = 0; j < filterSize; j++) {
                int64_t d = i64abs(((int64_t)xx * (1 << 17)) - xDstInSrc) << 13;
                int64_t coeff;
                const int64_t B = 0;                                /* param[0] default */
                const int64_t C = (int64_t)(0.6 * (1 << 24));       /* param[1] default */
                if (xInc > 1 << 16)
                    d = d * dstW / srcW;
                if (d >= 1LL << 31) {
                    coeff = 0;
                } else {
                    const int64_t dd = (d * d) >> 30;
                    const int64_t ddd = (dd * d) >> 30;
                    if (d < 1LL << 30)
                        coeff = (12 * (1 << 24) - 9 * B - 6 * C) * ddd +
                                (-18 * (1 << 24) + 12 * B + 6 * C) * dd +
                                (6 * (1 << 24) - 2 * B) * (1LL << 30);
                    else
                        coeff = (-B - 6 * C) * ddd + (6 * B + 30 * C) * dd +
                                (-12 * B - 48 * C) * d + (8 * B + 24 * C) * (1LL << 30);
                }
                coeff /= (1LL << 54) / fone;
                filter[(size_t)i * filterSize + j] = coeff;
                xx++;
            }
            xDstInSrc += 2 * xInc;
        }
    }
    /* no src/dst filter: filter2 == filter, filterPos unchanged (:516-543) */
    const int filter2Size = filterSize;
    filter2 = filter;
    /* step 1 of the size reduction (:545-584) */
    int minFilterSize = 0;
    for (i = dstW - 1; i >= 0; i--) {
        int min = filter2Size;
        int64_t cutOff = 0;
        for (j = 0; j < filter2Size; j++) {
            int k;
            cutOff += i64abs(filter2[(size_t)i * filter2Size]);
            if ((double)cutOff > SWS_MAX_REDUCE_CUTOFF * (double)fone) break;
            if (i < dstW - 1 && pos[i] >= pos[i + 1]) break;
            for (k = 1; k < filter2Size; k++)
                filter2[(size_t)i * filter2Size + k - 1] = filter2[(size_t)i * filter2Size + k];
            filter2[(size_t)i * filter2Size + k - 1] = 0;
            pos[i]++;
        }
        cutOff = 0;
        for (j = filter2Size - 1; j > 0; j--) {
            cutOff += i64abs(filter2[(size_t)i * filter2Size + j]);
            if ((double)cutOff > SWS_MAX_REDUCE_CUTOFF * (double)fone) break;
            min--;
        }
        if (min > minFilterSize) minFilterSize = min;
    }
    filterSize = minFilterSize;                                     /* filterAlign 1 */
    filter = malloc(sizeof(int64_t) * (size_t)dstW * (size_t)filterSize);
    for (i = 0; i < dstW; i++)
        for (j = 0; j < filterSize; j++)
            filter[(size_t)i * filterSize + j] = j >= filter2Size ? 0 : filter2[(size_t)i * filter2Size + j];
    free(filter2);
    /* fix borders (:630-671) */
    for (i = 0; i < dstW; i++) {
        if (pos[i] < 0) {
            for (j = 1; j < filterSize; j++) {
                int left = j + pos[i] > 0 ? j + pos[i] : 0;
                filter[(size_t)i * filterSize + left] += filter[(size_t)i * filterSize + j];
                filter[(size_t)i * filterSize + j] = 0;
            }
            pos[i] = 0;
        }
        if (pos[i] + filterSize > srcW) {
            int shift = pos[i] + (filterSize - srcW < 0 ? filterSize - srcW : 0);
            int64_t acc = 0;
            for (j = filterSize - 1; j >= 0; j--)
                if (pos[i] + j >= srcW) { acc += filter[(size_t)i * filterSize + j]; filter[(size_t)i * filterSize + j] = 0; }
            for (j = filterSize - 1; j >= 0; j--)
                filter[(size_t)i * filterSize + j] = j < shift ? 0 : filter[(size_t)i * filterSize + j - shift];
            pos[i] -= shift;
            filter[(size_t)i * filterSize + srcW - 1 - pos[i]] += acc;
        }
    }
    /* normalise with error feedback (:679-698) */
    int16_t *of = calloc((size_t)(dstW + 3) * (size_t)filterSize, sizeof(int16_t));
    for (i = 0; i < dstW; i++) {
        int64_t error = 0, sum = 0;
        for (j = 0; j < filterSize; j++) sum += filter[(size_t)i * filterSize + j];
        sum = (sum + one / 2) / one;
        if (!sum) sum = 1;
        for (j = 0; j < filterSize; j++) {
            const int64_t v = filter[(size_t)i * filterSize + j] + error;
            const int intV = (int)rounded_div(v, sum);
            of[(size_t)i * filterSize + j] = (int16_t)intV;
            error = v - intV * sum;
        }
    }
    free(filter);
    *out_filter = of;
    *out_pos = pos;
    return filterSize;
}

/* exported for the tests: the chroma filter of one axis (n = luma extent) */
int ffv2o_sws_chroma_filter(int n, int one, int16_t *filter_out, int32_t *pos_out, int cap_taps)
{
    const int srcN = (n + 1) >> 1;
    const int inc = (int)((((int64_t)srcN << 16) + (n >> 1)) / n);
    int16_t *f; int32_t *p;
    const int fs = init_filter_bicubic(&f, &p, inc, srcN, n, one, 128, 128);
    if (fs <= cap_taps) {
        memcpy(filter_out, f, sizeof(int16_t) * (size_t)n * (size_t)fs);
        memcpy(pos_out, p, sizeof(int32_t) * (size_t)n);
    }
    free(f); free(p);
    return fs;
}

/* yuv420p{,10le,12le} -> yuv444p{,10le,12le}.  src[0..2]: Y (w x h), U, V (ceil(w/2) x ceil(h/2));
 * strides in bytes; samples uint8 (depth 8) or native-endian uint16.  dst likewise, all w x h. */
int ffv2o_sws_420_to_444(const uint8_t *const src[3], const ptrdiff_t src_stride[3],
                         uint8_t *const dst[3], const ptrdiff_t dst_stride[3], int w, int h, int depth)
{
    if (w < 1 || h < 1 || (depth != 8 && depth != 10 && depth != 12)) return -22;
    const int bps = depth > 8 ? 2 : 1;
    const int cw = (w + 1) >> 1, ch = (h + 1) >> 1;
    int16_t *hf, *vf; int32_t *hp, *vp;
    const int xinc = (int)((((int64_t)cw << 16) + (w >> 1)) / w);
    const int yinc = (int)((((int64_t)ch << 16) + (h >> 1)) / h);
    const int hfs = init_filter_bicubic(&hf, &hp, xinc, cw, w, 1 << 14, 128, 128);
    const int vfs = init_filter_bicubic(&vf, &vp, yinc, ch, h, 1 << 12, 128, 128);
    for (int y = 0; y < h; y++)                                     /* luma: identity */
        memcpy(dst[0] + y * dst_stride[0], src[0] + y * src_stride[0], (size_t)w * bps);
    int16_t *hbuf = malloc(sizeof(int16_t) * (size_t)ch * (size_t)w);
    for (int p = 1; p < 3; p++) {
        for (int y = 0; y < ch; y++) {                              /* hScale8To15_c / hScale16To15_c */
            const uint8_t *row = src[p] + y * src_stride[p];
            for (int x = 0; x < w; x++) {
                int val = 0;
                for (int j = 0; j < hfs; j++) {
                    const int s = bps == 1 ? row[hp[x] + j] : ((const uint16_t *)row)[hp[x] + j];
                    val += s * hf[(size_t)x * hfs + j];
                }
                val >>= bps == 1 ? 7 : depth - 1;
                hbuf[(size_t)y * w + x] = (int16_t)(val < 32767 ? val : 32767);
            }
        }
        for (int y = 0; y < h; y++) {                               /* yuv2planeX_8_c / _10_c_template */
            uint8_t *orow = dst[p] + y * dst_stride[p];
            for (int x = 0; x < w; x++) {
                if (bps == 1) {
                    int val = 64 << 12;
                    for (int j = 0; j < vfs; j++) val += hbuf[(size_t)(vp[y] + j) * w + x] * vf[(size_t)y * vfs + j];
                    val >>= 19;
                    orow[x] = (uint8_t)(val < 0 ? 0 : val > 255 ? 255 : val);
                } else {
                    const int shift = 11 + 16 - depth;
                    int val = 1 << (shift - 1);
                    for (int j = 0; j < vfs; j++) val += hbuf[(size_t)(vp[y] + j) * w + x] * vf[(size_t)y * vfs + j];
                    val >>= shift;
                    const int hi = (1 << depth) - 1;
                    ((uint16_t *)orow)[x] = (uint16_t)(val < 0 ? 0 : val > hi ? hi : val);
                }
            }
        }
    }
    free(hbuf); free(hf); free(vf); free(hp); free(vp);
    return 0;
}
