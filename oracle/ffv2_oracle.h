/*
 * ffv2_oracle.h -- CPU restatement of the FFV2 encode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
 * executed by the product (ffmpeg_ffv2_amd/, include/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only
 * as the checker / the CPU baseline, never as the thing shipped.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   qp == 0 : pinned by the seven known-answer packets of SURVEY.md section 8
 *             (outputs of the compiled reference, captured by the survey in this
 *             container) -- tests/test_oracle_kat.py -- and, for the 1-D
 *             transform, by tests/golden/fdct64_vectors.npz (the reference's
 *             own statement text executed numerically, tools/derive_lifting_ir.py).
 *   qp  > 0 : PARITY UNPINNED.  The PVQ search restates x86/celt_pvq_search.asm,
 *             which cannot be assembled here (no nasm/yasm) and for which the
 *             reference holds no test or vector.
 *
 * The reference cannot be compiled by gcc on its own files (it needs the
 * configure-generated config.h / libavutil/avconfig.h), so there is no
 * oracle/_ref build.
 */
#ifndef FFV2_ORACLE_H
#define FFV2_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* AVPixelFormat values of the reference tree (libavutil/pixfmt.h). */
enum {
    FFV2O_PIX_GRAY8       = 8,
    FFV2O_PIX_YUV444P     = 5,
    FFV2O_PIX_YUV444P10LE = 70,
    FFV2O_PIX_YUV444P12LE = 133,
    FFV2O_PIX_GBRP        = 73,
    FFV2O_PIX_GBRP10LE    = 77,
    FFV2O_PIX_GBRP12LE    = 137,
};

#define FFV2O_ERR_PIXFMT   (-1)  /* not in allowed_pix_fmts (ffv2enc.c:596-601)      */
#define FFV2O_ERR_NOSPACE  (-2)  /* caller's output buffer too small                  */
#define FFV2O_ERR_ABORT    (-3)  /* the reference would hit av_assert0 -> abort()     */
#define FFV2O_ERR_NOMEM    (-4)

/* planes / bit depth of a pix_fmt; <0 when the encoder would reject it. */
int ffv2o_pixfmt_info(int pix_fmt, int *planes, int *depth);

/* 1-D pieces (checkasm-style unit tests). */
void ffv2o_fdct64(int32_t y[64], const int32_t *x, int xstride);   /* ffv2.c:4678 */
void ffv2o_lap_filter32(int32_t y[32], const int32_t x[32]);       /* ffv2.c:183  */
/* Exp-Golomb as coded by ffv2enc.c:105-123: returns code length, writes the
 * raw-bit pattern (LSB = first bit emitted) to *pattern (length <= 63). */
int  ffv2o_golomb(uint32_t val, uint64_t *pattern);
/* gain as coded: (uint32)(float)pow(sqrtf(igain)+FLT_EPSILON, 1/1.5f) (ffv2enc.c:166,174) */
uint32_t ffv2o_coded_gain(int64_t igain);

/* T-stage: level shift, lapping, 2-D DCT, scan, band energies.
 *   coef   : [nsb*planes][4096] coding-order coefficients, block-plane index
 *            bp = (sby*nsx + sbx)*planes + p                          (may be NULL)
 *   energy : [nsb*planes][13]   sum of squares per band, phantom W NOT included
 *                                                                     (may be NULL) */
int ffv2o_tstage(const uint8_t *const data[4], const ptrdiff_t linesize[4],
                 int width, int height, int pix_fmt,
                 int32_t *coef, int64_t *energy);

/* Decoder-side inverse of the T-stage (ffv2.c:81-98, :4962-4972, :216-239 with the
 * seam order of ffv2dec.c, :40-52): coding-order coefficients -> picture planes.
 * Not an exact inverse (the post-filter divides with truncation); used for round-trip
 * self checks. */
void ffv2o_idct64(int32_t *x, int xstride, const int32_t y[64]);
void ffv2o_inv_lap_filter32(int32_t x[32]);
int ffv2o_inverse_tstage(const int32_t *coef, int width, int height, int pix_fmt,
                         uint8_t *const data[4], const ptrdiff_t linesize[4]);

/* Whole frame -> one packet (ffv2enc.c:453-493).
 *   W : optional phantom coefficient per block-plane, NULL = 0. */
int ffv2o_encode_frame(const uint8_t *const data[4], const ptrdiff_t linesize[4],
                       int width, int height, int pix_fmt, int qp,
                       const int32_t *W,
                       uint8_t *out, size_t out_cap, size_t *out_size);

/* Decoder side (ffv2dec.c:76-136,275-280,315-377 over daala_entropy.c:79-105,200-224,273-326,382-396,
 * 413-425,564-578), for the FATE-style encode -> decode report: the entropy layer and dequant_block of
 * one packet -> coding-order coefficients [nblk][4096]; and the whole frame.  Reproduces the decoder's
 * quirks (stale pulses between bands, mag / sqrt(0) at qp 0 with x86's NaN -> int32 store), see the .c.
 * FFV2O_DEC_GRID: also the reference's `#define DEBUGGING` overwrite of each superblock's first row and
 * column.  PARITY UNPINNED (the reference holds no decoded-output hash for FFV2). */
#define FFV2O_DEC_GRID 1
int ffv2o_decode_coefficients(const uint8_t *pkt, size_t size, int width, int height, int expect_pix_fmt,
                              int *pix_fmt_out, int *qp_out, int32_t *coef);
int ffv2o_decode_frame(const uint8_t *pkt, size_t size, int width, int height, int expect_pix_fmt, int flags,
                       uint8_t *const data[4], const ptrdiff_t linesize[4], int *qp_out);

/* PVQ search restating ff_pvq_search_exact_avx (celt_pvq_search.asm:214-368). */
float ffv2o_pvq_search(float *X, int *y, int K, int N);

/* ffv2_swscale_oracle.c: what the reference tool chain does to a 4:2:0 frame before encode2()
 * (auto-inserted scale filter, flags=bicubic: libswscale/utils.c:332-727 initFilter,
 * swscale.c:96-139, output.c:333-393).  PARITY UNPINNED. */
int ffv2o_sws_chroma_filter(int n, int one, int16_t *filter_out, int32_t *pos_out, int cap_taps);
int ffv2o_sws_420_to_444(const uint8_t *const src[3], const ptrdiff_t src_stride[3],
                         uint8_t *const dst[3], const ptrdiff_t dst_stride[3], int w, int h, int depth);

#ifdef __cplusplus
}
#endif
#endif
