// Internal launcher interface between the C-ABI (ffv2_capi.cpp) and the gfx950
// kernels (ffv2_kernels.hip).  Not installed; the public surface is include/ffv2_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FFV2_NUM_BANDS     13
#define FFV2_CODES_PER_BP  16   // uint32 slots per block-plane record (see below)

// Per-block-plane record written by the T-stage for the E-stage:
//   [0]      c0 as int32 (coding index 0 = coefficient (x=0,y=1))
//   [1..13]  coded band gains (ffv2enc.c:174)
//   [14]     raw bits this block-plane emits (its 14 Exp-Golomb codes + sign)
//   [15]     0
struct FFV2Geom {
    int width, height, depth, planes, bytes_per_sample;
    int nsx, nsy, nblk;               // nblk = nsx*nsy*planes
    size_t row_pitch, plane_stride, frame_stride;
    uint32_t inv_planes, inv_nsx;     // floor(2^32/d)+1 for d = planes, nsx (unused when d == 1)
};

struct FFV2TStageArgs {
    FFV2Geom g;
    int nframes;
    const uint8_t *frames;
    int32_t  *coef;                   // optional [nframes][nblk][4096]
    int64_t  *energy;                 // optional [nframes][nblk][13]
    uint32_t *codes;                  // optional [nframes][nblk][16]
    uint32_t *bitcnt;                 // with codes: [nframes][nblk] raw bits per block-plane
    const int32_t *W;                 // optional [nframes][nblk]
    const int64_t *gain_thr;          // gain_thr[n] = least energy whose coded gain is >= n+1
    int gain_n;                       // entries in gain_thr
    const uint16_t *lds_scan;         // [8][64][8] byte offsets into the LDS raster, see ffv2_capi.cpp
    int32_t *status;                  // [nframes] sticky per-frame error
    uint32_t *zero;                   // optional: packet buffers the E-stage ORs into, cleared here (saves a memset launch)
    uint32_t zero_stride_dw;          // dwords per frame in `zero`
};

struct FFV2EStageArgs {
    FFV2Geom g;
    int nframes;
    const uint32_t *codes;            // [nframes][nblk][16]
    const uint32_t *bitoff;           // [nframes][nblk] raw bits per block-plane (T-stage bitcnt)
    uint8_t  *packets;                // [nframes][packet_stride], zeroed beforehand (FFV2TStageArgs::zero)
    size_t    packet_stride;
    uint32_t *sizes;                  // [nframes]
    int32_t  *status;                 // [nframes] written (not updated) by the E-stage
    int32_t  *err;                    // [nframes] the T-stage's sticky error flags (its FFV2TStageArgs::status); read and cleared
    const uint8_t *prefix;            // range-coded prefix bytes (data independent at qp 0)
    int prefix_len, slack_bits;
    uint32_t header_bits, header_nbits; // raw bits in front of the first superblock
};

#define FFV2_Y_STRIDE 4104        // int16 per block-plane of the PVQ output (4096 + phantom slot, padded)

hipError_t ffv2_launch_tstage(const FFV2TStageArgs &a, hipStream_t s);
void ffv2_tstage_force_variant(int mode);   // tests: 0 one-block kernel, 1 column-walking kernel, -1 automatic
const char *ffv2_tstage_kernel_name(const FFV2Geom &g, int nframes, bool coef_writeback);   // which T-stage kernel a launch would use
hipError_t ffv2_launch_inverse(const FFV2Geom &g, int nframes, const int32_t *coef, int32_t *plane,
                               uint8_t *frames, const uint16_t *lds_scan, hipStream_t s);
// decoder-side dequantisation (ffv2dec.c:135-136) in front of ffv2_launch_inverse: see ffv2_inverse.hip
hipError_t ffv2_launch_dequant(const int16_t *pulses, const float *mag, const int32_t *c0, int32_t *coef, long long nbp,
                               hipStream_t s);
// Per-block-plane index into a frame's compact symbol stream (qp > 0): the int8 pulses the
// range coder will read, band after band, start at stream[offset]; count[b] of them in band b.
struct FFV2SymRec {
    uint32_t offset;
    uint16_t count[FFV2_NUM_BANDS];
    uint16_t pad;
};
hipError_t ffv2_launch_compact(const int16_t *y, int qp, int nblk, int nframes, FFV2SymRec *rec, int8_t *stream,
                               size_t stream_stride, uint32_t *totals, hipStream_t s);
hipError_t ffv2_launch_pvq(const int32_t *coef, const int32_t *W, int16_t *y, int qp, long long nbp, hipStream_t s);
hipError_t ffv2_launch_pvq_counted(const int32_t *coef, const int32_t *W, int16_t *y, int qp, long long nbp, int nblk,
                                   const uint32_t *codes, FFV2SymRec *cnt, uint32_t *bits, int32_t *abort_, hipStream_t s);
hipError_t ffv2_launch_pvq_vectors(const float *X, int stride, int N, int K, int count, int16_t *y, hipStream_t s);
hipError_t ffv2_launch_estage_qp0(const FFV2EStageArgs &a, hipStream_t s);

// 4:2:0 -> 4:4:4 up-conversion in front of the T-stage (ffv2_upconv.hip)
struct FFV2Upconv;
FFV2Upconv *ffv2_upconv_create(int w, int h, int depth);
void ffv2_upconv_destroy(FFV2Upconv *u);
size_t ffv2_upconv_src_frame_bytes(int w, int h, int depth);
hipError_t ffv2_launch_upconv(const FFV2Upconv *u, const FFV2Geom &g, int nframes, const uint8_t *src,
                              size_t src_frame_stride, uint8_t *dst, hipStream_t s);
hipError_t ffv2_launch_upconv_chroma(const FFV2Upconv *u, const FFV2Geom &g, int nframes, const uint8_t *src_u,
                                     size_t c_pitch, size_t c_plane_stride, size_t src_frame_stride, uint8_t *dst,
                                     hipStream_t s);

// qp > 0 entropy coder on the device (ffv2_rangecoder.hip): one wavefront per frame
struct FFV2RangeCoderArgs {
    const uint32_t *codes;            // [nframes][nblk][16] T-stage records
    const FFV2SymRec *rec;            // [nframes][nblk]
    const int8_t *stream;             // [nframes][stream_stride] compact pulses
    size_t stream_stride;
    const int32_t *status_in;         // [nframes] T-stage status (a failed frame is skipped)
    uint16_t *pre;                    // [nframes][cap] scratch: pre-carry range words
    uint8_t *raw;                     // [nframes][cap] scratch: raw bytes in write order
    size_t cap;
    uint8_t *packets;                 // [nframes][packet_stride]
    size_t packet_stride;
    uint32_t *sizes;                  // [nframes]
    int32_t *status;                  // [nframes]
    int nblk, nsb, planes, pix_fmt, qp;
};
hipError_t ffv2_launch_rangecoder(const FFV2RangeCoderArgs &a, int nframes, hipStream_t s);

// qp > 0 entropy coder with many frames in flight (ffv2_lanecoder.hip): the serial range chain
// runs one frame per lane, everything else is data parallel.  Frames in flight are numbered
// 0..F-1; `width` of them share a wavefront of the chain kernel.
struct FFV2LaneState { uint32_t woff, o, rng, full, acc, pad0, pad1, pad2; };   // where the next symbol would land (word, bit), the range, words exhausted, the word so far
struct FFV2LaneCoderArgs {
    int nblk, planes, qp, width, f0;
    const uint32_t *codes;            // [F][nblk][16] T-stage records
    const int32_t *status_in;         // [F] T-stage status
    int32_t *abort_;                  // [F] != 0: the reference would av_assert0 (daala_entropy.c:336)
    FFV2SymRec *cnt;                  // [F][nblk] count[13] = pulses read per band, offset = their sum
    uint32_t *bits;                   // [F][nblk] raw bits of the block-plane
    uint32_t *rowbase;                // [F][13][nblk+1]
    uint32_t *gbase;                  // [F][nblk+1]
    uint32_t *rawbase;                // [F][nblk+1]
    uint32_t *delta;                  // [F][13][nblk] coding-order index minus row index of a band's first symbol
    uint8_t *rows;                    // [F][row_stride] |pulse| per CDF row, rows back to back
    size_t row_stride;
    uint2 *recs;                      // coding-order records {fl | fh << 16, ft} of ONE WINDOW of the coding order, interleaved
                                      // (lc_record); two buffers, buf_stride uint2 apart (cdf of window i+1 beside the chain of window i)
    size_t group_stride;              // uint2 per group of `width` frames (window symbols x width)
    size_t buf_stride;
    uint32_t win0, win1;              // this launch's window: symbols [win0, win1) of every frame's coding order
    int win_buf;                      // which record buffer
    uint32_t *cdfstate;               // [F][13][68] a CDF row between windows: 64 entries, total, position, block-plane
    const uint2 *split;               // [superblocks] the data-independent "no split" symbols (ffv2enc.c:222)
    uint2 header;                     // ff_daalaent_encode_uint(pix_fmt, 196)'s range-coded part (ffv2enc.c:449)
    uint32_t header_bits, header_nbits; // raw: pix_fmt & 15, Exp-Golomb(qp)
    uint32_t *raw;                    // [F][raw_words] raw-bit tail, LSB first, cleared beforehand
    uint32_t raw_words;
    uint32_t *words;                  // [F][wcap] the range code as anchored 32-bit words
    uint32_t wcap;
    FFV2LaneState *state;             // [F]
    uint8_t *packets;                 // packed: packet f at packets + offs[f] (16-byte aligned); capacity F * packet_stride
    size_t packet_stride;             // most one packet may take
    uint32_t *sizes;                  // [F]
    int32_t *status;                  // [F]
    unsigned long long *offs;         // [F + 1] packet offsets, [F] = bytes in all
    uint4 *fin;                       // [F] range bytes, slack bits, bytes the carry chain covers
};
hipError_t ffv2_launch_lc_front(const FFV2LaneCoderArgs &a, const int16_t *y, int nframes, bool counted, hipStream_t s);   // count, scan, scatter of frames f0..
// the back of frames 0..nframes-1, window by window (symbols [w0, w1) of every frame, record buffer buf), then finish
hipError_t ffv2_launch_lc_cdf(const FFV2LaneCoderArgs &a, int nframes, uint32_t w0, uint32_t w1, int buf, hipStream_t s);
hipError_t ffv2_launch_lc_chain(const FFV2LaneCoderArgs &a, int nframes, uint32_t w0, uint32_t w1, int buf, hipStream_t s);
hipError_t ffv2_launch_lc_finish(const FFV2LaneCoderArgs &a, int nframes, hipStream_t s);

// The T-stage of ONE frame in plain wrapping int32 (ffv2_wide.hip): any 16-bit sample, any gain.
// plane: int32 [planes][64 nsy][64 nsx] workspace; coef optional [nblk][4096]; energy [nblk][13]
// (phantom W excluded); c0 [nblk].
hipError_t ffv2_launch_wide_tstage(const FFV2Geom &g, const uint8_t *d_frame, int32_t *plane, int32_t *coef,
                                   int64_t *energy, int32_t *c0, hipStream_t s);
