/*
 * ffv2_amd.h -- C-ABI of the MI355X-native FFV2 encode hot path.
 *
 * This is the drop-in boundary for the path SURVEY.md section 8 scopes: what
 * libavcodec/ffv2enc.c does between AVCodec.init / .encode2 / .close
 * (reference libavcodec/ffv2enc.c:495-513, :453-493, :515-580; the AVCodec
 * contract itself is libavcodec/avcodec.h:3546-3693, caller encode.c:264-346).
 * Plain pointers and sizes only: no torch, no C++ types.  INTEGRATION.md shows
 * the ~40-line AVCodec glue a maintainer adds on the FFmpeg side.
 *
 * All entry points return 0 on success or a negative FFV2AMD_ERR_* code; the
 * values follow AVERROR(errno) so they can be returned from encode2() as is.
 * There is no CPU fallback: without a usable HIP device every call that needs
 * the GPU fails with FFV2AMD_ERR_DEVICE.
 */
#ifndef FFV2_AMD_H
#define FFV2_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFV2AMD_OK               0
#define FFV2AMD_ERR_INVAL      (-22)  /* AVERROR(EINVAL): bad geometry / pix_fmt / argument     */
#define FFV2AMD_ERR_NOMEM      (-12)  /* AVERROR(ENOMEM)                                        */
#define FFV2AMD_ERR_DEVICE     (-5)   /* AVERROR(EIO): HIP runtime / device failure             */
#define FFV2AMD_ERR_NOSPACE    (-28)  /* AVERROR(ENOSPC): caller's packet buffer too small      */
#define FFV2AMD_ERR_RANGE      (-34)  /* AVERROR(ERANGE): a sample exceeds the declared depth, or a band
                                          gain left the quantiser table, on a path that cannot rerun
                                          the frame in int32 (see ffv2amd_tstage_wide_device)       */
#define FFV2AMD_ERR_ABORT      (-1)   /* AVERROR(EPERM): the reference would av_assert0 -> abort
                                          (daala_entropy.c:336; qp > 0 only)                     */
#define FFV2AMD_ERR_UNSUPPORTED (-38) /* AVERROR(ENOSYS)                                        */
#define FFV2AMD_ERR_AGAIN      (-11)  /* AVERROR(EAGAIN): ring full (send) / nothing ready (receive) */

/* AVPixelFormat values the reference encoder accepts (ffv2enc.c:596-601; numeric
 * values of libavutil/pixfmt.h in the reference tree, little-endian host). */
#define FFV2AMD_PIX_YUV444P      5
#define FFV2AMD_PIX_GRAY8        8
#define FFV2AMD_PIX_YUV444P10LE 70
#define FFV2AMD_PIX_GBRP        73
#define FFV2AMD_PIX_GBRP10LE    77
#define FFV2AMD_PIX_YUV444P12LE 133
#define FFV2AMD_PIX_GBRP12LE    137

typedef struct ffv2amd_encoder ffv2amd_encoder;

/* Geometry derived at init (ffv2enc.c:500-504). */
typedef struct ffv2amd_info {
    int width, height, pix_fmt;
    int planes, depth;
    int num_sb_x, num_sb_y;        /* ceil(w/64), ceil(h/64)                              */
    int block_planes;              /* num_sb_x*num_sb_y*planes: 64x64 blocks per frame     */
    int max_batch;                 /* frames one *_device call may carry                   */
    size_t packet_cap;             /* upper bound of one packet in bytes (qp == 0)         */
    size_t packet_cap_qp;          /* upper bound for any qp (host-assembled packets)      */
    size_t tstage_bytes_per_frame; /* algorithmic HBM bytes of the T-stage, SURVEY.md 8(d):
                                      P*W*H*bytes_in + P*(64nsx)*(64nsy)*4                 */
    /* layout the *_device entry points expect for frames resident in HBM */
    size_t row_pitch;              /* bytes, multiple of 16, >= width*bytes_per_sample     */
    size_t plane_stride;           /* bytes, row_pitch*height rounded up to 256            */
    size_t frame_stride;           /* bytes, plane_stride*planes                           */
} ffv2amd_info;

/* == AVCodec.init (ffv2enc.c:495).  device = HIP ordinal.  max_batch >= 1 sizes the
 * device workspace (frames in flight per call; BASELINE.json's "slices" in SURVEY's
 * reading). */
int  ffv2amd_encoder_create(ffv2amd_encoder **enc, int width, int height, int pix_fmt,
                            int device, int max_batch);
/* == AVCodec.close (ffv2enc.c:515). NULL is accepted. */
void ffv2amd_encoder_destroy(ffv2amd_encoder *enc);
int  ffv2amd_encoder_info(const ffv2amd_encoder *enc, ffv2amd_info *info);

/* == AVCodec.encode2 (ffv2enc.c:453): one frame in host memory -> one packet in host
 * memory.  Reads data[p]/linesize[p] for p < planes exactly as ref2coeff does
 * (ffv2enc.c:471-474), qp = avctx->global_quality (ffv2enc.c:460).
 * W: optional phantom coefficient per block-plane, index (sby*nsx+sbx)*planes+p
 * (the reference reads temp2[4096]); NULL means 0 = clang-built
 * reference. */
int  ffv2amd_encode_frame(ffv2amd_encoder *enc,
                          const uint8_t *const data[4], const ptrdiff_t linesize[4],
                          int qp, const int32_t *W,
                          uint8_t *out, size_t out_cap, size_t *out_size);

/* Threads, streams, devices.  One thread at a time per encoder (as libavcodec calls encode2);
 * different encoders may be used from different threads.  Every entry point selects the
 * encoder's device for its own duration and restores the caller's current device.
 * *_batch_device and tstage_device are ordered on the stream passed in: the frames must have
 * been produced on that stream (or be complete).  ffv2amd_encode_frame,
 * ffv2amd_encode_batch_to_host and the ring run on streams the encoder owns and synchronise
 * with the host only; for encode_batch_to_host the device frames must be complete when it is
 * called.  The encoder's internal hand-off buffers (and the default status words used when
 * d_status == NULL) belong to one call at a time, two in pipelined mode: do not issue
 * *_batch_device calls of one encoder on several streams at once. */

/* Same step for `nframes` (<= max_batch) frames already resident in HBM in the
 * layout ffv2amd_info describes; packets stay in HBM:
 *   d_packets[f*packet_stride ...], d_sizes[f] (uint32 bytes).
 * d_W: device int32[nframes][block_planes] or NULL.  stream: hipStream_t (NULL = the default stream).
 * Asynchronous with respect to the host; per-frame status lands in d_status[f]
 * (0 or a negative FFV2AMD_ERR_*), if d_status != NULL. */
int  ffv2amd_encode_batch_device(ffv2amd_encoder *enc, int nframes, const void *d_frames,
                                 int qp, const int32_t *d_W,
                                 void *d_packets, size_t packet_stride,
                                 uint32_t *d_sizes, int32_t *d_status, void *stream);

/* T-stage alone (level shift, lapping, 2-D lifting DCT, scan, band energies):
 *   d_coef   int32[nframes][block_planes][4096] coding order      (may be NULL)
 *   d_energy int64[nframes][block_planes][13]   phantom W excluded (may be NULL)
 * Used by the parity tests and the roofline measurement. */
int  ffv2amd_tstage_device(ffv2amd_encoder *enc, int nframes, const void *d_frames,
                           int32_t *d_coef, int64_t *d_energy, void *stream);

/* The T-stage of ONE frame in plain wrapping int32 (ffv2_wide.hip), as the reference computes it for
 * any 16-bit sample (ffv2.c:26-38 level-shifts whatever it is given).  The fast kernels stage samples
 * as int16 and refuse a frame with samples above its declared depth, or a band gain beyond their
 * 32 768-entry table, with FFV2AMD_ERR_RANGE in the frame's status; the entry points that end in host
 * memory (ffv2amd_encode_frame, _encode_frame_420, _encode_batch_to_host, _ring_receive, _qp_finish with
 * the host coder, the codec shim) then rerun that frame through this path -- at qp > 0 followed by the PVQ
 * search, which takes any int32 coefficient -- and code it on the host, so the caller gets the reference's
 * packet, not an error.  ffv2amd_encode_batch_device (packets stay in HBM, no host in the loop) and the
 * device coders of qp > 0 (lane coder, one-wavefront coder) report the status instead.  Outputs as
 * ffv2amd_tstage_device; a test hook. */
int  ffv2amd_tstage_wide_device(ffv2amd_encoder *enc, const void *d_frame, int32_t *d_coef, int64_t *d_energy,
                                void *stream);

/* Batch variant of encode2 that ends in host memory and accepts any qp >= 0
 * (synchronous).  Frames are resident in HBM (layout of ffv2amd_info); packets are
 * written to h_packets[f*packet_stride ...], h_sizes[f], h_status[f].
 * qp == 0 : everything on the GPU, packets copied back.
 * qp  > 0 : T-stage and PVQ search (ffv2enc.c:163-171, celt_pvq_search.asm) on the GPU;
 *           the adaptive range coder (daala_entropy.c:328-379,428-440), which is one
 *           serial chain per frame, runs on host threads, one frame per thread.
 * A frame on which the reference would abort (daala_entropy.c:336,342: a band whose
 * pulses all land on one coefficient, or qp == 1) gets status FFV2AMD_ERR_ABORT. */
int  ffv2amd_encode_batch_to_host(ffv2amd_encoder *enc, int nframes, const void *d_frames,
                                  int qp, const int32_t *d_W,
                                  uint8_t *h_packets, size_t packet_stride,
                                  uint32_t *h_sizes, int32_t *h_status);

/* 4:2:0 front end.  encode2() takes 4:4:4 planar only, as the reference's
 * does (ffv2enc.c:596-601, utils.c:814-822); handed yuv420p / yuv420p10le / yuv420p12le the
 * reference TOOL converts first -- choose_pixel_fmt (fftools/ffmpeg_filter.c:63-131) picks
 * yuv444p* of the same depth and an auto-inserted scale filter runs libswscale's default bicubic
 * (libswscale/utils.c:332-727 initFilter; swscale.c:96-139; output.c:333-393): luma unchanged,
 * chroma 2x up both ways.  These calls are that step on the GPU, for an encoder created with
 * the yuv444p format of the same depth.  PARITY UNPINNED (no libswscale binary or vector here).
 *   4:2:0 frames are tightly packed: Y (w x h), U, V (ceil(w/2) x ceil(h/2)), uint8 or uint16le,
 *   ffv2amd_frame_bytes_420() bytes each. */
size_t ffv2amd_frame_bytes_420(const ffv2amd_encoder *enc);
int  ffv2amd_upconvert_420_device(ffv2amd_encoder *enc, int nframes, const void *d_src420,
                                  void *d_frames444, void *stream);
int  ffv2amd_encode_frame_420(ffv2amd_encoder *enc, const uint8_t *const data[3], const ptrdiff_t linesize[3],
                              int qp, uint8_t *out, size_t out_cap, size_t *out_size);

/* The same batch step for 1 <= qp <= 64 split in two, so that consecutive batches overlap:
 *   qp_submit : T-stage, PVQ search and symbol compaction of one batch, asynchronous on the
 *               encoder's stream (frames must be complete when it is called); at most two
 *               batches in flight (FFV2AMD_ERR_AGAIN beyond that).
 *   qp_finish : oldest submitted batch -> packets in host memory; runs the range coder on host
 *               threads, one frame per thread.  FFV2AMD_ERR_AGAIN when nothing is submitted.
 * submit(n+1) issued before finish(n) hides the GPU work behind the host coder. */
/* Where the adaptive range coder of qp > 0 runs.  0 (default): on host threads, one frame per
 * thread, behind the GPU.  1: on the device, one wavefront per frame (ffv2_rangecoder.hip: the
 * serial symbol loop on one lane, the carry propagation of encode_done as a wavefront prefix
 * scan) -- the north_star's device coder; several times slower per frame than a host core, because
 * the coder is one dependent chain per frame, but nothing crosses PCIe except the finished packets. */
int  ffv2amd_encoder_set_device_coder(ffv2amd_encoder *enc, int on);
int  ffv2amd_qp_submit(ffv2amd_encoder *enc, int nframes, const void *d_frames, int qp, const int32_t *d_W);
int  ffv2amd_qp_finish(ffv2amd_encoder *enc, uint8_t *h_packets, size_t packet_stride,
                       uint32_t *h_sizes, int32_t *h_status);
/* == avcodec_send_frame / avcodec_receive_packet (encode.c:420,449) for global_quality 1..64: host
 * frames through that pipeline, one frame per batch, at most two in flight (FFV2AMD_ERR_AGAIN).
 * receive runs the oldest frame's range coder on the calling thread while the GPU works on the
 * frame sent after it; a frame the reference would abort on returns FFV2AMD_ERR_ABORT and leaves
 * the pipeline.  Do not mix with ffv2amd_qp_submit on one encoder.  Parity unpinned (qp > 0). */
int  ffv2amd_qp_send_frame(ffv2amd_encoder *enc, const uint8_t *const data[4], const ptrdiff_t linesize[4],
                           int qp, const int32_t *W, int64_t tag);
/* a yuv420p* frame (Y, U, V) through the same pipeline, up-converted on the device first (see ffv2amd_ring_send_420) */
int  ffv2amd_qp_send_frame_420(ffv2amd_encoder *enc, const uint8_t *const data[3], const ptrdiff_t linesize[3], int qp, int64_t tag);
int  ffv2amd_qp_receive_packet(ffv2amd_encoder *enc, uint8_t *out, size_t out_cap, size_t *out_size, int64_t *tag);
int  ffv2amd_qp_pending(const ffv2amd_encoder *enc);

/* == avcodec_send_frame / avcodec_receive_packet at qp > 0 AT THE DEVICE CODER'S RATE (reference encode.c:420,449
 * around ffv2enc.c:453; round 3).  ffv2amd_qp_send_frame above codes one frame per call with the host-thread coder
 * (tens of milliseconds per 1080p frame); this ring is the same boundary on top of the lane coder below: frames in
 * host memory are collected `frames_per_call` at a time in device memory as they arrive (H2D on the ring's own
 * stream, a 4:2:0 frame up-converted there), every full batch is one lane coder call (two to four in flight), and the
 * packets come back in send order, byte-identical to ffv2amd_encode_batch_to_host at that qp.
 *   qpring_open    : qp 1..64.  Below 3 800 frames per call: four calls in flight with a range chain each (a call
 *                    lasts one frame's chain whatever it holds; four side by side cost little more than one) --
 *                    HBM for five batches of frames plus ffv2amd_lanecoder_bytes_per_frame_ex(enc, packet_cap,
 *                    4, 4) per frame; three or two calls where the device cannot hold that.  From 3 800 frames:
 *                    two calls, one chain, three batches of frames.  FFV2AMD_QPRING_CALLS=2..4 and
 *                    FFV2AMD_LC_BACKS=1..calls in the environment override.  Page-locked host memory
 *                    for one batch of packets; FFV2AMD_ERR_NOMEM if nothing fits.  Throughput grows with
 *                    frames_per_call (1080p / qp 16: 6.9 Gpix/s at 512, 8.8 at 1 024, 10.1 at 2 048), and so
 *                    does the delay: a packet comes back once its whole batch is through.
 *   qpring_send    : flags FFV2AMD_FRAME_PINNED (planes page-locked and untouched until the frame's packet has
 *                    been received: the DMA engine reads them in place), FFV2AMD_FRAME_REGISTER (the same promise for
 *                    ordinary memory from a pool of long-lived buffers, page-locked here on first sight) and/or
 *                    FFV2AMD_FRAME_YUV420 (data = Y, U, V of a yuv420p* frame); otherwise the rows are copied before
 *                    the call returns (into about 256 MB of page-locked bounce frames the ring owns, by helper
 *                    threads as in ring_send: FFV2AMD_GATHER_THREADS).
 *                    FFV2AMD_ERR_AGAIN: a batch is full, two calls are in flight and the packets of the one
 *                    before them have not all been received -- receive, then send the frame again.
 *   qpring_flush   : end of stream (avcodec_send_frame(NULL)): the partly filled batch goes out.  FFV2AMD_ERR_AGAIN
 *                    as for send.
 *   qpring_receive : next packet in send order with the tag given at send; FFV2AMD_ERR_AGAIN when none is ready
 *                    (wait == 0) or nothing has been submitted (send more frames, or flush); with wait != 0 it
 *                    blocks until the oldest batch in flight is through.  A frame the reference would abort on
 *                    returns FFV2AMD_ERR_ABORT and leaves the ring.
 * One thread drives a ring; while it is open the encoder's lane coder entry points are the ring's.  PARITY
 * UNPINNED like all of qp > 0. */
int  ffv2amd_qpring_open(ffv2amd_encoder *enc, int qp, int frames_per_call, size_t packet_cap);
int  ffv2amd_qpring_send(ffv2amd_encoder *enc, const uint8_t *const data[4], const ptrdiff_t linesize[4],
                         const int32_t *W, int64_t tag, unsigned flags);
int  ffv2amd_qpring_flush(ffv2amd_encoder *enc);
int  ffv2amd_qpring_receive(ffv2amd_encoder *enc, uint8_t *out, size_t out_cap, size_t *out_size, int64_t *tag, int wait);
int  ffv2amd_qpring_pending(const ffv2amd_encoder *enc);
int  ffv2amd_qpring_close(ffv2amd_encoder *enc);

/* qp > 0 with MANY FRAMES IN FLIGHT (ffv2_lanecoder.hip).  The range coder
 * is one dependent chain per frame (ffv2enc.c:461,466), so the device codes many frames side by
 * side, one per lane of a wavefront, and does the rest (CDF rows as prefix counts, raw bits,
 * carry propagation) data-parallel.  Throughput grows with the frames in flight until the other
 * kernels bound it; a call takes at least one frame's chain (about 73 ns per symbol).
 *   lanecoder_open   : sizes the HBM scratch for `frames_in_flight` frames per call (plus, once, the coder's own
 *                      transform / pulse workspace for up to 64 frames, at most 2.5 GB: its launches do not
 *                      depend on the encoder's max_batch)
 *                      (ffv2amd_lanecoder_bytes_per_frame() each: 38 MB per 1080p frame with the
 *                      default packet_cap and two calls in flight; round 2: 84 MB, before the range
 *                      chain's input was produced window by window);
 *                      FFV2AMD_ERR_NOMEM if the device cannot hold it.  packet_cap = 0 reserves
 *                      ffv2amd_info.packet_cap_qp bytes per packet (2 100 per block-plane, six
 *                      buffers of that size per frame); a smaller packet_cap saves HBM, and a frame
 *                      whose packet would not fit comes back as FFV2AMD_ERR_NOSPACE (noise at
 *                      qp 16 / 64 codes to 160 / 370 bytes per block-plane).
 *   lanecoder_submit : up to that many device-resident frames (layout of ffv2amd_info), qp 1..64.
 *                      Asynchronous; the frames (and W) stay untouched until the call's finish, and they are
 *                      COMPLETE when the call is made: the coder's streams are its own and wait for no stream
 *                      of the caller's (the same holds for ffv2amd_qp_submit and ffv2amd_encode_batch_to_host,
 *                      which take no stream argument either).
 *                      calls_in_flight (2, or 3; 0 = 2) calls may be in flight (FFV2AMD_ERR_AGAIN
 *                      for one more; each holds its own copy of the smaller buffers): the transform,
 *                      PVQ search and symbol bookkeeping of call n+1 then run beside the range
 *                      chain of call n, which occupies a small part of the chip.
 *   lanecoder_finish : the oldest submitted call -> packets in host memory (packet f at h_packets +
 *                      f * packet_stride), byte-identical to ffv2amd_encode_batch_to_host at the
 *                      same qp; one device-to-host copy per packet.  Per-frame status as there
 *                      (FFV2AMD_ERR_ABORT where the reference would av_assert0).
 *   lanecoder_finish_packed : the same packets as they lie on the device: back to back, packet f
 *                      at h_buf + h_offsets[f] (16-byte aligned), brought over in a few large
 *                      copies -- the fast way out for thousands of packets; page-locked h_buf
 *                      (ffv2amd_host_alloc) for full PCIe rate.  FFV2AMD_ERR_NOSPACE (the call stays
 *                      queued) when h_cap is too small: frames * (packet_cap + 16) always suffices.
 *   lanecoder_encode : submit + finish.
 *   lanecoder_open_ex / bytes_per_frame_ex : calls_in_flight 2..4 and `backs` 1..calls_in_flight range chains
 *                      side by side (call n takes back n % backs; each back holds its own records, code words
 *                      and lane state: 2 * window * 8 + 2 * packet_cap bytes per frame more).  A call lasts one
 *                      frame's chain at least, whatever it holds; where the memory holds too few frames for
 *                      the next call's front to fill that time (4K and larger pictures, or a few hundred
 *                      frames per call) a second chain beside the first does.  lanecoder_open() is
 *                      open_ex(..., backs = 1) unless the environment says FFV2AMD_LC_BACKS=n.
 * One thread drives a coder.  PARITY UNPINNED like all of qp > 0. */
int    ffv2amd_lanecoder_open(ffv2amd_encoder *enc, int frames_in_flight, size_t packet_cap, int calls_in_flight);
int    ffv2amd_lanecoder_open_ex(ffv2amd_encoder *enc, int frames_in_flight, size_t packet_cap, int calls_in_flight, int backs);
int    ffv2amd_lanecoder_close(ffv2amd_encoder *enc);
size_t ffv2amd_lanecoder_bytes_per_frame(const ffv2amd_encoder *enc, size_t packet_cap, int calls_in_flight);
size_t ffv2amd_lanecoder_bytes_per_frame_ex(const ffv2amd_encoder *enc, size_t packet_cap, int calls_in_flight, int backs);
int    ffv2amd_lanecoder_submit(ffv2amd_encoder *enc, int nframes, const void *d_frames, int qp, const int32_t *d_W);
int    ffv2amd_lanecoder_finish(ffv2amd_encoder *enc, uint8_t *h_packets, size_t packet_stride,
                                uint32_t *h_sizes, int32_t *h_status);
int    ffv2amd_lanecoder_finish_packed(ffv2amd_encoder *enc, uint8_t *h_buf, size_t h_cap, uint64_t *h_offsets,
                                       uint32_t *h_sizes, int32_t *h_status);
/* Timing of the call finished last: its range-chain kernel and its whole back (cdf, chain, packets), in ms on
 * the device clock, and the symbols the coder read for the call's first frame.  For benchmarks. */
int    ffv2amd_lanecoder_stats(const ffv2amd_encoder *enc, float *chain_ms, float *back_ms, uint32_t *symbols_frame0);
/* Test hook (process-wide, read by the next lanecoder_open): symbols of the coding order the back works on at a
 * time (0 = the default, 2^18; the environment variable FFV2AMD_LC_WINDOW sets the same at start-up). */
void   ffv2amd_debug_lanecoder_window(uint32_t symbols);
/* Benchmark aid: the Q-stage (PVQ search) kernel alone on `nframes` (<= max_batch) device-resident frames:
 * average ms per launch over `reps` launches (bench.py --qp reports it against the f32 division rate). */
int    ffv2amd_debug_pvq_time(ffv2amd_encoder *enc, int nframes, const void *d_frames, int qp, int reps, float *ms_per_launch);
int    ffv2amd_lanecoder_encode(ffv2amd_encoder *enc, int nframes, const void *d_frames, int qp,
                                const int32_t *d_W, uint8_t *h_packets, size_t packet_stride,
                                uint32_t *h_sizes, int32_t *h_status);


/* Decoder-side inverse of the T-stage (reference ffv2.c:81-98 coding_to_raster, :4962-4972
 * tx_inv_2d / od_bin_idct64, :216-239 lapping post-filter in ffv2dec.c's seam order,
 * :40-52 coeffs_2_ref): coding-order coefficients d_coef[nframes][block_planes][4096] ->
 * pictures d_frames_out in the layout of ffv2amd_info.  A round-trip self check for the
 * encoder (on picture data inverse(tstage(x)) == x exactly), not a decoder. */
int  ffv2amd_inverse_tstage_device(ffv2amd_encoder *enc, int nframes, const int32_t *d_coef,
                                   void *d_frames_out, void *stream);

/* Decoder-side check of a finished packet, the shape of FATE's enc_dec (reference tests/fate-run.sh:188-210):
 * ffv2_decode_frame (ffv2dec.c:315-377) -- entropy layer parsed on the host (daala_entropy.c:273-326,
 * 413-425 in the symbol order of dequant_block, ffv2dec.c:100-136), scaling of the pulses, inverse
 * T-stage and coeffs_2_ref on the device -- into the caller's planes (the encoder's geometry and
 * pix_fmt; the packet's header must agree).  Reproduces the reference decoder as it is, quirks
 * included: at qp == 0 it divides by sqrt(0) and every coefficient becomes 0x80000000 (ffv2dec.c:134-136),
 * unread pulse slots carry over between bands.  FFV2AMD_DECODE_GRID: also its `#define DEBUGGING`
 * overwrite of each superblock's first row and column (:258-273; the text overlay of 8-bit pictures
 * contains the decoding time and is not reproducible).  *qp_out: the packet's qp.  A self check and the
 * PSNR line of tools/fate_report.py, not a product decoder.  PARITY UNPINNED. */
#define FFV2AMD_DECODE_GRID 1u
/* The host half of ffv2amd_decode_frame on its own (no GPU): the packet's entropy layer in dequant_block's symbol order.
 *   pulses int16 [nblk][4096]  the decoder's pulses[] value for every coding position when its band is scaled
 *   mag    float [nblk][13]    (float)pow(gain, 1.5f) / sqrt(sum of squares of the pulses read) -- inf / NaN at qp 0
 *   c0     int32 [nblk]        the "DC" slot
 * sized by the caller for the planes of the format it expects; check *pix_fmt_out. */
int  ffv2amd_parse_packet(const uint8_t *packet, size_t size, int width, int height, int *pix_fmt_out, int *qp_out,
                          int16_t *pulses, float *mag, int32_t *c0);
int  ffv2amd_decode_frame(ffv2amd_encoder *enc, const uint8_t *packet, size_t size, uint8_t *const data[4],
                          const ptrdiff_t linesize[4], unsigned flags, int *qp_out);

/* Test hook: the device PVQ search on `count` float vectors of N <= 2049 elements
 * (d_X[v*stride + i]); writes int16 pulses to d_y[v*stride + i]. */
int  ffv2amd_pvq_search_device(ffv2amd_encoder *enc, const float *d_X, int stride, int N, int K,
                               int count, int16_t *d_y, void *stream);

/* Optional: also materialise the coding-order coefficients of every batch encode
 * in HBM (what the reference keeps in temp2[], ffv2enc.c:194,201, and what the
 * qp > 0 quantiser consumes).  d_coef: int32[max_batch][block_planes][4096] or NULL
 * to turn it off again.  Same kernel launch, no extra pass. */
int  ffv2amd_encoder_set_coef_sink(ffv2amd_encoder *enc, int32_t *d_coef);

/* Pipelined mode ("one batch per stream"): the E-stage of *_batch_device call n is issued on
 * an internal stream behind an event, so the caller's stream can already run the T-stage of
 * call n+1.  Consecutive calls must then use different d_packets/d_sizes/d_status buffers
 * (two sets suffice), and ffv2amd_encoder_flush(enc, stream) makes `stream` wait for every
 * E-stage issued so far (a device-wide synchronise does too). Off by default. */
int  ffv2amd_encoder_set_pipelined(ffv2amd_encoder *enc, int on);
int  ffv2amd_encoder_flush(ffv2amd_encoder *enc, void *stream);

/* == avcodec_send_frame / avcodec_receive_packet (reference encode.c:420,449 around
 * ffv2enc.c:453) as an asynchronous ring of `depth` frames, qp == 0: while frame n+1 crosses
 * PCIe, frame n is transformed and coded and the packet of frame n-1 returns, each on its own
 * HIP stream.  One thread drives a ring; packets come back in send order with the tag given at
 * send.  encode2() itself stays one-in/one-out (the reference sets no AV_CODEC_CAP_DELAY).
 *   ring_send    : FFV2AMD_ERR_AGAIN when `depth` frames are in flight (receive one first).
 *                  flags & FFV2AMD_FRAME_PINNED: the planes are page-locked (ffv2amd_host_alloc,
 *                  hipHostMalloc/hipHostRegister) and stay untouched until the frame's packet has
 *                  been received -- the DMA engine then reads them in place (planes that follow
 *                  each other in memory without a gap, rows at the device pitch, travel as ONE copy:
 *                  they must then belong to one page-locked allocation, as the planes of an
 *                  ffv2amd_host_alloc / av_image_alloc style frame do); otherwise the rows
 *                  are gathered into a pinned staging frame before send returns, by a pool of
 *                  host threads the ring owns (FFV2AMD_GATHER_THREADS, caller included; default 6)
 *                  that is started on the first such send and joined by ring_close.
 *   ring_receive : oldest frame in flight; FFV2AMD_ERR_AGAIN if none, or (wait == 0) not finished.
 *                  Copies back the packet's own size, not the capacity.  A frame that fails
 *                  (status < 0) is dropped from the ring and its error returned. */
#define FFV2AMD_FRAME_PINNED 1u
#define FFV2AMD_FRAME_YUV420 2u      /* ffv2amd_codec_send_frame only: a 4:2:0 frame (== ffv2amd_ring_send_420) */
#define FFV2AMD_FRAME_REGISTER 4u    /* the planes are ordinary (pageable) memory from a pool of long-lived buffers -- what
                                        libavcodec's get_buffer2 hands out: the ring page-locks each distinct buffer the first
                                        time it sees it (hipHostRegister, milliseconds) and lets the DMA engine read it in place
                                        from then on, like FFV2AMD_FRAME_PINNED; same promise: untouched until the frame's packet
                                        has been received, and the buffers outlive the ring (registrations end at ring_close;
                                        at most 256 buffers, further ones -- and planes below 256 KB, which would page-lock bits
                                        of the C library's heap -- are gathered like unflagged frames) */
int   ffv2amd_ring_open(ffv2amd_encoder *enc, int depth);
int   ffv2amd_ring_send(ffv2amd_encoder *enc, const uint8_t *const data[4], const ptrdiff_t linesize[4],
                        const int32_t *W, int64_t tag, unsigned flags);
int   ffv2amd_ring_receive(ffv2amd_encoder *enc, uint8_t *out, size_t out_cap, size_t *out_size,
                           int64_t *tag, int wait);
/* The same for yuv420p / yuv420p10le / yuv420p12le frames (encoder created with the yuv444p* format of
 * the same depth; data[0..2] = Y, U, V with their own linesizes): what the ffmpeg tool's auto-inserted
 * bicubic scale filter does in front of encode2 (see ffv2amd_encode_frame_420), on the frame's compute
 * stream.  Half the bytes cross PCIe; luma is copied straight into plane 0.  May be mixed with
 * ring_send on one ring.  Parity unpinned. */
int   ffv2amd_ring_send_420(ffv2amd_encoder *enc, const uint8_t *const data[3], const ptrdiff_t linesize[3],
                            const int32_t *W, int64_t tag, unsigned flags);
int   ffv2amd_ring_pending(const ffv2amd_encoder *enc);
void  ffv2amd_ring_close(ffv2amd_encoder *enc);
/* page-locked host memory for frame pools feeding ring_send(FFV2AMD_FRAME_PINNED) */
void *ffv2amd_host_alloc(size_t bytes);
void  ffv2amd_host_free(void *p);

/* Per-kernel timing with HIP events recorded on the launch stream around the
 * T-stage kernel and around the E-stage kernels of every *_batch_device call.
 * profile_read waits for the recorded events, returns the summed durations (ms)
 * and the number of batch launches since the last read, and resets the counters. */
int  ffv2amd_profile_enable(ffv2amd_encoder *enc, int on);   /* on = n > 1: only every n-th call is timed (the timing
                                                                 events themselves cost ~3 % of a 0.37 ms step) */
/* Name of the T-stage kernel a *_batch_device call of `nframes` frames launches (two variants:
 * one 64x64 block-plane per wavefront, or wavefronts walking down columns of superblocks). */
const char *ffv2amd_tstage_kernel_name(ffv2amd_encoder *enc, int nframes);
/* Test hook (process-wide): 0 = always the one-block kernel, 1 = always the column-walking kernel,
 * -1 = automatic (the default; the environment variable FFV2AMD_TSTAGE sets the same at start-up). */
void ffv2amd_debug_force_tstage(int mode);
int  ffv2amd_profile_read(ffv2amd_encoder *enc, double *tstage_ms, double *estage_ms, int *launches);
/* The same, plus the shortest and the longest single T-stage launch among them (ms). */
int  ffv2amd_profile_read_ex(ffv2amd_encoder *enc, double *tstage_ms, double *estage_ms, int *launches,
                             double *tstage_min_ms, double *tstage_max_ms);

/* Host helpers with no GPU work (unit-tested on CPU):
 * coded band gain for an integer band energy, bit-identical to
 * (uint32)(float)pow(sqrtf(e)+FLT_EPSILON, 1/1.5f) (ffv2enc.c:166,174) ... */
uint32_t ffv2amd_coded_gain(int64_t energy);
/* ... and the data-independent range-coded prefix of a qp==0 packet
 * (ffv2enc.c:449, :222 through daala_entropy.c:328-379,428-440,624-674):
 * writes the bytes, returns their count (<0 on error), *slack_bits = free low
 * bits of the last byte that raw bits are OR-ed into (daala_entropy.c:719-721). */
int  ffv2amd_range_prefix(int pix_fmt, int num_sb, uint8_t *out, size_t cap, int *slack_bits);

const char *ffv2amd_version(void);

#ifdef __cplusplus
}
#endif
#endif
