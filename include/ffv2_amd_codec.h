/*
 * ffv2_amd_codec.h -- AVCodec-shaped host surface of the MI355X FFV2 encoder.
 *
 * On a machine with the FFmpeg tree, libavcodec/ffv2enc_amd.c
 * fills a real `AVCodec` with these three functions.  The GPU box has no FFmpeg,
 * so this header mirrors only the fields the reference encoder touches
 * (SURVEY.md section 8(b) "Inputs read" / "Ownership"):
 *   AVCodecContext: width, height, pix_fmt, global_quality, priv_data
 *                   (libavcodec/avcodec.h; read at ffv2enc.c:460,474,500-504)
 *   AVFrame       : data[], linesize[]                       (ffv2enc.c:471-474)
 *   AVPacket      : data, size + an owner handle in place of AVBufferRef
 *                   (daala_entropy.c:727-732)
 * Same names, argument meaning and error behaviour as AVCodec.init / encode2 /
 * close (avcodec.h:3546-3693): 0 on success, negative AVERROR otherwise,
 * *got_packet_ptr = 1 with exactly one packet per frame, no delay.
 */
#ifndef FFV2_AMD_CODEC_H
#define FFV2_AMD_CODEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFV2AMD_MAX_DEVICES 16

typedef struct FFV2AMDCodecContext {
    int width, height;
    int pix_fmt;            /* enum AVPixelFormat value                       */
    int global_quality;     /* qp (ffv2enc.c:460); default 0                  */
    int hip_device;         /* extension: HIP ordinal, default 0              */
    int ring_depth;         /* extension: frames in flight PER DEVICE for send_frame/receive_packet, default 4 */
    void *priv_data;        /* owned by init/close                            */
    /* extension: frame-level fan-out over several GPUs behind ONE context (frames are independent,
     * ffv2enc.c:461-469).  nb_devices > 1: send_frame deals frame n to hip_devices[n % nb_devices]
     * (one encoder + ring per entry; an ordinal may appear more than once), receive_packet delivers
     * in send order.  nb_devices <= 1: hip_device alone.  encode2 always runs on the first device. */
    int nb_devices;
    int hip_devices[FFV2AMD_MAX_DEVICES];
    /* extension: global_quality > 0 through send_frame/receive_packet at the device coder's rate.  0 (default): one
     * frame per call, coded by host threads (ffv2amd_qp_send_frame).  N > 0: frames are collected N at a time PER
     * DEVICE and coded side by side on the device (ffv2amd_qpring_*, ffv2_lanecoder.hip); packets come back once their
     * batch is through, so the caller drains with send_frame(NULL) at the end of the stream, as with any encoder
     * that delays (the reference's encoder does not: ffv2enc.c:603-617 sets no AV_CODEC_CAP_DELAY). */
    int qp_frames_per_call;
} FFV2AMDCodecContext;

typedef struct FFV2AMDFrame {
    const uint8_t *data[4];
    ptrdiff_t linesize[4];
    int64_t pts;
} FFV2AMDFrame;

typedef struct FFV2AMDPacket {
    uint8_t *data;          /* malloc'ed by encode2, freed by ffv2amd_packet_unref */
    int size;
    int64_t pts, dts;       /* stamped from the frame, as encode.c:329-330 does */
} FFV2AMDPacket;

typedef struct FFV2AMDCodecDescriptor {
    const char *name;           /* "ffv2"                                      */
    const char *long_name;
    const int *pix_fmts;        /* terminated by -1 (AV_PIX_FMT_NONE)          */
    int capabilities;           /* AV_CODEC_CAP_DR1 | AV_CODEC_CAP_EXPERIMENTAL */
    int caps_internal;          /* INIT_THREADSAFE | INIT_CLEANUP              */
    int priv_data_size;
} FFV2AMDCodecDescriptor;

const FFV2AMDCodecDescriptor *ffv2amd_codec_descriptor(void);   /* ffv2enc.c:603-617 */
int  ffv2amd_codec_init(FFV2AMDCodecContext *avctx);             /* ffv2enc.c:495     */
int  ffv2amd_codec_encode2(FFV2AMDCodecContext *avctx, FFV2AMDPacket *avpkt,
                           const FFV2AMDFrame *frame, int *got_packet_ptr);  /* :453 */
int  ffv2amd_codec_close(FFV2AMDCodecContext *avctx);            /* ffv2enc.c:515     */
/* avcodec_send_frame / avcodec_receive_packet (encode.c:420,449): asynchronous, up to ring_depth
 * frames in flight per device, FFV2AMD_ERR_AGAIN (= AVERROR(EAGAIN)) when full (receive a packet,
 * then send again) / nothing ready; packets in send order with the frame's pts, whichever device
 * finishes first.  flags: FFV2AMD_FRAME_PINNED, FFV2AMD_FRAME_REGISTER, FFV2AMD_FRAME_YUV420 of ffv2_amd.h (the last:
 * frame->data[0..2] = Y, U, V of a yuv420p* frame of the context's depth, see
 * ffv2amd_codec_encode_yuv420).
 * global_quality 1..64 goes through ffv2amd_qp_send_frame / _receive_packet: two frames in flight per
 * device, receive_packet always waits (it runs the frame's range coder), a frame the reference
 * would abort on comes back as FFV2AMD_ERR_ABORT.  global_quality must not change while frames
 * are in flight.  A frame that fails leaves the pipeline: the next receive is the next frame. */
int  ffv2amd_codec_send_frame(FFV2AMDCodecContext *avctx, const FFV2AMDFrame *frame, unsigned flags);
int  ffv2amd_codec_receive_packet(FFV2AMDCodecContext *avctx, FFV2AMDPacket *avpkt, int wait);
/* The ffmpeg tool's format step + encode2 for yuv420p / yuv420p10le / yuv420p12le sources
 * (fftools/ffmpeg_filter.c:63-131, auto-inserted bicubic scale filter): avctx initialised with the
 * yuv444p* format of the same depth, frame->data[0..2] = Y, U, V.  Parity unpinned. */
int  ffv2amd_codec_encode_yuv420(FFV2AMDCodecContext *avctx, FFV2AMDPacket *avpkt,
                                 const FFV2AMDFrame *frame, int *got_packet_ptr);
void ffv2amd_packet_unref(FFV2AMDPacket *pkt);

#ifdef __cplusplus
}
#endif
#endif
