/*
 * ffv2enc_amd.c -- plain-C host shim with the shape of libavcodec's AVCodec
 * init / encode2 / close for the FFV2 encoder, calling the HIP path through the
 * extern "C" entry points of include/ffv2_amd.h.  Mirrors reference
 * libavcodec/ffv2enc.c:495-513 (init), :453-493 (encode2), :515-580 (close),
 * :603-617 (codec descriptor).  Compiled by gcc; no HIP headers needed here.
 */
#include "ffv2_amd.h"
#include "ffv2_amd_codec.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct FFV2AMDEncCtx {      /* the role of FFV2EncCtx, ffv2enc.c:29-53 */
    ffv2amd_encoder *enc;           /* == encs[0]: encode2's device */
    ffv2amd_encoder *encs[FFV2AMD_MAX_DEVICES];   /* one per entry of the device list (frame fan-out) */
    int ndev;
    ffv2amd_info info;
    uint8_t *scratch;
    size_t scratch_cap;
    int ring_open;
    int mode;                       /* frames in flight came in through: 0 nothing yet, 1 the ring (qp 0), 2 the qp > 0 pipeline, 3 the qp > 0 ring */
    int qpring_qp;                  /* the qp > 0 rings are open for this qp (0: closed) */
    uint64_t sent, received;        /* frame n lives on device n % ndev */
    int verbose;                    /* FFV2AMD_VERBOSE: the reference's per-frame size line */
} FFV2AMDEncCtx;

static const int allowed_pix_fmts[] = {     /* ffv2enc.c:596-601 */
    FFV2AMD_PIX_GBRP, FFV2AMD_PIX_GBRP10LE, FFV2AMD_PIX_GBRP12LE,
    FFV2AMD_PIX_YUV444P, FFV2AMD_PIX_YUV444P10LE, FFV2AMD_PIX_YUV444P12LE,
    FFV2AMD_PIX_GRAY8,
    -1,
};

static const FFV2AMDCodecDescriptor descriptor = {
    .name           = "ffv2",
    .long_name      = "FFv2 (MI355X HIP)",
    .pix_fmts       = allowed_pix_fmts,
    .capabilities   = (1 << 1) | (1 << 9),    /* AV_CODEC_CAP_DR1 | AV_CODEC_CAP_EXPERIMENTAL */
    .caps_internal  = (1 << 0) | (1 << 1),    /* FF_CODEC_CAP_INIT_THREADSAFE | INIT_CLEANUP   */
    .priv_data_size = sizeof(FFV2AMDEncCtx),
};

const FFV2AMDCodecDescriptor *ffv2amd_codec_descriptor(void)
{
    return &descriptor;
}

int ffv2amd_codec_close(FFV2AMDCodecContext *avctx)
{
    /* also runs after a failed init (FF_CODEC_CAP_INIT_CLEANUP, utils.c:1048-1051) */
    FFV2AMDEncCtx *s;
    if (!avctx || !avctx->priv_data)
        return 0;
    s = avctx->priv_data;
    for (int d = 0; d < FFV2AMD_MAX_DEVICES; d++)
        ffv2amd_encoder_destroy(s->encs[d]);
    free(s->scratch);
    free(s);
    avctx->priv_data = NULL;
    return 0;
}

int ffv2amd_codec_init(FFV2AMDCodecContext *avctx)
{
    FFV2AMDEncCtx *s;
    int ok = 0, ret;
    if (!avctx)
        return FFV2AMD_ERR_INVAL;
    for (const int *p = allowed_pix_fmts; *p >= 0; p++)   /* utils.c:814-822 */
        ok |= *p == avctx->pix_fmt;
    if (!ok)
        return FFV2AMD_ERR_INVAL;
    s = calloc(1, sizeof(*s));
    if (!s)
        return FFV2AMD_ERR_NOMEM;
    avctx->priv_data = s;
    s->ndev = avctx->nb_devices > 1 ? avctx->nb_devices : 1;
    if (s->ndev > FFV2AMD_MAX_DEVICES) {
        ret = FFV2AMD_ERR_INVAL;
        goto fail;
    }
    for (int d = 0; d < s->ndev; d++) {
        ret = ffv2amd_encoder_create(&s->encs[d], avctx->width, avctx->height, avctx->pix_fmt,
                                     avctx->nb_devices > 1 ? avctx->hip_devices[d] : avctx->hip_device, 1);
        if (ret < 0)
            goto fail;
    }
    s->enc = s->encs[0];
    if ((ret = ffv2amd_encoder_info(s->enc, &s->info)) < 0)
        goto fail;
    /* qp = global_quality > 0 packets are larger than the qp == 0 bound (ffv2amd_info) */
    s->scratch_cap = avctx->global_quality > 0 ? s->info.packet_cap_qp : s->info.packet_cap;
    s->scratch = malloc(s->scratch_cap);
    if (!s->scratch) {
        ret = FFV2AMD_ERR_NOMEM;
        goto fail;
    }
    s->verbose = getenv("FFV2AMD_VERBOSE") != NULL;
    return 0;
fail:
    ffv2amd_codec_close(avctx);
    return ret;
}

static int grow_scratch(FFV2AMDEncCtx *s, size_t want)
{
    uint8_t *g;
    if (s->scratch_cap >= want)
        return 0;
    g = realloc(s->scratch, want);
    if (!g)
        return FFV2AMD_ERR_NOMEM;
    s->scratch = g;
    s->scratch_cap = want;
    return 0;
}

/* the encoder owns the payload and hands it over (daala_entropy.c:727-732) */
static int hand_over(FFV2AMDEncCtx *s, FFV2AMDPacket *avpkt, size_t n, int64_t pts)
{
    avpkt->data = malloc(n ? n : 1);
    if (!avpkt->data)
        return FFV2AMD_ERR_NOMEM;
    memcpy(avpkt->data, s->scratch, n);
    avpkt->size = (int)n;
    avpkt->pts = avpkt->dts = pts;                              /* encode.c:329-330 */
    if (s->verbose)
        fprintf(stderr, "Packet size = %f kib\n", n / 1024.0f);     /* ffv2enc.c:488 */
    return 0;
}

int ffv2amd_codec_encode2(FFV2AMDCodecContext *avctx, FFV2AMDPacket *avpkt,
                          const FFV2AMDFrame *frame, int *got_packet_ptr)
{
    FFV2AMDEncCtx *s;
    size_t n = 0;
    int ret;
    if (!avctx || !avctx->priv_data || !avpkt || !frame || !got_packet_ptr)
        return FFV2AMD_ERR_INVAL;
    s = avctx->priv_data;
    *got_packet_ptr = 0;
    /* global_quality raised after init: grow to the bound for any qp */
    if (avctx->global_quality > 0 && (ret = grow_scratch(s, s->info.packet_cap_qp)) < 0)
        return ret;
    ret = ffv2amd_encode_frame(s->enc, frame->data, frame->linesize, avctx->global_quality,
                               NULL, s->scratch, s->scratch_cap, &n);
    if (ret < 0)
        return ret;
    ret = hand_over(s, avpkt, n, frame->pts);
    if (ret < 0)
        return ret;
    *got_packet_ptr = 1;
    return 0;
}

/* What the ffmpeg TOOL does around encode2() for a yuv420p / yuv420p10le / yuv420p12le source:
 * choose_pixel_fmt() (fftools/ffmpeg_filter.c:63-131) selects yuv444p* of the same depth, the
 * auto-inserted scale filter (flags=bicubic) converts, then encode2() runs.  avctx must have been
 * initialised with that yuv444p* format -- encode2() itself keeps refusing 4:2:0, as the reference
 * does (utils.c:814-822).  frame: data[0..2] = Y, U, V with their own linesizes.  PARITY UNPINNED. */
int ffv2amd_codec_encode_yuv420(FFV2AMDCodecContext *avctx, FFV2AMDPacket *avpkt,
                                const FFV2AMDFrame *frame, int *got_packet_ptr)
{
    FFV2AMDEncCtx *s;
    size_t n = 0;
    int ret;
    if (!avctx || !avctx->priv_data || !avpkt || !frame || !got_packet_ptr)
        return FFV2AMD_ERR_INVAL;
    s = avctx->priv_data;
    *got_packet_ptr = 0;
    if (avctx->global_quality > 0 && s->scratch_cap < s->info.packet_cap_qp)
        return FFV2AMD_ERR_NOSPACE;
    ret = ffv2amd_encode_frame_420(s->enc, frame->data, frame->linesize, avctx->global_quality,
                                   s->scratch, s->scratch_cap, &n);
    if (ret < 0)
        return ret;
    ret = hand_over(s, avpkt, n, frame->pts);
    if (ret < 0)
        return ret;
    *got_packet_ptr = 1;
    return 0;
}

/* avcodec_send_frame / avcodec_receive_packet (encode.c:420,449): the caller is ONE thread feeding
 * frames and collecting packets; frames are independent (ffv2enc.c:461-469), so frame n goes to
 * device n % ndev -- each device has its own encoder and asynchronous ring (global_quality 0) or
 * two-deep qp > 0 pipeline -- and packets are handed back in send order with the frame's pts. */
int ffv2amd_codec_send_frame(FFV2AMDCodecContext *avctx, const FFV2AMDFrame *frame, unsigned flags)
{
    FFV2AMDEncCtx *s;
    ffv2amd_encoder *enc;
    const int qp = avctx ? avctx->global_quality : 0;
    int ret, mode;
    if (!avctx || !avctx->priv_data || qp < 0)
        return FFV2AMD_ERR_INVAL;
    s = avctx->priv_data;
    if (!frame) {                               /* end of stream: batches that are not full yet go out */
        if (s->mode == 3)
            for (int d = 0; d < s->ndev; d++)
                if ((ret = ffv2amd_qpring_flush(s->encs[d])) < 0)
                    return ret;
        return 0;
    }
    mode = qp > 0 ? (avctx->qp_frames_per_call > 0 ? 3 : 2) : 1;
    if (s->sent != s->received && s->mode != mode)
        return FFV2AMD_ERR_INVAL;               /* global_quality changed with frames in flight */
    enc = s->encs[s->sent % (uint64_t)s->ndev];
    if (mode == 3) {
        if (s->qpring_qp != qp) {
            if (s->sent != s->received)
                return FFV2AMD_ERR_INVAL;
            for (int d = 0; d < s->ndev; d++)
                ffv2amd_qpring_close(s->encs[d]);
            s->qpring_qp = 0;
            for (int d = 0; d < s->ndev; d++)
                if ((ret = ffv2amd_qpring_open(s->encs[d], qp, avctx->qp_frames_per_call, 0)) < 0) {
                    while (d-- > 0)
                        ffv2amd_qpring_close(s->encs[d]);
                    return ret;
                }
            s->qpring_qp = qp;
        }
        ret = ffv2amd_qpring_send(enc, frame->data, frame->linesize, NULL, frame->pts,
                                  flags & (FFV2AMD_FRAME_PINNED | FFV2AMD_FRAME_YUV420 | FFV2AMD_FRAME_REGISTER));
    } else if (mode == 2) {
        ret = flags & FFV2AMD_FRAME_YUV420
            ? ffv2amd_qp_send_frame_420(enc, frame->data, frame->linesize, qp, frame->pts)
            : ffv2amd_qp_send_frame(enc, frame->data, frame->linesize, qp, NULL, frame->pts);
    } else {
        if (!s->ring_open) {
            for (int d = 0; d < s->ndev; d++)
                if ((ret = ffv2amd_ring_open(s->encs[d], avctx->ring_depth > 0 ? avctx->ring_depth : 4)) < 0) {
                    while (d-- > 0)
                        ffv2amd_ring_close(s->encs[d]);
                    return ret;
                }
            s->ring_open = 1;
        }
        ret = flags & FFV2AMD_FRAME_YUV420
            ? ffv2amd_ring_send_420(enc, frame->data, frame->linesize, NULL, frame->pts, flags & (FFV2AMD_FRAME_PINNED | FFV2AMD_FRAME_REGISTER))
            : ffv2amd_ring_send(enc, frame->data, frame->linesize, NULL, frame->pts, flags & (FFV2AMD_FRAME_PINNED | FFV2AMD_FRAME_REGISTER));
    }
    if (ret < 0)
        return ret;
    s->mode = mode;
    s->sent++;
    return 0;
}

int ffv2amd_codec_receive_packet(FFV2AMDCodecContext *avctx, FFV2AMDPacket *avpkt, int wait)
{
    FFV2AMDEncCtx *s;
    ffv2amd_encoder *enc;
    size_t n = 0;
    int64_t pts = 0;
    int ret;
    if (!avctx || !avctx->priv_data || !avpkt)
        return FFV2AMD_ERR_INVAL;
    s = avctx->priv_data;
    if (s->received == s->sent)
        return FFV2AMD_ERR_AGAIN;
    enc = s->encs[s->received % (uint64_t)s->ndev];
    if (s->mode == 3) {
        if ((ret = grow_scratch(s, s->info.packet_cap_qp)) < 0)
            return ret;
        ret = ffv2amd_qpring_receive(enc, s->scratch, s->scratch_cap, &n, &pts, wait);
    } else if (s->mode == 2) {
        if ((ret = grow_scratch(s, s->info.packet_cap_qp)) < 0)
            return ret;
        ret = ffv2amd_qp_receive_packet(enc, s->scratch, s->scratch_cap, &n, &pts);
    } else {
        ret = ffv2amd_ring_receive(enc, s->scratch, s->scratch_cap, &n, &pts, wait);
    }
    if (ret == FFV2AMD_ERR_AGAIN || ret == FFV2AMD_ERR_INVAL)
        return ret;
    s->received++;                              /* delivered or failed, the oldest frame has left */
    if (ret < 0)
        return ret;
    return hand_over(s, avpkt, n, pts);
}

void ffv2amd_packet_unref(FFV2AMDPacket *pkt)
{
    if (!pkt)
        return;
    free(pkt->data);
    pkt->data = NULL;
    pkt->size = 0;
}
