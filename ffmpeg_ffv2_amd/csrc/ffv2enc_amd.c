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

typedef struct FFV2AMDEncCtx {      /* the role of FFV2EncCtx, ffv2enc.c:29-53 */
    ffv2amd_encoder *enc;
    ffv2amd_info info;
    uint8_t *scratch;
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
    ffv2amd_encoder_destroy(s->enc);
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
    ret = ffv2amd_encoder_create(&s->enc, avctx->width, avctx->height, avctx->pix_fmt,
                                 avctx->hip_device, 1);
    if (ret < 0)
        goto fail;
    if ((ret = ffv2amd_encoder_info(s->enc, &s->info)) < 0)
        goto fail;
    s->scratch = malloc(s->info.packet_cap);
    if (!s->scratch) {
        ret = FFV2AMD_ERR_NOMEM;
        goto fail;
    }
    return 0;
fail:
    ffv2amd_codec_close(avctx);
    return ret;
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
    ret = ffv2amd_encode_frame(s->enc, frame->data, frame->linesize, avctx->global_quality,
                               NULL, s->scratch, s->info.packet_cap, &n);
    if (ret < 0)
        return ret;
    /* the encoder owns the payload and hands it over (daala_entropy.c:727-732) */
    avpkt->data = malloc(n ? n : 1);
    if (!avpkt->data)
        return FFV2AMD_ERR_NOMEM;
    for (size_t i = 0; i < n; i++)
        avpkt->data[i] = s->scratch[i];
    avpkt->size = (int)n;
    avpkt->pts = avpkt->dts = frame->pts;
    fprintf(stderr, "Packet size = %f kib\n", n / 1024.0f);     /* ffv2enc.c:488 */
    *got_packet_ptr = 1;
    return 0;
}

void ffv2amd_packet_unref(FFV2AMDPacket *pkt)
{
    if (!pkt)
        return;
    free(pkt->data);
    pkt->data = NULL;
    pkt->size = 0;
}
