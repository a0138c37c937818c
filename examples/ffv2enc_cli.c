/*
 * ffv2enc_cli.c -- raw planar 4:4:4 frames in, concatenated FFV2 packets out, through the
 * AVCodec-shaped host shim (include/ffv2_amd_codec.h).  The equivalent of
 *   ffmpeg -f rawvideo -s WxH -pix_fmt FMT -i in.yuv -c:v ffv2 -strict -2 -f rawvideo out.ffv2
 * on a machine that has an MI355X but no FFmpeg: the rawvideo
 * muxer writes packets back to back, which is what the reference's known-answer md5s are taken over.
 *
 *   ffv2enc_cli WIDTH HEIGHT PIX_FMT IN.yuv OUT.ffv2 [QP] [HIP_DEVICE] [--async N]
 *   --async N: avcodec_send_frame / avcodec_receive_packet instead of encode2 (same bytes out): N frames in flight
 *            at QP 0 (the asynchronous ring), batches of N frames coded side by side on the device at QP > 0
 *            (FFV2AMDCodecContext.qp_frames_per_call; drained with send_frame(NULL) at the end of the input).
 *   PIX_FMT: gray | yuv444p | yuv444p10le | yuv444p12le | gbrp | gbrp10le | gbrp12le
 *            yuv420p | yuv420p10le | yuv420p12le are converted first, as the ffmpeg tool does
 *            (choose_pixel_fmt -> yuv444p* + auto-inserted bicubic scale filter); --no-convert
 *            as a last argument hands them to encode2 unconverted, which refuses them (exit 2).
 * An output name ending in ".mkv" selects the Matroska writer (include/ffv2_amd_mkv.h,
 * "V_FFV2", 25 frames per second) instead of the back-to-back packet stream.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ffv2_amd.h"
#include "ffv2_amd_codec.h"
#include "ffv2_amd_mkv.h"

static int parse_fmt(const char *s, int *planes, int *bps, int *is420)
{
    static const struct { const char *n; int id, bps; } sub[] = {
        { "yuv420p", FFV2AMD_PIX_YUV444P, 1 }, { "yuv420p10le", FFV2AMD_PIX_YUV444P10LE, 2 },
        { "yuv420p12le", FFV2AMD_PIX_YUV444P12LE, 2 },
    };
    *is420 = 0;
    for (size_t i = 0; i < sizeof(sub) / sizeof(sub[0]); i++)
        if (!strcmp(s, sub[i].n)) { *planes = 3; *bps = sub[i].bps; *is420 = 1; return sub[i].id; }
    static const struct { const char *n; int id, planes, bps; } tab[] = {
        { "gray", FFV2AMD_PIX_GRAY8, 1, 1 },           { "yuv444p", FFV2AMD_PIX_YUV444P, 3, 1 },
        { "gbrp", FFV2AMD_PIX_GBRP, 3, 1 },            { "yuv444p10le", FFV2AMD_PIX_YUV444P10LE, 3, 2 },
        { "gbrp10le", FFV2AMD_PIX_GBRP10LE, 3, 2 },    { "yuv444p12le", FFV2AMD_PIX_YUV444P12LE, 3, 2 },
        { "gbrp12le", FFV2AMD_PIX_GBRP12LE, 3, 2 },
    };
    for (size_t i = 0; i < sizeof(tab) / sizeof(tab[0]); i++)
        if (!strcmp(s, tab[i].n)) { *planes = tab[i].planes; *bps = tab[i].bps; return tab[i].id; }
    return -1;
}

int main(int argc, char **argv)
{
    if (argc < 6) {
        fprintf(stderr, "usage: %s WIDTH HEIGHT PIX_FMT IN.yuv OUT.ffv2 [QP] [HIP_DEVICE] [--async N]\n", argv[0]);
        return 2;
    }
    int planes = 0, bps = 0, is420 = 0;
    FFV2AMDCodecContext ctx = { 0 };
    ctx.width = atoi(argv[1]);
    ctx.height = atoi(argv[2]);
    ctx.pix_fmt = parse_fmt(argv[3], &planes, &bps, &is420);
    if (is420 && !strcmp(argv[argc - 1], "--no-convert")) {
        fprintf(stderr, "%s is not an encoder input (ffv2enc.c:596-601)\n", argv[3]);
        return 2;
    }
    int async_n = 0;
    for (int i = 6; i + 1 < argc; i++)
        if (!strcmp(argv[i], "--async")) { async_n = atoi(argv[i + 1]); argc = i; break; }
    ctx.global_quality = argc > 6 ? atoi(argv[6]) : 0;
    ctx.hip_device = argc > 7 ? atoi(argv[7]) : 0;
    if (async_n > 0) { ctx.ring_depth = async_n; ctx.qp_frames_per_call = async_n; }
    if (ctx.pix_fmt < 0) { fprintf(stderr, "unsupported pix_fmt %s\n", argv[3]); return 2; }
    FILE *in = strcmp(argv[4], "-") ? fopen(argv[4], "rb") : stdin;
    const size_t olen = strlen(argv[5]);
    const int as_mkv = olen > 4 && !strcmp(argv[5] + olen - 4, ".mkv");
    FILE *out = as_mkv ? NULL : (strcmp(argv[5], "-") ? fopen(argv[5], "wb") : stdout);
    ffv2amd_mkv *mkv = NULL;
    if (as_mkv && ffv2amd_mkv_open(&mkv, argv[5], ctx.width, ctx.height, 25, 1) < 0) { perror("open"); return 1; }
    if (!in || (!out && !mkv)) { perror("open"); return 1; }
    int ret = ffv2amd_codec_init(&ctx);
    if (ret < 0) { fprintf(stderr, "init failed: %d\n", ret); return 1; }

    const size_t plane_bytes = (size_t)ctx.width * ctx.height * bps;
    const int cw = (ctx.width + 1) / 2, ch = (ctx.height + 1) / 2;
    const size_t cplane_bytes = (size_t)cw * ch * bps;
    const size_t frame_bytes = is420 ? plane_bytes + 2 * cplane_bytes : plane_bytes * planes;
    uint8_t *buf = malloc(frame_bytes);
    if (!buf) return 1;
    long nframes = 0;
    size_t nbytes = 0;
    if (async_n > 0) {
        /* one thread sends and receives, as with avcodec_send_frame / avcodec_receive_packet (encode.c:420,449): a frame
           is sent until the encoder takes it (EAGAIN: take a packet first), packets come back in send order; at the end
           of the input send_frame(NULL) lets batches that are not full go, then everything in flight is received.
           The frame buffer is reused at once: without FFV2AMD_FRAME_PINNED / _REGISTER the rows are copied by send. */
        long sent = 0;
        int eof = 0, drained = 0, done = 0;
        ret = 0;
        while (!done) {
            int have = !eof && fread(buf, 1, frame_bytes, in) == frame_bytes;
            FFV2AMDFrame fr = { 0 };
            if (!have) eof = 1;
            for (int p = 0; p < planes && have; p++) {
                fr.data[p] = is420 ? (p ? buf + plane_bytes + (p - 1) * cplane_bytes : buf) : buf + p * plane_bytes;
                fr.linesize[p] = (ptrdiff_t)(is420 && p ? cw : ctx.width) * bps;
            }
            fr.pts = sent;
            for (;;) {
                if (have) {
                    ret = ffv2amd_codec_send_frame(&ctx, &fr, is420 ? FFV2AMD_FRAME_YUV420 : 0);
                    if (ret == 0) { sent++; break; }                    /* next frame */
                    if (ret != FFV2AMD_ERR_AGAIN) { done = 1; break; }
                } else if (!drained) {
                    ret = ffv2amd_codec_send_frame(&ctx, NULL, 0);
                    if (ret == 0) drained = 1;
                    else if (ret != FFV2AMD_ERR_AGAIN) { done = 1; break; }
                }
                FFV2AMDPacket pkt = { 0 };
                ret = ffv2amd_codec_receive_packet(&ctx, &pkt, 1);
                if (ret == FFV2AMD_ERR_AGAIN) {
                    if (drained && nframes == sent) { ret = 0; done = 1; break; }
                    continue;
                }
                if (ret < 0) { fprintf(stderr, "receive_packet failed on frame %ld: %d\n", nframes, ret); done = 1; break; }
                if (mkv) ret = ffv2amd_mkv_write_packet(mkv, pkt.data, (size_t)pkt.size, pkt.pts);
                else     fwrite(pkt.data, 1, (size_t)pkt.size, out);
                nbytes += (size_t)pkt.size;
                ffv2amd_packet_unref(&pkt);
                nframes++;
            }
        }
    } else
    while (fread(buf, 1, frame_bytes, in) == frame_bytes) {
        FFV2AMDFrame fr = { 0 };
        FFV2AMDPacket pkt = { 0 };
        int got = 0;
        for (int p = 0; p < planes; p++) {
            fr.data[p] = is420 ? (p ? buf + plane_bytes + (p - 1) * cplane_bytes : buf) : buf + p * plane_bytes;
            fr.linesize[p] = (ptrdiff_t)(is420 && p ? cw : ctx.width) * bps;
        }
        fr.pts = nframes;
        ret = is420 ? ffv2amd_codec_encode_yuv420(&ctx, &pkt, &fr, &got) : ffv2amd_codec_encode2(&ctx, &pkt, &fr, &got);
        if (ret < 0 || !got) { fprintf(stderr, "encode2 failed on frame %ld: %d\n", nframes, ret); break; }
        if (mkv) ret = ffv2amd_mkv_write_packet(mkv, pkt.data, (size_t)pkt.size, pkt.pts);
        else     fwrite(pkt.data, 1, (size_t)pkt.size, out);
        nbytes += (size_t)pkt.size;
        ffv2amd_packet_unref(&pkt);
        nframes++;
    }
    ffv2amd_codec_close(&ctx);
    free(buf);
    if (mkv) { const int r2 = ffv2amd_mkv_close(mkv); if (ret >= 0) ret = r2; }
    else if (out != stdout) fclose(out);
    fprintf(stderr, "%ld frames, %zu bytes\n", nframes, nbytes);
    return ret < 0 ? 1 : 0;
}
