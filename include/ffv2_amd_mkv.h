/*
 * ffv2_amd_mkv.h -- Matroska wire step for FFV2 packets.
 *
 * The reference carries FFV2 in Matroska under the codec id "V_FFV2"
 * (libavformat/matroska.c:83) with no CodecPrivate (the encoder has no extradata,
 * ffv2enc.c:495-513) and without the keyframe bit on its SimpleBlocks (the encoder
 * does not set AV_PKT_FLAG_KEY; matroskaenc.c:2160).  This writer emits the same
 * elements, in the order libavformat/matroskaenc.c writes them for one video stream
 * (mkv_write_header :1835, mkv_write_track :1151, mkv_write_block :2080), minus the
 * optional machinery a single intra-only video track does not need (SeekHead, Cues - the
 * reference only cues keyframes -, Tags, CRC-32 children).  Host code only, no GPU.
 *
 * PARITY UNPINNED: no libavformat binary exists in this environment to compare bytes with;
 * tests/test_mkv.py checks the files with an independent EBML reader instead.
 */
#ifndef FFV2_AMD_MKV_H
#define FFV2_AMD_MKV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ffv2amd_mkv ffv2amd_mkv;

/* Creates `path` and writes the EBML header, Segment, Info and Tracks.
 * The time base of the packets is fps_den / fps_num seconds per pts tick.
 * Returns 0 or a negative AVERROR-style code (-EINVAL, -ENOMEM, -EIO). */
int ffv2amd_mkv_open(ffv2amd_mkv **m, const char *path, int width, int height, int fps_num, int fps_den);

/* One FFV2 packet = one SimpleBlock (track 1, no lacing, flags 0).  pts must not decrease.
 * A new Cluster starts every 5 s or 5 MiB (matroskaenc.c's defaults) and whenever the
 * 16-bit relative timestamp would overflow. */
int ffv2amd_mkv_write_packet(ffv2amd_mkv *m, const uint8_t *data, size_t size, int64_t pts);

/* Flushes the last Cluster, patches the Segment size and the Duration, closes the file. */
int ffv2amd_mkv_close(ffv2amd_mkv *m);

#ifdef __cplusplus
}
#endif
#endif
