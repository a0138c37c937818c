/*
 * ffv2mkv.c -- minimal Matroska writer for "V_FFV2" packets (include/ffv2_amd_mkv.h).
 * Element ids and defaults follow libavformat/matroska.h / matroskaenc.c of the reference
 * tree (file:line in the comments); nothing here is copied from it.
 */
#include "ffv2_amd_mkv.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ids: libavformat/matroska.h */
#define ID_EBML            0x1A45DFA3u
#define ID_EBMLVERSION     0x4286u
#define ID_EBMLREADVERSION 0x42F7u
#define ID_EBMLMAXIDLEN    0x42F2u
#define ID_EBMLMAXSIZELEN  0x42F3u
#define ID_DOCTYPE         0x4282u
#define ID_DOCTYPEVERSION  0x4287u
#define ID_DOCTYPEREADVER  0x4285u
#define ID_SEGMENT         0x18538067u
#define ID_SEEKHEAD        0x114D9B74u
#define ID_SEEKENTRY       0x4DBBu
#define ID_SEEKID          0x53ABu
#define ID_SEEKPOSITION    0x53ACu
#define ID_INFO            0x1549A966u
#define ID_TIMECODESCALE   0x2AD7B1u
#define ID_MUXINGAPP       0x4D80u
#define ID_WRITINGAPP      0x5741u
#define ID_DURATION        0x4489u
#define ID_TRACKS          0x1654AE6Bu
#define ID_TRACKENTRY      0xAEu
#define ID_TRACKNUMBER     0xD7u
#define ID_TRACKUID        0x73C5u
#define ID_FLAGLACING      0x9Cu
#define ID_LANGUAGE        0x22B59Cu
#define ID_CODECID         0x86u
#define ID_TRACKTYPE       0x83u
#define ID_DEFAULTDURATION 0x23E383u
#define ID_VIDEO           0xE0u
#define ID_PIXELWIDTH      0xB0u
#define ID_PIXELHEIGHT     0xBAu
#define ID_DISPLAYUNIT     0x54B2u
#define ID_CLUSTER         0x1F43B675u
#define ID_CLUSTERTIMECODE 0xE7u
#define ID_SIMPLEBLOCK     0xA3u

#define CLUSTER_BYTES_MAX  (5u * 1024u * 1024u)   /* matroskaenc.c: cluster_size_limit default */
#define CLUSTER_MS_MAX     5000                   /* matroskaenc.c: cluster_time_limit default */

typedef struct { uint8_t *p; size_t n, cap; } Buf;

struct ffv2amd_mkv {
    FILE *f;
    int fps_num, fps_den;
    long segment_size_pos, duration_pos, segment_data_pos;
    Buf cluster;
    int64_t cluster_ms;       /* timestamp of the open cluster, -1 = none */
    int64_t last_pts, max_end_ms;
    int err;
};

static int buf_put(Buf *b, const void *src, size_t n)
{
    if (b->n + n > b->cap) {
        size_t cap = b->cap ? b->cap : 1u << 16;
        while (cap < b->n + n) cap *= 2;
        uint8_t *q = (uint8_t *)realloc(b->p, cap);
        if (!q) return -ENOMEM;
        b->p = q; b->cap = cap;
    }
    memcpy(b->p + b->n, src, n);
    b->n += n;
    return 0;
}

static int id_bytes(uint32_t id) { return id > 0xFFFFFFu ? 4 : id > 0xFFFFu ? 3 : id > 0xFFu ? 2 : 1; }

static int put_id(Buf *b, uint32_t id)
{
    uint8_t t[4];
    const int n = id_bytes(id);
    for (int i = 0; i < n; i++) t[i] = (uint8_t)(id >> (8 * (n - 1 - i)));
    return buf_put(b, t, (size_t)n);
}

/* EBML size in the fewest bytes (all-ones patterns are reserved: matroskaenc.c ebml_num_size) */
static int put_size(Buf *b, uint64_t v)
{
    int n = 1;
    while (n < 8 && v + 1 >= (1ull << (7 * n))) n++;
    uint8_t t[8];
    for (int i = 0; i < n; i++) t[i] = (uint8_t)(v >> (8 * (n - 1 - i)));
    t[0] |= (uint8_t)(0x80u >> (n - 1));
    return buf_put(b, t, (size_t)n);
}

static int put_size8(Buf *b, uint64_t v)         /* fixed 8-byte size, patchable */
{
    uint8_t t[8] = { 0x01 };
    for (int i = 1; i < 8; i++) t[i] = (uint8_t)(v >> (8 * (7 - i)));
    return buf_put(b, t, 8);
}

static int put_uint(Buf *b, uint32_t id, uint64_t v)
{
    int n = 1;
    while (n < 8 && (v >> (8 * n))) n++;
    uint8_t t[8];
    for (int i = 0; i < n; i++) t[i] = (uint8_t)(v >> (8 * (n - 1 - i)));
    int r = put_id(b, id);
    if (!r) r = put_size(b, (uint64_t)n);
    if (!r) r = buf_put(b, t, (size_t)n);
    return r;
}

static int put_str(Buf *b, uint32_t id, const char *s)
{
    int r = put_id(b, id);
    if (!r) r = put_size(b, strlen(s));
    if (!r) r = buf_put(b, s, strlen(s));
    return r;
}

static int put_f64(Buf *b, uint32_t id, double v)
{
    uint64_t u;
    memcpy(&u, &v, 8);
    uint8_t t[8];
    for (int i = 0; i < 8; i++) t[i] = (uint8_t)(u >> (8 * (7 - i)));
    int r = put_id(b, id);
    if (!r) r = put_size(b, 8);
    if (!r) r = buf_put(b, t, 8);
    return r;
}

static int put_master(Buf *b, uint32_t id, const Buf *body)
{
    int r = put_id(b, id);
    if (!r) r = put_size(b, body->n);
    if (!r) r = buf_put(b, body->p, body->n);
    return r;
}

static int64_t pts_to_ms(const ffv2amd_mkv *m, int64_t pts)
{
    /* rounded rescale to the 1 ms TimecodeScale, as av_rescale_q does for the muxer time base */
    const int64_t num = pts * 1000 * m->fps_den;
    return (2 * num + m->fps_num) / (2 * (int64_t)m->fps_num);
}

static int flush_cluster(ffv2amd_mkv *m)
{
    if (m->cluster_ms < 0) return 0;
    Buf head = { 0 }, body = { 0 };
    int r = put_uint(&body, ID_CLUSTERTIMECODE, (uint64_t)m->cluster_ms);
    if (!r) r = buf_put(&body, m->cluster.p, m->cluster.n);
    if (!r) r = put_master(&head, ID_CLUSTER, &body);
    if (!r && fwrite(head.p, 1, head.n, m->f) != head.n) r = -EIO;
    free(head.p); free(body.p);
    m->cluster.n = 0;
    m->cluster_ms = -1;
    return r;
}

int ffv2amd_mkv_open(ffv2amd_mkv **out, const char *path, int width, int height, int fps_num, int fps_den)
{
    if (!out || !path || width < 1 || height < 1 || fps_num < 1 || fps_den < 1) return -EINVAL;
    ffv2amd_mkv *m = (ffv2amd_mkv *)calloc(1, sizeof(*m));
    if (!m) return -ENOMEM;
    m->f = fopen(path, "wb");
    if (!m->f) { free(m); return -EIO; }
    m->fps_num = fps_num; m->fps_den = fps_den;
    m->cluster_ms = -1; m->last_pts = -1;

    Buf file = { 0 }, b = { 0 }, t = { 0 }, v = { 0 };
    int r = 0;
    /* EBML header, matroskaenc.c:1865-1874 (DocType "matroska", version 4, read version 2) */
    r |= put_uint(&b, ID_EBMLVERSION, 1);
    r |= put_uint(&b, ID_EBMLREADVERSION, 1);
    r |= put_uint(&b, ID_EBMLMAXIDLEN, 4);
    r |= put_uint(&b, ID_EBMLMAXSIZELEN, 8);
    r |= put_str(&b, ID_DOCTYPE, "matroska");
    r |= put_uint(&b, ID_DOCTYPEVERSION, 4);
    r |= put_uint(&b, ID_DOCTYPEREADVER, 2);
    r |= put_master(&file, ID_EBML, &b);
    /* Segment with an 8-byte size that close() patches */
    r |= put_id(&file, ID_SEGMENT);
    m->segment_size_pos = (long)file.n;
    r |= put_size8(&file, 0);
    m->segment_data_pos = (long)file.n;
    /* Info, matroskaenc.c:1896-1946: TimecodeScale 1 ms, app strings, Duration as a double */
    Buf info = { 0 }, tracks = { 0 }, seek = { 0 }, one = { 0 };
    b.n = 0;
    r |= put_uint(&b, ID_TIMECODESCALE, 1000000);
    r |= put_str(&b, ID_MUXINGAPP, "ffv2_amd");
    r |= put_str(&b, ID_WRITINGAPP, "ffv2_amd");
    const size_t dur_in_info = b.n;
    r |= put_f64(&b, ID_DURATION, 0.0);
    r |= put_master(&info, ID_INFO, &b);
    /* Tracks, matroskaenc.c:1192-1374: one video TrackEntry */
    r |= put_uint(&v, ID_PIXELWIDTH, (uint64_t)width);
    r |= put_uint(&v, ID_PIXELHEIGHT, (uint64_t)height);
    r |= put_uint(&v, ID_DISPLAYUNIT, 4);          /* MATROSKA_VIDEO_DISPLAYUNIT_UNKNOWN, :1361 */
    r |= put_uint(&t, ID_TRACKNUMBER, 1);
    r |= put_uint(&t, ID_TRACKUID, 1);
    r |= put_uint(&t, ID_FLAGLACING, 0);
    r |= put_str(&t, ID_LANGUAGE, "und");
    r |= put_str(&t, ID_CODECID, "V_FFV2");        /* libavformat/matroska.c:83 */
    r |= put_uint(&t, ID_TRACKTYPE, 1);
    r |= put_uint(&t, ID_DEFAULTDURATION, (uint64_t)(1000000000LL * fps_den / fps_num));   /* :1297 */
    r |= put_master(&t, ID_VIDEO, &v);
    b.n = 0;
    r |= put_master(&b, ID_TRACKENTRY, &t);
    r |= put_master(&tracks, ID_TRACKS, &b);
    /* SeekHead in front (matroskaenc.c:1881-1894 mkv_start_seekhead / :463-517 mkv_write_seekhead): where
     * Info and Tracks start, relative to the first byte of the segment's data.  The stock muxer reserves
     * room for 10 entries there (10 * MAX_SEEKENTRY_SIZE + 19 = 229 bytes, :170,439), fills in the SeekHead
     * when it finishes and leaves the rest as an EBML Void (:509-513, put_ebml_void :294-309): same here, so
     * Info starts 229 bytes into the segment as in a stock file.  (For tracks with key frames the stock
     * muxer also adds Cues at the end; FFV2 packets carry no key-frame flag -- ffv2enc.c never sets
     * AV_PKT_FLAG_KEY and the codec descriptor has no INTRA_ONLY property, codec_desc.c:1757-1763 -- so a
     * stock file has no Cues either.) */
    {
        const uint32_t ids[2] = { ID_INFO, ID_TRACKS };
        const size_t reserved = 10 * 21 + 19;
        const size_t entry = id_bytes(ID_SEEKENTRY) + 1 + (id_bytes(ID_SEEKID) + 1 + 4) + (id_bytes(ID_SEEKPOSITION) + 1 + 4);
        const size_t seekhead_size = id_bytes(ID_SEEKHEAD) + 1 + 2 * entry;
        const size_t pos[2] = { reserved, reserved + info.n };
        for (int i = 0; i < 2; i++) {
            const uint8_t idb[4] = { (uint8_t)(ids[i] >> 24), (uint8_t)(ids[i] >> 16), (uint8_t)(ids[i] >> 8), (uint8_t)ids[i] };
            const uint8_t pb[4] = { (uint8_t)(pos[i] >> 24), (uint8_t)(pos[i] >> 16), (uint8_t)(pos[i] >> 8), (uint8_t)pos[i] };
            one.n = 0;
            r |= put_id(&one, ID_SEEKID);       r |= put_size(&one, 4); r |= buf_put(&one, idb, 4);
            r |= put_id(&one, ID_SEEKPOSITION); r |= put_size(&one, 4); r |= buf_put(&one, pb, 4);
            r |= put_master(&seek, ID_SEEKENTRY, &one);
        }
        r |= put_master(&file, ID_SEEKHEAD, &seek);
        if (!r && file.n - (size_t)m->segment_data_pos != seekhead_size) r = -EINVAL;
        if (!r && seekhead_size + 10 > reserved) r = -EINVAL;
        if (!r) {
            /* Void over what is left of the reservation: id 0xEC, an 8-byte size field (size >= 10), zeros */
            const size_t vsize = reserved - seekhead_size;
            uint8_t hdr[9] = { 0xEC, 0x01, 0, 0, 0, 0, 0, 0, (uint8_t)(vsize - 9) };
            static const uint8_t zeros[256];
            r |= buf_put(&file, hdr, 9);
            r |= buf_put(&file, zeros, vsize - 9);
        }
    }
    const size_t info_at = file.n;
    r |= buf_put(&file, info.p, info.n);
    m->duration_pos = (long)(info_at + id_bytes(ID_INFO) + 1 + dur_in_info + id_bytes(ID_DURATION) + 1);
    r |= buf_put(&file, tracks.p, tracks.n);
    if (r > 0) r = -ENOMEM;
    if (!r && info_at + id_bytes(ID_INFO) + 1 >= file.n) r = -EINVAL;
    if (!r && fwrite(file.p, 1, file.n, m->f) != file.n) r = -EIO;
    free(file.p); free(b.p); free(t.p); free(v.p); free(info.p); free(tracks.p); free(seek.p); free(one.p);
    if (r) { fclose(m->f); free(m); return r; }
    *out = m;
    return 0;
}

int ffv2amd_mkv_write_packet(ffv2amd_mkv *m, const uint8_t *data, size_t size, int64_t pts)
{
    if (!m || (!data && size) || pts < 0 || pts < m->last_pts) return -EINVAL;
    if (m->err) return m->err;
    const int64_t ms = pts_to_ms(m, pts);
    int r = 0;
    if (m->cluster_ms >= 0 &&
        (ms - m->cluster_ms > 32767 || ms - m->cluster_ms > CLUSTER_MS_MAX || m->cluster.n + size > CLUSTER_BYTES_MAX))
        r = flush_cluster(m);
    if (!r && m->cluster_ms < 0) m->cluster_ms = ms;
    /* SimpleBlock: track number (EBML-coded), int16 relative timestamp, flags (matroskaenc.c:2151-2160) */
    const int16_t rel = (int16_t)(ms - m->cluster_ms);
    const uint8_t hdr[4] = { 0x81, (uint8_t)((uint16_t)rel >> 8), (uint8_t)rel, 0x00 };
    if (!r) r = put_id(&m->cluster, ID_SIMPLEBLOCK);
    if (!r) r = put_size(&m->cluster, size + 4);
    if (!r) r = buf_put(&m->cluster, hdr, 4);
    if (!r && size) r = buf_put(&m->cluster, data, size);
    m->last_pts = pts;
    const int64_t end_ms = pts_to_ms(m, pts + 1);
    if (end_ms > m->max_end_ms) m->max_end_ms = end_ms;
    if (r) m->err = r;
    return r;
}

int ffv2amd_mkv_close(ffv2amd_mkv *m)
{
    if (!m) return -EINVAL;
    int r = m->err ? m->err : flush_cluster(m);
    const long end = ftell(m->f);
    if (!r && end < 0) r = -EIO;
    if (!r) {
        Buf b = { 0 };
        r = put_size8(&b, (uint64_t)(end - m->segment_data_pos));
        if (!r && (fseek(m->f, m->segment_size_pos, SEEK_SET) || fwrite(b.p, 1, 8, m->f) != 8)) r = -EIO;
        b.n = 0;
        const double d = (double)m->max_end_ms;
        uint64_t u;
        memcpy(&u, &d, 8);
        uint8_t t[8];
        for (int i = 0; i < 8; i++) t[i] = (uint8_t)(u >> (8 * (7 - i)));
        if (!r && (fseek(m->f, m->duration_pos, SEEK_SET) || fwrite(t, 1, 8, m->f) != 8)) r = -EIO;
        free(b.p);
    }
    if (fclose(m->f) && !r) r = -EIO;
    free(m->cluster.p);
    free(m);
    return r;
}
