/*
 * amvlib_compat.c -- the amvlib call surface of the reference, in host C, on top of the
 * batch C ABI of libamvhip (include/amvhip.h).
 *
 * Mirrors C-AMVDecoder/amvlib/AMVDec.c (AmvOpen :15, AmvClose :131, AmvReadNextFrame :150,
 * AmvRewindFrameStart :240, AmvVideoDecode :259, AmvAudioDecode :288), AmvJpeg.c
 * (PrepareForVideoDecode :1396, AmvJpegDecode :1515) and AdpcmIma.c (AdpcmImaDecodeFrame
 * :206, AdpcmImaEncodeFrame :92): same names, arguments, ownership rules and return codes.
 * Every codec computation goes to the GPU through the batch ABI; this file only moves
 * buffers and walks the container.  One process-wide context is created on first use on
 * the device named by AMVHIP_DEVICE (default 0); like amvlib, these entry points are not
 * re-entrant.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>

#include "../../include/amvhip.h"

static amvhip_ctx *g_ctx;
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void make_ctx(void)
{
    const char *e = getenv("AMVHIP_DEVICE");
    if (amvhip_create(&g_ctx, e ? atoi(e) : 0) != AMVHIP_OK) g_ctx = NULL;
}

static amvhip_ctx *ctx(void)
{
    pthread_once(&g_once, make_ctx);
    return g_ctx;
}

/* ---- AmvJpeg.h ------------------------------------------------------------------------- */

void PrepareForVideoDecode(AMVInfo *info)
{
    (void)info; /* the reference rebuilt its constant tables here; ours live in the code object */
}

int AmvJpegDecode(AMVInfo *info, FRAMEBUFF *inbuff, VIDEOBUFF *video)
{
    amvhip_ctx *c;
    uint64_t off = 0, need;
    uint32_t len;
    int32_t st = 0;
    unsigned char *tmp;
    int rc;

    if (info == NULL) return -1;                                  /* AmvJpeg.c:1519 */
    if (inbuff == NULL || video == NULL || inbuff->videobuff == NULL || video->fbmpdat == NULL) return -1;
    if ((c = ctx()) == NULL) return -1;
    len = inbuff->videobufflen;
    need = amvhip_frame_bytes(info->dwWidth, info->dwHeight);
    if (need == 0) return -1;
    /* amvlib sizes the caller's buffer as W*H*3 (AMVDec.c:277); rows are padded to 4 bytes
     * (AmvJpeg.c:1524), so decode into a full-stride frame and hand back what fits */
    if (video->len != 0 && video->len < need) {
        tmp = (unsigned char *)malloc(need);
        if (tmp == NULL) return -1;
        rc = amvhip_decode_batch(c, inbuff->videobuff, len, &off, &len, 1, info->dwWidth, info->dwHeight, 0, tmp, &st);
        if (rc == AMVHIP_OK) memcpy(video->fbmpdat, tmp, video->len);
        free(tmp);
    } else {
        rc = amvhip_decode_batch(c, inbuff->videobuff, len, &off, &len, 1, info->dwWidth, info->dwHeight, 0,
                                 video->fbmpdat, &st);
    }
    return (rc == AMVHIP_OK && st == 0) ? 0 : -1;                 /* AmvJpeg.c:1531-1538 */
}

int decode_amv_frame(const unsigned char *chunk, unsigned int len, unsigned int width, unsigned int height,
                     unsigned char *bgr_out)
{
    amvhip_ctx *c = ctx();
    uint64_t off = 0;
    int32_t st = 0;
    if (c == NULL || chunk == NULL || bgr_out == NULL) return -1;
    if (amvhip_decode_batch(c, chunk, len, &off, &len, 1, width, height, 0, bgr_out, &st) != AMVHIP_OK) return -1;
    return st == 0 ? 0 : -1;
}

int encode_amv_frame(const unsigned char *pixels, unsigned int stride, unsigned int width, unsigned int height,
                     int is_bgr, unsigned char *chunk_out, unsigned int cap)
{
    amvhip_ctx *c = ctx();
    uint64_t off = 0;
    uint32_t len = 0;
    if (c == NULL || pixels == NULL || chunk_out == NULL) return -1;
    if (amvhip_encode_batch(c, pixels, stride, is_bgr, 1, width, height, AMVHIP_QBIAS_AMV, chunk_out, cap, &off, &len) != AMVHIP_OK)
        return -1;
    return (int)len;
}

/* ---- AdpcmIma.h ------------------------------------------------------------------------ */

int AdpcmImaDecodeFrame(ADPCMContext *c, void *data, int *data_size, unsigned char *buf, int buf_size)
{
    amvhip_ctx *h;
    unsigned char *chunk;
    uint64_t off[2], pcm_off[2];
    uint32_t len[2];
    int32_t fin[4] = { 0, 0, 0, 0 };
    const int st = c != NULL && c->channel == 2;                  /* AdpcmIma.c:222 */
    int n, half, ch, i, rc;
    int16_t *pcm;

    if (data == NULL || !buf_size) return -1;                     /* AdpcmIma.c:216-217 */
    if (c == NULL || buf == NULL || buf_size < 0) return -1;
    if ((h = ctx()) == NULL) return -1;
    /* The reference consumes input 4 bytes at a time, 8 when stereo (:225-237), and so reads up to 3 (7) bytes
     * past buf_size; those bytes are taken as zero here.  Stereo is two mono streams side by side: of every 8 input
     * bytes the first 4 are the left channel's nibbles, the last 4 the right's (src[4*i], :231-234), each channel
     * with its own predictor and step index, and the samples leave interleaved L R.  So each channel's bytes go to
     * the device as a chunk of their own (one batch call) and the host only interleaves. */
    n = st ? (buf_size + 7) & ~7 : (buf_size + 3) & ~3;
    half = st ? n / 2 : n;                                        /* bytes per channel */
    chunk = (unsigned char *)calloc((size_t)(st + 1), (size_t)half + 8);
    pcm = st ? (int16_t *)malloc(4u * (size_t)n) : (int16_t *)data;
    if (chunk == NULL || pcm == NULL) { free(chunk); if (st) free(pcm); return -1; }
    for (ch = 0; ch <= st; ch++) {
        unsigned char *k = chunk + (size_t)ch * ((size_t)half + 8);
        int idx = c->status[ch].step_index;
        if (idx < 0) idx = 0;
        if (idx > 88) idx = 88;
        k[0] = (unsigned char)(c->status[ch].predictor & 0xff);
        k[1] = (unsigned char)((c->status[ch].predictor >> 8) & 0xff);
        k[2] = (unsigned char)idx;
        if (!st) {
            memcpy(k + 8, buf, (size_t)buf_size);
        } else {
            for (i = 0; i < buf_size; i++)
                if (((i >> 2) & 1) == ch) k[8 + ((i >> 3) << 2) + (i & 3)] = buf[i];
        }
        off[ch] = (uint64_t)ch * ((uint64_t)half + 8);
        len[ch] = (uint32_t)half + 8;
        pcm_off[ch] = (uint64_t)ch * 2u * (uint64_t)half;
    }
    rc = amvhip_adpcm_decode_batch(h, chunk, (uint64_t)(st + 1) * ((uint64_t)half + 8), off, len, (uint32_t)(st + 1), pcm,
                                   2ull * (uint64_t)n, pcm_off, fin);
    free(chunk);
    if (rc != AMVHIP_OK) { if (st) free(pcm); return -1; }
    if (st) {
        int16_t *out = (int16_t *)data;
        for (i = 0; i < 2 * half; i++) {
            out[2 * i] = pcm[i];
            out[2 * i + 1] = pcm[2 * half + i];
        }
        free(pcm);
    }
    for (ch = 0; ch <= st; ch++) {
        c->status[ch].predictor = fin[2 * ch];
        c->status[ch].step_index = (short)fin[2 * ch + 1];
    }
    if (data_size) *data_size = 4 * n;                            /* :239 */
    return n;                                                     /* :241 src - buf */
}

int AdpcmImaEncodeFrame(ADPCMContext *c, int channels, int frame_size, unsigned char *frame, int buf_size, void *data)
{
    amvhip_ctx *h;
    int32_t st[2];
    int rc;
    if (c == NULL || frame == NULL || data == NULL || channels != 1) return -1;
    if ((h = ctx()) == NULL) return -1;
    st[0] = c->status[0].prev_sample;
    st[1] = c->status[0].step_index;
    rc = amvhip_adpcm_wav_encode_frame(h, (const int16_t *)data, frame_size, st, frame, buf_size);
    if (rc < 0) return -1;
    c->status[0].prev_sample = st[0];
    c->status[0].step_index = (short)st[1];
    return rc;
}

/* ---- AMVDec.h: container reader --------------------------------------------------------
 * On-disk layout (AMVHeader.h:18-139 with 32-bit DWORDs; offsets in bytes):
 *   0 'RIFF' 8 'AMV ' 12 'LIST' 20 'hdrl' 24 'amvh' 32 us/frame 64 width 68 height 72 fps
 *   84 sec 85 min 86 hour(u16) 88 'LIST' 96 'strl' 100 'strh' 164 'strf' 208 'LIST' 216 'strl'
 *   220 'strh' 276 'strf' 284 WAVEFORMATEX fields 304 'LIST' 312 'movi' 316 first chunk      */

#define AMV_HDR_BYTES 304

static uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static int is4(const unsigned char *p, const char *cc) { return memcmp(p, cc, 4) == 0; }

AMVDecoder *AmvOpen(const char *amvname)
{
    FILE *fp;
    AMVDecoder *amv;
    unsigned char h[AMV_HDR_BYTES + 12];
    size_t got;

    if (amvname == NULL) return NULL;
    amv = (AMVDecoder *)calloc(1, sizeof(AMVDecoder));
    if (amv == NULL) return NULL;
    fp = fopen(amvname, "rb");
    if (!fp) { free(amv); return NULL; }
    got = fread(h, 1, sizeof h, fp);
    fclose(fp);
    /* AMVDec.c:50-93: every four-cc of the fixed header must match */
    if (got != sizeof h || !is4(h, "RIFF") || !is4(h + 8, "AMV ") || !is4(h + 12, "LIST") || !is4(h + 20, "hdrl") ||
        !is4(h + 24, "amvh") || !is4(h + 88, "LIST") || !is4(h + 96, "strl") || !is4(h + 100, "strh") ||
        !is4(h + 164, "strf") || !is4(h + 208, "LIST") || !is4(h + 216, "strl") || !is4(h + 220, "strh") ||
        !is4(h + 276, "strf") || !is4(h + 304, "LIST") || !is4(h + 312, "movi")) {
        free(amv);
        return NULL;
    }
    amv->amvinfo.dwMicroSecPerFrame = rd32(h + 32);               /* :95-101 */
    amv->amvinfo.dwWidth = rd32(h + 64);
    amv->amvinfo.dwHeight = rd32(h + 68);
    amv->amvinfo.dwSpeed = rd32(h + 72);
    amv->amvinfo.dwTimeSec = h[84];
    amv->amvinfo.dwTimeMin = h[85];
    amv->amvinfo.dwTimeHour = rd16(h + 86);
    amv->amvinfo.wFormatTag = rd16(h + 284);                      /* :103-110 */
    amv->amvinfo.nChannels = rd16(h + 286);
    amv->amvinfo.nSamplesPerSec = rd32(h + 288);
    amv->amvinfo.nAvgBytesPerSec = rd32(h + 292);
    amv->amvinfo.nBlockAlign = rd16(h + 296);
    amv->amvinfo.wBitsPerSample = rd16(h + 298);
    amv->amvinfo.cbSize = rd16(h + 300);
    amv->amvinfo.wSamplesPerBlock = rd16(h + 302);
    amv->totalframe = (amv->amvinfo.dwTimeHour * 60 * 60 + amv->amvinfo.dwTimeMin * 60 + amv->amvinfo.dwTimeSec) *
                      amv->amvinfo.dwSpeed;                        /* :112-114 */
    amv->amvfilename = strdup(amvname);
    amv->opened = 1;
    amv->fileseekpos = AMV_HDR_BYTES + 12;
    amv->dataseekpos = amv->fileseekpos;                          /* :117 */
    return amv;
}

/* ---- read-ahead -------------------------------------------------------------------------
 * The reference reopens the file, reads ONE frame and decodes it on the CPU, per call (AMVDec.c:164,259,288).
 * One frame per call is the worst shape for a GPU: a chunk is a few kilobytes and a decode call would be three
 * copies, five launches and a synchronisation for it.  So behind the same three calls the reader takes a WINDOW of
 * frames from the file at once, into page-locked buffers, and the first AmvVideoDecode (AmvAudioDecode) that falls
 * inside the window sends all of its video (audio) chunks through the batch ABI in one go; every decode call then
 * hands out its frame where it lies in the window's result (round 4; a copy into a fresh malloc before).  What the caller
 * sees is unchanged: framebuf holds the chunk's bytes in memory the decoder owns, videobuf / audiobuf are set by the
 * decode calls and stay valid until the next decode call of their kind (the reference frees and mallocs them there,
 * AMVDec.c:277-283,326), positions advance chunk by chunk, AmvRewindFrameStart and edited framebuf contents work (a
 * decode call whose chunk is not byte for byte the window's falls back to decoding that one chunk).
 * AMVHIP_READAHEAD=<frames> sets the window (default 1 024 -- putting a window on the stream costs ~0.4 ms of API calls
 * whatever its size --, never more than 64 MB of decoded frames per window, of which there are three: the one in hand and
 * two in flight; 1 = no read-ahead, one frame per GPU round trip, the copying path). */

typedef struct ra_entry {
    long pos;                 /* file position of the frame's "00dc" */
    uint32_t voff, vlen;      /* video chunk in vblob */
    uint32_t aoff, alen;      /* audio chunk in ablob, alen = real length; the slot is zero-padded to 8 + round4(alen - 8) */
} ra_entry;

/* One window of frames: the chunks as they lie in the file, and the decoded frames / PCM of all of them.  There are two,
 * used in turn (round 4): the decode calls hand out POINTERS into the current window's results instead of copying
 * them into a fresh malloc (the reference's per-frame malloc + memset, AMVDec.c:277-283,326-329, was 10 GB/s of
 * single-thread memory traffic here), and what a caller holds stays untouched until its next decode call of the same
 * kind because the window after is decoded into the OTHER set of buffers -- whose chunks are read, and whose decode is
 * put on the context's stream, from inside that next decode call, so that it runs while the caller walks the frames of
 * this window. */
typedef struct ra_window {
    long start, end;          /* file range of its frames: first frame's position, position behind the last one */
    uint32_t n;               /* frames */
    int vstate, astate;       /* 0 = not sent, 2 = on the stream, 1 = result in vout / aout, -1 = the batch call failed */
    ra_entry *e;
    /* page-locked */
    uint8_t *vblob, *ablob, *vout;
    int16_t *aout;
    size_t vblob_cap, ablob_cap, vout_cap, aout_cap;
    size_t aout_need;         /* what the window's samples take when that is more than aout holds (set by ra_refill, met by ra_issue) */
    size_t vbytes, abytes;    /* filled part of vblob / ablob */
    uint64_t *voffs, *aoffs, *pcm_offs;
    uint32_t *vlens, *alens;
    int32_t *vstatus;
} ra_window;

/* Three windows (round 5; two before): the one frames are handed out of, and TWO in flight behind it -- a window's results
 * come back over PCIe in ~1.6 ms (59 MB of frames), a caller walks its 1 024 frames in ~1 ms, so with one window in flight the
 * caller waited 0.6 ms per window for the link to finish what it had only just been given.  With two in flight the copy
 * stream never runs dry.  A window's buffers of a kind are still only written from inside a decode call of that kind. */
#define RA_WINDOWS 3

typedef struct readahead {
    AMVDecoder *owner;
    struct readahead *next;
    FILE *fp;
    long fsize;
    uint32_t cap_frames;
    uint32_t w, h;
    uint64_t fb;              /* bytes of a decoded frame */
    int pinned;               /* buffers came from amvhip_host_alloc (else malloc: no device in this process) */
    ra_window win[RA_WINDOWS];
    int nwin;                 /* windows in use: RA_WINDOWS, or 1 with AMVHIP_READAHEAD=1 */
    int c;                    /* the window frames are handed out of */
    uint32_t cur;             /* next frame of it to hand out */
    int last;                 /* index (in win[c]) of the frame in framebuf, -1 = none */
    long no_more_at;          /* a look-ahead read at this file position found no complete frame (the end of the stream): not tried again; -1 = none */
    /* what the caller holds of ours (never freed by anybody but the window's owner) */
    unsigned char *lent_v;
    short *lent_a;
    /* framebuf's chunk buffers, kept from frame to frame */
    unsigned char *fbv, *fba;
    size_t fbv_cap, fba_cap;
    size_t frame_bytes;       /* file bytes per frame of the last window read: what the next read is sized by */
} readahead;

static readahead *g_ra;

/* AMVHIP_TRACE_WINDOWS=1: where the reader's time goes (seconds in the file reads, in putting windows on the stream, in
 * waiting for the stream), printed to stderr by AmvClose */
static double g_t_read, g_t_issue, g_t_wait;
static unsigned long g_n_windows;
static int g_trace = -1;
static double ra_now(void)
{
    struct timespec ts;
    if (g_trace < 0) g_trace = getenv("AMVHIP_TRACE_WINDOWS") != NULL;
    if (!g_trace) return 0.0;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static readahead *ra_find(const AMVDecoder *amv)
{
    readahead *r;
    for (r = g_ra; r != NULL; r = r->next)
        if (r->owner == amv) return r;
    return NULL;
}

static void *ra_alloc(readahead *r, size_t bytes)
{
    void *p = NULL;
    if (r->pinned && amvhip_host_alloc(ctx(), &p, bytes) == AMVHIP_OK) return p;
    return r->pinned ? NULL : malloc(bytes ? bytes : 1);
}

static void ra_release(readahead *r, void *p)
{
    if (p == NULL) return;
    if (r->pinned) amvhip_host_free(ctx(), p);
    else free(p);
}

static void ra_free(readahead *r)
{
    int k;
    if (r == NULL) return;
    if (ctx() != NULL) amvhip_sync(ctx());
    if (r->fp) fclose(r->fp);
    for (k = 0; k < RA_WINDOWS; k++) {
        ra_window *w = &r->win[k];
        ra_release(r, w->vblob); ra_release(r, w->ablob); ra_release(r, w->vout); ra_release(r, w->aout);
        ra_release(r, w->voffs); ra_release(r, w->aoffs); ra_release(r, w->pcm_offs);
        ra_release(r, w->vlens); ra_release(r, w->alens); ra_release(r, w->vstatus);
        free(w->e);
    }
    free(r);
}

static uint32_t round4(uint32_t n) { return (n + 3u) & ~3u; }

/* state of `amv`, made on first use */
static readahead *ra_get(AMVDecoder *amv)
{
    readahead *r = ra_find(amv);
    const char *env;
    uint64_t cap;
    int k, ok = 1;
    if (r != NULL) return r;
    r = (readahead *)calloc(1, sizeof *r);
    if (r == NULL) return NULL;
    if (amv->amvinfo.dwWidth > AMVHIP_MAX_DIM || amv->amvinfo.dwHeight > AMVHIP_MAX_DIM) { free(r); return NULL; }   /* (sizes below are products of the two) */
    r->owner = amv;
    r->last = -1;
    r->w = amv->amvinfo.dwWidth;
    r->h = amv->amvinfo.dwHeight;
    r->fb = amvhip_frame_bytes(r->w, r->h);
    r->pinned = ctx() != NULL;
    env = getenv("AMVHIP_READAHEAD");
    cap = env ? (uint64_t)strtoul(env, NULL, 10) : 1024u;
    if (cap < 1) cap = 1;
    if (cap > 4096) cap = 4096;
    while (cap > 1 && cap * (r->fb ? r->fb : 1) > (64u << 20)) cap /= 2;   /* at most 64 MB of decoded frames per window */
    r->cap_frames = (uint32_t)cap;
    r->fp = fopen(amv->amvfilename, "rb");
    if (r->fp != NULL && fseek(r->fp, 0, SEEK_END) == 0) r->fsize = ftell(r->fp);
    r->nwin = cap > 1 ? RA_WINDOWS : 1;                         /* AMVHIP_READAHEAD=1: one frame per round trip, one window */
    r->no_more_at = -1;
    for (k = 0; k < r->nwin; k++) {
        ra_window *w = &r->win[k];
        /* chunk space: AMV streams run at ~0.2 byte per pixel; a window that meets fatter chunks just ends early */
        w->vblob_cap = (size_t)cap * ((size_t)r->w * r->h / 2 + 4096) + 64;
        w->ablob_cap = (size_t)cap * 8192 + 64;
        w->vout_cap = (size_t)cap * (size_t)r->fb + 64;
        w->aout_cap = (size_t)cap * 4 * 8192;                   /* 4 bytes of PCM per chunk byte, chunks of at most 8 KB */
        w->e = (ra_entry *)calloc(cap, sizeof(ra_entry));
        w->vblob = (uint8_t *)ra_alloc(r, w->vblob_cap);
        w->ablob = (uint8_t *)ra_alloc(r, w->ablob_cap);
        w->vout = (uint8_t *)ra_alloc(r, w->vout_cap);
        w->aout = (int16_t *)ra_alloc(r, w->aout_cap);
        w->voffs = (uint64_t *)ra_alloc(r, cap * 8);
        w->aoffs = (uint64_t *)ra_alloc(r, cap * 8);
        w->pcm_offs = (uint64_t *)ra_alloc(r, cap * 8);
        w->vlens = (uint32_t *)ra_alloc(r, cap * 4);
        w->alens = (uint32_t *)ra_alloc(r, cap * 4);
        w->vstatus = (int32_t *)ra_alloc(r, cap * 4);
        ok = ok && w->e && w->vblob && w->ablob && w->vout && w->aout && w->voffs && w->aoffs && w->pcm_offs && w->vlens &&
             w->alens && w->vstatus;
    }
    if (!r->fp || r->fsize <= 0 || !ok) {
        ra_free(r);
        return NULL;
    }
    r->next = g_ra;
    g_ra = r;
    return r;
}

/* a window buffer that the FIRST frame of a window does not fit is replaced by a larger one (nothing is in it yet) */
static int ra_grow(readahead *r, void **buf, size_t *cap, size_t need)
{
    void *p;
    if (need <= *cap) return 0;
    need += need / 2 + 64;
    p = ra_alloc(r, need);
    if (p == NULL) return -1;
    ra_release(r, *buf);
    *buf = p;
    *cap = need;
    return 0;
}

/* everything on the context's stream has finished: results that were on their way are there */
static void ra_landed(readahead *r, int ok)
{
    int k;
    for (k = 0; k < RA_WINDOWS; k++) {
        if (r->win[k].vstate == 2) r->win[k].vstate = ok ? 1 : -1;
        if (r->win[k].astate == 2) r->win[k].astate = ok ? 1 : -1;
    }
}

/* Fill window `w` with the frames that start at `pos`.  Returns the number of frames (0: none complete there),
 * -1 when `pos` holds the end marker.  The file is read a window at a time -- ONE fread of what the frames are expected
 * to take (the last window's bytes per frame, with a margin) into the window's chunk buffer, the container walked in
 * memory; the video chunks are used where they lie (the batch ABI takes chunks anywhere in its blob), the audio chunks
 * are copied to slots of their own, zero-padded to what the reference's 4-byte loop reads.  (Four freads per frame were
 * a third of the reader's time.)  The buffers are sized for the ~0.2 byte per pixel AMV streams run at; a window that
 * meets fatter chunks ends early, and a frame that does not fit an EMPTY window (a noisy picture with
 * AMVHIP_READAHEAD=1, an audio chunk of more than 8 KB) gets larger buffers: the reader has no size limit of its own,
 * as the reference's has none (AMVDec.c:196-231 mallocs what the chunk header says).  Only the chunk buffers are written
 * -- or replaced -- here; vout / aout, what a caller may still hold a pointer into, are written (and aout, when a first
 * frame needs more of it, replaced) by the decode calls of their kind. */
static int ra_refill(readahead *r, ra_window *w, long pos)
{
    const double t0 = ra_now();
    const long start = pos;
    size_t want, have = 0, p = 0, ao = 0, pcm = 0;
    int attempt, result = 0;
    if (ctx() != NULL && (w->vstate == 2 || w->astate == 2)) ra_landed(r, amvhip_sync(ctx()) == AMVHIP_OK);   /* nothing in flight out of the buffers */
    w->n = 0;
    w->vstate = w->astate = 0;
    w->vbytes = w->abytes = 0;
    w->aout_need = 0;
    w->start = w->end = pos;
    want = (size_t)r->cap_frames * (r->frame_bytes ? r->frame_bytes + r->frame_bytes / 8 + 64 : (size_t)r->w * r->h / 4 + 2048) + 65536;
    for (attempt = 0; attempt < 4 && w->n == 0; attempt++) {
        if (want + 64 > w->vblob_cap) {
            if (attempt == 0) want = w->vblob_cap - 64;                                   /* an ordinary window: what the buffer holds */
            else if (ra_grow(r, (void **)&w->vblob, &w->vblob_cap, want + 64) != 0) break; /* a first frame that did not fit */
        }
        if ((long)want > r->fsize - start) want = (size_t)(r->fsize - start);
        if (fseek(r->fp, start, SEEK_SET) != 0) break;
        have = fread(w->vblob, 1, want, r->fp);
        p = 0; ao = 0; pcm = 0; pos = start;
        result = 0;
        while (w->n < r->cap_frames) {
            ra_entry *e = &w->e[w->n];
            const unsigned char *hd = w->vblob + p;
            uint32_t vlen, alen, slot;
            if (p + 8 > have) break;
            if (is4(hd, "AMV_") && is4(hd + 4, "END_")) { result = w->n ? (int)w->n : -1; break; }   /* AMVDec.c:173-190 */
            if (!is4(hd, "00dc")) break;                                                 /* :171,196-208 */
            vlen = rd32(hd + 4);
            if ((long)vlen > r->fsize - pos - 8) break;                                  /* truncated file */
            if (p + 8 + (size_t)vlen + 8 > have) {                                       /* the read ends inside this frame */
                if (w->n == 0) want = 8 + (size_t)vlen + 8 + 8192;
                break;
            }
            hd = w->vblob + p + 8 + vlen;
            if (!is4(hd, "01wb")) { want = 0; break; }                                   /* :213-231 */
            alen = rd32(hd + 4);
            if ((long)alen > r->fsize - pos - 16 - (long)vlen) { want = 0; break; }
            if (p + 16 + (size_t)vlen + alen > have) {
                if (w->n == 0) want = 16 + (size_t)vlen + alen + 64;
                break;
            }
            slot = alen > 8 ? 8 + round4(alen - 8) : 8;
            if (ao + slot + 16 > w->ablob_cap || pcm + 4u * (size_t)(slot - 8) > w->aout_cap) {
                /* a later frame: the window ends here.  The window's FIRST frame gets room: the chunk buffer (nobody holds
                 * a pointer into it) at once, the sample buffer NOT here -- this runs inside AmvReadNextFrame and inside
                 * VIDEO decode calls, while the caller may still hold audiodata out of this window's aout (the last audio
                 * frame of the window before the one before); the need is noted and met by ra_issue, which for audio only
                 * runs inside AmvAudioDecode, the call that ends that pointer's life */
                if (w->n != 0 || ra_grow(r, (void **)&w->ablob, &w->ablob_cap, (size_t)slot + 16) != 0) break;
                if (4u * (size_t)(slot - 8) > w->aout_cap) w->aout_need = 4u * (size_t)(slot - 8);
            }
            memcpy(w->ablob + ao, hd + 8, alen);
            memset(w->ablob + ao + alen, 0, slot - alen);   /* the bytes the reference's 4-byte loop reads past the chunk */
            e->pos = pos;
            e->voff = (uint32_t)(p + 8); e->vlen = vlen;
            e->aoff = (uint32_t)ao; e->alen = alen;
            ao += slot;
            pcm += 4u * (size_t)(slot - 8);
            p += 16 + (size_t)vlen + alen;
            pos += 16 + (long)vlen + (long)alen;
            w->n++;
            w->vbytes = e->voff + (size_t)round4(vlen);
            w->abytes = ao;
            w->end = pos;
        }
        if (result != 0 || want == 0 || (long)want <= 0 || have < 8) break;
        if (w->n == 0 && want + 64 <= w->vblob_cap && want <= have) break;                /* nothing more a larger read would bring */
    }
    if (w->n) r->frame_bytes = (size_t)(w->end - w->start) / w->n;
    g_t_read += ra_now() - t0;
    g_n_windows++;
    return result != 0 ? result : (int)w->n;
}

/* vb->fbmpdat / ab->audiodata are the caller's to read, ours to replace: free what malloc made, forget what was lent */
static void drop_video(AMVDecoder *amv, readahead *r)
{
    if (r == NULL || amv->videobuf.fbmpdat != r->lent_v) free(amv->videobuf.fbmpdat);
    amv->videobuf.fbmpdat = NULL;
    if (r != NULL) r->lent_v = NULL;
}

static void drop_audio(AMVDecoder *amv, readahead *r)
{
    if (r == NULL || amv->audiobuf.audiodata != r->lent_a) free(amv->audiobuf.audiodata);
    amv->audiobuf.audiodata = NULL;
    if (r != NULL) r->lent_a = NULL;
}

static void drop_chunks(AMVDecoder *amv, readahead *r)
{
    FRAMEBUFF *fb = &amv->framebuf;
    if (r == NULL || fb->videobuff != r->fbv) free(fb->videobuff);
    if (r == NULL || fb->audiobuff != r->fba) free(fb->audiobuff);
    fb->videobuff = fb->audiobuff = NULL;
    fb->videobufflen = fb->audiobufflen = 0;
}

void AmvClose(AMVDecoder *amv)
{
    readahead *r, **pp;
    if (amv == NULL) return;                                      /* AMVDec.c:131-148 */
    r = ra_find(amv);
    drop_video(amv, r);
    drop_audio(amv, r);
    drop_chunks(amv, r);
    if (r != NULL) {
        free(r->fbv);
        free(r->fba);
    }
    for (pp = &g_ra; (r = *pp) != NULL; pp = &r->next)
        if (r->owner == amv) { *pp = r->next; ra_free(r); break; }
    free(amv->amvfilename);
    free(amv);
    if (g_trace > 0)
        fprintf(stderr, "amvhip reader: %lu windows, %.3f ms reading the file, %.3f ms putting them on the stream, %.3f ms waiting for it\n",
                g_n_windows, 1e3 * g_t_read, 1e3 * g_t_issue, 1e3 * g_t_wait);
}

/* the chunk's bytes into memory framebuf owns (the reference frees and mallocs per frame, AMVDec.c:196-231; here the
 * buffer is kept and grown).  A copy, not a pointer into the window: a caller may edit framebuf in place, and the
 * decode calls tell an edited chunk from the window's by comparing the two. */
static int hand_out(unsigned char **buf, unsigned int *len, unsigned char **mine, size_t *cap, const unsigned char *src, uint32_t n)
{
    if (*buf != *mine) {                                          /* somebody else's malloc: replaced as the reference would */
        free(*buf);
        *buf = NULL;
    }
    if (*mine == NULL || *cap < n) {
        const size_t want = (size_t)n + (size_t)n / 2 + 64;
        unsigned char *p = (unsigned char *)malloc(want);
        if (p == NULL) { *buf = NULL; *len = 0; return -1; }
        free(*mine);
        *mine = p;
        *cap = want;
    }
    *buf = *mine;
    memcpy(*buf, src, n);
    *len = n;
    return 0;
}

int AmvReadNextFrame(AMVDecoder *amv)
{
    FRAMEBUFF *fb;
    readahead *r;
    ra_window *w;
    const ra_entry *e;

    if (amv == NULL) return -1;                                   /* AMVDec.c:157-160 */
    if (!amv->opened || amv->amvfilename == NULL) return -1;
    fb = &amv->framebuf;
    if ((r = ra_get(amv)) == NULL) return -1;
    w = &r->win[r->c];
    if (!(r->cur < w->n && w->e[r->cur].pos == amv->fileseekpos)) {   /* window used up, or the position was moved */
        const int o = (r->c + 1) % r->nwin;
        ra_window *nw = &r->win[o];
        int got;
        if (r->nwin > 1 && nw->n > 0 && nw->start == amv->fileseekpos) {
            got = (int)nw->n;                                     /* read (and being decoded) since a decode call of the window before */
        } else {
            got = ra_refill(r, nw, amv->fileseekpos);             /* (a window that holds other frames -- the position was moved -- is simply read again) */
        }
        if (got < 0) {                                            /* :173-190 end of stream */
            drop_chunks(amv, r);
            fb->framenum = -1;
            amv->fileseekpos += 8;
            return 0;
        }
        if (got == 0) return -1;                                  /* no complete frame here: nothing changes */
        r->c = o;
        r->cur = 0;
        r->last = -1;
        w = nw;
    }
    e = &w->e[r->cur];
    /* both chunks or neither (a reader that took the video chunk and then failed on the audio chunk would be out of
     * step with the file) */
    if (hand_out(&fb->videobuff, &fb->videobufflen, &r->fbv, &r->fbv_cap, w->vblob + e->voff, e->vlen) != 0) return -1;
    if (hand_out(&fb->audiobuff, &fb->audiobufflen, &r->fba, &r->fba_cap, w->ablob + e->aoff, e->alen) != 0) {
        fb->videobufflen = 0;
        return -1;
    }
    amv->fileseekpos += 16 + (long)e->vlen + (long)e->alen;
    r->last = (int)r->cur++;
    fb->framenum++;                                               /* :233-234 */
    amv->currentframe = (unsigned int)fb->framenum;
    return 0;
}

int AmvRewindFrameStart(AMVDecoder *amv)
{
    if (amv == NULL) return -1;                                   /* AMVDec.c:244-247 */
    if (!amv->opened || amv->amvfilename == NULL) return -1;
    amv->fileseekpos = amv->dataseekpos;                          /* :253 */
    return 0;
}

/* the window's frame that framebuf still holds, byte for byte; NULL -> decode the single chunk */
static readahead *ra_current(AMVDecoder *amv, int video)
{
    readahead *r = ra_find(amv);
    const ra_window *w;
    const ra_entry *e;
    const FRAMEBUFF *fb = &amv->framebuf;
    if (r == NULL || r->last < 0 || ctx() == NULL) return NULL;
    if (r->w != amv->amvinfo.dwWidth || r->h != amv->amvinfo.dwHeight) return NULL;
    w = &r->win[r->c];
    e = &w->e[r->last];
    if (video) return (fb->videobufflen == e->vlen && memcmp(fb->videobuff, w->vblob + e->voff, e->vlen) == 0) ? r : NULL;
    return (fb->audiobufflen == e->alen && e->alen > 8 && memcmp(fb->audiobuff, w->ablob + e->aoff, e->alen) == 0) ? r : NULL;
}

/* put the window's video (audio) chunks on the context's stream through the batch ABI, once per window */
static void ra_issue(readahead *r, ra_window *w, int video)
{
    uint32_t i;
    const uint32_t width = r->w, height = r->h;
    int *state = video ? &w->vstate : &w->astate;
    const double t0 = ra_now();
    if (*state != 0 || w->n == 0) return;
    /* (audio) inside AmvAudioDecode: nothing the caller holds of this window's samples outlives this call, and nothing is
     * in flight into them (state 0) -- the moment a sample buffer too small for the window's first frame is replaced */
    if (!video && w->aout_need > w->aout_cap && ra_grow(r, (void **)&w->aout, &w->aout_cap, w->aout_need) != 0) {
        *state = -1;
        return;
    }
    if (video) {
        for (i = 0; i < w->n; i++) { w->voffs[i] = w->e[i].voff; w->vlens[i] = w->e[i].vlen; }
        *state = amvhip_decode_batch_async(ctx(), w->vblob, w->vbytes + 16, w->voffs, w->vlens, w->n, width, height, 0, w->vout,
                                           w->vstatus) == AMVHIP_OK ? 2 : -1;
    } else {
        uint64_t po = 0;
        for (i = 0; i < w->n; i++) {
            const uint32_t slot = w->e[i].alen > 8 ? 8 + round4(w->e[i].alen - 8) : 8;
            w->aoffs[i] = w->e[i].aoff; w->alens[i] = slot; w->pcm_offs[i] = po;
            po += 2ull * (slot - 8);
        }
        *state = amvhip_adpcm_decode_batch_async(ctx(), w->ablob, w->abytes + 16, w->aoffs, w->alens, w->n, w->aout, po, w->pcm_offs,
                                                 NULL) == AMVHIP_OK ? 2 : -1;
    }
    g_t_issue += ra_now() - t0;
}

/* the current window's video (audio) result: 1 = there */
static int ra_decode_window(readahead *r, int video)
{
    ra_window *w = &r->win[r->c];
    int *state = video ? &w->vstate : &w->astate;
    ra_issue(r, w, video);
    if (*state == 2) {
        const double t0 = ra_now();
        ra_landed(r, amvhip_sync(ctx()) == AMVHIP_OK);
        g_t_wait += ra_now() - t0;
    }
    return *state;
}

/* From inside a decode call of kind `video`: whatever the caller held of that kind from the window before is dead by
 * the call's contract, so the OTHER windows' buffers of that kind may be written now -- the frames behind this window are
 * read into them (once: a window that already holds the frames expected at its place is left alone) and their decode of this
 * kind is put on the stream; it runs while the caller walks this window and the next.  Called per frame: everything is a few
 * comparisons once the windows are in place. */
static void ra_look_ahead(readahead *r, int video)
{
    int k;
    if (r->nwin <= 1) return;
    for (k = 1; k < r->nwin; k++) {
        const ra_window *before = &r->win[(r->c + k - 1) % r->nwin];
        ra_window *o = &r->win[(r->c + k) % r->nwin];
        if (before->n == 0) return;
        if (!(o->n > 0 && o->start == before->end)) {
            int got;
            if (r->no_more_at == before->end) return;             /* the end of the stream: the reader finds out itself */
            got = ra_refill(r, o, before->end);
            if (got <= 0) { o->n = 0; r->no_more_at = before->end; return; }
        }
        ra_issue(r, o, video);
    }
}

int AmvVideoDecode(AMVDecoder *amv)
{
    FRAMEBUFF *fb;
    VIDEOBUFF *vb;
    readahead *r, *any;
    uint64_t full;
    size_t size;

    if (amv == NULL) return -1;                                   /* AMVDec.c:265-268 */
    if (!amv->opened) return -1;
    fb = &amv->framebuf;
    if (fb->videobuff == NULL || fb->videobufflen == 0) return -1; /* :271-272 */
    vb = &amv->videobuf;
    any = ra_find(amv);
    /* a header that claims a picture the decoder refuses (a damaged file: 8 million pixels wide) is refused HERE, before
     * W*H*3 bytes are allocated and cleared for it frame after frame (what AMVDec.c:277-283 would do: gigabytes) */
    if (amv->amvinfo.dwWidth == 0 || amv->amvinfo.dwHeight == 0 || amv->amvinfo.dwWidth > AMVHIP_MAX_DIM ||
        amv->amvinfo.dwHeight > AMVHIP_MAX_DIM) {
        drop_video(amv, any);
        vb->len = 0;
        return -1;
    }
    full = amvhip_frame_bytes(amv->amvinfo.dwWidth, amv->amvinfo.dwHeight);
    vb->len = amv->amvinfo.dwHeight * amv->amvinfo.dwWidth * 3;   /* :277 */
    size = full > vb->len ? (size_t)full : (vb->len ? vb->len : 1);
    if ((r = ra_current(amv, 1)) != NULL && ra_decode_window(r, 1) == 1) {
        /* the whole frame, padding and undecoded remainder included, comes zeroed from the decoder (:283) */
        const ra_window *w = &r->win[r->c];
        unsigned char *frame = w->vout + (size_t)r->last * (size_t)r->fb;
        const int rc = w->vstatus[r->last] == 0 ? 0 : -1;         /* AmvJpeg.c:1531-1538 */
        if (r->cap_frames > 1) {                                  /* lend the window's frame: valid until the next AmvVideoDecode */
            drop_video(amv, r);
            vb->fbmpdat = r->lent_v = frame;
            ra_look_ahead(r, 1);
            return rc;
        }
        drop_video(amv, any);
        vb->fbmpdat = (unsigned char *)malloc(size);
        if (vb->fbmpdat == NULL) return -2;                       /* :281-282 */
        memcpy(vb->fbmpdat, frame, (size_t)full);
        return rc;
    }
    drop_video(amv, any);
    vb->fbmpdat = (unsigned char *)malloc(size);
    if (vb->fbmpdat == NULL) return -2;                           /* :281-282 */
    memset(vb->fbmpdat, 0, size);                                 /* :283 */
    if (full > vb->len) {                                         /* room for the padded rows: decode in place */
        unsigned int keep = vb->len;
        int rc;
        vb->len = (unsigned int)full;
        rc = AmvJpegDecode(&amv->amvinfo, fb, vb);
        vb->len = keep;
        return rc;
    }
    return AmvJpegDecode(&amv->amvinfo, fb, vb);                  /* :285 */
}

int AmvAudioDecode(AMVDecoder *amv)
{
    FRAMEBUFF *fb;
    AUDIOBUFF *ab;
    ADPCMContext audio;
    readahead *r, *any;
    int rtn, declen = 0;

    if (amv == NULL) return -1;                                   /* AMVDec.c:296-299 */
    if (!amv->opened) return -1;
    fb = &amv->framebuf;
    if (fb->audiobuff == NULL || fb->audiobufflen == 0) return -1; /* :302-303 */
    if (fb->audiobufflen <= 8) return -1;
    ab = &amv->audiobuf;
    any = ra_find(amv);
    ab->len = rd32(fb->audiobuff + 4) * 2;                        /* :319-321 */
    if (ab->len < (fb->audiobufflen - 8) * 4) ab->len = (fb->audiobufflen - 8) * 4;  /* :322-323 */
    if (amv->amvinfo.nChannels != 2 && (r = ra_current(amv, 0)) != NULL && ra_decode_window(r, 0) == 1) {
        const ra_window *w = &r->win[r->c];
        const uint32_t n4 = round4(fb->audiobufflen - 8);         /* what AdpcmImaDecodeFrame consumes, AdpcmIma.c:225-241 */
        short *pcm = w->aout + w->pcm_offs[r->last];
        if (r->cap_frames > 1) {                                  /* lend the window's samples: valid until the next AmvAudioDecode */
            drop_audio(amv, r);
            ab->audiodata = r->lent_a = pcm;
            ab->len = 4u * n4;                                    /* :333-337 */
            ra_look_ahead(r, 0);
            return 0;
        }
        drop_audio(amv, any);
        ab->audiodata = (short *)malloc((size_t)ab->len + 32);    /* :326 */
        if (ab->audiodata == NULL) return -2;
        memset(ab->audiodata, 0, ab->len);                        /* :329 */
        memcpy(ab->audiodata, pcm, 4u * (size_t)n4);
        ab->len = 4u * n4;                                        /* :333-337 */
        return 0;
    }
    drop_audio(amv, any);
    ab->audiodata = (short *)malloc((size_t)ab->len + 32);        /* :326 (+ what the 4 / 8-byte loop writes past it) */
    if (ab->audiodata == NULL) return -2;
    memset(ab->audiodata, 0, ab->len);                            /* :329 */
    memset(&audio, 0, sizeof audio);
    audio.channel = amv->amvinfo.nChannels;
    audio.status[0].predictor = (short)rd16(fb->audiobuff);       /* :312 */
    audio.status[0].step_index = fb->audiobuff[2];                /* :313 */
    audio.status[1] = audio.status[0];                            /* :316-317 */
    rtn = AdpcmImaDecodeFrame(&audio, ab->audiodata, &declen, fb->audiobuff + 8, (int)fb->audiobufflen - 8);
    if (rtn > 0) {                                                /* :333-337 */
        ab->len = (unsigned int)declen;
        return 0;
    }
    return rtn;
}

/* ---- export helpers: AMVDec.c:342-547, AmvJpeg.c:315-414,1289-1393 ---------------------- */

void AmvJpegPutHeader(FILE *fp, unsigned short height, unsigned short width)
{
    unsigned char hdr[1024];
    uint32_t n = amvhip_jpeg_header(height, width, hdr, sizeof hdr);
    if (fp != NULL && n <= sizeof hdr) fwrite(hdr, n, 1, fp);
}

int AmvCreateJpegFileFromBuffer(AMVInfo *amvinfo, FRAMEBUFF *framebuf, const char *filename)
{
    FILE *fp;
    if (amvinfo == NULL || framebuf == NULL || filename == NULL || framebuf->videobuff == NULL ||
        framebuf->videobufflen < 2)
        return -1;
    fp = fopen(filename, "wb");                                   /* AMVDec.c:365-367 */
    if (fp == NULL) return -1;
    AmvJpegPutHeader(fp, (unsigned short)amvinfo->dwHeight, (unsigned short)amvinfo->dwWidth);
    fwrite(framebuf->videobuff + 2, framebuf->videobufflen - 2, 1, fp);   /* :371 the chunk minus its SOI */
    fclose(fp);
    return 0;
}

int AmvCreateJpegFileFromFrameBuffer(AMVDecoder *amv, const char *dirname)
{
    char name[512];
    if (amv == NULL || dirname == NULL) return -1;
    snprintf(name, sizeof name, "%s-amvjpg_%06d_.jpg", dirname, amv->framebuf.framenum);   /* AMVDec.c:347 */
    return AmvCreateJpegFileFromBuffer(&amv->amvinfo, &amv->framebuf, name);
}

static void wr16(unsigned char *p, unsigned v) { p[0] = (unsigned char)v; p[1] = (unsigned char)(v >> 8); }
static void wr32(unsigned char *p, uint32_t v) { wr16(p, v & 0xffffu); wr16(p + 2, v >> 16); }

/* AmvJpeg.h:94 / AmvJpeg.c:1289-1393: the name amvlib's own AmvConvertJpegFileToBmpFile forwards to (AMVDec.c:376-382);
 * same file-level contract as that one (it has no NULL check of its own in the reference: here it gets the wrapper's) */
int ConvertJpegFileToBmpFile(const char *jpgname, const char *bmpname)
{
    return AmvConvertJpegFileToBmpFile(jpgname, bmpname);
}

int AmvConvertJpegFileToBmpFile(const char *jpgname, const char *bmpname)
{
    FILE *fp;
    long size;
    unsigned char *jpg = NULL, *bmp = NULL, want[1024];
    uint32_t hdr, w, h, stride, img;
    int rc = -1;

    if (jpgname == NULL || bmpname == NULL) return -1;            /* AMVDec.c:378-379 */
    fp = fopen(jpgname, "rb");
    if (fp == NULL) return -1;
    fseek(fp, 0, SEEK_END);
    size = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    hdr = amvhip_jpeg_header(0, 0, NULL, 0);
    if (size > (long)hdr + 2 && (jpg = (unsigned char *)malloc((size_t)size)) != NULL &&
        fread(jpg, 1, (size_t)size, fp) == (size_t)size) {
        /* SOF0 sits at a fixed place in the header this library (and amvlib) writes: SOI 2, APP0 18, DQT 2 x 69 */
        const uint32_t sof = 2 + 18 + 2 * 69;
        h = ((uint32_t)jpg[sof + 5] << 8) | jpg[sof + 6];
        w = ((uint32_t)jpg[sof + 7] << 8) | jpg[sof + 8];
        amvhip_jpeg_header((uint16_t)h, (uint16_t)w, want, sizeof want);
        if (w && h && w <= AMVHIP_MAX_DIM && h <= AMVHIP_MAX_DIM &&  /* (stride * h below fits 32 bits; a larger picture is refused by the decoder anyway) */
            memcmp(jpg, want, hdr) == 0) {                        /* amvlib's tables and 4:2:0, nothing else */
            stride = amvhip_stride(w);                            /* WIDTHBYTES, AmvJpeg.c:1343 */
            img = stride * h;
            bmp = (unsigned char *)calloc(1, 54 + (size_t)img);
            if (bmp != NULL) {
                /* the scan behind the header is a chunk without its SOI: give it one in place */
                unsigned char *chunk = jpg + hdr - 2;
                chunk[0] = 0xff;
                chunk[1] = 0xd8;
                if (decode_amv_frame(chunk, (unsigned int)((uint32_t)size - hdr + 2), w, h, bmp + 54) == 0) {
                    bmp[0] = 'B'; bmp[1] = 'M';                   /* BITMAPFILEHEADER :1346-1348 */
                    wr32(bmp + 2, 54 + img);
                    wr32(bmp + 10, 54);
                    wr32(bmp + 14, 40);                           /* BITMAPINFOHEADER :1334-1341 */
                    wr32(bmp + 18, w);
                    wr32(bmp + 22, h);
                    wr16(bmp + 26, 1);
                    wr16(bmp + 28, 24);
                    fclose(fp);
                    fp = fopen(bmpname, "wb");                    /* :1374-1377 */
                    if (fp != NULL && fwrite(bmp, 1, 54 + (size_t)img, fp) == 54 + (size_t)img) rc = 0;
                }
            }
        }
    }
    if (fp != NULL) fclose(fp);
    free(jpg);
    free(bmp);
    return rc;
}

int AmvCreateWavFileFromAmvFile(AMVDecoder *amv, int type, const char *wavfile)
{
    static const unsigned char adpcminfo[2] = {0xF9, 0x03};      /* AMVDec.c:386-390: wSamplesPerBlock 1017 */
    unsigned char h[64], pre_index[4] = {0, 0, 0, 0};
    long dataseekpos_save, fileseekpos_save;
    uint32_t totlen = 0, n = 0;
    int first = 0, adpcm;
    AMVInfo *info;
    FILE *fp;

    if (amv == NULL || wavfile == NULL) return -1;                /* :409-414 */
    if (!amv->opened) return -1;
    if (!(type == AUDIO_FILE_TYPE_PCM || type == AUDIO_FILE_TYPE_ADPCM_IMA)) return -1;
    adpcm = type == AUDIO_FILE_TYPE_ADPCM_IMA;
    dataseekpos_save = amv->dataseekpos;                          /* :416-417 */
    fileseekpos_save = amv->fileseekpos;
    fp = fopen(wavfile, "wb");
    if (fp == NULL) return -1;
    info = &amv->amvinfo;

    memcpy(h + n, "RIFF", 4); n += 4;                             /* :426-477 */
    wr32(h + n, 38); n += 4;
    memcpy(h + n, "WAVEfmt ", 8); n += 8;
    wr32(h + n, adpcm ? 0x14 : 18); n += 4;
    wr16(h + n, adpcm ? 0x11 : info->wFormatTag); n += 2;
    wr16(h + n, info->nChannels); n += 2;
    wr32(h + n, info->nSamplesPerSec); n += 4;
    wr32(h + n, adpcm ? info->nAvgBytesPerSec / 4 : info->nAvgBytesPerSec); n += 4;
    wr16(h + n, info->nBlockAlign); n += 2;
    wr16(h + n, adpcm ? info->wBitsPerSample / 4 : info->wBitsPerSample); n += 2;
    wr16(h + n, adpcm ? 2 : info->cbSize); n += 2;
    if (adpcm) { memcpy(h + n, adpcminfo, 2); n += 2; }
    memcpy(h + n, "data", 4); n += 4;
    wr32(h + n, 0); n += 4;
    if (adpcm) { wr32(h + n, 0); n += 4; }                        /* :486-490 room for the first chunk's predictor/index */
    fwrite(h, 1, n, fp);

    for (;;) {                                                    /* :492-529 */
        FRAMEBUFF *fb = &amv->framebuf;
        if (AmvReadNextFrame(amv) != 0) break;
        if (fb->framenum == -1) break;
        if (fb->audiobuff == NULL || fb->audiobufflen < 8) continue;
        if (adpcm) {
            if (!first) { memcpy(pre_index, fb->audiobuff, 4); first = 1; }
            fwrite(fb->audiobuff + 8, 1, fb->audiobufflen - 8, fp);
            totlen += fb->audiobufflen - 8;
        } else if (AmvAudioDecode(amv) == 0) {
            fwrite(amv->audiobuf.audiodata, 1, amv->audiobuf.len, fp);
            totlen += amv->audiobuf.len;
        }
    }
    if (adpcm && (totlen & 1u)) totlen -= 1;                      /* :533-535 */
    fseek(fp, 4, SEEK_SET);
    wr32(h, adpcm ? totlen + 0x28 : totlen + 38);                 /* :536-539 */
    fwrite(h, 1, 4, fp);
    fseek(fp, adpcm ? 0x2C : 42, SEEK_SET);                       /* :541-545 */
    wr32(h, totlen);
    fwrite(h, 1, 4, fp);
    if (adpcm) fwrite(pre_index, 1, 4, fp);                       /* :547-550 */
    fclose(fp);
    amv->dataseekpos = dataseekpos_save;                          /* :554-555 */
    amv->fileseekpos = fileseekpos_save;
    return 0;
}
