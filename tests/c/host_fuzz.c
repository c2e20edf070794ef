/*
 * host_fuzz.c -- TEST INFRASTRUCTURE: walks mutations of an AMV file through the product's host C (the amvlib reader
 * and decode calls of host/amvlib_compat.c, the muxer of host/amv_container.c), built with AddressSanitizer + UBSan
 * against tests/c/host_stub.c (tests/test_abi_and_host.py::test_host_c_under_sanitizers).  The reader parses untrusted
 * files -- chunk lengths from the file size its reads and its buffers (the reference's own reader: AMVDec.c:150-238,
 * which mallocs what a chunk header says) -- so every structural field is attacked: truncation at every header byte and
 * around chunk headers, chunk lengths of 0 / 1 / 7..9 / 2^31 - 1 / 2^31 / 2^32 - 1 / "what is left" +- 1, the end
 * marker missing, 00dc and 01wb swapped, every header byte forced to 00 / 7f / ff, and seeded random damage.
 * A mutation passes when every call returns one of the codes the reference's API has for it and the sanitizers stay
 * silent; the program prints one "ok ..." line and exits 0.
 *
 *     host_fuzz <base.amv> <workdir> <random mutations> <seed>
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "amvhip.h"

static uint8_t *g_base;
static size_t g_len;
static char g_path[512], g_aux[512], g_aux2[512];
static unsigned long n_mut, n_open, n_frames, n_end, n_vok, n_aok, n_bad;
static uint64_t g_rng;

static uint32_t rnd(void)
{
    g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17;
    return (uint32_t)(g_rng >> 16);
}

static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t get32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

static void fail(const char *what, long v)
{
    printf("FAIL mutation %lu: %s = %ld\n", n_mut, what, v);
    n_bad++;
}

static void walk(const char *path, int extras)
{
    AMVDecoder *d = AmvOpen(path);
    volatile unsigned sink = 0;
    int pass, k, rc, most;
    if (d == NULL) return;
    n_open++;
    /* a header that claims a huge picture makes every decode call clear W*H*3 bytes, as the reference's does (AMVDec.c:283):
     * a few frames of that are as good a test as many */
    most = (uint64_t)d->amvinfo.dwWidth * d->amvinfo.dwHeight > 512u * 512u ? 4 : 150;
    for (pass = 0; pass < 2; pass++) {
        /* what the caller holds from the decode calls before: the contract (AMVDec.c:277-283, :326-329 free and malloc
         * them in the NEXT decode call of their kind) keeps them readable across AmvReadNextFrame and across the decode
         * call of the OTHER kind -- read here, under AddressSanitizer, at both points */
        const unsigned char *held_v = NULL;
        const short *held_a = NULL;
        unsigned held_vlen = 0, held_alen = 0;
        for (k = 0; k < (pass ? 3 : most); k++) {
            rc = AmvReadNextFrame(d);
            if (held_v != NULL && held_vlen) sink += held_v[0] + held_v[held_vlen - 1];
            if (held_a != NULL && held_alen >= 2) sink += (unsigned)held_a[0] + (unsigned)held_a[held_alen / 2 - 1];
            if (rc != 0) { if (rc != -1) fail("AmvReadNextFrame", rc); break; }
            if (d->framebuf.framenum == -1) { n_end++; break; }
            n_frames++;
            if (d->framebuf.videobuff == NULL || d->framebuf.audiobuff == NULL) { fail("framebuf pointer NULL after a successful read", 0); break; }
            if (d->framebuf.videobufflen) sink += d->framebuf.videobuff[0] + d->framebuf.videobuff[d->framebuf.videobufflen - 1];
            if (d->framebuf.audiobufflen) sink += d->framebuf.audiobuff[0] + d->framebuf.audiobuff[d->framebuf.audiobufflen - 1];
            rc = AmvVideoDecode(d);
            held_v = NULL;
            if (rc == 0) {
                n_vok++;
                if (d->videobuf.len) sink += d->videobuf.fbmpdat[0] + d->videobuf.fbmpdat[d->videobuf.len - 1];
                held_v = d->videobuf.fbmpdat; held_vlen = d->videobuf.len;
            } else if (rc != -1 && rc != -2) fail("AmvVideoDecode", rc);
            if (held_a != NULL && held_alen >= 2) sink += (unsigned)held_a[0] + (unsigned)held_a[held_alen / 2 - 1];   /* still the caller's */
            rc = AmvAudioDecode(d);
            held_a = NULL;
            if (rc == 0) {
                n_aok++;
                if (d->audiobuf.len >= 2) sink += (unsigned)d->audiobuf.audiodata[0] + (unsigned)d->audiobuf.audiodata[d->audiobuf.len / 2 - 1];
                held_a = d->audiobuf.audiodata; held_alen = d->audiobuf.len;
            } else if (rc > 0 || rc < -2) fail("AmvAudioDecode", rc);
            if (held_v != NULL && held_vlen) sink += held_v[0] + held_v[held_vlen - 1];
            if (extras && k == 1) {
                (void)AmvCreateJpegFileFromBuffer(&d->amvinfo, &d->framebuf, g_aux);
                (void)AmvConvertJpegFileToBmpFile(g_aux, g_aux2);
            }
        }
        if (AmvRewindFrameStart(d) != 0) fail("AmvRewindFrameStart", -1);
        d->framebuf.framenum = 0;
    }
    if (extras) {
        rc = AmvCreateWavFileFromAmvFile(d, AUDIO_FILE_TYPE_ADPCM_IMA, g_aux);
        if (rc != 0 && rc != -1) fail("AmvCreateWavFileFromAmvFile", rc);
    }
    AmvClose(d);
    (void)sink;
}

static void run(const uint8_t *data, size_t len)
{
    FILE *f = fopen(g_path, "wb");
    if (f == NULL) { printf("cannot write %s\n", g_path); exit(2); }
    if (len) fwrite(data, 1, len, f);
    fclose(f);
    walk(g_path, (n_mut & 63) == 0);
    n_mut++;
}

/* chunk header positions of the (undamaged) base file: hdr[i] = offset of frame i's "00dc" */
static size_t g_hdr[4096];
static unsigned g_nhdr;
static size_t g_data;

static void index_base(void)
{
    size_t p;
    for (p = 0; p + 8 <= g_len; p++)
        if (memcmp(g_base + p, "movi", 4) == 0) break;
    g_data = p + 4;
    p = g_data;
    while (p + 16 <= g_len && g_nhdr < 4096 && memcmp(g_base + p, "00dc", 4) == 0) {
        const uint32_t vl = get32(g_base + p + 4);
        uint32_t al;
        g_hdr[g_nhdr++] = p;
        if (p + 8 + vl + 8 > g_len) break;
        al = get32(g_base + p + 8 + vl + 4);
        p += 16 + (size_t)vl + al;
    }
}

int main(int argc, char **argv)
{
    FILE *f;
    uint8_t *m;
    unsigned i, j, k, iters;
    static const uint32_t lens[] = {0u, 1u, 7u, 8u, 9u, 0x7fffffffu, 0x80000000u, 0xffffffffu};
    static const uint8_t force[] = {0x00, 0x7f, 0xff};
    static const int around[] = {-1, 0, 1, 3, 4, 7, 8, 9, 12, 15, 16};
    amvhip_muxer *mx;

    if (argc < 5) { printf("usage: host_fuzz base.amv workdir iterations seed\n"); return 2; }
    f = fopen(argv[1], "rb");
    if (f == NULL) return 2;
    fseek(f, 0, SEEK_END);
    g_len = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    g_base = (uint8_t *)malloc(g_len + 64);
    m = (uint8_t *)malloc(g_len + 64);
    if (g_base == NULL || m == NULL || fread(g_base, 1, g_len, f) != g_len) return 2;
    fclose(f);
    snprintf(g_path, sizeof g_path, "%s/mut.amv", argv[2]);
    snprintf(g_aux, sizeof g_aux, "%s/aux.out", argv[2]);
    snprintf(g_aux2, sizeof g_aux2, "%s/aux2.out", argv[2]);
    iters = (unsigned)atoi(argv[3]);
    g_rng = 0x9e3779b97f4a7c15ull ^ (uint64_t)strtoull(argv[4], NULL, 0);
    index_base();
    if (g_nhdr < 8) { printf("base file has no frames\n"); return 2; }

    run(g_base, g_len);                                             /* the file itself */
    /* A. cut short: every byte of the header, around chunk headers of the first and last frames, the tail */
    for (i = 0; i <= g_data + 40 && i < g_len; i++) run(g_base, i);
    for (j = 0; j < 6; j++)
        for (k = 0; k < sizeof around / sizeof around[0]; k++) {
            const unsigned fr = j < 3 ? j : g_nhdr - (j - 2);
            const long at = (long)g_hdr[fr] + around[k];
            const long at2 = (long)g_hdr[fr] + 8 + (long)get32(g_base + g_hdr[fr] + 4) + around[k];
            if (at >= 0 && (size_t)at <= g_len) run(g_base, (size_t)at);
            if (at2 >= 0 && (size_t)at2 <= g_len) run(g_base, (size_t)at2);
        }
    for (i = 1; i <= 24 && i < g_len; i++) run(g_base, g_len - i);
    /* B. chunk lengths */
    for (j = 0; j < 5; j++) {
        const unsigned fr = j < 3 ? j : (j == 3 ? g_nhdr / 2 : g_nhdr - 1);
        const size_t vh = g_hdr[fr], ah = vh + 8 + get32(g_base + vh + 4);
        for (k = 0; k < sizeof lens / sizeof lens[0] + 3; k++) {
            const uint32_t left_v = (uint32_t)(g_len - vh - 8), left_a = (uint32_t)(g_len - ah - 8);
            memcpy(m, g_base, g_len);
            put32(m + vh + 4, k < 8 ? lens[k] : left_v + (k - 9));
            run(m, g_len);
            memcpy(m, g_base, g_len);
            put32(m + ah + 4, k < 8 ? lens[k] : left_a + (k - 9));
            run(m, g_len);
        }
    }
    /* C. the end marker: gone, damaged, early */
    memcpy(m, g_base, g_len);
    run(m, g_len - 8);
    memcpy(m + g_len - 8, "AMV_ENDX", 8);
    run(m, g_len);
    memcpy(m, g_base, g_len);
    memcpy(m + g_hdr[3], "AMV_END_", 8);
    run(m, g_len);
    /* D. stream tags swapped / unknown */
    for (j = 0; j < 3; j++) {
        const unsigned fr = j == 2 ? 5 : j;
        const size_t vh = g_hdr[fr], ah = vh + 8 + get32(g_base + vh + 4);
        memcpy(m, g_base, g_len); memcpy(m + vh, "01wb", 4); run(m, g_len);
        memcpy(m, g_base, g_len); memcpy(m + ah, "00dc", 4); run(m, g_len);
        memcpy(m, g_base, g_len); memcpy(m + vh, "01wb", 4); memcpy(m + ah, "00dc", 4); run(m, g_len);
        memcpy(m, g_base, g_len); memcpy(m + ah, "JUNK", 4); run(m, g_len);
    }
    /* E. every header byte forced */
    for (i = 0; i < g_data && i < 0x160; i++)
        for (k = 0; k < 3; k++) {
            memcpy(m, g_base, g_len);
            m[i] = force[k];
            run(m, g_len);
        }
    /* F. seeded random damage */
    for (i = 0; i < iters; i++) {
        size_t len = g_len;
        const unsigned kind = rnd() % 5u;
        memcpy(m, g_base, g_len);
        if (kind == 0) {                                            /* a few byte flips anywhere */
            for (k = 0; k < 1 + rnd() % 8u; k++) m[rnd() % g_len] ^= (uint8_t)(1u << (rnd() & 7u));
        } else if (kind == 1) {                                     /* a chunk length replaced by a random number */
            const size_t vh = g_hdr[rnd() % g_nhdr];
            const size_t at = (rnd() & 1u) ? vh + 4 : vh + 8 + get32(g_base + vh + 4) + 4;
            put32(m + at, (rnd() & 3u) ? rnd() % (uint32_t)(2 * g_len) : rnd() * 65536u + rnd());
        } else if (kind == 2) {                                     /* cut anywhere */
            len = rnd() % g_len;
        } else if (kind == 3) {                                     /* header byte flips + a cut */
            for (k = 0; k < 1 + rnd() % 4u; k++) m[rnd() % g_data] = (uint8_t)rnd();
            if (rnd() & 1u) len = g_data + rnd() % (g_len - g_data);
        } else {                                                    /* a block overwritten with noise */
            const size_t at = rnd() % g_len, cnt = 1 + rnd() % 64u;
            for (k = 0; k < cnt && at + k < g_len; k++) m[at + k] = (uint8_t)rnd();
        }
        run(m, len);
    }
    /* G. the muxer: ordinary use, empty chunks, odd sizes; what it writes goes back through the reader */
    for (i = 0; i < 24; i++) {
        const uint32_t w = (i & 1u) ? 16u + 2u * (rnd() % 200u) : 128u, h = (i & 2u) ? 16u + 2u * (rnd() % 150u) : 96u;
        mx = amvhip_mux_open(g_path, w, h, i == 5 ? 0u : 1u + rnd() % 60u, i == 7 ? 0u : 22050u, 200000u, 64000u);
        if (mx == NULL) continue;
        for (k = 0; k < 1 + rnd() % 40u; k++) {
            const uint32_t vl = (rnd() & 7u) ? 1u + rnd() % 3000u : 0u, al = (rnd() & 7u) ? 8u + rnd() % 800u : rnd() % 9u;
            uint8_t *v = (uint8_t *)malloc(vl ? vl : 1), *a = (uint8_t *)malloc(al ? al : 1);
            for (j = 0; j < vl; j++) v[j] = (uint8_t)rnd();
            for (j = 0; j < al; j++) a[j] = (uint8_t)rnd();
            if (al >= 8) put32(a + 4, (al - 8) * 2u);
            (void)amvhip_mux_write_frame(mx, vl ? v : NULL, vl, al ? a : NULL, al);
            free(v); free(a);
        }
        if (amvhip_mux_close(mx) != 0) fail("amvhip_mux_close", -1);
        walk(g_path, 1);
        n_mut++;
    }
    /* H. fat chunks where the windows change: an audio chunk (20 KB) or a video chunk (0.6 MB) that no window buffer is sized
     * for, as frame 0, in the middle, and as the first frame of every window for windows of 1 / 2 / 32 frames -- the reader
     * has to find room for them without taking anything away from what the caller still holds (the held pointers above) */
    for (i = 0; i < 12; i++) {
        const uint32_t period = (i % 3u == 0) ? 1u : (i % 3u == 1) ? 2u : 32u;
        mx = amvhip_mux_open(g_path, 128, 96, 16, 22050, 200000u, 64000u);
        if (mx == NULL) continue;
        for (k = 0; k < 70; k++) {
            const int fat = (k % period) == (i / 3u) % period || k == 37;
            const uint32_t vl = (fat && (i & 4u)) ? 600000u : 1u + rnd() % 3000u;
            const uint32_t al = (fat && !(i & 4u)) || (fat && (i & 8u)) ? 8u + 20000u + (rnd() & 3u) : 8u + rnd() % 800u;
            uint8_t *v = (uint8_t *)malloc(vl), *a = (uint8_t *)malloc(al);
            for (j = 0; j < vl; j++) v[j] = (uint8_t)rnd();
            for (j = 0; j < al; j++) a[j] = (uint8_t)rnd();
            put32(a + 4, (al - 8) * 2u);
            if (amvhip_mux_write_frame(mx, v, vl, a, al) != 0) fail("amvhip_mux_write_frame", -1);
            free(v); free(a);
        }
        if (amvhip_mux_close(mx) != 0) fail("amvhip_mux_close", -1);
        walk(g_path, 0);
        n_mut++;
    }
    if (amvhip_mux_open("/nonexistent-dir/x.amv", 128, 96, 16, 22050, 1, 1) != NULL) fail("mux_open into a missing directory", 0);
    if (amvhip_mux_close(NULL) != -1) fail("mux_close(NULL)", 0);
    if (amvhip_mux_write_frame(NULL, NULL, 0, NULL, 0) != -1) fail("mux_write_frame(NULL)", 0);
    if (AmvOpen(NULL) != NULL || AmvOpen("/nonexistent-dir/x.amv") != NULL) fail("AmvOpen of nothing", 0);
    if (AmvReadNextFrame(NULL) != -1 || AmvVideoDecode(NULL) != -1 || AmvAudioDecode(NULL) != -1 || AmvRewindFrameStart(NULL) != -1)
        fail("NULL decoder", 0);
    AmvClose(NULL);

    printf("%s mutations=%lu opened=%lu frames=%lu ends=%lu video_ok=%lu audio_ok=%lu violations=%lu\n", n_bad ? "FAILED" : "ok", n_mut,
           n_open, n_frames, n_end, n_vok, n_aok, n_bad);
    free(g_base);
    free(m);
    return n_bad ? 1 : 0;
}
