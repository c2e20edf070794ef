/*
 * sanitize_main.c -- drives the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer.  TEST INFRASTRUCTURE.
 *
 * the reference has real out-of-bounds accesses on this path (the 4-byte ADPCM loop reads past
 * the chunk, AdpcmIma.c:225-237; iclp[] is indexed unguarded, AmvJpeg.c:1167; HufBlock runs past 64 on corrupt data,
 * :967-969).  The restatement defines those cases; this program proves it does so without touching memory it does
 * not own or relying on undefined arithmetic: every video and audio chunk of an AMV file, the same chunks truncated
 * and damaged, random bytes, the encoders at awkward geometries.  Built by `make -C oracle sanitize`; run by
 * tests/test_oracle_pin.py::test_oracle_under_sanitizers.  Exit code 0 = nothing reported.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "amv_oracle.h"

static uint32_t rnd_state = 12345u;
static uint32_t rnd(void) { return rnd_state = rnd_state * 1664525u + 1013904223u; }
static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* every call gets buffers of exactly the documented size, so that an overrun of one byte is seen */
static void decode_all_ways(const uint8_t *chunk, uint32_t len, uint32_t w, uint32_t h)
{
    uint8_t *in = malloc(len ? len : 1);
    uint8_t *out = malloc((size_t)amvo_stride(w) * h);
    uint8_t *yuv = malloc(amvo_yuv420_frame_bytes(w, h));
    int16_t *coef = malloc((size_t)amvo_mcus_per_row(w) * amvo_mcu_rows(h) * 6 * 64 * sizeof(int16_t));
    uint32_t ok, st;
    memcpy(in, chunk, len);
    amvo_decode_frame(in, len, w, h, 0, out, coef, &ok, &st);
    amvo_decode_frame(in, len, w, h, AMVO_FLAG_ZIGZAG_FIXED, out, NULL, NULL, NULL);
    amvo_decode_frame_ffmpeg(in, len, w, h, yuv, &ok, &st);
    free(in); free(out); free(yuv); free(coef);
}

static void adpcm_all_ways(const uint8_t *chunk, uint32_t len)
{
    uint8_t *in = malloc(len ? len : 1);
    int16_t *pcm = malloc(len > 8 ? (size_t)(len - 8) * 2 * sizeof(int16_t) : 2);
    uint32_t hdr;
    memcpy(in, chunk, len);
    amvo_adpcm_decode_chunk(in, len, pcm, &hdr);
    free(in); free(pcm);
}

int main(int argc, char **argv)
{
    FILE *fp;
    uint8_t *file;
    long size, p;
    uint32_t w, h, nv = 0, na = 0, i;
    if (argc != 2 || !(fp = fopen(argv[1], "rb"))) { fprintf(stderr, "usage: sanitize_main file.amv\n"); return 2; }
    fseek(fp, 0, SEEK_END); size = ftell(fp); fseek(fp, 0, SEEK_SET);
    file = malloc((size_t)size);
    if (fread(file, 1, (size_t)size, fp) != (size_t)size) return 2;
    fclose(fp);
    w = le32(file + 64); h = le32(file + 68);
    for (p = 316; p + 8 <= size && memcmp(file + p, "AMV_", 4) != 0;) {
        const uint32_t n = le32(file + p + 4);
        const uint8_t *c = file + p + 8;
        if (p + 8 + (long)n > size) break;
        if (!memcmp(file + p, "00dc", 4)) {
            ++nv;
            decode_all_ways(c, n, w, h);
            if (nv % 16 == 1) {   /* truncated at every kind of place, damaged, at a geometry the stream was not made for */
                uint8_t *d = malloc(n);
                uint32_t cut;
                for (cut = 0; cut < 12; ++cut) decode_all_ways(c, cut, w, h);
                decode_all_ways(c, n / 2, w, h);
                decode_all_ways(c, n - 1, w, h);
                for (i = 0; i < 6; ++i) {
                    memcpy(d, c, n);
                    d[2 + rnd() % (n - 2)] ^= (uint8_t)(1u << (rnd() % 8));
                    d[2 + rnd() % (n - 2)] = 0xff;
                    decode_all_ways(d, n, w, h);
                }
                decode_all_ways(c, n, w + 13, h + 7);
                decode_all_ways(c, n, 16, 16);
                decode_all_ways(c, n, 1, 1);
                free(d);
            }
        } else {
            ++na;
            adpcm_all_ways(c, n);
            if (na % 32 == 1) {
                uint8_t hdr[12] = {0xff, 0x7f, 200, 0, 0, 0, 0, 0, 0x77, 0x88, 0xff, 0x00};   /* step index past 88 */
                for (i = 0; i <= 12; ++i) adpcm_all_ways(hdr, i);
                adpcm_all_ways(c, n / 2);
            }
        }
        p += 8 + (long)n;
    }
    {   /* random bytes behind an SOI: nothing a decoder can assume holds */
        uint8_t junk[4096];
        uint32_t t;
        for (t = 0; t < 40; ++t) {
            for (i = 0; i < sizeof junk; ++i) junk[i] = (uint8_t)(rnd() >> 24);
            junk[0] = 0xff; junk[1] = 0xd8;
            decode_all_ways(junk, 64 + rnd() % (sizeof junk - 64), 160, 120);
        }
        memset(junk, 0xff, sizeof junk);
        decode_all_ways(junk, sizeof junk, 48, 32);
    }
    {   /* encoders: even geometries incl. partial MCUs, extreme content; round trip through the decoder */
        static const uint32_t geo[][2] = {{160, 120}, {16, 16}, {18, 2}, {2, 34}, {130, 98}, {320, 240}};
        uint32_t g, t;
        for (g = 0; g < sizeof geo / sizeof geo[0]; ++g) {
            const uint32_t gw = geo[g][0], gh = geo[g][1];
            uint8_t *rgb = malloc((size_t)gw * gh * 3), *chunk = malloc(amvo_encode_bound(gw, gh));
            int16_t *coef = malloc((size_t)amvo_mcus_per_row(gw) * amvo_mcu_rows(gh) * 6 * 64 * sizeof(int16_t));
            for (t = 0; t < 4; ++t) {
                int n;
                if (t == 0) amvo_synth_frame(0xA11CE, g, gw, gh, rgb);
                else for (i = 0; i < gw * gh * 3; ++i) rgb[i] = t == 1 ? (uint8_t)(rnd() >> 24) : (t == 2 ? 255 : (uint8_t)(((i / 3) & 1) * 255));
                n = amvo_encode_frame(rgb, gw * 3, gw, gh, (int)(t & 1), t & 2 ? 128 : 0, chunk, coef);
                if (n <= 4) { fprintf(stderr, "encode failed\n"); return 1; }
                decode_all_ways(chunk, (uint32_t)n, gw, gh);
            }
            free(rgb); free(chunk); free(coef);
        }
    }
    {   /* audio encoders */
        int16_t pcm[4097];
        uint8_t out[8 + 2048 + 16];
        int32_t st[2] = {0, 0};
        int idx = 0;
        uint32_t extra = 0, t;
        uint64_t written = 0;
        amvo_synth_audio(0xA11CE, 0, 4097, pcm);
        for (t = 0; t < 50; ++t) {
            const uint32_t pairs = amvo_adpcm_amv_pairs(1471, 22050, &extra, &written);
            if (2 * pairs > 4096) return 1;
            for (i = 0; i < 2 * pairs; ++i) pcm[i] = (int16_t)(t % 3 == 0 ? (int)(rnd() >> 16) - 32768 : pcm[i]);
            amvo_adpcm_encode_chunk(pcm, 2 * pairs, &idx, out);
            adpcm_all_ways(out, 8 + pairs);
        }
        amvo_adpcm_wav_encode_frame(pcm, 1016, st, out);
    }
    printf("sanitized: %u video chunks, %u audio chunks\n", nv, na);
    free(file);
    return 0;
}
