/*
 * amvlib_host.c -- a plain C host of the amvlib call surface.
 *
 * It makes the calls the reference's console test makes (C-AMVDecoder/AmvLibTest/AmvLibTest.cpp:11-85: open,
 * export the audio as an ADPCM WAV, print the header, walk the frames, close) and, per frame, the two decode
 * calls of its player (AMVDecoderDlg.cpp FillBuffer).  tests/test_gpu_parity.py builds it with gcc against
 * include/amvhip.h and links it with libamvhip.so: what a maintainer's program does when it swaps amvlib for
 * this library.
 *
 *     amvlib_host <file.amv> <out.wav>
 *     amvlib_host <file.amv> --bench <passes>     the same read / video / audio loop, timed, nothing hashed
 *                                                 (bench.py --workload amvlib)
 *
 * Output: "key: value" lines -- header fields, per-stream totals, and the chained FNV-1a-64 of every decoded
 * BGR frame, seeded as the survey's harness was so that it can be compared with the figure amvlib produced.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "amvhip.h"

struct totals {
    unsigned long frames, video_in, audio_in, pcm_out;
    uint64_t fnv;
};

static void fold(struct totals *t, const unsigned char *p, unsigned int n)
{
    unsigned int i;
    for (i = 0; i < n; ++i) t->fnv = (t->fnv ^ p[i]) * 1099511628211ull;
}

static int play(AMVDecoder *d, struct totals *t)
{
    while (AmvReadNextFrame(d) == 0 && d->framebuf.framenum != -1) {
        t->video_in += d->framebuf.videobufflen;
        t->audio_in += d->framebuf.audiobufflen;
        if (AmvVideoDecode(d) != 0 || AmvAudioDecode(d) != 0) {
            printf("error: decode failed at frame %d\n", d->framebuf.framenum);
            return -1;
        }
        fold(t, d->videobuf.fbmpdat, d->videobuf.len);
        t->pcm_out += 4ul * (d->framebuf.audiobufflen - 8);   /* the chunk's defined samples: two per byte */
        t->frames++;
    }
    return 0;
}

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* passes over the file through AmvReadNextFrame / AmvVideoDecode / AmvAudioDecode; one untimed pass first */
static int bench(const char *path, int passes)
{
    AMVDecoder *d = AmvOpen(path);
    unsigned long frames = 0, sink = 0;
    double t0 = 0;
    int p;
    if (d == NULL) return 1;
    for (p = -1; p < passes; p++) {
        if (p == 0) { t0 = now(); frames = 0; }
        if (AmvRewindFrameStart(d) != 0) return 1;
        d->framebuf.framenum = 0;
        while (AmvReadNextFrame(d) == 0 && d->framebuf.framenum != -1) {
            if (AmvVideoDecode(d) != 0 || AmvAudioDecode(d) != 0) return 1;
            sink += d->videobuf.fbmpdat[d->videobuf.len / 2] + (unsigned long)d->audiobuf.audiodata[0];
            frames++;
        }
    }
    printf("bench frames: %lu\nbench seconds: %.6f\nbench sink: %lu\n", frames, now() - t0, sink);
    AmvClose(d);
    return 0;
}

int main(int argc, char **argv)
{
    struct totals t = {0, 0, 0, 0, 1469598103934665603ull};
    const AMVInfo *hdr;
    AMVDecoder *d;

    if (argc < 3) {
        printf("usage: amvlib_host file.amv out.wav\n");
        return 2;
    }
    if (argc >= 4 && strcmp(argv[2], "--bench") == 0) return bench(argv[1], atoi(argv[3]));
    d = AmvOpen(argv[1]);
    if (d == NULL) return 1;
    if (AmvCreateWavFileFromAmvFile(d, AUDIO_FILE_TYPE_ADPCM_IMA, argv[2]) != 0) return 1;
    hdr = &d->amvinfo;
    printf("frame interval: %u us\n", hdr->dwMicroSecPerFrame);
    printf("size: %u x %u\n", hdr->dwWidth, hdr->dwHeight);
    printf("speed: %u frames/s\n", hdr->dwSpeed);
    printf("duration: %u h %u min %u s\n", hdr->dwTimeHour, hdr->dwTimeMin, hdr->dwTimeSec);
    printf("total frames: %u\n", d->totalframe);
    printf("audio: %u ch, %u Hz, %u bits, %u bytes/s\n", hdr->nChannels, hdr->nSamplesPerSec, hdr->wBitsPerSample,
           hdr->nAvgBytesPerSec);
    if (play(d, &t) != 0) return 1;
    printf("decoded frames: %lu\n", t.frames);
    printf("video chunk bytes: %lu\n", t.video_in);
    printf("audio chunk bytes: %lu\n", t.audio_in);
    printf("pcm bytes: %lu\n", t.pcm_out);
    printf("video fnv1a64: %016llx\n", (unsigned long long)t.fnv);
    AmvClose(d);
    return 0;
}
