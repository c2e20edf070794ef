/*
 * amvlib_host.c -- a C host of the amvlib call surface, in the shape of the reference's own test program
 * (C-AMVDecoder/AmvLibTest/AmvLibTest.cpp:11-85: AmvOpen, AmvCreateWavFileFromAmvFile, the header print-out,
 * the AmvReadNextFrame loop, AmvClose) plus the two decode calls its player makes per frame
 * (AMVDecoderDlg.cpp FillBuffer).  Built by tests/test_gpu_parity.py with plain gcc against include/amvhip.h and
 * linked with libamvhip.so: what a maintainer's program does when it swaps amvlib for this library.
 *
 *     amvlib_host <file.amv> <out.wav>
 *
 * prints the header fields, per-stream totals and the chained FNV-1a-64 of every decoded BGR frame (seeded as the
 * survey's harness was, so the figure can be compared with the one amvlib itself produced).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "amvhip.h"

int main(int argc, char **argv)
{
    AMVDecoder *amvdec;
    AMVInfo *amvinfo;
    FRAMEBUFF *fbuff;
    uint64_t hash = 1469598103934665603ull;
    unsigned long video_bytes = 0, audio_bytes = 0, pcm_bytes = 0, frames = 0;
    unsigned int i;

    if (argc < 3) {
        printf("usage: amvlib_host file.amv out.wav\n");
        return 2;
    }
    amvdec = AmvOpen(argv[1]);
    if (amvdec == NULL) return 1;
    amvinfo = &amvdec->amvinfo;
    if (AmvCreateWavFileFromAmvFile(amvdec, AUDIO_FILE_TYPE_ADPCM_IMA, argv[2]) != 0) return 1;

    printf("frame interval: %u us\n", amvinfo->dwMicroSecPerFrame);
    printf("size: %u x %u\n", amvinfo->dwWidth, amvinfo->dwHeight);
    printf("speed: %u frames/s\n", amvinfo->dwSpeed);
    printf("duration: %u h %u min %u s\n", amvinfo->dwTimeHour, amvinfo->dwTimeMin, amvinfo->dwTimeSec);
    printf("total frames: %u\n", amvdec->totalframe);
    printf("audio: %u ch, %u Hz, %u bits, %u bytes/s\n", amvinfo->nChannels, amvinfo->nSamplesPerSec,
           amvinfo->wBitsPerSample, amvinfo->nAvgBytesPerSec);

    for (;;) {
        if (AmvReadNextFrame(amvdec) != 0) break;
        fbuff = &amvdec->framebuf;
        if (fbuff->framenum == -1) break;
        video_bytes += fbuff->videobufflen;
        audio_bytes += fbuff->audiobufflen;
        if (AmvVideoDecode(amvdec) != 0) { printf("video decode failed at frame %d\n", fbuff->framenum); return 1; }
        for (i = 0; i < amvdec->videobuf.len; ++i) {
            hash ^= amvdec->videobuf.fbmpdat[i];
            hash *= 1099511628211ull;
        }
        if (AmvAudioDecode(amvdec) != 0) { printf("audio decode failed at frame %d\n", fbuff->framenum); return 1; }
        pcm_bytes += 4ul * (fbuff->audiobufflen - 8);     /* the defined samples of the chunk: 2 per byte */
        ++frames;
    }
    printf("decoded frames: %lu\n", frames);
    printf("video chunk bytes: %lu\n", video_bytes);
    printf("audio chunk bytes: %lu\n", audio_bytes);
    printf("pcm bytes: %lu\n", pcm_bytes);
    printf("video fnv1a64: %016llx\n", (unsigned long long)hash);
    AmvClose(amvdec);
    return 0;
}
