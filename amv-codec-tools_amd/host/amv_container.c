/*
 * amv_container.c -- AMV container writer in host C: the muxer half of the path
 * (SURVEY.md section 8(f), row 1).  Pure host code, no device work.
 *
 * Follows the reference muxer AMVmuxer/ffmpeg/libavformat/amvenc.c field by field (cited
 * below as amvenc.c:line) with the RIFF helpers of riff.c:221-236 (start_tag/end_tag) and the
 * WAVEFORMAT writer riff.c:240-330 (put_wav_header: 16 bytes for adpcm_ima_amv, which has no
 * extradata).  The result has the fixed 304-byte header the amvlib reader checks four-cc by
 * four-cc (AMVDec.c:50-93), "movi" at 0x138 (what compare_amv.c:29-44 tests), unpadded
 * 00dc / 01wb chunks in strict video/audio alternation and the "AMV_END_" trailer.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/amvhip.h"

/* amv.h:27-32 */
#define AMVF_HASINDEX 0x00000010u
#define AMVF_ISINTERLEAVED 0x00000100u
#define AMVF_TRUSTCKTYPE 0x00000800u

struct amvhip_muxer {
    FILE *fp;
    uint32_t fps;
    long riff_start, movi_list;              /* positions just after a tag's size field (riff.c:221-226) */
    long frames_hdr_all, frames_hdr_strm[2]; /* amvenc.c:35: counters patched at close */
    long seconds, minutes, hours;            /* amvenc.c:36 */
    uint32_t packet_count[2];
    int failed;
};

static void put8(amvhip_muxer *m, unsigned v) { if (fputc((int)(v & 0xff), m->fp) == EOF) m->failed = 1; }
static void put16(amvhip_muxer *m, unsigned v) { put8(m, v); put8(m, v >> 8); }
static void put32(amvhip_muxer *m, uint32_t v) { put16(m, v); put16(m, v >> 16); }
static void put_tag(amvhip_muxer *m, const char *t) { if (fwrite(t, 1, 4, m->fp) != 4) m->failed = 1; }

static long start_tag(amvhip_muxer *m, const char *t)   /* riff.c:221-226 */
{
    put_tag(m, t);
    put32(m, 0);
    return ftell(m->fp);
}

static void patch32(amvhip_muxer *m, long at, uint32_t v)
{
    long pos = ftell(m->fp);
    if (fseek(m->fp, at, SEEK_SET) != 0) { m->failed = 1; return; }
    put32(m, v);
    if (fseek(m->fp, pos, SEEK_SET) != 0) m->failed = 1;
}

static void end_tag(amvhip_muxer *m, long start)        /* riff.c:228-236 */
{
    patch32(m, start - 4, (uint32_t)(ftell(m->fp) - start));
}

amvhip_muxer *amvhip_mux_open(const char *path, uint32_t width, uint32_t height, uint32_t fps,
                              uint32_t sample_rate, uint32_t video_bit_rate, uint32_t audio_bit_rate)
{
    amvhip_muxer *m;
    long list1, list2, strh, strf;
    int i;

    if (path == NULL || width == 0 || height == 0 || fps == 0 || sample_rate == 0) return NULL;
    m = (amvhip_muxer *)calloc(1, sizeof *m);
    if (m == NULL) return NULL;
    m->fp = fopen(path, "wb+");
    if (m->fp == NULL) { free(m); return NULL; }
    m->fps = fps;

    /* avi_start_new_riff, amvenc.c:41-54 */
    m->riff_start = start_tag(m, "RIFF");
    put_tag(m, "AMV ");
    list1 = start_tag(m, "LIST");
    put_tag(m, "hdrl");

    /* main header, amvenc.c:128-177: 14 dwords */
    put_tag(m, "amvh");
    put32(m, 14 * 4);
    put32(m, (uint32_t)(1000000ull * 1u / fps));          /* :146 microseconds per frame, time base 1/fps */
    put32(m, (video_bit_rate + audio_bit_rate) / 8);      /* :150 "not quite exact" */
    put32(m, 0);                                          /* :151 padding */
    put32(m, AMVF_TRUSTCKTYPE | AMVF_HASINDEX | AMVF_ISINTERLEAVED);   /* :155 seekable output */
    m->frames_hdr_all = ftell(m->fp);                     /* :156 */
    put32(m, 0);                                          /* frames, patched at close */
    put32(m, 0);                                          /* :158 initial frame */
    put32(m, 2);                                          /* :159 streams */
    put32(m, 1024 * 1024);                                /* :160 suggested buffer size */
    put32(m, width);                                      /* :162-163 */
    put32(m, height);
    put32(m, fps);                                        /* :169 frame rate where AVI has a reserved word */
    put32(m, 1);                                          /* :170 "always 1 in a real AMV" */
    put32(m, 0);                                          /* :171 */
    m->seconds = ftell(m->fp);                            /* :173-178 duration, patched at close */
    put8(m, 0);
    m->minutes = ftell(m->fp);
    put8(m, 0);
    m->hours = ftell(m->fp);
    put16(m, 0);

    for (i = 0; i < 2; ++i) {                             /* stream lists, amvenc.c:181-262: 0 video, 1 audio */
        list2 = start_tag(m, "LIST");
        put_tag(m, "strl");
        strh = start_tag(m, "strh");
        put_tag(m, i == 0 ? "vids" : "auds");             /* :189-192 */
        put32(m, i == 0 ? 0u : 1u);                       /* :193-196 codec tag (riff.c has none for AMV video) / 1 */
        put32(m, 0);                                      /* flags */
        put16(m, 0);                                      /* priority */
        put16(m, 0);                                      /* language */
        put32(m, 0);                                      /* initial frame */
        put32(m, 1);                                      /* :210 scale: both streams run on the video time base */
        put32(m, fps);                                    /* :211 rate */
        put32(m, 0);                                      /* :214 start */
        m->frames_hdr_strm[i] = ftell(m->fp);             /* :215 */
        put32(m, 0);                                      /* length, patched at close */
        if (i == 0) {                                     /* :222-224 video only: buffer size, quality */
            put32(m, 1024 * 1024);
            put32(m, 0xffffffffu);
        }
        put32(m, i == 0 ? 0u : 2u);                       /* :229 sample size (audio: forced to 2, :203) */
        put32(m, 0);
        put16(m, i == 0 ? width : 0u);                    /* :231-232 */
        put16(m, i == 0 ? height : 0u);
        end_tag(m, strh);

        strf = start_tag(m, "strf");
        if (i == 0) {
            int k;
            for (k = 0; k < 9; ++k) put32(m, 0);          /* :238-246 */
        } else {                                          /* put_wav_header, riff.c:248-289, then :253 */
            put16(m, 0x01);                               /* riff.c:191 tag of adpcm_ima_amv */
            put16(m, 1);                                  /* mono (adpcm.c:191) */
            put32(m, sample_rate);
            put32(m, audio_bit_rate / 8);                 /* riff.c:285 */
            put16(m, 2);                                  /* riff.c:278 channels * 16 >> 3 */
            put16(m, 16);                                 /* riff.c:264 */
            put32(m, 0);
        }
        end_tag(m, strf);
        end_tag(m, list2);
    }
    end_tag(m, list1);

    m->movi_list = start_tag(m, "LIST");                  /* :266-267 */
    put_tag(m, "movi");
    if (m->failed) { fclose(m->fp); free(m); return NULL; }
    return m;
}

/* avi_write_packet, amvenc.c:285-321: tag, le32 size, payload, no padding byte */
static void write_chunk(amvhip_muxer *m, int stream, const uint8_t *data, uint32_t len)
{
    m->packet_count[stream]++;
    put_tag(m, stream == 0 ? "00dc" : "01wb");            /* avi_stream2fourcc :56-70 */
    put32(m, len);
    if (len && fwrite(data, 1, len, m->fp) != len) m->failed = 1;
}

/* One video chunk and its audio chunk: the order amv_interleave_packet (amvenc.c:378-406) enforces. */
int amvhip_mux_write_frame(amvhip_muxer *m, const uint8_t *video, uint32_t video_len, const uint8_t *audio,
                           uint32_t audio_len)
{
    if (m == NULL || m->fp == NULL || (video_len && !video) || (audio_len && !audio)) return -1;
    write_chunk(m, 0, video, video_len);
    write_chunk(m, 1, audio, audio_len);
    return m->failed ? -1 : 0;
}

/* avi_write_trailer + avi_write_counters, amvenc.c:323-338 and 72-114 */
int amvhip_mux_close(amvhip_muxer *m)
{
    int rc;
    uint32_t nb_frames, duration;
    if (m == NULL) return -1;
    end_tag(m, m->movi_list);
    if (fwrite("AMV_END_", 1, 8, m->fp) != 8) m->failed = 1;          /* :332 */
    end_tag(m, m->riff_start);
    nb_frames = m->packet_count[0];
    patch32(m, m->frames_hdr_strm[0], m->packet_count[0]);            /* :88 sample size 0: packet count */
    patch32(m, m->frames_hdr_strm[1], m->packet_count[1]);            /* block_align of adpcm_ima_amv is 0 */
    patch32(m, m->frames_hdr_all, nb_frames);                         /* :96-98 */
    duration = nb_frames / m->fps;                                    /* :101 */
    if (fseek(m->fp, m->seconds, SEEK_SET) == 0) put8(m, duration % 60);   /* :103-104 */
    if (fseek(m->fp, m->minutes, SEEK_SET) == 0) put8(m, duration / 60);   /* :106-107 (not reduced mod 60 there either) */
    if (fseek(m->fp, m->hours, SEEK_SET) == 0) put16(m, duration / 3600);  /* :109-110 */
    rc = (m->failed || fclose(m->fp) != 0) ? -1 : 0;
    free(m);
    return rc;
}

/* ---------------------------------------------------------------------------------------------
 * AMV audio framing of the reference's encoder and muxer, in host arithmetic (no device work)
 * ------------------------------------------------------------------------------------------- */

/* adpcm.c:469-477,497: sample PAIRS of the next chunk for a nominal frame_size -- an odd frame_size puts its extra
 * sample into every second chunk, and a chunk that would straddle a whole second of audio is stretched to end on it.
 * extra / samples_written are the caller's stream state (c->extra_amv_samples, c->samples_written). */
uint32_t amvhip_amv_audio_pairs(uint32_t frame_size, uint32_t sample_rate, uint32_t *extra, uint64_t *samples_written)
{
    uint32_t n = frame_size >> 1;                           /* :469 */
    *extra += frame_size & 1u;                              /* :470 */
    n += *extra >> 1;                                       /* :471 */
    *extra &= 1u;                                           /* :472 */
    if (sample_rate) {
        const uint32_t i = (uint32_t)((*samples_written + 2ull * n) % sample_rate);   /* :474 */
        if (i && i + frame_size > sample_rate) n += (sample_rate - i) >> 1;           /* :476-477 */
    }
    *samples_written += 2ull * n;                           /* :497 */
    return n;
}

/* frame_size the AMV muxer imposes on the audio encoder: sample_rate * time_base (amvenc.c:276-281, av_rescale
 * rounds to nearest) */
uint32_t amvhip_amv_audio_frame_size(uint32_t sample_rate, uint32_t tb_num, uint32_t tb_den)
{
    if (!tb_den) return 0;
    return (uint32_t)(((uint64_t)sample_rate * tb_num + tb_den / 2) / tb_den);
}
