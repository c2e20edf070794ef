/*
 * lavc_host.c -- a C host that drives the four AMV codec tables of libamvhip_lavc.so THROUGH `struct AVCodec`,
 * the way libavcodec/utils.c does (avcodec_open :839-882 allocates priv_data_size bytes and calls init;
 * avcodec_decode_video :926-947, avcodec_encode_video :900-924, avcodec_encode_audio :886-898,
 * avcodec_decode_audio2 :949-983 call the table's decode / encode).  Compiled by amv-codec-tools_amd/build.py
 * against the reference's own avcodec.h where it lies, like the plugin itself.
 *
 *     lavc_host <file.amv> <outdir>
 *
 * decode leg: walks the AMV file with the library's container reader and writes
 *     dec_video.yuv   every frame's three planes, rows tight      dec_audio.pcm   every chunk's samples
 * encode leg: a deterministic 160x120 YUVJ420P clip in padded buffers and a deterministic PCM track
 *     enc_src.yuv     the clip, rows tight                        enc_video.bin   per frame: le32 length + chunk
 *     enc_pcm.raw     the PCM track                               enc_audio.bin   per chunk: le32 length + chunk
 *     (audio twice over: frame_size 1378 = 22050/16 as the muxer sets it, then an odd frame_size, 1471)
 * tests/test_gpu_parity.py compares every file with what the batch ABI and the oracle give for the same input.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "avcodec.h"

#include "amvhip.h"

extern AVCodec amv_decoder, amv_encoder, adpcm_ima_amv_decoder, adpcm_ima_amv_encoder;

static int n_get, n_release;

/* a caller-side get_buffer (the field is the user's to set, avcodec.h:1096): padded rows, 16-byte edges */
static int my_get_buffer(AVCodecContext *c, AVFrame *pic)
{
    const int w[3] = {c->width, (c->width + 1) / 2, (c->width + 1) / 2};
    const int h[3] = {c->height, (c->height + 1) / 2, (c->height + 1) / 2};
    int i;
    for (i = 0; i < 3; i++) {
        pic->linesize[i] = (w[i] + 32 + 15) & ~15;
        pic->base[i] = malloc((size_t)pic->linesize[i] * (h[i] + 32));
        if (!pic->base[i]) return -1;
        memset(pic->base[i], 0xAB, (size_t)pic->linesize[i] * (h[i] + 32));
        pic->data[i] = pic->base[i] + 16 * pic->linesize[i] + 16;
    }
    n_get++;
    return 0;
}

static void my_release_buffer(AVCodecContext *c, AVFrame *pic)
{
    int i;
    (void)c;
    for (i = 0; i < 3; i++) {
        free(pic->base[i]);
        pic->base[i] = pic->data[i] = NULL;
    }
    n_release++;
}

static AVCodecContext *open_codec(AVCodec *codec)   /* avcodec_alloc_context + avcodec_open, the parts that matter */
{
    AVCodecContext *c = calloc(1, sizeof *c);
    if (!c) return NULL;
    c->codec = codec;
    c->codec_id = codec->id;
    c->codec_type = codec->type;
    c->priv_data = calloc(1, codec->priv_data_size);
    c->get_buffer = my_get_buffer;
    c->release_buffer = my_release_buffer;
    return c;
}

static void close_codec(AVCodecContext *c)
{
    if (c->codec->close) c->codec->close(c);
    free(c->priv_data);
    free(c);
}

static FILE *out_file(const char *dir, const char *name)
{
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    return fopen(path, "wb");
}

static void put_le32(FILE *f, uint32_t v)
{
    uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
    fwrite(b, 1, 4, f);
}

static int decode_leg(const char *amv, const char *dir)
{
    AMVDecoder *d = AmvOpen(amv);
    AVCodecContext *vc = open_codec(&amv_decoder), *ac = open_codec(&adpcm_ima_amv_decoder);
    FILE *fv = out_file(dir, "dec_video.yuv"), *fa = out_file(dir, "dec_audio.pcm");
    int16_t *pcm = malloc(192000);
    int frames = 0, y, i;

    if (!d || !vc || !ac || !fv || !fa || !pcm) return -1;
    vc->width = vc->coded_width = (int)d->amvinfo.dwWidth;
    vc->height = vc->coded_height = (int)d->amvinfo.dwHeight;
    ac->channels = 1;
    ac->sample_rate = (int)d->amvinfo.nSamplesPerSec;
    if (amv_decoder.init(vc) < 0 || adpcm_ima_amv_decoder.init(ac) < 0) return -2;
    while (AmvReadNextFrame(d) == 0 && d->framebuf.framenum != -1) {
        AVFrame pic;
        int got = 0, bytes = 0, r;
        r = vc->codec->decode(vc, &pic, &got, d->framebuf.videobuff, (int)d->framebuf.videobufflen);
        if (r != (int)d->framebuf.videobufflen || got != (int)sizeof(AVFrame)) return -3;
        for (i = 0; i < 3; i++) {
            const int w = i ? (vc->width + 1) / 2 : vc->width, h = i ? (vc->height + 1) / 2 : vc->height;
            for (y = 0; y < h; y++) fwrite(pic.data[i] + y * pic.linesize[i], 1, w, fv);
        }
        r = ac->codec->decode(ac, pcm, &bytes, d->framebuf.audiobuff, (int)d->framebuf.audiobufflen);
        if (r != (int)d->framebuf.audiobufflen) return -4;
        fwrite(pcm, 1, bytes, fa);
        frames++;
    }
    printf("decoded frames: %d\n", frames);
    printf("pix_fmt is yuvj420p: %d\n", vc->pix_fmt == PIX_FMT_YUVJ420P);
    close_codec(vc);
    close_codec(ac);
    printf("get_buffer calls: %d\nrelease_buffer calls: %d\n", n_get, n_release);
    AmvClose(d);
    fclose(fv);
    fclose(fa);
    free(pcm);
    return 0;
}

static uint32_t lcg(uint32_t *s) { return *s = *s * 1664525u + 1013904223u; }

static int encode_video_leg(const char *dir)
{
    enum { W = 160, H = 120, N = 24, LY = 192, LC = 112 };
    AVCodecContext *c = open_codec(&amv_encoder);
    FILE *fs = out_file(dir, "enc_src.yuv"), *fo = out_file(dir, "enc_video.bin");
    uint8_t *planes[3], *buf = malloc(1 << 20);
    uint32_t seed = 12345;
    int t, x, y, i;

    if (!c || !fs || !fo || !buf) return -1;
    c->width = W;
    c->height = H;
    c->pix_fmt = PIX_FMT_YUVJ420P;
    c->time_base.num = 1;
    c->time_base.den = 16;
    /* the reference refuses EMU_EDGE (mjpegenc.c:462-464); so does the plugin */
    if (amv_encoder.init(c) < 0) return -2;
    c->flags |= CODEC_FLAG_EMU_EDGE;
    {
        AVFrame dummy;
        memset(&dummy, 0, sizeof dummy);
        printf("emu_edge rejected: %d\n", c->codec->encode(c, buf, 1 << 20, &dummy) == -1);
    }
    c->flags &= ~CODEC_FLAG_EMU_EDGE;
    planes[0] = malloc(LY * H);
    planes[1] = malloc(LC * H / 2);
    planes[2] = malloc(LC * H / 2);
    for (t = 0; t < N; t++) {
        AVFrame pic;
        int len;
        memset(&pic, 0, sizeof pic);
        memset(planes[0], 0xEE, LY * H);                      /* the padding must never be read */
        memset(planes[1], 0xEE, LC * H / 2);
        memset(planes[2], 0xEE, LC * H / 2);
        for (y = 0; y < H; y++)
            for (x = 0; x < W; x++)
                planes[0][y * LY + x] = (uint8_t)(128 + ((x * 3 + t * 5) & 63) - ((y * 2 + t) & 31) + (int)(lcg(&seed) >> 28));
        for (y = 0; y < H / 2; y++)
            for (x = 0; x < W / 2; x++) {
                planes[1][y * LC + x] = (uint8_t)(100 + ((x + y + t) & 63));
                planes[2][y * LC + x] = (uint8_t)(160 - ((x * 2 - y + 3 * t) & 63));
            }
        for (i = 0; i < 3; i++) {
            const int w = i ? W / 2 : W, h = i ? H / 2 : H, ls = i ? LC : LY;
            pic.data[i] = planes[i];
            pic.linesize[i] = ls;
            for (y = 0; y < h; y++) fwrite(planes[i] + y * ls, 1, w, fs);
        }
        len = c->codec->encode(c, buf, 1 << 20, &pic);
        if (len <= 4) return -3;
        put_le32(fo, (uint32_t)len);
        fwrite(buf, 1, len, fo);
    }
    printf("encoded frames: %d\ncoded_frame key: %d\n", N, c->coded_frame && c->coded_frame->key_frame);
    close_codec(c);
    fclose(fs);
    fclose(fo);
    {   /* the other pixel format of pix_fmts (mjpegenc.c:493): YUVJ422P, chroma planes W/2 x H */
        enum { N2 = 6 };
        AVCodecContext *c2 = open_codec(&amv_encoder);
        FILE *fs2 = out_file(dir, "enc_src422.yuv"), *fo2 = out_file(dir, "enc_video422.bin");
        uint8_t *cb = malloc(LC * H), *cr = malloc(LC * H);
        if (!c2 || !fs2 || !fo2 || !cb || !cr) return -4;
        c2->width = W;
        c2->height = H;
        c2->pix_fmt = PIX_FMT_YUVJ422P;
        c2->time_base.num = 1;
        c2->time_base.den = 16;
        printf("pix_fmts: %d %d %d\n", amv_encoder.pix_fmts[0] == PIX_FMT_YUVJ420P, amv_encoder.pix_fmts[1] == PIX_FMT_YUVJ422P,
               amv_encoder.pix_fmts[2] == -1);
        if (amv_encoder.init(c2) < 0) return -5;
        for (t = 0; t < N2; t++) {
            AVFrame pic;
            int len;
            memset(&pic, 0, sizeof pic);
            memset(cb, 0xEE, LC * H);
            memset(cr, 0xEE, LC * H);
            for (y = 0; y < H; y++)
                for (x = 0; x < W; x++)
                    planes[0][y * LY + x] = (uint8_t)(128 + ((x * 5 + t * 3) & 63) - ((y * 3 + t) & 31) + (int)(lcg(&seed) >> 28));
            for (y = 0; y < H; y++)
                for (x = 0; x < W / 2; x++) {
                    cb[y * LC + x] = (uint8_t)(90 + ((x + 3 * y + t) & 127) + (int)(lcg(&seed) >> 30));
                    cr[y * LC + x] = (uint8_t)(200 - ((x * 2 + y * 5 + 3 * t) & 127));
                }
            pic.data[0] = planes[0]; pic.linesize[0] = LY;
            pic.data[1] = cb; pic.linesize[1] = LC;
            pic.data[2] = cr; pic.linesize[2] = LC;
            for (y = 0; y < H; y++) fwrite(planes[0] + y * LY, 1, W, fs2);
            for (y = 0; y < H; y++) fwrite(cb + y * LC, 1, W / 2, fs2);
            for (y = 0; y < H; y++) fwrite(cr + y * LC, 1, W / 2, fs2);
            len = c2->codec->encode(c2, buf, 1 << 20, &pic);
            if (len <= 4) return -6;
            put_le32(fo2, (uint32_t)len);
            fwrite(buf, 1, len, fo2);
        }
        printf("encoded 422 frames: %d\n", N2);
        close_codec(c2);
        fclose(fs2);
        fclose(fo2);
        free(cb);
        free(cr);
    }
    free(planes[0]);
    free(planes[1]);
    free(planes[2]);
    free(buf);
    return 0;
}

static int encode_audio_leg(const char *dir)
{
    enum { CHUNKS = 40, TOTAL = 2 * CHUNKS * 1500 + 4096 };
    FILE *fp = out_file(dir, "enc_pcm.raw"), *fo = out_file(dir, "enc_audio.bin");
    int16_t *pcm = malloc(TOTAL * 2);
    uint8_t buf[4096];
    uint32_t seed = 99;
    int v = 0, i, pass;

    if (!fp || !fo || !pcm) return -1;
    for (i = 0; i < TOTAL; i++) {                              /* a random walk with loud and quiet stretches */
        const int amp = (i / 5000) % 3 == 0 ? 4000 : ((i / 5000) % 3 == 1 ? 300 : 20);
        v += (int)(lcg(&seed) >> 16) % (2 * amp + 1) - amp;
        if (v > 32000) v = 32000;
        if (v < -32000) v = -32000;
        pcm[i] = (int16_t)v;
    }
    fwrite(pcm, 2, TOTAL, fp);
    for (pass = 0; pass < 2; pass++) {
        AVCodecContext *c = open_codec(&adpcm_ima_amv_encoder);
        int pos = 0;
        if (!c) return -1;
        c->channels = 1;
        c->sample_rate = 11025;
        if (pass == 0) printf("11025 Hz rejected: %d\n", adpcm_ima_amv_encoder.init(c) == -1);   /* adpcm.c:195-198 */
        c->sample_rate = 22050;
        if (adpcm_ima_amv_encoder.init(c) < 0) return -2;
        c->frame_size = pass == 0 ? (int)amvhip_amv_audio_frame_size(22050, 1, 16) : 1471;        /* amvenc.c:276-281 */
        for (i = 0; i < CHUNKS; i++) {
            const int len = c->codec->encode(c, buf, sizeof buf, pcm + pos);
            if (len <= 8) return -3;
            put_le32(fo, (uint32_t)len);
            fwrite(buf, 1, len, fo);
            pos += 2 * (len - 8);                              /* the samples the encoder consumed (:497) */
        }
        printf("audio pass %d: frame_size %d, samples consumed %d, pts of last chunk %lld\n", pass, c->frame_size, pos,
               (long long)c->coded_frame->pts);
        close_codec(c);
    }
    fclose(fp);
    fclose(fo);
    free(pcm);
    return 0;
}

int main(int argc, char **argv)
{
    int r;
    if (argc != 3) {
        fprintf(stderr, "usage: lavc_host <file.amv> <outdir>\n");
        return 2;
    }
    if ((r = decode_leg(argv[1], argv[2])) != 0) { printf("error: decode leg %d\n", r); return 1; }
    if ((r = encode_video_leg(argv[2])) != 0) { printf("error: video encode leg %d\n", r); return 1; }
    if ((r = encode_audio_leg(argv[2])) != 0) { printf("error: audio encode leg %d\n", r); return 1; }
    printf("ok\n");
    return 0;
}
