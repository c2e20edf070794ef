/*
 * ref_harness.c -- entry points INTO the reference's own code.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is ours; everything it calls is the reference's, compiled by oracle/Makefile from the
 * source files where they lie under /root/reference/AMVmuxer/ffmpeg (no copies, no stand-in headers,
 * no -D configuration substitutes):
 *
 *   libavcodec/mjpegenc.c   ff_mjpeg_encode_init :47, encode_block :379 via ff_mjpeg_encode_mb :437,
 *                           ff_mjpeg_encode_dc :357, ff_mjpeg_encode_stuffing :338, escape_FF :282 via
 *                           ff_mjpeg_encode_picture_trailer :345      
 *   libavcodec/mjpeg.c      the K.3 Huffman specifications + ff_mjpeg_build_huffman_codes :129
 *   libavcodec/simple_idct.c simple_idct :410 (idctRowCondDC :78 + idctSparseCol :249; the arithmetic
 *                           of simple_idct_put :390, which only adds the clip to 0..255)   (row a15)
 *   libavcodec/sp5x.h       sp5x_quant_table[10], [11] = the "Q60" tables sp5xdec.c:60-61 puts into the
 *                           JFIF it hands to the MJPEG decoder, and its DHT/SOF/SOS images  (row a14)
 *   libavcodec/imgresample.c img_resample_full_init :425, img_resample :474 (component_resample :341, h_resample
 *                           :312, v_resample :119) -- the rescaler behind the sws_scale shim    (row f3)
 *   libavcodec/resample2.c  av_build_filter :93, which builds its polyphase filters
 *   libavutil/mem.c, mathematics.c  av_malloc/av_free, ff_log2_tab
 *
 * The structures (MpegEncContext, MJpegContext, PutBitContext) are the reference's, from its own headers.
 * The harness only fills the fields the called functions read, the way MPV_encode_init / encode_thread do
 * (mpegvideo_enc.c:534-549 sampling, :2036 last_dc) -- except that the DC predictors start at 0 and the
 * coefficients arrive level-shifted, which is the convention this build's encoder documents.
 *
 * The reference objects are compiled with hidden visibility and one section per function, and the link
 * drops what the harness cannot reach (--gc-sections): that removes the AVCodec tables of mjpegenc.c, whose
 * MPV_encode_* pointers belong to mpegvideo_enc.c (needs the configure-generated config.h: not buildable
 * here), without writing a stub for them.  The link is checked with -z defs: nothing is left unresolved.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "avcodec.h"
#include "dsputil.h"
#include "mpegvideo.h"
#include "mjpeg.h"
#include "mjpegenc.h"
#include "simple_idct.h"
#include "sp5x.h"
#include "swscale.h"   /* libswscale/swscale.h: what imgresample.c itself includes for its shim */

#define EXPORT __attribute__((visibility("default")))

/* Entropy-code nmcu MCUs of 6 blocks (Y0 Y1 Y2 Y3 Cb Cr) with the reference's ff_mjpeg_encode_mb, then its
 * picture trailer (1-bit padding, FF escaping, EOI).  coef: [nmcu*6][64] quantised coefficients in
 * zig-zag order, DC not predicted.  Returns the number of bytes written to out (scan + FF D9), -1 on error. */
EXPORT int amvref_mjpeg_encode_scan(const int16_t *coef, int nmcu, uint8_t *out, int cap)
{
    MpegEncContext *s = av_mallocz(sizeof(MpegEncContext));
    DCTELEM (*block)[64] = av_malloc(sizeof(DCTELEM) * 6 * 64);
    int m, b, i, n = -1;

    if (!s || !block || ff_mjpeg_encode_init(s) < 0)
        goto done;
    init_put_bits(&s->pb, out, cap);
    s->header_bits = 0;                       /* the scan starts at out[0] (escape_FF's start, :352) */
    s->chroma_format = CHROMA_420;            /* ff_mjpeg_encode_mb :443 */
    s->last_dc[0] = s->last_dc[1] = s->last_dc[2] = 0;
    for (i = 0; i < 64; i++)
        s->intra_scantable.permutated[i] = i; /* the blocks arrive in scan order */
    for (m = 0; m < nmcu; m++) {
        for (b = 0; b < 6; b++) {
            int last = 0;
            for (i = 0; i < 64; i++) {
                block[b][i] = coef[(m * 6 + b) * 64 + i];
                if (block[b][i])
                    last = i;
            }
            s->block_last_index[b] = last;    /* what dct_quantize_c returns, mpegvideo_enc.c:3722 */
        }
        if ((put_bits_count(&s->pb) >> 3) + 6 * 64 * 4 + 16 > cap)
            goto done;
        ff_mjpeg_encode_mb(s, block);
    }
    ff_mjpeg_encode_picture_trailer(s);
    flush_put_bits(&s->pb);
    n = put_bits_count(&s->pb) >> 3;
done:
    if (s) {
        ff_mjpeg_encode_close(s);
        av_free(s);
    }
    av_free(block);
    return n;
}

/* simple_idct on n blocks of 64 int16, in place */
EXPORT void amvref_simple_idct(int16_t *blocks, int n)
{
    int i;
    for (i = 0; i < n; i++)
        simple_idct(blocks + 64 * i);
}

/* which: 0 luma Q60 table (sp5x_quant_table[10]), 1 chroma (sp5x_quant_table[11]) -- sp5xdec.c:40,60-61 */
EXPORT void amvref_sp5x_quant(int which, uint8_t out[64])
{
    memcpy(out, sp5x_quant_table[5 * 2 + (which ? 1 : 0)], 64);
}

/* the DHT / SOF / SOS images sp5xdec.c:64-73 copies in front of an AMV scan; returns the size */
EXPORT int amvref_sp5x_segment(int which, uint8_t *out, int cap)
{
    const uint8_t *src = which == 0 ? sp5x_data_dht : which == 1 ? sp5x_data_sof : sp5x_data_sos;
    int n = which == 0 ? (int)sizeof(sp5x_data_dht) : which == 1 ? (int)sizeof(sp5x_data_sof) : (int)sizeof(sp5x_data_sos);
    if (n > cap)
        return -1;
    memcpy(out, src, n);
    return n;
}

/* Huffman specification t (0 DC luma, 1 DC chroma, 2 AC luma, 3 AC chroma) of mjpeg.c:62-127:
 * bits[17] (index 1..16) and the symbol values; returns the number of symbols */
EXPORT int amvref_mjpeg_huffman_spec(int t, uint8_t bits[17], uint8_t vals[256])
{
    const uint8_t *b = t == 0 ? ff_mjpeg_bits_dc_luminance : t == 1 ? ff_mjpeg_bits_dc_chrominance
                     : t == 2 ? ff_mjpeg_bits_ac_luminance : ff_mjpeg_bits_ac_chrominance;
    const uint8_t *v = t == 0 ? ff_mjpeg_val_dc_luminance : t == 1 ? ff_mjpeg_val_dc_chrominance
                     : t == 2 ? ff_mjpeg_val_ac_luminance : ff_mjpeg_val_ac_chrominance;
    int i, n = 0;
    for (i = 0; i < 17; i++)
        bits[i] = b[i];
    for (i = 1; i <= 16; i++)
        n += b[i];
    memcpy(vals, v, n);
    return n;
}

/* the code book ff_mjpeg_build_huffman_codes derives for specification t: size[256], code[256] */
EXPORT void amvref_mjpeg_huffman_codes(int t, uint8_t size[256], uint16_t code[256])
{
    uint8_t bits[17], vals[256];
    memset(size, 0, 256);
    memset(code, 0, 512);
    amvref_mjpeg_huffman_spec(t, bits, vals);
    ff_mjpeg_build_huffman_codes(size, code, bits, vals);
}

/* img_resample on tight YUV420P frames: in = Y iw x ih, Cb, Cr (iw/2 x ih/2); out likewise at ow x oh */
EXPORT int amvref_img_resample(const uint8_t *in, int iw, int ih, uint8_t *out, int ow, int oh)
{
    ImgReSampleContext *s = img_resample_init(ow, oh, iw, ih);
    AVPicture src, dst;
    if (!s)
        return -1;
    memset(&src, 0, sizeof src);
    memset(&dst, 0, sizeof dst);
    src.data[0] = (uint8_t *)in;
    src.data[1] = src.data[0] + iw * ih;
    src.data[2] = src.data[1] + (iw / 2) * (ih / 2);
    src.linesize[0] = iw;
    src.linesize[1] = src.linesize[2] = iw / 2;
    dst.data[0] = out;
    dst.data[1] = dst.data[0] + ow * oh;
    dst.data[2] = dst.data[1] + (ow / 2) * (oh / 2);
    dst.linesize[0] = ow;
    dst.linesize[1] = dst.linesize[2] = ow / 2;
    img_resample(s, &dst, &src);
    img_resample_close(s);
    return 0;
}
