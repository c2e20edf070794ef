/*
 * amv_oracle.c -- CPU restatement of the AMV codec hot path.  TEST INFRASTRUCTURE ONLY.
 * See amv_oracle.h for the rules about who may load this and how it is pinned.
 *
 * Compile with -fwrapv (signed wrap-around is relied on, as the reference relies on
 * two's-complement behaviour of its compiler).  Reference citations are file:line
 * under /root/reference:
 *   amvlib = C-AMVDecoder/amvlib, lavc = AMVmuxer/ffmpeg/libavcodec
 */
#include "amv_oracle.h"
#include <string.h>
#include <stdlib.h>
#include <math.h>

/* ------------------------------------------------------------------------------------
 * constant tables (data of the format, amvlib/AmvJpeg.c:30-39,52-61,65-131,133-150)
 * ---------------------------------------------------------------------------------- */
static const uint8_t k_qt_luma[64] = { /* zig-zag order, AmvJpeg.c:30-39 */
     8,  6,  6,  7,  6,  5,  8,  7,  7,  7,  9,  9,  8, 10, 12, 20,
    13, 12, 11, 11, 12, 25, 18, 19, 15, 20, 29, 26, 31, 30, 29, 26,
    28, 28, 32, 36, 46, 39, 32, 34, 44, 39, 28, 28, 40, 55, 41, 44,
    48, 49, 52, 52, 52, 31, 39, 57, 61, 56, 50, 60, 46, 51, 52, 50 };
static const uint8_t k_qt_chroma[64] = { /* zig-zag order, AmvJpeg.c:52-61 */
     9,  9,  9, 12, 11, 12, 24, 13, 13, 24, 50, 33, 28, 33, 50, 50,
    50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50,
    50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50,
    50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50 };

/* JPEG K.3 tables: number of codes of each length 1..16, then the symbols (AmvJpeg.c:65-131) */
static const uint8_t k_bits[4][16] = {
    { 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0 },          /* DC luma   */
    { 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0 },          /* DC chroma */
    { 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d },       /* AC luma   */
    { 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77 } };     /* AC chroma */
static const uint8_t k_val_dc[12] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11 };
static const uint8_t k_val_ac_luma[162] = {
    0x01,0x02,0x03,0x00,0x04,0x11,0x05,0x12,0x21,0x31,0x41,0x06,0x13,0x51,0x61,0x07,
    0x22,0x71,0x14,0x32,0x81,0x91,0xa1,0x08,0x23,0x42,0xb1,0xc1,0x15,0x52,0xd1,0xf0,
    0x24,0x33,0x62,0x72,0x82,0x09,0x0a,0x16,0x17,0x18,0x19,0x1a,0x25,0x26,0x27,0x28,
    0x29,0x2a,0x34,0x35,0x36,0x37,0x38,0x39,0x3a,0x43,0x44,0x45,0x46,0x47,0x48,0x49,
    0x4a,0x53,0x54,0x55,0x56,0x57,0x58,0x59,0x5a,0x63,0x64,0x65,0x66,0x67,0x68,0x69,
    0x6a,0x73,0x74,0x75,0x76,0x77,0x78,0x79,0x7a,0x83,0x84,0x85,0x86,0x87,0x88,0x89,
    0x8a,0x92,0x93,0x94,0x95,0x96,0x97,0x98,0x99,0x9a,0xa2,0xa3,0xa4,0xa5,0xa6,0xa7,
    0xa8,0xa9,0xaa,0xb2,0xb3,0xb4,0xb5,0xb6,0xb7,0xb8,0xb9,0xba,0xc2,0xc3,0xc4,0xc5,
    0xc6,0xc7,0xc8,0xc9,0xca,0xd2,0xd3,0xd4,0xd5,0xd6,0xd7,0xd8,0xd9,0xda,0xe1,0xe2,
    0xe3,0xe4,0xe5,0xe6,0xe7,0xe8,0xe9,0xea,0xf1,0xf2,0xf3,0xf4,0xf5,0xf6,0xf7,0xf8,
    0xf9,0xfa };
static const uint8_t k_val_ac_chroma[162] = {
    0x00,0x01,0x02,0x03,0x11,0x04,0x05,0x21,0x31,0x06,0x12,0x41,0x51,0x07,0x61,0x71,
    0x13,0x22,0x32,0x81,0x08,0x14,0x42,0x91,0xa1,0xb1,0xc1,0x09,0x23,0x33,0x52,0xf0,
    0x15,0x62,0x72,0xd1,0x0a,0x16,0x24,0x34,0xe1,0x25,0xf1,0x17,0x18,0x19,0x1a,0x26,
    0x27,0x28,0x29,0x2a,0x35,0x36,0x37,0x38,0x39,0x3a,0x43,0x44,0x45,0x46,0x47,0x48,
    0x49,0x4a,0x53,0x54,0x55,0x56,0x57,0x58,0x59,0x5a,0x63,0x64,0x65,0x66,0x67,0x68,
    0x69,0x6a,0x73,0x74,0x75,0x76,0x77,0x78,0x79,0x7a,0x82,0x83,0x84,0x85,0x86,0x87,
    0x88,0x89,0x8a,0x92,0x93,0x94,0x95,0x96,0x97,0x98,0x99,0x9a,0xa2,0xa3,0xa4,0xa5,
    0xa6,0xa7,0xa8,0xa9,0xaa,0xb2,0xb3,0xb4,0xb5,0xb6,0xb7,0xb8,0xb9,0xba,0xc2,0xc3,
    0xc4,0xc5,0xc6,0xc7,0xc8,0xc9,0xca,0xd2,0xd3,0xd4,0xd5,0xd6,0xd7,0xd8,0xd9,0xda,
    0xe2,0xe3,0xe4,0xe5,0xe6,0xe7,0xe8,0xe9,0xea,0xf2,0xf3,0xf4,0xf5,0xf6,0xf7,0xf8,
    0xf9,0xfa };
static const uint8_t *const k_vals[4] = { k_val_dc, k_val_dc, k_val_ac_luma, k_val_ac_chroma };

/* natural position (row, col) -> index in the bitstream's zig-zag order.
 * amvlib's copy (AmvJpeg.c:133-143) has 37 at [3][4] where the standard has 31. */
static const uint8_t k_zigzag_std[64] = {
     0,  1,  5,  6, 14, 15, 27, 28,
     2,  4,  7, 13, 16, 26, 29, 42,
     3,  8, 12, 17, 25, 30, 41, 43,
     9, 11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54,
    20, 22, 33, 38, 46, 51, 55, 60,
    21, 34, 37, 47, 50, 56, 59, 61,
    35, 36, 48, 49, 57, 58, 62, 63 };
#define AMVLIB_QUIRK_POS (3 * 8 + 4)
#define AMVLIB_QUIRK_VAL 37

static inline int zz_index(int nat, uint32_t flags)
{
    if (nat == AMVLIB_QUIRK_POS && !(flags & AMVO_FLAG_ZIGZAG_FIXED))
        return AMVLIB_QUIRK_VAL;
    return k_zigzag_std[nat];
}

/* ------------------------------------------------------------------------------------
 * geometry
 * ---------------------------------------------------------------------------------- */
uint32_t amvo_stride(uint32_t w) { return (w * 24 + 31) / 32 * 4; }   /* AmvJpeg.c:420,1524 */
uint32_t amvo_mcus_per_row(uint32_t w) { return (w + 15) / 16; }      /* AmvJpeg.c:1276-1281 */
uint32_t amvo_mcu_rows(uint32_t h) { return (h + 15) / 16; }          /* AmvJpeg.c:1280-1284 */

/* ------------------------------------------------------------------------------------
 * canonical Huffman bounds, PrepareForVideoDecode AmvJpeg.c:1454-1481
 * ---------------------------------------------------------------------------------- */
typedef struct {
    uint16_t minc[16], maxc[16];
    int16_t pos[16];
    uint8_t cnt[16];
} hufbounds;

static void build_bounds(hufbounds *hb, const uint8_t bits[16])
{
    int i = 0, j;
    memset(hb, 0, sizeof *hb);
    for (j = 0; j < 16; j++) hb->cnt[j] = bits[j];
    while (hb->cnt[i] == 0) i++;                       /* :1464 */
    hb->minc[i] = 0;                                    /* :1471 */
    hb->maxc[i] = (uint16_t)(hb->cnt[i] - 1);           /* :1472 */
    for (j = i + 1; j < 16; j++) {                      /* :1473-1477 */
        hb->minc[j] = (uint16_t)((hb->maxc[j - 1] + 1) << 1);
        hb->maxc[j] = (uint16_t)(hb->minc[j] + hb->cnt[j] - 1);
    }
    hb->pos[0] = 0;                                     /* :1478-1480 */
    for (j = 1; j < 16; j++) hb->pos[j] = (int16_t)(hb->cnt[j - 1] + hb->pos[j - 1]);
}

/* ------------------------------------------------------------------------------------
 * bit reader: ReadByte AmvJpeg.c:1061-1071 and the BitPos/CurByte handling of
 * DecodeElement :850-863.  Bytes past the chunk read as zero (flagged TRUNCATED).
 * ---------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *buf;
    uint32_t len, pos;
    int bitpos;
    unsigned cur;
    uint64_t consumed, valid;
} bitrd;

static void rd_byte(bitrd *b)
{
    unsigned v = 0;
    if (b->pos < b->len) { v = b->buf[b->pos]; b->valid += 8; }
    b->pos++;
    if (v == 0xff) b->pos++;            /* :1066-1067: the byte after FF is dropped unseen */
    b->bitpos = 8;
    b->cur = v;
}

static unsigned rd_bit(bitrd *b)
{
    unsigned bit;
    if (b->bitpos < 1) rd_byte(b);
    b->bitpos--;
    bit = (b->cur >> b->bitpos) & 1u;
    b->cur &= (1u << b->bitpos) - 1u;
    b->consumed++;
    return bit;
}

/* DecodeElement AmvJpeg.c:842-936 */
static int decode_element(bitrd *b, const hufbounds *hb, const uint8_t *vals, int *run, int *val)
{
    int code = (int)rd_bit(b), len = 1, size, sym;
    unsigned v = 0;
    while (hb->cnt[len - 1] == 0 || code < hb->minc[len - 1] || code > hb->maxc[len - 1]) { /* :867-869 */
        code = (code << 1) + (int)rd_bit(b);
        len++;
        if (len > 16) return AMVO_ST_FORMAT;                                               /* :887 */
    }
    sym = vals[(uint16_t)(code - hb->minc[len - 1] + hb->pos[len - 1])];                   /* :891-892 */
    *run = sym >> 4;
    size = sym & 15;
    if (size == 0) { *val = 0; return 0; }                                                 /* :895-899 */
    for (int i = 0; i < size; i++) v = (v << 1) | rd_bit(b);                               /* :902-922 */
    if (v >> (size - 1)) *val = (int16_t)v;                                                /* :924-927 */
    else *val = (int16_t)-(int)(((1u << size) - 1u) - v);                                  /* :928-933 */
    return 0;
}

/* HufBlock AmvJpeg.c:939-974 */
static int huf_block(bitrd *b, const hufbounds hb[4], int dctab, int actab, int16_t blk[64])
{
    int run, val, count = 0, st;
    st = decode_element(b, &hb[dctab], k_vals[dctab], &run, &val);
    if (st) return st;
    blk[count++] = (int16_t)val;
    while (count < 64) {
        st = decode_element(b, &hb[actab], k_vals[actab], &run, &val);
        if (st) return st;
        if (run == 0 && val == 0) {              /* :959-964 end of block */
            while (count < 64) blk[count++] = 0;
        } else {
            if (count + run > 63) return AMVO_ST_OVERRUN; /* reference writes out of bounds here */
            for (int i = 0; i < run; i++) blk[count++] = 0;
            blk[count++] = (int16_t)val;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * IDCT: Fast_IDCT / idctrow / idctcol, AmvJpeg.c:1050-1059,1082-1175; W1..W7 :145-150
 * ---------------------------------------------------------------------------------- */
#define W1 2841
#define W2 2676
#define W3 2408
#define W5 1609
#define W6 1108
#define W7 565

/* iclp[] (AmvJpeg.c:1073-1080) covers -512..511; outside it the reference reads out of
 * bounds, defined here as saturation. */
static inline int32_t iclp(int32_t i) { return i < -256 ? -256 : (i > 255 ? 255 : i); }

static void idct_row(int32_t *blk) /* :1082-1128 */
{
    int32_t x0, x1, x2, x3, x4, x5, x6, x7, x8;
    x1 = blk[4] * 2048; x2 = blk[6]; x3 = blk[2]; x4 = blk[1]; x5 = blk[7]; x6 = blk[5]; x7 = blk[3];
    if (!(x1 | x2 | x3 | x4 | x5 | x6 | x7)) {                 /* :1087-1092 */
        blk[0] = blk[1] = blk[2] = blk[3] = blk[4] = blk[5] = blk[6] = blk[7] = blk[0] * 8;
        return;
    }
    x0 = blk[0] * 2048 + 128;
    x8 = W7 * (x4 + x5);
    x4 = x8 + (W1 - W7) * x4;
    x5 = x8 - (W1 + W7) * x5;
    x8 = W3 * (x6 + x7);
    x6 = x8 - (W3 - W5) * x6;
    x7 = x8 - (W3 + W5) * x7;
    x8 = x0 + x1;
    x0 -= x1;
    x1 = W6 * (x3 + x2);
    x2 = x1 - (W2 + W6) * x2;
    x3 = x1 + (W2 - W6) * x3;
    x1 = x4 + x6;
    x4 -= x6;
    x6 = x5 + x7;
    x5 -= x7;
    x7 = x8 + x3;
    x8 -= x3;
    x3 = x0 + x2;
    x0 -= x2;
    x2 = (181 * (x4 + x5) + 128) >> 8;
    x4 = (181 * (x4 - x5) + 128) >> 8;
    blk[0] = (x7 + x1) >> 8;
    blk[1] = (x3 + x2) >> 8;
    blk[2] = (x0 + x4) >> 8;
    blk[3] = (x8 + x6) >> 8;
    blk[4] = (x8 - x6) >> 8;
    blk[5] = (x0 - x4) >> 8;
    blk[6] = (x3 - x2) >> 8;
    blk[7] = (x7 - x1) >> 8;
}

static void idct_col(int32_t *blk) /* :1130-1175 */
{
    int32_t x0, x1, x2, x3, x4, x5, x6, x7, x8;
    x1 = blk[8 * 4] * 256; x2 = blk[8 * 6]; x3 = blk[8 * 2]; x4 = blk[8 * 1];
    x5 = blk[8 * 7]; x6 = blk[8 * 5]; x7 = blk[8 * 3];
    if (!(x1 | x2 | x3 | x4 | x5 | x6 | x7)) {                 /* :1134-1140 */
        blk[8 * 0] = blk[8 * 1] = blk[8 * 2] = blk[8 * 3] = blk[8 * 4] = blk[8 * 5] =
            blk[8 * 6] = blk[8 * 7] = iclp((blk[8 * 0] + 32) >> 6);
        return;
    }
    x0 = blk[8 * 0] * 256 + 8192;
    x8 = W7 * (x4 + x5) + 4;
    x4 = (x8 + (W1 - W7) * x4) >> 3;
    x5 = (x8 - (W1 + W7) * x5) >> 3;
    x8 = W3 * (x6 + x7) + 4;
    x6 = (x8 - (W3 - W5) * x6) >> 3;
    x7 = (x8 - (W3 + W5) * x7) >> 3;
    x8 = x0 + x1;
    x0 -= x1;
    x1 = W6 * (x3 + x2) + 4;
    x2 = (x1 - (W2 + W6) * x2) >> 3;
    x3 = (x1 + (W2 - W6) * x3) >> 3;
    x1 = x4 + x6;
    x4 -= x6;
    x6 = x5 + x7;
    x5 -= x7;
    x7 = x8 + x3;
    x8 -= x3;
    x3 = x0 + x2;
    x0 -= x2;
    x2 = (181 * (x4 + x5) + 128) >> 8;
    x4 = (181 * (x4 - x5) + 128) >> 8;
    blk[8 * 0] = iclp((x7 + x1) >> 14);
    blk[8 * 1] = iclp((x3 + x2) >> 14);
    blk[8 * 2] = iclp((x0 + x4) >> 14);
    blk[8 * 3] = iclp((x8 + x6) >> 14);
    blk[8 * 4] = iclp((x8 - x6) >> 14);
    blk[8 * 5] = iclp((x0 - x4) >> 14);
    blk[8 * 6] = iclp((x3 - x2) >> 14);
    blk[8 * 7] = iclp((x7 - x1) >> 14);
}

void amvo_idct_block(int32_t blk[64]) /* Fast_IDCT :1050-1059 */
{
    for (int i = 0; i < 8; i++) idct_row(blk + 8 * i);
    for (int i = 0; i < 8; i++) idct_col(blk + i);
}

/* IQtIZzBlock AmvJpeg.c:1010-1048 */
void amvo_dequant_idct_block(const int16_t coef[64], int comp, uint32_t flags, int32_t out[64])
{
    const uint8_t *qt = comp == 0 ? k_qt_luma : k_qt_chroma;   /* :1430,1434,1438 */
    int32_t offset = comp == 0 ? 128 : 0;                      /* :1023,1027,1031 */
    for (int i = 0; i < 64; i++) {
        int tag = zz_index(i, flags);                          /* :1039 */
        out[i] = (int32_t)coef[tag] * (int32_t)qt[tag];        /* :1040 */
    }
    amvo_idct_block(out);
    for (int i = 0; i < 64; i++) out[i] += offset;             /* :1047 */
}

/* StoreBuffer's pixel maths AmvJpeg.c:805-831 */
void amvo_yuv_to_bgr(int32_t y, int32_t u, int32_t v, uint8_t bgr[3])
{
    int32_t rr = (y * 256 + 18 * u + 367 * v) >> 8;
    int32_t gg = (y * 256 - 159 * u - 220 * v) >> 8;
    int32_t bb = (y * 256 + 411 * u - 29 * v) >> 8;
    bgr[0] = (uint8_t)(bb < 0 ? 0 : (bb > 255 ? 255 : bb));
    bgr[1] = (uint8_t)(gg < 0 ? 0 : (gg > 255 ? 255 : gg));
    bgr[2] = (uint8_t)(rr < 0 ? 0 : (rr > 255 ? 255 : rr));
}

/* ------------------------------------------------------------------------------------
 * frame decode: AmvJpegDecode AmvJpeg.c:1515-1539, Decode :1244-1287,
 * DecodeMCUBlock :1177-1242, GetYUV :754-787, StoreBuffer :789-840
 * ---------------------------------------------------------------------------------- */
int amvo_decode_frame(const uint8_t *chunk, uint32_t len, uint32_t w, uint32_t h,
                      uint32_t flags, uint8_t *out, int16_t *coef_out,
                      uint32_t *nmcu_ok, uint32_t *status)
{
    hufbounds hb[4];
    bitrd b;
    const uint32_t stride = amvo_stride(w);
    const uint32_t mcw = amvo_mcus_per_row(w), mch = amvo_mcu_rows(h);
    int16_t pred[3] = { 0, 0, 0 };                              /* :1511 */
    uint32_t st = 0, mcu = 0;
    static const int comp_of[6] = { 0, 0, 0, 0, 1, 2 };

    for (int t = 0; t < 4; t++) build_bounds(&hb[t], k_bits[t]);
    memset(out, 0, (size_t)stride * h);                         /* AMVDec.c:283 */
    if (coef_out) memset(coef_out, 0, (size_t)mcw * mch * 6 * 64 * sizeof(int16_t));
    memset(&b, 0, sizeof b);
    b.buf = chunk; b.len = len; b.pos = 2;                      /* :1527 skip FF D8 */

    for (uint32_t my = 0; my < mch && !st; my++) {
        for (uint32_t mx = 0; mx < mcw; mx++) {
            int16_t mcub[6][64];
            int32_t px[6][64];
            for (int k = 0; k < 6 && !st; k++) {                /* :1195-1224 */
                int c = comp_of[k];
                int r = huf_block(&b, hb, c ? 1 : 0, c ? 3 : 2, mcub[k]);
                if (r) { st |= (uint32_t)r; break; }
                mcub[k][0] = (int16_t)(mcub[k][0] + pred[c]);   /* :1200-1201 etc. */
                pred[c] = mcub[k][0];
            }
            if (st) break;
            if (coef_out) memcpy(coef_out + (size_t)mcu * 384, mcub, sizeof mcub);
            for (int k = 0; k < 6; k++)                         /* :1266-1268 */
                amvo_dequant_idct_block(mcub[k], comp_of[k], flags, px[k]);
            /* GetYUV + StoreBuffer: Y tile is 16x16 from blocks 0..3, chroma is nearest (i/2, j/2) */
            for (uint32_t i = 0; i < 16; i++) {
                uint32_t row = my * 16 + i;
                if (row >= h) break;                            /* :798,837-838 */
                uint8_t *dst = out + (size_t)(h - 1 - row) * stride + (size_t)mx * 16 * 3; /* :800 */
                for (uint32_t j = 0; j < 16; j++) {
                    if (mx * 16 + j >= w) break;                /* :803,833-834 */
                    int32_t y = px[(i >> 3) * 2 + (j >> 3)][(i & 7) * 8 + (j & 7)];
                    int32_t u = px[4][(i >> 1) * 8 + (j >> 1)];
                    int32_t v = px[5][(i >> 1) * 8 + (j >> 1)];
                    amvo_yuv_to_bgr(y, u, v, dst + 3 * j);
                }
            }
            mcu++;
        }
    }
    if (b.consumed > b.valid) st |= AMVO_ST_TRUNCATED;
    if (nmcu_ok) *nmcu_ok = mcu;
    if (status) *status = st;
    return st ? -1 : 0;
}

/* ------------------------------------------------------------------------------------
 * FFmpeg-compat decode: what the patched FFmpeg's amv_decoder produces (sp5xdec.c:33-93 ->
 * ff_mjpeg_decode_frame -> mjpeg_decode_scan mjpegdec.c:660-736 -> decode_block :376-430 ->
 * simple_idct_put simple_idct.c:390-408).  Same bitstream and Huffman tables as above; different
 * quantiser tables ("Q60", sp5x.h:187-195 via sp5xdec.c:40,60-61), the standard zig-zag
 * (dsputil.c:50-59), DC kept in dequantised units from 1024 (mjpegdec.c:805,388-390), DCTELEM = int16
 * everywhere, FFmpeg's simple_idct, planar YUVJ420P output flipped with the formula of :672-677.
 * ---------------------------------------------------------------------------------- */
static const uint8_t k_q60_luma[64] = { /* zig-zag order, sp5x.h:187-190 = sp5x_quant_table[10] */
    13,  9, 10, 11, 10,  8, 13, 11, 10, 11, 14, 14, 13, 15, 19, 32,
    21, 19, 18, 18, 19, 39, 28, 30, 23, 32, 46, 41, 49, 48, 46, 41,
    45, 44, 51, 58, 74, 62, 51, 54, 70, 55, 44, 45, 64, 87, 65, 70,
    76, 78, 82, 83, 82, 50, 62, 90, 97, 90, 80, 96, 74, 81, 82, 79 };
static const uint8_t k_q60_chroma[64] = { /* sp5x.h:191-194 = sp5x_quant_table[11] */
    14, 14, 14, 19, 17, 19, 38, 21, 21, 38, 79, 53, 45, 53, 79, 79,
    79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79,
    79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79,
    79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79 };

void amvo_q60_table(int chroma, uint8_t out[64]) { memcpy(out, chroma ? k_q60_chroma : k_q60_luma, 64); }

#define SW1 22725 /* simple_idct.c:47-55 */
#define SW2 21407
#define SW3 19266
#define SW4 16383
#define SW5 12873
#define SW6 8867
#define SW7 4520
#define SROW_SHIFT 11
#define SCOL_SHIFT 20

/* idctRowCondDC simple_idct.c:78-181 (results stored back into int16) */
static void sidct_row(int16_t *row)
{
    int a0, a1, a2, a3, b0, b1, b2, b3;
    if (!(row[1] | row[2] | row[3] | row[4] | row[5] | row[6] | row[7])) {   /* :107-117: NOT the general formula */
        int16_t v = (int16_t)((row[0] << 3) & 0xffff);
        for (int i = 0; i < 8; i++) row[i] = v;
        return;
    }
    a0 = SW4 * row[0] + (1 << (SROW_SHIFT - 1));
    a1 = a0; a2 = a0; a3 = a0;
    a0 += SW2 * row[2]; a1 += SW6 * row[2]; a2 -= SW6 * row[2]; a3 -= SW2 * row[2];
    b0 = SW1 * row[1] + SW3 * row[3];
    b1 = SW3 * row[1] - SW7 * row[3];
    b2 = SW5 * row[1] - SW1 * row[3];
    b3 = SW7 * row[1] - SW5 * row[3];
    /* :152-171, the guard only skips additions of zero */
    a0 += SW4 * row[4] + SW6 * row[6];
    a1 += -SW4 * row[4] - SW2 * row[6];
    a2 += -SW4 * row[4] + SW2 * row[6];
    a3 += SW4 * row[4] - SW6 * row[6];
    b0 += SW5 * row[5] + SW7 * row[7];
    b1 += -SW1 * row[5] - SW5 * row[7];
    b2 += SW7 * row[5] + SW3 * row[7];
    b3 += SW3 * row[5] - SW1 * row[7];
    row[0] = (int16_t)((a0 + b0) >> SROW_SHIFT);
    row[7] = (int16_t)((a0 - b0) >> SROW_SHIFT);
    row[1] = (int16_t)((a1 + b1) >> SROW_SHIFT);
    row[6] = (int16_t)((a1 - b1) >> SROW_SHIFT);
    row[2] = (int16_t)((a2 + b2) >> SROW_SHIFT);
    row[5] = (int16_t)((a2 - b2) >> SROW_SHIFT);
    row[3] = (int16_t)((a3 + b3) >> SROW_SHIFT);
    row[4] = (int16_t)((a3 - b3) >> SROW_SHIFT);
}

/* idctSparseCol / idctSparseColPut simple_idct.c:183-247,320-388: the eight sums before the clip */
static void sidct_col(const int16_t *col, int out[8])
{
    int a0, a1, a2, a3, b0, b1, b2, b3;
    a0 = SW4 * (col[8 * 0] + ((1 << (SCOL_SHIFT - 1)) / SW4));                /* :190 */
    a1 = a0; a2 = a0; a3 = a0;
    a0 += SW2 * col[8 * 2]; a1 += SW6 * col[8 * 2]; a2 -= SW6 * col[8 * 2]; a3 -= SW2 * col[8 * 2];
    b0 = SW1 * col[8 * 1] + SW3 * col[8 * 3];
    b1 = SW3 * col[8 * 1] - SW7 * col[8 * 3];
    b2 = SW5 * col[8 * 1] - SW1 * col[8 * 3];
    b3 = SW7 * col[8 * 1] - SW5 * col[8 * 3];
    a0 += SW4 * col[8 * 4]; a1 -= SW4 * col[8 * 4]; a2 -= SW4 * col[8 * 4]; a3 += SW4 * col[8 * 4];
    b0 += SW5 * col[8 * 5]; b1 -= SW1 * col[8 * 5]; b2 += SW7 * col[8 * 5]; b3 += SW3 * col[8 * 5];
    a0 += SW6 * col[8 * 6]; a1 -= SW2 * col[8 * 6]; a2 += SW2 * col[8 * 6]; a3 -= SW6 * col[8 * 6];
    b0 += SW7 * col[8 * 7]; b1 -= SW5 * col[8 * 7]; b2 += SW3 * col[8 * 7]; b3 -= SW1 * col[8 * 7];
    out[0] = (a0 + b0) >> SCOL_SHIFT; out[1] = (a1 + b1) >> SCOL_SHIFT;
    out[2] = (a2 + b2) >> SCOL_SHIFT; out[3] = (a3 + b3) >> SCOL_SHIFT;
    out[4] = (a3 - b3) >> SCOL_SHIFT; out[5] = (a2 - b2) >> SCOL_SHIFT;
    out[6] = (a1 - b1) >> SCOL_SHIFT; out[7] = (a0 - b0) >> SCOL_SHIFT;
}

/* simple_idct simple_idct.c:410-419: in place, no clip */
void amvo_simple_idct(int16_t blk[64])
{
    int t[8];
    for (int i = 0; i < 8; i++) sidct_row(blk + 8 * i);
    for (int i = 0; i < 8; i++) {
        sidct_col(blk + i, t);
        for (int r = 0; r < 8; r++) blk[8 * r + i] = (int16_t)t[r];
    }
}

/* simple_idct_put :390-398.  ff_cropTbl covers -1024..1279 (dsputil.h MAX_NEG_CROP); beyond it the reference
 * reads out of bounds, defined here as saturation */
void amvo_simple_idct_put(uint8_t *dest, int line_size, int16_t blk[64])
{
    int t[8];
    for (int i = 0; i < 8; i++) sidct_row(blk + 8 * i);
    for (int i = 0; i < 8; i++) {
        sidct_col(blk + i, t);
        for (int r = 0; r < 8; r++) dest[r * line_size + i] = (uint8_t)(t[r] < 0 ? 0 : (t[r] > 255 ? 255 : t[r]));
    }
}

/* decode_block's dequantisation mjpegdec.c:388-390,417,424 from a block of quantised coefficients in scan
 * order whose DC is already the running sum of the differences (amvo_decode_frame's coef_out): FFmpeg keeps
 * last_dc = 1024 + q0 * (that sum), and every store goes through DCTELEM = int16 */
void amvo_ffmpeg_dequant_block(const int16_t coef[64], int comp, int16_t out[64])
{
    const uint8_t *qt = comp == 0 ? k_q60_luma : k_q60_chroma;
    for (int nat = 0; nat < 64; nat++) {
        int scan = k_zigzag_std[nat];
        int v = (int)coef[scan] * (int)qt[scan];
        if (scan == 0) v += 1024;                                             /* :805 */
        out[nat] = (int16_t)v;
    }
}

uint32_t amvo_yuv420_frame_bytes(uint32_t w, uint32_t h) { return w * h + 2 * ((w + 1) / 2) * ((h + 1) / 2); }

/* Decode one chunk the way FFmpeg's amv_decoder does.  out: Y plane w*h, then Cb, then Cr, each
 * ((w+1)/2) x ((h+1)/2), rows tight (linesize = width), zero-filled first.  Plane row p of a component with
 * vertical factor v (2 luma, 1 chroma) receives scan row v*(8*mcu_rows - ((h/2)&7)) - 1 - p (mjpegdec.c:672-677;
 * rows the formula sends outside the plane fall into FFmpeg's edge area and are dropped here; columns beyond
 * the plane width likewise).  Errors, statuses and nmcu_ok are defined as in amvo_decode_frame: MCUs before the
 * first error are stored, the rest stays zero (FFmpeg logs the error and keeps whatever the buffer held). */
/* the entropy stage alone, block by block (HufBlock + the DC prediction of DecodeMCUBlock, AmvJpeg.c:939-974,1200-1221):
 * coef[b] = block b's 64 quantised coefficients in scan order, for every WHOLE block before the first error -- the blocks
 * of a failing MCU in front of the failing one too, which amvo_decode_frame does not hand out.  Returns the count. */
uint32_t amvo_entropy_blocks(const uint8_t *chunk, uint32_t len, uint32_t nblocks, int16_t *coef, uint32_t *status)
{
    hufbounds hb[4];
    bitrd b;
    int16_t pred[3] = { 0, 0, 0 };
    static const int comp_of[6] = { 0, 0, 0, 0, 1, 2 };
    uint32_t st = 0, done = 0;
    for (int t = 0; t < 4; t++) build_bounds(&hb[t], k_bits[t]);
    memset(&b, 0, sizeof b);
    b.buf = chunk; b.len = len; b.pos = 2;
    for (; done < nblocks; done++) {
        int c = comp_of[done % 6];
        int16_t *blk = coef + (size_t)done * 64;
        int r = huf_block(&b, hb, c ? 1 : 0, c ? 3 : 2, blk);
        if (r) { st |= (uint32_t)r; break; }
        blk[0] = (int16_t)(blk[0] + pred[c]);
        pred[c] = blk[0];
    }
    if (b.consumed > b.valid) st |= AMVO_ST_TRUNCATED;
    if (status) *status = st;
    return done;
}

static int decode_frame_ffmpeg(const uint8_t *chunk, uint32_t len, uint32_t w, uint32_t h,
                               uint8_t *out, uint32_t *nmcu_ok, uint32_t *status, int keep);
int amvo_decode_frame_ffmpeg(const uint8_t *chunk, uint32_t len, uint32_t w, uint32_t h,
                             uint8_t *out, uint32_t *nmcu_ok, uint32_t *status)
{
    return decode_frame_ffmpeg(chunk, len, w, h, out, nmcu_ok, status, 0);
}
/* The same with what mjpegdec.c leaves behind on a damaged chunk (AMVHIP_FLAG_FFMPEG_KEEP): mjpeg_decode_scan returns
 * at the block whose decode_block fails (:699-706); the blocks before it -- of the same MCU too -- have been put into the
 * picture (:708-716), nothing else of `out` is touched (not cleared first either).  blocks_ok: whole blocks decoded.
 * Where the chunk fails is amvo_decode_frame's rule, as in the plain mode.  RESTATEMENT ONLY: no reference build pins it. */
int amvo_decode_frame_ffmpeg_keep(const uint8_t *chunk, uint32_t len, uint32_t w, uint32_t h,
                                  uint8_t *out, uint32_t *blocks_ok, uint32_t *status)
{
    return decode_frame_ffmpeg(chunk, len, w, h, out, blocks_ok, status, 1);
}
static int decode_frame_ffmpeg(const uint8_t *chunk, uint32_t len, uint32_t w, uint32_t h,
                               uint8_t *out, uint32_t *nmcu_ok, uint32_t *status, int keep)
{
    hufbounds hb[4];
    bitrd b;
    const uint32_t mcw = amvo_mcus_per_row(w), mch = amvo_mcu_rows(h);
    const uint32_t cw = (w + 1) / 2, chh = (h + 1) / 2;
    uint8_t *plane[3] = { out, out + (size_t)w * h, out + (size_t)w * h + (size_t)cw * chh };
    const uint32_t pw[3] = { w, cw, cw }, ph[3] = { h, chh, chh };
    int16_t pred[3] = { 0, 0, 0 };
    uint32_t st = 0, mcu = 0, blocks = 0;
    static const int comp_of[6] = { 0, 0, 0, 0, 1, 2 };

    for (int t = 0; t < 4; t++) build_bounds(&hb[t], k_bits[t]);
    if (!keep) memset(out, 0, amvo_yuv420_frame_bytes(w, h));
    memset(&b, 0, sizeof b);
    b.buf = chunk; b.len = len; b.pos = 2;                                    /* sp5xdec.c:75-77 copies [2, n-2) */

    for (uint32_t my = 0; my < mch && !st; my++) {
        for (uint32_t mx = 0; mx < mcw; mx++) {
            int16_t mcub[6][64];
            int good = 0;                                                      /* whole blocks of this MCU */
            for (int k = 0; k < 6 && !st; k++) {
                int c = comp_of[k];
                int r = huf_block(&b, hb, c ? 1 : 0, c ? 3 : 2, mcub[k]);
                if (r) { st |= (uint32_t)r; break; }
                mcub[k][0] = (int16_t)(mcub[k][0] + pred[c]);
                pred[c] = mcub[k][0];
                good++;
            }
            blocks += (uint32_t)good;
            if (st && !keep) break;
            for (int k = 0; k < good; k++) {
                int c = comp_of[k];
                int v = c == 0 ? 2 : 1;
                int16_t blk[64];
                uint8_t px[64];
                amvo_ffmpeg_dequant_block(mcub[k], c, blk);
                amvo_simple_idct_put(px, 8, blk);
                /* block origin in scan coordinates (mjpegdec.c:708-710) */
                int32_t sy = (int32_t)((c == 0 ? 2 * my + (uint32_t)(k >> 1) : my) * 8);
                uint32_t sx = (c == 0 ? 2 * mx + (uint32_t)(k & 1) : mx) * 8;
                int32_t start = v * (int32_t)(8 * mch - ((h / 2) & 7)) - 1;   /* :675 */
                for (int i = 0; i < 8; i++) {
                    int32_t p = start - (sy + i);
                    if (p < 0 || p >= (int32_t)ph[c]) continue;
                    for (uint32_t j = 0; j < 8; j++)
                        if (sx + j < pw[c]) plane[c][(size_t)p * pw[c] + sx + j] = px[8 * i + j];
                }
            }
            if (st) break;
            mcu++;
        }
    }
    if (b.consumed > b.valid) st |= AMVO_ST_TRUNCATED;
    if (nmcu_ok) *nmcu_ok = keep ? blocks : mcu;
    if (status) *status = st;
    return st ? -1 : 0;
}

/* ------------------------------------------------------------------------------------
 * picture rescale: img_resample (lavc/imgresample.c:340-505), the routine behind the sws_scale shim
 * (:515-660) that ffmpeg.c:757 calls before amv_encode_picture when the source is not 160x120.
 * Four-tap polyphase filter, 16 phases, 16.16 positions; every line is filtered horizontally into
 * bytes (clipped), then four such lines vertically (clipped again).
 * ---------------------------------------------------------------------------------- */
#define RS_PHASE_BITS 4
#define RS_TAPS 4
#define RS_POS_BITS 16
#define RS_FILTER_BITS 8

/* av_build_filter(filter, factor, 4, 16, 256, 0) -- lavc/resample2.c:93-140, type 0 = cubic, d = -0.5 */
void amvo_build_resample_filter(int16_t filter[16 * 4], int out_size, int in_size)
{
    double factor = (float)out_size / (float)in_size;               /* imgresample.c:468-471 */
    const int center = (RS_TAPS - 1) / 2;
    if (factor > 1.0) factor = 1.0;                                  /* resample2.c:100-101 */
    for (int ph = 0; ph < 16; ph++) {
        double tab[RS_TAPS], norm = 0;
        for (int i = 0; i < RS_TAPS; i++) {
            const float d = -0.5f;
            double x = fabs(((double)(i - center) - (double)ph / 16) * factor), y;
            if (x < 1.0) y = 1 - 3 * x * x + 2 * x * x * x + d * (-x * x + x * x * x);
            else y = d * (-4 + 8 * x - 5 * x * x + x * x * x);
            tab[i] = y;
            norm += y;
        }
        for (int i = 0; i < RS_TAPS; i++) {
            long v = lrintf((float)(tab[i] * 256 / norm));           /* :137 */
            filter[ph * RS_TAPS + i] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
        }
    }
}

static inline int rs_phase(int pos) { return (pos >> (RS_POS_BITS - RS_PHASE_BITS)) & 15; }
static inline uint8_t rs_clip(int sum) { sum >>= RS_FILTER_BITS; return (uint8_t)(sum < 0 ? 0 : (sum > 255 ? 255 : sum)); }

/* one output sample of h_resample (:312-339; the slow form :288-310 is the general one, the fast forms are the
 * same sums where no clamping is needed) */
static uint8_t rs_h(const uint8_t *src, int iw, int x, int h_incr, const int16_t *hf)
{
    const int pos = -(1 << RS_POS_BITS) + x * h_incr;               /* src_start = -FCENTER * POS_FRAC, :367 */
    const int s0 = pos >> RS_POS_BITS;
    const int16_t *f = hf + rs_phase(pos) * RS_TAPS;
    int sum = 0;
    for (int j = 0; j < RS_TAPS; j++) {
        int s = s0 + j;
        s = s < 0 ? 0 : (s >= iw ? iw - 1 : s);
        sum += src[s] * f[j];
    }
    return rs_clip(sum);
}

/* component_resample (:341-405) */
void amvo_resample_plane(const uint8_t *in, int iwrap, int iw, int ih, uint8_t *out, int owrap, int ow, int oh,
                         int h_incr, int v_incr, const int16_t *hf, const int16_t *vf)
{
    for (int y = 0; y < oh; y++) {
        const int src_y = 2 * (1 << RS_POS_BITS) + y * v_incr;      /* (last_src_y + NB_TAPS) * POS_FRAC, :350 */
        const int y1 = src_y >> RS_POS_BITS;
        const int16_t *f = vf + rs_phase(src_y) * RS_TAPS;
        for (int x = 0; x < ow; x++) {
            int sum = 0;
            for (int j = 0; j < RS_TAPS; j++) {
                int line = y1 - 3 + j;                               /* the ring's last four lines, :400 */
                line = line < 0 ? 0 : (line >= ih ? ih - 1 : line);  /* :361-366 */
                sum += rs_h(in + (size_t)line * iwrap, iw, x, h_incr, hf) * f[j];
            }
            out[(size_t)y * owrap + x] = rs_clip(sum);
        }
    }
}

/* img_resample (:474-495) on tight YUV420P planes: Y iw x ih, then Cb, Cr of (iw >> 1) x (ih >> 1) */
void amvo_img_resample_yuv420(const uint8_t *in, int iw, int ih, uint8_t *out, int ow, int oh)
{
    int16_t hf[64], vf[64];
    const int h_incr = (iw * (1 << RS_POS_BITS)) / ow, v_incr = (ih * (1 << RS_POS_BITS)) / oh;   /* :465-466 */
    amvo_build_resample_filter(hf, ow, iw);
    amvo_build_resample_filter(vf, oh, ih);
    const size_t iy = (size_t)iw * ih, ic = (size_t)(iw >> 1) * (ih >> 1);
    const size_t oy = (size_t)ow * oh, oc = (size_t)(ow >> 1) * (oh >> 1);
    amvo_resample_plane(in, iw, iw, ih, out, ow, ow, oh, h_incr, v_incr, hf, vf);
    for (int c = 0; c < 2; c++)
        amvo_resample_plane(in + iy + c * ic, iw >> 1, iw >> 1, ih >> 1, out + oy + c * oc, ow >> 1, ow >> 1, oh >> 1,
                            h_incr, v_incr, hf, vf);
}

/* ------------------------------------------------------------------------------------
 * IMA ADPCM
 * ---------------------------------------------------------------------------------- */
static const int8_t k_index_table[16] = { -1, -1, -1, -1, 2, 4, 6, 8, -1, -1, -1, -1, 2, 4, 6, 8 }; /* AdpcmIma.c:20-23 */
static const int16_t k_step_table[89] = {                                                          /* AdpcmIma.c:29-39 */
    7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 21, 23, 25, 28, 31, 34, 37, 41, 45,
    50, 55, 60, 66, 73, 80, 88, 97, 107, 118, 130, 143, 157, 173, 190, 209, 230, 253, 279, 307,
    337, 371, 408, 449, 494, 544, 598, 658, 724, 796, 876, 963, 1060, 1166, 1282, 1411, 1552, 1707, 1878, 2066,
    2272, 2499, 2749, 3024, 3327, 3660, 4026, 4428, 4871, 5358, 5894, 6484, 7132, 7845, 8630, 9493, 10442, 11487, 12635, 13899,
    15289, 16818, 18500, 20350, 22385, 24623, 27086, 29794, 32767 };

static inline int clip_s16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
static inline int clip_idx(int v) { return v < 0 ? 0 : (v > 88 ? 88 : v); }

/* AdpcmImaExpandNibble AdpcmIma.c:170-204 (shift = 3) */
static int16_t expand_nibble(int *predictor, int *step_index, unsigned nibble)
{
    int step = k_step_table[*step_index];
    int idx = clip_idx(*step_index + k_index_table[nibble]);
    int diff = ((2 * (int)(nibble & 7) + 1) * step) >> 3;
    int p = *predictor;
    p = (nibble & 8) ? p - diff : p + diff;
    p = clip_s16(p);
    *predictor = p;
    *step_index = idx;
    return (int16_t)p;
}

int amvo_adpcm_decode_chunk(const uint8_t *chunk, uint32_t len, int16_t *pcm, uint32_t *nsamples_hdr)
{
    if (len <= 8) return -1;                                        /* AdpcmIma.c:216 (!buf_size) */
    int predictor = (int16_t)(chunk[0] | (chunk[1] << 8));          /* AMVDec.c:312 */
    int step_index = clip_idx(chunk[2]);                            /* AMVDec.c:313; >88 is out of bounds in the reference */
    if (nsamples_hdr)
        *nsamples_hdr = (uint32_t)chunk[4] | ((uint32_t)chunk[5] << 8) | ((uint32_t)chunk[6] << 16) | ((uint32_t)chunk[7] << 24);
    uint32_t n = len - 8;
    for (uint32_t i = 0; i < n; i++) {                              /* AdpcmIma.c:225-237, mono: high nibble first */
        unsigned byte = chunk[8 + i];
        *pcm++ = expand_nibble(&predictor, &step_index, byte >> 4);
        *pcm++ = expand_nibble(&predictor, &step_index, byte & 15);
    }
    return (int)(2 * n);
}

/* adpcm_ima_compress_sample lavc/adpcm.c:219-227 (yamaha_difflookup :124-127) */
static unsigned compress_sample(int *prev, int *step_index, int sample)
{
    int delta = sample - *prev;
    int step = k_step_table[*step_index];
    int q = abs(delta) * 4 / step;
    unsigned nibble = (unsigned)(q > 7 ? 7 : q) + (delta < 0 ? 8u : 0u);
    int look = 2 * (int)(nibble & 7) + 1;
    if (nibble & 8) look = -look;
    *prev = clip_s16(*prev + (step * look) / 8);
    *step_index = clip_idx(*step_index + k_index_table[nibble]);
    return nibble;
}

int amvo_adpcm_encode_chunk(const int16_t *samples, uint32_t nsamp, int *step_index, uint8_t *out)
{
    uint8_t *dst = out;
    int prev = samples[0];                                          /* adpcm.c:464 */
    uint32_t n = nsamp >> 1;
    *dst++ = (uint8_t)(prev & 0xff); *dst++ = (uint8_t)((prev >> 8) & 0xff);           /* :465 */
    *dst++ = (uint8_t)(*step_index & 0xff); *dst++ = (uint8_t)((*step_index >> 8) & 0xff); /* :466 */
    *dst++ = (uint8_t)((n << 1) & 0xff); *dst++ = (uint8_t)(((n << 1) >> 8) & 0xff);   /* :479 */
    *dst++ = (uint8_t)(((n << 1) >> 16) & 0xff); *dst++ = (uint8_t)(((n << 1) >> 24) & 0xff);
    for (uint32_t i = 0; i < n; i++) {                              /* :489-493 */
        unsigned hi = compress_sample(&prev, step_index, samples[2 * i]);
        unsigned lo = compress_sample(&prev, step_index, samples[2 * i + 1]);
        *dst++ = (uint8_t)((hi << 4) | (lo & 15));
    }
    return (int)(dst - out);
}

/* adpcm_compress_trellis lavc/adpcm.c:287-443, the IMA branch (LOOP_NODES :373-385, STORE_NODE :336-369), as the
 * AMV case of adpcm_encode_frame calls it (:482-487): a beam search over the 2^trellis best decoder states, frozen
 * into the output every 128 samples.  nodes[] is kept sorted by error; a candidate that decodes to a sample value some
 * node of the new frontier already has is dropped.  Unpinned restatement: adpcm.c's encoders sit behind
 * CONFIG_ENCODERS, which only ./configure defines. */
#define TRELLIS_FREEZE 128
#define TRELLIS_MAX 5
typedef struct { uint32_t ssd; int path, sample1, sample2, step; } trellis_node;
typedef struct { int nibble, prev; } trellis_path;

static void compress_trellis(const int16_t *samples, uint8_t *dst, int n, int trellis, int *prev_sample, int *step_index)
{
    const int frontier = 1 << trellis;
    static const int difflookup[16] = { 1, 3, 5, 7, 9, 11, 13, 15, -1, -3, -5, -7, -9, -11, -13, -15 };   /* :124-127 */
    trellis_path *paths = malloc(sizeof(trellis_path) * (size_t)frontier * TRELLIS_FREEZE);
    trellis_node node_buf[2][1 << TRELLIS_MAX];
    trellis_node *nodep_buf[2][1 << TRELLIS_MAX];
    trellis_node **nodes = nodep_buf[0], **nodes_next = nodep_buf[1];
    int pathn = 0, froze = -1;
    memset(nodep_buf, 0, sizeof nodep_buf);
    nodes[0] = &node_buf[1][0];                                     /* :309-316 */
    nodes[0]->ssd = 0; nodes[0]->path = 0; nodes[0]->step = *step_index;
    nodes[0]->sample1 = *prev_sample; nodes[0]->sample2 = 0;
    for (int i = 0; i < n; i++) {
        trellis_node *t = node_buf[i & 1];
        const int sample = samples[i];
        memset(nodes_next, 0, (size_t)frontier * sizeof(trellis_node *));
        for (int j = 0; j < frontier && nodes[j]; j++) {
            const int range = j < frontier / 2 ? 1 : 0;            /* :333 */
            const int step = nodes[j]->step, st = k_step_table[step];
            const int predictor = nodes[j]->sample1;
            const int div = (sample - predictor) * 4 / st;          /* :376 */
            int nmin = div - range, nmax = div + range;
            nmin = nmin < -7 ? -7 : (nmin > 6 ? 6 : nmin);
            nmax = nmax < -6 ? -6 : (nmax > 7 ? 7 : nmax);
            if (nmin <= 0) nmin--;                                  /* distinguish -0 from +0 */
            if (nmax < 0) nmax--;
            for (int nidx = nmin; nidx <= nmax; nidx++) {
                const int nibble = nidx < 0 ? 7 - nidx : nidx;
                int dec = clip_s16(predictor + (st * difflookup[nibble]) / 8);
                const int d = sample - dec;
                const uint32_t ssd = nodes[j]->ssd + (uint32_t)(d * d);
                int k, dup = 0;
                if (nodes_next[frontier - 1] && ssd >= nodes_next[frontier - 1]->ssd) continue;   /* :342 */
                for (k = 0; k < frontier && nodes_next[k]; k++)     /* collapse equal previous samples, :347-352 */
                    if (dec == nodes_next[k]->sample1) { dup = 1; break; }
                if (dup) continue;
                for (k = 0; k < frontier; k++) {
                    if (!nodes_next[k] || ssd < nodes_next[k]->ssd) {
                        trellis_node *u = nodes_next[frontier - 1];
                        if (!u) { u = t++; u->path = pathn++; }
                        u->ssd = ssd;
                        u->step = clip_idx(step + k_index_table[nibble]);
                        u->sample2 = nodes[j]->sample1;
                        u->sample1 = dec;
                        paths[u->path].nibble = nibble;
                        paths[u->path].prev = nodes[j]->path;
                        memmove(&nodes_next[k + 1], &nodes_next[k], (size_t)(frontier - k - 1) * sizeof(trellis_node *));
                        nodes_next[k] = u;
                        break;
                    }
                }
            }
        }
        { trellis_node **u = nodes; nodes = nodes_next; nodes_next = u; }
        if (nodes[0]->ssd > (1u << 28)) {                          /* :398-402 */
            for (int j = 1; j < frontier && nodes[j]; j++) nodes[j]->ssd -= nodes[0]->ssd;
            nodes[0]->ssd = 0;
        }
        if (i == froze + TRELLIS_FREEZE) {                         /* :405-417 */
            const trellis_path *p = &paths[nodes[0]->path];
            for (int k = i; k > froze; k--) { dst[k] = (uint8_t)p->nibble; p = &paths[p->prev]; }
            froze = i;
            pathn = 0;
            memset(nodes + 1, 0, (size_t)(frontier - 1) * sizeof(trellis_node *));
        }
    }
    {
        const trellis_path *p = &paths[nodes[0]->path];
        for (int i = n - 1; i > froze; i--) { dst[i] = (uint8_t)p->nibble; p = &paths[p->prev]; }
    }
    *prev_sample = nodes[0]->sample1;                               /* :426-431 (prev_sample itself is reset per frame, :464) */
    *step_index = nodes[0]->step;
    free(paths);
}

/* the AMV case of adpcm_encode_frame with avctx->trellis > 0 (adpcm.c:461-487): same header, nibbles from the search */
int amvo_adpcm_encode_chunk_trellis(const int16_t *samples, uint32_t nsamp, int *step_index, int trellis, uint8_t *out)
{
    uint8_t *dst = out;
    int prev = samples[0];
    const uint32_t n = nsamp >> 1;
    uint8_t *nib;
    if (trellis < 1 || trellis > TRELLIS_MAX) return -1;
    nib = malloc(2 * (size_t)n + 1);
    *dst++ = (uint8_t)(prev & 0xff); *dst++ = (uint8_t)((prev >> 8) & 0xff);
    *dst++ = (uint8_t)(*step_index & 0xff); *dst++ = (uint8_t)((*step_index >> 8) & 0xff);
    *dst++ = (uint8_t)((n << 1) & 0xff); *dst++ = (uint8_t)(((n << 1) >> 8) & 0xff);
    *dst++ = (uint8_t)(((n << 1) >> 16) & 0xff); *dst++ = (uint8_t)(((n << 1) >> 24) & 0xff);
    if (n) compress_trellis(samples, nib, (int)(2 * n), trellis, &prev, step_index);
    for (uint32_t i = 0; i < n; i++) *dst++ = (uint8_t)((nib[2 * i] << 4) | nib[2 * i + 1]);   /* :485-486 */
    free(nib);
    return (int)(dst - out);
}

/* AdpcmImaCompressSample AdpcmIma.c:43-89: the quotient goes through an unsigned char before it
 * is limited to 7 (:62-65) and the predicted delta uses the already updated step (:73) */
static unsigned wav_compress(int *prev, int *step_index, int sample)
{
    int delta = sample - *prev, sign = 0, idx = *step_index, pd;
    unsigned char nibble;
    if (delta < 0) { sign = 1; delta = -delta; }
    nibble = (unsigned char)((delta << 2) / k_step_table[clip_idx(idx)]);
    if (nibble > 7) nibble = 7;
    idx = clip_idx(idx + k_index_table[nibble]);
    pd = (k_step_table[idx] * nibble) / 4 + k_step_table[idx] / 8;
    *prev = clip_s16(sign ? *prev - pd : *prev + pd);
    *step_index = idx;
    return (unsigned)nibble + ((unsigned)sign << 3);
}

int amvo_adpcm_wav_encode_frame(const int16_t *samples, int frame_size, int32_t state[2], uint8_t *frame)
{
    uint8_t *dst = frame;                                           /* AdpcmIma.c:92-160, mono */
    int n = frame_size / 8, prev = samples[0], idx = state[1];
    *dst++ = (uint8_t)(prev & 0xff); *dst++ = (uint8_t)((prev >> 8) & 0xff);
    *dst++ = (uint8_t)idx; *dst++ = 0;
    samples++;
    for (; n > 0; n--, samples += 8)
        for (int k = 0; k < 4; k++) {
            unsigned lo = wav_compress(&prev, &idx, samples[2 * k]) & 0x0f;
            unsigned hi = wav_compress(&prev, &idx, samples[2 * k + 1]);
            *dst++ = (uint8_t)(lo | ((hi << 4) & 0xf0));
        }
    state[0] = prev; state[1] = idx;
    return (int)(dst - frame);
}

uint32_t amvo_adpcm_amv_pairs(uint32_t frame_size, uint32_t sample_rate, uint32_t *extra, uint64_t *samples_written)
{
    uint32_t n = frame_size >> 1;                                   /* adpcm.c:469-472 */
    *extra += frame_size & 1;
    n += *extra >> 1;
    *extra &= 1;
    uint32_t i = (uint32_t)((*samples_written + 2ull * n) % sample_rate);   /* :474 */
    if (i && i + frame_size > sample_rate) n += (sample_rate - i) >> 1;    /* :476-477 */
    *samples_written += 2ull * n;                                   /* :495 */
    return n;
}

/* ------------------------------------------------------------------------------------
 * encoder front end: rgb24_to_yuvj420p lavc/imgconvert_template.h:654-, colorspace.h:30-97
 * ---------------------------------------------------------------------------------- */
#define SCALEBITS 10
#define ONE_HALF (1 << (SCALEBITS - 1))
#define FIXC(x) ((int)((x) * (1 << SCALEBITS) + 0.5))

void amvo_rgb24_to_yuvj420p(const uint8_t *rgb, uint32_t src_stride, uint32_t w, uint32_t h,
                            int bgr, uint8_t *yp, uint8_t *cb, uint8_t *cr)
{
    const int ro = bgr ? 2 : 0, bo = bgr ? 0 : 2;
    for (uint32_t y = 0; y < h; y += 2) {
        for (uint32_t x = 0; x < w; x += 2) {
            int r1 = 0, g1 = 0, b1 = 0;
            for (int dy = 0; dy < 2; dy++)
                for (int dx = 0; dx < 2; dx++) {
                    const uint8_t *p = rgb + (size_t)(y + dy) * src_stride + (size_t)(x + dx) * 3;
                    int r = p[ro], g = p[1], b = p[bo];
                    r1 += r; g1 += g; b1 += b;
                    yp[(size_t)(y + dy) * w + x + dx] = (uint8_t)((FIXC(0.29900) * r + FIXC(0.58700) * g +
                                                                   FIXC(0.11400) * b + ONE_HALF) >> SCALEBITS);
                }
            cb[(size_t)(y / 2) * (w / 2) + x / 2] = (uint8_t)(((-FIXC(0.16874) * r1 - FIXC(0.33126) * g1 + FIXC(0.50000) * b1 +
                                                                (ONE_HALF << 2) - 1) >> (SCALEBITS + 2)) + 128);
            cr[(size_t)(y / 2) * (w / 2) + x / 2] = (uint8_t)(((FIXC(0.50000) * r1 - FIXC(0.41869) * g1 - FIXC(0.08131) * b1 +
                                                                (ONE_HALF << 2) - 1) >> (SCALEBITS + 2)) + 128);
        }
    }
}

/* ------------------------------------------------------------------------------------
 * forward DCT: ff_jpeg_fdct_islow lavc/jfdctint.c:184-343 (CONST_BITS 13, PASS1_BITS 4)
 * ---------------------------------------------------------------------------------- */
#define CONST_BITS 13
#define PASS1_BITS 4
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172

static void fdct_1d(int16_t *d, int stride, int pass)
{
    int32_t t0 = d[0] + d[7 * stride], t7 = d[0] - d[7 * stride];
    int32_t t1 = d[stride] + d[6 * stride], t6 = d[stride] - d[6 * stride];
    int32_t t2 = d[2 * stride] + d[5 * stride], t5 = d[2 * stride] - d[5 * stride];
    int32_t t3 = d[3 * stride] + d[4 * stride], t4 = d[3 * stride] - d[4 * stride];
    int32_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    int32_t z1, z2, z3, z4, z5;
    const int sh = pass == 0 ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS;

    if (pass == 0) {                                         /* jfdctint.c:214-215 */
        d[0] = (int16_t)((t10 + t11) << PASS1_BITS);
        d[4 * stride] = (int16_t)((t10 - t11) << PASS1_BITS);
    } else {                                                 /* :296-297 */
        d[0] = (int16_t)DESCALE(t10 + t11, PASS1_BITS);
        d[4 * stride] = (int16_t)DESCALE(t10 - t11, PASS1_BITS);
    }
    z1 = (t12 + t13) * FIX_0_541196100;
    d[2 * stride] = (int16_t)DESCALE(z1 + t13 * FIX_0_765366865, sh);
    d[6 * stride] = (int16_t)DESCALE(z1 + t12 * -FIX_1_847759065, sh);

    z1 = t4 + t7; z2 = t5 + t6; z3 = t4 + t6; z4 = t5 + t7;
    z5 = (z3 + z4) * FIX_1_175875602;
    t4 *= FIX_0_298631336; t5 *= FIX_2_053119869; t6 *= FIX_3_072711026; t7 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
    z3 += z5; z4 += z5;
    d[7 * stride] = (int16_t)DESCALE(t4 + z1 + z3, sh);
    d[5 * stride] = (int16_t)DESCALE(t5 + z2 + z4, sh);
    d[3 * stride] = (int16_t)DESCALE(t6 + z2 + z3, sh);
    d[1 * stride] = (int16_t)DESCALE(t7 + z1 + z4, sh);
}

void amvo_fdct_islow(int16_t blk[64])
{
    for (int r = 0; r < 8; r++) fdct_1d(blk + 8 * r, 1, 0);   /* row_fdct :184-258 */
    for (int c = 0; c < 8; c++) fdct_1d(blk + c, 8, 1);       /* pass 2 :273-341 */
}

/* ------------------------------------------------------------------------------------
 * quantiser: dct_quantize_c lavc/mpegvideo_enc.c:3647-3724 for AMV (bias 0, :492-496),
 * qmat = (1<<QMAT_SHIFT)/(qscale*Q) with qscale 8 (:80-91, :2866-2877), QMAT_SHIFT 22
 * (mpegvideo.h:51).  Differences, required to be decodable by amvlib:
 * amvlib's fixed tables instead of mpeg1_default_intra*qscale>>3, and a true level
 * shift, so DC may be negative and is rounded symmetrically.
 * ---------------------------------------------------------------------------------- */
#define QMAT_SHIFT 22

void amvo_quantize_block(const int16_t dct[64], int comp, uint32_t qbias, int16_t zz[64])
{
    const uint8_t *qt = comp == 0 ? k_qt_luma : k_qt_chroma;
    const int32_t bias = (int32_t)(qbias << (QMAT_SHIFT - 8));      /* :3679, QUANT_BIAS_SHIFT 8 */
    for (int nat = 0; nat < 64; nat++) {
        int i = k_zigzag_std[nat];
        int32_t c = dct[nat];
        if (i == 0) {
            int32_t q = (int32_t)qt[0] << 3;                       /* :3670-3676: (b + q/2) / q */
            int32_t a = (abs(c) + (q >> 1)) / q;
            zz[0] = (int16_t)(c < 0 ? -a : a);
        } else {
            int32_t qmat = (int32_t)((1u << QMAT_SHIFT) / (8u * qt[i]));
            int32_t level = c * qmat;                              /* :3702 */
            int32_t a = ((level < 0 ? -level : level) + bias) >> QMAT_SHIFT; /* :3706-3712 */
            zz[i] = (int16_t)(level < 0 ? -a : a);
        }
    }
}

/* ------------------------------------------------------------------------------------
 * entropy coder: encode_block / ff_mjpeg_encode_dc lavc/mjpegenc.c:357-435,
 * code assignment ff_mjpeg_build_huffman_codes lavc/mjpeg.c:129-147
 * ---------------------------------------------------------------------------------- */
typedef struct { uint8_t size[256]; uint16_t code[256]; } hufenc;

static void build_enc(hufenc *he, const uint8_t bits[16], const uint8_t *vals)
{
    int code = 0, k = 0;
    memset(he, 0, sizeof *he);
    for (int i = 1; i <= 16; i++) {
        for (int j = 0; j < bits[i - 1]; j++) {
            int sym = vals[k++];
            he->size[sym] = (uint8_t)i;
            he->code[sym] = (uint16_t)code;
            code++;
        }
        code <<= 1;
    }
}

typedef struct { uint8_t *buf; size_t pos; uint32_t acc; int nacc; } bitwr;

static void put_bits(bitwr *bw, int n, uint32_t v)
{
    for (int i = n - 1; i >= 0; i--) {
        bw->acc = (bw->acc << 1) | ((v >> i) & 1u);
        if (++bw->nacc == 8) { bw->buf[bw->pos++] = (uint8_t)bw->acc; bw->acc = 0; bw->nacc = 0; }
    }
}

static int nbits_of(int v) { int n = 0; while (v) { n++; v >>= 1; } return n; }

static void encode_coef(bitwr *bw, const hufenc *he, int run, int val)
{
    int mant = val, nb;
    if (val < 0) { val = -val; mant--; }                      /* mjpegenc.c:366-369,414-417 */
    nb = nbits_of(val);
    put_bits(bw, he->size[(run << 4) | nb], he->code[(run << 4) | nb]);
    put_bits(bw, nb, (uint32_t)mant & ((1u << nb) - 1u));
}

static void encode_block(bitwr *bw, const hufenc *dc, const hufenc *ac, const int16_t zz[64], int16_t *pred)
{
    int diff = zz[0] - *pred, run = 0, last = 0;             /* :390-401 */
    *pred = zz[0];
    if (diff == 0) put_bits(bw, dc->size[0], dc->code[0]);    /* :362-363 */
    else encode_coef(bw, dc, 0, diff);
    for (int i = 63; i > 0; i--) if (zz[i]) { last = i; break; }
    for (int i = 1; i <= last; i++) {                         /* :405-427 */
        int v = zz[i];
        if (v == 0) { run++; continue; }
        while (run >= 16) { put_bits(bw, ac->size[0xf0], ac->code[0xf0]); run -= 16; }
        encode_coef(bw, ac, run, v);
        run = 0;
    }
    if (last < 63) put_bits(bw, ac->size[0], ac->code[0]);    /* :430-431 */
}

uint32_t amvo_encode_bound(uint32_t w, uint32_t h)
{
    /* worst case per coefficient: 16-bit code + 11 magnitude bits, doubled by FF escaping */
    return 4 + amvo_mcus_per_row(w) * amvo_mcu_rows(h) * 6 * 64 * 4 * 2;
}

/* planes (tight rows: Y w x h, Cb / Cr w/2 x h/2) -> chunk: what amv_encode_picture does with the picture it is given
 * (mjpegenc.c:454-472 flip, :379-450 blocks), behind whatever produced the planes */
static int encode_planes(const uint8_t *yp, const uint8_t *cb, const uint8_t *cr, uint32_t w, uint32_t h, uint32_t qbias, uint8_t *out,
                         int16_t *coef_out)
{
    const uint32_t mcw = amvo_mcus_per_row(w), mch = amvo_mcu_rows(h);
    const uint32_t cw = w / 2, ch = h / 2;
    uint8_t *raw = (uint8_t *)malloc(amvo_encode_bound(w, h));
    hufenc he[4];
    bitwr bw = { raw, 0, 0, 0 };
    int16_t pred[3] = { 0, 0, 0 };
    uint32_t mcu = 0;

    for (int t = 0; t < 4; t++) build_enc(&he[t], k_bits[t], k_vals[t]);
    for (uint32_t my = 0; my < mch; my++)
        for (uint32_t mx = 0; mx < mcw; mx++, mcu++)
            for (int k = 0; k < 6; k++) {                     /* block order Y0..Y3,Cb,Cr mjpegenc.c:437-450 */
                int16_t blk[64], zz[64];
                const uint8_t *plane = k < 4 ? yp : (k == 4 ? cb : cr);
                const uint32_t pw = k < 4 ? w : cw, ph = k < 4 ? h : ch;
                const uint32_t bx = k < 4 ? mx * 16 + (k & 1) * 8 : mx * 8;
                const uint32_t by = k < 4 ? my * 16 + (k >> 1) * 8 : my * 8;
                for (uint32_t i = 0; i < 8; i++)
                    for (uint32_t j = 0; j < 8; j++) {
                        /* the bitstream holds the picture bottom-up (amv_encode_picture mjpegenc.c:454-472);
                         * rows/columns beyond the picture replicate the nearest edge sample */
                        uint32_t r = by + i, c = bx + j;
                        uint32_t sy = r < ph ? ph - 1 - r : 0;
                        uint32_t sx = c < pw ? c : pw - 1;
                        blk[i * 8 + j] = (int16_t)((int)plane[(size_t)sy * pw + sx] - 128);
                    }
                amvo_fdct_islow(blk);
                amvo_quantize_block(blk, k < 4 ? 0 : 1, qbias, zz);
                if (coef_out) memcpy(coef_out + ((size_t)mcu * 6 + k) * 64, zz, sizeof zz);
                encode_block(&bw, k < 4 ? &he[0] : &he[1], k < 4 ? &he[2] : &he[3], zz, &pred[k < 4 ? 0 : k - 3]);
            }
    if (bw.nacc) put_bits(&bw, 8 - bw.nacc, (1u << (8 - bw.nacc)) - 1u);  /* ff_mjpeg_encode_stuffing :338-343 */

    size_t o = 0;
    out[o++] = 0xff; out[o++] = 0xd8;                          /* SOI only, :201-204 */
    for (size_t i = 0; i < bw.pos; i++) {                      /* escape_FF :282-336 */
        out[o++] = raw[i];
        if (raw[i] == 0xff) out[o++] = 0;
    }
    out[o++] = 0xff; out[o++] = 0xd9;                          /* EOI :354 */
    free(raw);
    return (int)o;
}

int amvo_encode_frame(const uint8_t *src, uint32_t src_stride, uint32_t w, uint32_t h, int bgr,
                      uint32_t qbias, uint8_t *out, int16_t *coef_out)
{
    if (w == 0 || h == 0 || (w & 1) || (h & 1)) return -1;
    const uint32_t cw = w / 2, ch = h / 2;
    uint8_t *yp = (uint8_t *)malloc((size_t)w * h + 2 * (size_t)cw * ch);
    uint8_t *cb = yp + (size_t)w * h, *cr = cb + (size_t)cw * ch;
    amvo_rgb24_to_yuvj420p(src, src_stride, w, h, bgr, yp, cb, cr);
    const int n = encode_planes(yp, cb, cr, w, h, qbias, out, coef_out);
    free(yp);
    return n;
}

/* The picture amv_encoder is handed as PIX_FMT_YUVJ420P (mjpegenc.c:493 pix_fmts, get_pixels mpegvideo_enc.c:1539-1549):
 * planes with the caller's line sizes, no colour conversion. */
int amvo_encode_frame_yuv420(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, uint32_t y_stride, uint32_t c_stride,
                             uint32_t w, uint32_t h, uint32_t qbias, uint8_t *out)
{
    if (w == 0 || h == 0 || (w & 1) || (h & 1)) return -1;
    const uint32_t cw = w / 2, ch = h / 2;
    uint8_t *yp = (uint8_t *)malloc((size_t)w * h + 2 * (size_t)cw * ch);
    uint8_t *pb = yp + (size_t)w * h, *pr = pb + (size_t)cw * ch;
    for (uint32_t r = 0; r < h; r++) memcpy(yp + (size_t)r * w, y + (size_t)r * y_stride, w);
    for (uint32_t r = 0; r < ch; r++) {
        memcpy(pb + (size_t)r * cw, cb + (size_t)r * c_stride, cw);
        memcpy(pr + (size_t)r * cw, cr + (size_t)r * c_stride, cw);
    }
    const int n = encode_planes(yp, pb, pr, w, h, qbias, out, NULL);
    free(yp);
    return n;
}

/* PIX_FMT_YUVJ422P, the other format amv_encoder declares (mjpegenc.c:493; chroma planes w/2 x h, mpegvideo_enc.c:534-543).
 * The reference itself would code it as MCUs of EIGHT blocks (ff_mjpeg_encode_mb, mjpegenc.c:437-450: blocks 6 and 7 when
 * the CHROMA_420 test fails) -- a scan no AMV decoder reads: the container has no frame header and both amvlib
 * (AmvJpeg.c:1406-1420) and the reference's own amv decoder (sp5xdec.c:51-91, a fixed 4:2:0 SOF) take six blocks per MCU.
 * The product's rule, restated here: chroma row r of the 4:2:0 picture = ((row 2r + row 2r+1 + 1) >> 1) of the 4:2:2 plane,
 * sample by sample, rounding up (h is even: every row has its pair); the picture is then coded as 4:2:0.  A DEFINITION,
 * not reference behaviour -- there is none that an AMV player could show. */
void amvo_yuv422_to_420(const uint8_t *c422, uint32_t stride422, uint32_t cw, uint32_t h, uint8_t *c420, uint32_t stride420)
{
    for (uint32_t r = 0; r < h / 2; r++)
        for (uint32_t x = 0; x < cw; x++)
            c420[(size_t)r * stride420 + x] =
                (uint8_t)(((uint32_t)c422[(size_t)(2 * r) * stride422 + x] + c422[(size_t)(2 * r + 1) * stride422 + x] + 1u) >> 1);
}

int amvo_encode_frame_yuv422(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, uint32_t y_stride, uint32_t c_stride,
                             uint32_t w, uint32_t h, uint32_t qbias, uint8_t *out)
{
    if (w == 0 || h == 0 || (w & 1) || (h & 1)) return -1;
    const uint32_t cw = w / 2, ch = h / 2;
    uint8_t *pb = (uint8_t *)malloc(2 * (size_t)cw * ch), *pr = pb + (size_t)cw * ch;
    amvo_yuv422_to_420(cb, c_stride, cw, h, pb, cw);
    amvo_yuv422_to_420(cr, c_stride, cw, h, pr, cw);
    const int n = amvo_encode_frame_yuv420(y, pb, pr, y_stride, cw, w, h, qbias, out);
    free(pb);
    return n;
}

/* ------------------------------------------------------------------------------------
 * synthetic sources: integer only, so every platform and the
 * HIP generator produce the same bytes.
 * ---------------------------------------------------------------------------------- */
static const int16_t k_sin_q[65] = { /* round(16384*sin(2*pi*i/256)), i = 0..64 */
    0, 402, 804, 1205, 1606, 2006, 2404, 2801, 3196, 3590, 3981, 4370, 4756, 5139, 5520, 5897,
    6270, 6639, 7005, 7366, 7723, 8076, 8423, 8765, 9102, 9434, 9760, 10080, 10394, 10702, 11003, 11297,
    11585, 11866, 12140, 12406, 12665, 12916, 13160, 13395, 13623, 13842, 14053, 14256, 14449, 14635, 14811, 14978,
    15137, 15286, 15426, 15557, 15679, 15791, 15893, 15986, 16069, 16143, 16207, 16261, 16305, 16340, 16364, 16379,
    16384 };

static inline int32_t isin(uint32_t a) /* a in 1/256 turns */
{
    a &= 255;
    uint32_t q = a & 63;
    switch (a >> 6) {
    case 0: return k_sin_q[q];
    case 1: return k_sin_q[64 - q];
    case 2: return -k_sin_q[q];
    default: return -k_sin_q[64 - q];
    }
}
static inline int32_t icos(uint32_t a) { return isin(a + 64); }

static inline uint32_t mix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
static inline int tri(uint32_t v) { v &= 511; return (int)(v < 256 ? v : 511 - v); }
static inline uint8_t clip_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

void amvo_synth_frame(uint32_t seed, uint32_t t, uint32_t w, uint32_t h, uint8_t *rgb)
{
    const int cx = (int)w / 2, cy = (int)h / 2;
    const int rad = (int)h * 3 / 8, cell = (int)w / 10 > 0 ? (int)w / 10 : 1;
    const int32_t s = isin(t * 2), c = icos(t * 2);
    /* moving gradients are scaled so that their slope in 1/256ths per pixel is the same at every size */
    const uint32_t gx = 512u * 256u / w, gy = 512u * 256u / h;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            int r = 48 + (tri(((x * gx) >> 8) + t * 3) * 5 >> 3);
            int g = 48 + (tri(((y * gy) >> 8) + t * 2) * 5 >> 3);
            int b = 48 + (tri((((x * gx) + (y * gy)) >> 9) + t * 5) * 5 >> 3);
            int dx = (int)x - cx, dy = (int)y - cy;
            if (dx * dx + dy * dy < rad * rad) {            /* rotating checker inside a disc */
                int u = (dx * c + dy * s) >> 14, v = (dy * c - dx * s) >> 14;
                int chk = (((u + 4096) / cell) ^ ((v + 4096) / cell)) & 1;
                r = chk ? 230 - (r >> 3) : 25 + (r >> 3);
                g = chk ? 230 - (g >> 3) : 25 + (g >> 3);
                b = chk ? 230 - (b >> 3) : 25 + (b >> 3);
            }
            uint32_t n = mix32(seed ^ mix32(t * 0x9e3779b9u + y * 65537u + x));
            uint8_t *p = rgb + ((size_t)y * w + x) * 3;    /* 5 % uniform noise: +-12 of 255 */
            p[0] = clip_u8(r + (int)(n % 25u) - 12);
            p[1] = clip_u8(g + (int)((n >> 8) % 25u) - 12);
            p[2] = clip_u8(b + (int)((n >> 16) % 25u) - 12);
        }
}

void amvo_synth_audio(uint32_t seed, uint64_t first, uint32_t n, int16_t *pcm)
{
    for (uint32_t k = 0; k < n; k++) {                     /* three sines + noise, 22050 Hz mono */
        uint64_t i = first + k;
        int32_t v = (6000 * isin((uint32_t)((i * 1301u) >> 8)) +     /* ~437 Hz  */
                     3000 * isin((uint32_t)((i * 3907u) >> 8)) +     /* ~1314 Hz */
                     1500 * isin((uint32_t)((i * 9973u) >> 8))) >> 14; /* ~3355 Hz */
        uint32_t r = mix32(seed ^ mix32((uint32_t)i * 0x85ebca6bu + (uint32_t)(i >> 32)));
        v += (int)(r % 401u) - 200;
        pcm[k] = (int16_t)clip_s16(v);
    }
}

/* ------------------------------------------------------------------------------------
 * helpers
 * ---------------------------------------------------------------------------------- */
uint64_t amvo_fnv1a64(uint64_t hsh, const uint8_t *p, size_t n)
{
    for (size_t i = 0; i < n; i++) { hsh ^= p[i]; hsh *= 0x100000001b3ull; }
    return hsh;
}

double amvo_psnr(const uint8_t *a, const uint8_t *b, size_t n)
{
    double se = 0;
    for (size_t i = 0; i < n; i++) { double d = (double)a[i] - (double)b[i]; se += d * d; }
    if (se == 0) return 99.0;
    return 10.0 * log10(255.0 * 255.0 * (double)n / se);
}

int amvo_decode_batch(const uint8_t *blob, const uint64_t *offs, const uint32_t *lens, uint32_t n,
                      uint32_t w, uint32_t h, uint32_t flags, uint8_t *out, int32_t *status, int threads)
{
    const size_t fsz = (size_t)amvo_stride(w) * h;
    int bad = 0;
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16) reduction(+ : bad)
    for (uint32_t i = 0; i < n; i++) {
        uint32_t st = 0;
        amvo_decode_frame(blob + offs[i], lens[i], w, h, flags, out + fsz * i, NULL, NULL, &st);
        if (status) status[i] = (int32_t)st;
        bad += st != 0;
    }
    return bad;
}

int amvo_synth_encode_batch(uint32_t seed, uint32_t first_frame, uint32_t n, uint32_t w, uint32_t h,
                            uint32_t qbias, uint8_t *blob, uint64_t cap, uint64_t *offs, uint32_t *lens, int threads)
{
    const uint32_t bound = amvo_encode_bound(w, h);
    int fail = 0;
    if (threads < 1) threads = 1;
    uint8_t **tmp = (uint8_t **)calloc(n, sizeof *tmp);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16)
    for (uint32_t i = 0; i < n; i++) {
        uint8_t *rgb = (uint8_t *)malloc((size_t)w * h * 3);
        uint8_t *buf = (uint8_t *)malloc(bound);
        amvo_synth_frame(seed, first_frame + i, w, h, rgb);
        int l = amvo_encode_frame(rgb, w * 3, w, h, 0, qbias, buf, NULL);
        lens[i] = l < 0 ? 0 : (uint32_t)l;
        tmp[i] = (uint8_t *)realloc(buf, lens[i] ? lens[i] : 1);
        free(rgb);
    }
    uint64_t o = 0;
    for (uint32_t i = 0; i < n; i++) {
        offs[i] = o;
        if (o + lens[i] > cap) fail = 1;
        else memcpy(blob + o, tmp[i], lens[i]);
        o += lens[i];
        free(tmp[i]);
    }
    free(tmp);
    return fail ? -1 : 0;
}
