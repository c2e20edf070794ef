// amv_tables.h -- format constants shared by the HIP kernels and the host side of libamvhip.
//
// These are data of the AMV wire format, not code: the fixed quantiser tables of
// C-AMVDecoder/amvlib/AmvJpeg.c:30-39,52-61, the JPEG Annex K.3 Huffman specifications the
// reference hard-codes (AmvJpeg.c:65-131, libavcodec/mjpeg.c:62-127), the zig-zag scan
// (AmvJpeg.c:133-143 with its [3][4] entry; libavcodec/dsputil.c:50-59 without), and the
// IMA step/index tables (amvlib/AdpcmIma.c:20-39, libavcodec/adpcm.c:56-75).
#pragma once
#include <stdint.h>

namespace amv {

// quantiser steps in bitstream (zig-zag) order
static constexpr uint8_t kQuantLuma[64] = {
    8,  6,  6,  7,  6,  5,  8,  7,  7,  7,  9,  9,  8,  10, 12, 20, 13, 12, 11, 11, 12, 25,
    18, 19, 15, 20, 29, 26, 31, 30, 29, 26, 28, 28, 32, 36, 46, 39, 32, 34, 44, 39, 28, 28,
    40, 55, 41, 44, 48, 49, 52, 52, 52, 31, 39, 57, 61, 56, 50, 60, 46, 51, 52, 50};
static constexpr uint8_t kQuantChroma[64] = {
    9,  9,  9,  12, 11, 12, 24, 13, 13, 24, 50, 33, 28, 33, 50, 50, 50, 50, 50, 50, 50, 50,
    50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50,
    50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50, 50};

// the "Q60" tables the patched FFmpeg's AMV decoder dequantises with (libavcodec/sp5x.h:187-194 =
// sp5x_quant_table[10], [11], selected by sp5xdec.c:40,60-61), bitstream order: AMVHIP_FLAG_FFMPEG only
static constexpr uint8_t kQ60Luma[64] = {
    13, 9,  10, 11, 10, 8,  13, 11, 10, 11, 14, 14, 13, 15, 19, 32, 21, 19, 18, 18, 19, 39,
    28, 30, 23, 32, 46, 41, 49, 48, 46, 41, 45, 44, 51, 58, 74, 62, 51, 54, 70, 55, 44, 45,
    64, 87, 65, 70, 76, 78, 82, 83, 82, 50, 62, 90, 97, 90, 80, 96, 74, 81, 82, 79};
static constexpr uint8_t kQ60Chroma[64] = {
    14, 14, 14, 19, 17, 19, 38, 21, 21, 38, 79, 53, 45, 53, 79, 79, 79, 79, 79, 79, 79, 79,
    79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79,
    79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79, 79};

// scan position of each natural (row-major) coefficient position
static constexpr uint8_t kScanOfNatural[64] = {
    0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30,
    41, 43, 9,  11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38,
    46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
// amvlib reads scan position 37 for natural position (3,4); the standard says 31
static constexpr int kAmvlibQuirkNatural = 3 * 8 + 4;
static constexpr int kAmvlibQuirkScan = 37;

// Huffman specifications: table 0 = DC luma, 1 = DC chroma, 2 = AC luma, 3 = AC chroma.
// kHuffCount[t][l-1] = number of codes of length l.
static constexpr uint8_t kHuffCount[4][16] = {
    {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0},
    {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0},
    {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d},
    {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77}};
static constexpr uint8_t kHuffDcSymbols[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static constexpr uint8_t kHuffAcLumaSymbols[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61,
    0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52,
    0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25,
    0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45,
    0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64,
    0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
    0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8,
    0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static constexpr uint8_t kHuffAcChromaSymbols[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61,
    0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33,
    0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18,
    0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44,
    0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63,
    0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
    0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7,
    0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

// IMA ADPCM
static constexpr int8_t kImaIndexAdjust[16] = {-1, -1, -1, -1, 2, 4, 6, 8, -1, -1, -1, -1, 2, 4, 6, 8};
static constexpr int16_t kImaStep[89] = {
    7,     8,     9,     10,    11,    12,    13,    14,    16,    17,    19,    21,    23,
    25,    28,    31,    34,    37,    41,    45,    50,    55,    60,    66,    73,    80,
    88,    97,    107,   118,   130,   143,   157,   173,   190,   209,   230,   253,   279,
    307,   337,   371,   408,   449,   494,   544,   598,   658,   724,   796,   876,   963,
    1060,  1166,  1282,  1411,  1552,  1707,  1878,  2066,  2272,  2499,  2749,  3024,  3327,
    3660,  4026,  4428,  4871,  5358,  5894,  6484,  7132,  7845,  8630,  9493,  10442, 11487,
    12635, 13899, 15289, 16818, 18500, 20350, 22385, 24623, 27086, 29794, 32767};

// quarter wave of the synthetic sources' sine, Q14 (round(16384*sin(2*pi*i/256)), i=0..64)
static constexpr int16_t kSinQ14[65] = {
    0,     402,   804,   1205,  1606,  2006,  2404,  2801,  3196,  3590,  3981,  4370,  4756,
    5139,  5520,  5897,  6270,  6639,  7005,  7366,  7723,  8076,  8423,  8765,  9102,  9434,
    9760,  10080, 10394, 10702, 11003, 11297, 11585, 11866, 12140, 12406, 12665, 12916, 13160,
    13395, 13623, 13842, 14053, 14256, 14449, 14635, 14811, 14978, 15137, 15286, 15426, 15557,
    15679, 15791, 15893, 15986, 16069, 16143, 16207, 16261, 16305, 16340, 16364, 16379, 16384};

// ---- device-side table images built once by the host (amvhip_api.hip) ------------------

// Two-level Huffman decode tables.  Level 1 is indexed by the next 9 bits, level 2 by the
// 7 bits after those.  Entry: bits 0-7 symbol (or level-2 page), bits 8-12 code length,
// bit 15 "go to level 2".  0 = no code with this prefix.
static constexpr int kLut1Bits = 9;
static constexpr int kLut2Bits = 7;
static constexpr int kLut2Pages = 16;
// m1/m2 are the same codes in the form the synchronising kernel wants (amv_decode_sync.hip):
// bits 0-4 code length + magnitude bits (0 = no such code), bits 5-10 how far the coefficient index
// moves (run + 1; 1 for a DC symbol; 63 for end-of-block), bits 11-14 magnitude bits, m1 bit 15 "the
// code is longer than 9 bits".  Long codes occupy the top prefixes of every table (at most the last 5 of
// the 512: they all begin with six one-bits), so m2 is indexed by the 10 bits behind those six.
static constexpr int kLut2PagesPerTable = 5;
static constexpr int kM2Bits = 10;
// fast[t] is table t once more, for the one-lane-per-frame walk (amv_huffman_fast_kernel), as 32-bit entries laid out
// so that a symbol step is arithmetic, not selects: words 0-511 are indexed by the next 9 bits; words 512-1536 by
// max(next 16 bits, 0xfbff) - 0xfbff, i.e. word 512 (always 0) for every stream that does not begin with six one-bits
// and the code's entry for one that does -- the two reads are OR-ed, the entry of a 9-bit prefix that only long codes
// share is 0.  Entry: bits 0-3 magnitude bits (bit 4 is 0: the whole word can serve as a bit-field width operand),
// bit 5 "no such code", bit 8 "carries a value" (a record's stride in the staging area), byte 2 how far the
// coefficient index moves (run + 1; 1 for a DC symbol; 192 for end-of-block: bit 6 of index + 192 is set for every
// index 1..63 and the sum stays clear of 65..79, an over-long run; 0 for "no such code"), byte 3 code length +
// magnitude bits (1 for "no such code": a walk from a guessed start slips one bit there, the strict walk stops).
static constexpr uint32_t kFastWords = 2048;          // per table: 8 KB, a power of two (the table is chosen by OR-ing address bits)
static constexpr uint32_t kFastM2Word = 512;
static constexpr uint32_t kFastLongFirst = 0xfc00u;   // 16-bit windows from here on begin with six one-bits
static constexpr uint32_t kFastInvalid = 1u << 5, kFastEmit = 1u << 8, kFastEobAdvance = 192u;
struct HuffDecodeImage {
    uint16_t l1[4][1 << kLut1Bits];
    uint16_t l2[kLut2Pages][1 << kLut2Bits];
    uint16_t m1[4][1 << kLut1Bits];
    uint16_t m2[4][1 << kM2Bits];
    uint32_t fast[4][kFastWords];
};

// Encoder code book: for symbol s of table t, code | (length << 16)
struct HuffEncodeImage {
    uint32_t code[4][256];
};

// geometry shared by host and kernels
struct FrameGeom {
    uint32_t width, height;
    uint32_t stride;      // bytes per output row, ((w*24+31)/32)*4
    uint32_t mcu_cols;    // ceil(w/16)
    uint32_t mcu_rows;    // ceil(h/16)
    uint32_t mcus;        // per frame
    uint32_t blocks;      // 6 * mcus
    uint64_t frame_bytes; // stride * height
};

inline FrameGeom make_geom(uint32_t w, uint32_t h) {
    FrameGeom g;
    g.width = w;
    g.height = h;
    g.stride = (w * 24 + 31) / 32 * 4;
    g.mcu_cols = (w + 15) / 16;
    g.mcu_rows = (h + 15) / 16;
    g.mcus = g.mcu_cols * g.mcu_rows;
    g.blocks = g.mcus * 6;
    g.frame_bytes = (uint64_t)g.stride * h;
    return g;
}

}  // namespace amv
