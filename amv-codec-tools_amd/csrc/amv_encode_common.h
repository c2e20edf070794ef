// amv_encode_common.h -- the front half of the AMV video encoder as device functions: what amv_forward_kernel
// (dense coefficient lines out) and amv_encode_frame_kernel (coefficients never leave the chip) both run.
//
// Reference path: AMVmuxer/ffmpeg (lavc = libavcodec):
//   RGB24 -> YUVJ420P          lavc/imgconvert_template.h:654-, colorspace.h:30-97
//   vertical flip              lavc/mjpegenc.c:454-472 (amv_encode_picture)
//   forward DCT                lavc/jfdctint.c:184-343 (ff_jpeg_fdct_islow)
//   quantise                   lavc/mpegvideo_enc.c:3647-3724 (dct_quantize_c), :70-91, :492-496
// The reference encoder quantises with a matrix no AMV decoder uses; this one uses amvlib's
// fixed tables (AmvJpeg.c:30-61) and a true -128 level shift so that amvlib and the patched FFmpeg both decode it.
#pragma once
#include "amv_kernels.h"

namespace amv {
namespace enc {

constexpr int kWave = 64;
constexpr int kSegMcus = 10;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// a.lo * w.lo + a.hi * w.hi + c over the 16-bit signed halves of a and w, 32-bit wrap: two of a pass's products and
// their sum in one instruction
constexpr uint32_t pair16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ int dot2(uint32_t a, uint32_t w, int c) {
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(w), "v"(c));
    return d;
}
// x * w + c with x inside 24 bits and w a constant: one full-rate instruction, the low 32 bits of the exact result
__device__ __forceinline__ int mad24s(int x, int w, int c) {
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(x), "s"(w), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t pack16(int lo, int hi) {   // the low halves of two values side by side
    return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);
}

// One 8-point LL&M pass of jfdctint.c: kPass 0 = row_fdct (:184-258), 1 = column pass (:273-341).
// The reference forms its outputs from shared partial products (z1 .. z5); in 32-bit wrap arithmetic each output is
// just as well the plain linear combination of the pass's inputs those add up to -- d[7] = -11363 t4 + 9633 t5 -
// 6436 t6 + 2260 t7 and so on (every combined weight fits 16 bits) -- which is two dot products of 16-bit pairs.
// That needs the inputs themselves inside 16 bits: true of the whole row pass (sums of eight samples of -128..127),
// and of the odd half of the column pass (differences of two row-pass outputs, each inside +-16384); the column
// pass's t12, t13 are sums of four and can reach +-65536, so its d[2], d[6] keep the 32-bit form.
template <int kPass>
__device__ __forceinline__ void fdct8(int (&d)[8]) {
    constexpr int kConstBits = 13, kPass1Bits = 4;
    constexpr int kShift = kPass == 0 ? kConstBits - kPass1Bits : kConstBits + kPass1Bits;
    constexpr int kHalf = 1 << (kShift - 1);          // descale's rounding term rides in the accumulator
    const int t0 = d[0] + d[7], t7 = d[0] - d[7];
    const int t1 = d[1] + d[6], t6 = d[1] - d[6];
    const int t2 = d[2] + d[5], t5 = d[2] - d[5];
    const int t3 = d[3] + d[4], t4 = d[3] - d[4];
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    if (kPass == 0) {
        d[0] = (t10 + t11) << kPass1Bits;
        d[4] = (t10 - t11) << kPass1Bits;
        const uint32_t e = pack16(t12, t13);
        d[2] = dot2(e, pair16(4433, 4433 + 6270), kHalf) >> kShift;        // (t12 + t13) * 4433 + t13 * 6270
        d[6] = dot2(e, pair16(4433 - 15137, 4433), kHalf) >> kShift;       // (t12 + t13) * 4433 - t12 * 15137
    } else {
        d[0] = descale(t10 + t11, kPass1Bits);
        d[4] = descale(t10 - t11, kPass1Bits);
        // t12, t13 are sums of four row-pass outputs (each inside +-16385: see transform_block), their sum stays inside 19
        // bits: every factor fits 24 bits and the products 31, so these are full-rate 24-bit multiply-adds too (as plain
        // C the compiler made them v_mul_lo_u32 / v_mad_u64_u32, a quarter of the rate)
        const int z1 = mad24s(t12 + t13, 4433, kHalf);
        d[2] = mad24s(t13, 6270, z1) >> kShift;
        d[6] = mad24s(t12, -15137, z1) >> kShift;
    }
    // z5 = (t4 + t5 + t6 + t7) * 9633; z1 = (t4 + t7) * -7373; z2 = (t5 + t6) * -20995; z3 = (t4 + t6) * -16069 + z5;
    // z4 = (t5 + t7) * -3196 + z5; d[7] = t4 * 2446 + z1 + z3; d[5] = t5 * 16819 + z2 + z4; d[3] = t6 * 25172 + z2 + z3;
    // d[1] = t7 * 12299 + z1 + z4
    const uint32_t o45 = pack16(t4, t5), o67 = pack16(t6, t7);
    d[7] = dot2(o67, pair16(-16069 + 9633, -7373 + 9633), dot2(o45, pair16(2446 - 7373 - 16069 + 9633, 9633), kHalf)) >> kShift;
    d[5] = dot2(o67, pair16(-20995 + 9633, -3196 + 9633), dot2(o45, pair16(9633, 16819 - 20995 - 3196 + 9633), kHalf)) >> kShift;
    d[3] = dot2(o67, pair16(25172 - 20995 - 16069 + 9633, 9633), dot2(o45, pair16(-16069 + 9633, -20995 + 9633), kHalf)) >> kShift;
    d[1] = dot2(o67, pair16(9633, 12299 - 7373 - 3196 + 9633), dot2(o45, pair16(-7373 + 9633, -3196 + 9633), kHalf)) >> kShift;
}

// The row pass (jfdctint.c:184-258) straight off the eight samples as they lie in the planes -- four dwords of two int16
// each, (d0, d1) (d2, d3) (d4, d5) (d6, d7) -- with the butterflies in packed 16-bit arithmetic (round 6): samples are
// -128..127, so t0 .. t7 stay inside +-255 and t10 .. t13 inside +-510; every pair the dot products want is a register the
// packed adds and subtracts leave behind (their halves the other way round: the weights swap places instead), and
// d[0], d[4] -- (t10 +- t11) << 4 -- are dot products with (16, +-16).  27 instructions instead of 33.
__device__ __forceinline__ uint32_t pk_add16(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_add_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t pk_sub16(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ void fdct_row_packed(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, int (&d)[8]) {
    constexpr int kShift = 13 - 4, kHalf = 1 << (kShift - 1);
    const uint32_t r3 = __builtin_amdgcn_alignbit(w3, w3, 16), r2 = __builtin_amdgcn_alignbit(w2, w2, 16);   // (d7, d6), (d5, d4)
    const uint32_t a = pk_add16(w0, r3), s = pk_sub16(w0, r3);     // (t0, t1), (t7, t6)
    const uint32_t b = pk_add16(w1, r2), t = pk_sub16(w1, r2);     // (t2, t3), (t5, t4)
    const uint32_t rb = __builtin_amdgcn_alignbit(b, b, 16);       // (t3, t2)
    const uint32_t e = pk_add16(a, rb), f = pk_sub16(a, rb);       // (t10, t11), (t13, t12)
    d[0] = dot2(e, pair16(16, 16), 0);                             // (t10 + t11) << PASS1_BITS
    d[4] = dot2(e, pair16(16, -16), 0);
    d[2] = dot2(f, pair16(4433 + 6270, 4433), kHalf) >> kShift;    // (t12 + t13) * 4433 + t13 * 6270
    d[6] = dot2(f, pair16(4433, 4433 - 15137), kHalf) >> kShift;   // (t12 + t13) * 4433 - t12 * 15137
    // the odd half as in fdct8: t = (t5, t4), s = (t7, t6)
    d[7] = dot2(s, pair16(-7373 + 9633, -16069 + 9633), dot2(t, pair16(9633, 2446 - 7373 - 16069 + 9633), kHalf)) >> kShift;
    d[5] = dot2(s, pair16(-3196 + 9633, -20995 + 9633), dot2(t, pair16(16819 - 20995 - 3196 + 9633, 9633), kHalf)) >> kShift;
    d[3] = dot2(s, pair16(9633, 25172 - 20995 - 16069 + 9633), dot2(t, pair16(-20995 + 9633, -16069 + 9633), kHalf)) >> kShift;
    d[1] = dot2(s, pair16(12299 - 7373 - 3196 + 9633, 9633), dot2(t, pair16(-3196 + 9633, -7373 + 9633), kHalf)) >> kShift;
}

struct __attribute__((packed, aligned(1))) Px12 { uint32_t w[3]; };   // four RGB pixels, any alignment

// LDS plane pitches in samples: multiples of 8 (16-byte rows for ds_read_b128), padded so that the
// two luma block rows of an MCU do not start on the same bank
constexpr uint32_t kPitchY = kSegMcus * 16 + 8, kPitchC = kSegMcus * 8 + 8;
// a segment's planes: Y (16 rows), Cb, Cr (8 rows each), int16 -- 8 192 bytes, which is also 64 coefficient lines
constexpr uint32_t kPlaneSamples = 16 * kPitchY + 2 * 8 * kPitchC;
static_assert(kPlaneSamples * 2 == 64 * 128, "a segment's planes and its 64 coefficient lines share one LDS region");

// RGB_TO_Y / RGB_TO_U / RGB_TO_V of colorspace.h:78-88 with the channel order folded into the weights
struct Weights { int y0, y2, u0, u2, v0, v2; };

__device__ __forceinline__ int luma(const Weights& k, int c0, int c1, int c2) {
    // ((sum + 512) >> 10) - 128 with the level shift inside the (arithmetic) shift: 128 * 1024 is a multiple of 1024
    return (k.y0 * c0 + 601 * c1 + k.y2 * c2 + (512 - 128 * 1024)) >> 10;
}
__device__ __forceinline__ int chroma_u(const Weights& k, int s0, int s1, int s2) {   // 2x2 sums, shift 2
    return (k.u0 * s0 - 339 * s1 + k.u2 * s2 + 2047) >> 12;                           // +128 -128
}
__device__ __forceinline__ int chroma_v(const Weights& k, int s0, int s1, int s2) {
    return (k.v0 * s0 - 429 * s1 + k.v2 * s2 + 2047) >> 12;
}

__device__ __forceinline__ void unpack12(const Px12& v, int (&b)[12]) {
#pragma unroll
    for (int i = 0; i < 12; ++i) b[i] = (int)((v.w[i >> 2] >> (8 * (i & 3))) & 0xffu);
}

struct Source {           // where a frame's pixels are
    const uint8_t* pix;   // RGB24 / BGR24 frames, rows pix_stride apart (kYuv == false)
    uint32_t pix_stride;
    int is_bgr;
    YuvSource yuv;        // planar YUVJ420P (kYuv == true)
};

// Stage 1 of a segment (MCU row `my`, MCUs m0 .. m0 + cnt) of frame f: colour conversion of 4x2-pixel patches (two
// unaligned 12-byte loads each) into the LDS planes -- 8 luma and 2 + 2 chroma samples per patch.  Bitstream row k is
// picture row h-1-k (mjpegenc.c:462-467); rows and columns outside the picture repeat the nearest edge sample.
// kYuv: the source is planar YUVJ420P, what amv_encoder itself takes (mjpegenc.c:493) -- samples are copied (level shift
// 128) instead of converted; everything else, edge repetition included, is the same, so that rgb24_to_yuvj420p followed
// by this form equals the RGB form bit for bit.  The caller synchronises the wave before the planes are read.
// In two halves: a lane takes up to kTrips patches, and load_patches requests all their pixels at once (a loop that
// loaded and converted a patch per trip waited for memory five times in a row) -- a caller with other work in hand
// puts it between the two.
constexpr int kTrips = 5;                                      // 8 rows pairs x cnt * 4 patches <= 320 = 5 * 64
struct Patches { Px12 a[kTrips], b[kTrips]; };

__device__ __forceinline__ void load_patches(const Source& in, uint32_t f, const FrameGeom& g, uint32_t my, uint32_t m0, uint32_t cnt,
                                             uint32_t lane, Patches& px) {
    const uint32_t w = g.width, h = g.height;
    const uint8_t* src = in.pix + (uint64_t)f * in.pix_stride * h;
    const uint32_t d4 = cnt * 4u, inv = (65536u + d4 - 1u) / d4;   // t / d4 == (t * inv) >> 16 for t < 8 * d4 <= 320
#pragma unroll
    for (int it = 0; it < kTrips; ++it) {
        const uint32_t t = lane + (uint32_t)it * kWave;
        px.a[it] = px.b[it] = Px12{{0u, 0u, 0u}};
        if (t < 8u * d4) {
            const uint32_t i2 = (t * inv) >> 16, p = t - i2 * d4;
            const uint32_t k0 = my * 16u + 2u * i2;
            const bool inside = k0 < h;
            const uint32_t row_a = inside ? h - 1u - k0 : 1u, row_b = inside ? h - 2u - k0 : 0u;
            const uint32_t c = m0 * 16u + 4u * p;
            if (c + 3u < w) {
                px.a[it] = *reinterpret_cast<const Px12*>(src + (uint64_t)row_a * in.pix_stride + c * 3u);
                px.b[it] = *reinterpret_cast<const Px12*>(src + (uint64_t)row_b * in.pix_stride + c * 3u);
            }
        }
    }
}

template <bool kYuv>
__device__ __forceinline__ void convert_patches(const Source& in, const Patches& px, uint32_t f, const FrameGeom& g, uint32_t my, uint32_t m0,
                                                uint32_t cnt, uint32_t lane, int16_t* s_y, int16_t* s_cb, int16_t* s_cr) {
    const uint32_t w = g.width, h = g.height, cw = w >> 1;
    const uint8_t* src = in.pix + (uint64_t)f * in.pix_stride * h;
    const uint32_t pix_stride = in.pix_stride;
    const YuvSource& yuv = in.yuv;
    const Weights k = in.is_bgr ? Weights{117, 306, 512, -173, -83, 512} : Weights{306, 117, -173, 512, 512, -83};
    const uint32_t d4 = cnt * 4u, inv = (65536u + d4 - 1u) / d4;
#pragma unroll
    for (int it = 0; it < kTrips; ++it) {
        const uint32_t t = lane + (uint32_t)it * kWave;
        if (t >= 8u * d4) break;
        const uint32_t i2 = (t * inv) >> 16, p = t - i2 * d4;
        const uint32_t k0 = my * 16u + 2u * i2;                    // bitstream rows k0, k0 + 1
        const bool inside = k0 < h;                                // h is even
        const uint32_t row_a = inside ? h - 1u - k0 : 1u, row_b = inside ? h - 2u - k0 : 0u;
        const uint32_t c = m0 * 16u + 4u * p;
        const uint8_t* pa = src + (uint64_t)row_a * pix_stride;
        const uint8_t* pb = src + (uint64_t)row_b * pix_stride;
        int ya[4], yb[4], u[2], v[2];
        if (kYuv) {
            const uint8_t* ya_p = yuv.y + (uint64_t)f * yuv.y_frame + (uint64_t)row_a * yuv.y_stride;
            const uint8_t* yb_p = yuv.y + (uint64_t)f * yuv.y_frame + (uint64_t)row_b * yuv.y_stride;
            // 4:2:0: the chroma row under both luma rows; 4:2:2: the rows of both, averaged (below the picture row_a, row_b
            // are rows 1 and 0: the 4:2:0 picture's chroma row 0, which is what repeats there)
            const uint32_t ca = yuv.c_rows422 ? row_a : row_b >> 1, cb_row = yuv.c_rows422 ? row_b : row_b >> 1;
            const uint64_t co = (uint64_t)f * yuv.c_frame + (uint64_t)cb_row * yuv.c_stride;
            const uint64_t co2 = (uint64_t)f * yuv.c_frame + (uint64_t)ca * yuv.c_stride;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t x = min(c + (uint32_t)q, w - 1u);
                ya[q] = (int)ya_p[x] - 128;
                yb[q] = (int)yb_p[x] - 128;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t x = min((c >> 1) + (uint32_t)e, cw - 1u);
                u[e] = (int)yuv.cb[co + x] - 128;
                v[e] = (int)yuv.cr[co + x] - 128;
                if (yuv.c_rows422) {
                    u[e] = (((int)yuv.cb[co + x] + (int)yuv.cb[co2 + x] + 1) >> 1) - 128;
                    v[e] = (((int)yuv.cr[co + x] + (int)yuv.cr[co2 + x] + 1) >> 1) - 128;
                }
            }
        } else if (c + 3u < w) {
            int a[12], b[12];
            unpack12(px.a[it], a);
            unpack12(px.b[it], b);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ya[q] = luma(k, a[3 * q], a[3 * q + 1], a[3 * q + 2]);
                yb[q] = luma(k, b[3 * q], b[3 * q + 1], b[3 * q + 2]);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int s0 = a[6 * e] + a[6 * e + 3] + b[6 * e] + b[6 * e + 3];
                const int s1 = a[6 * e + 1] + a[6 * e + 4] + b[6 * e + 1] + b[6 * e + 4];
                const int s2 = a[6 * e + 2] + a[6 * e + 5] + b[6 * e + 2] + b[6 * e + 5];
                u[e] = chroma_u(k, s0, s1, s2);
                v[e] = chroma_v(k, s0, s1, s2);
            }
        } else {                                                   // right edge of a picture whose width is not 0 mod 16
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t x = min(c + (uint32_t)q, w - 1u) * 3u;
                ya[q] = luma(k, pa[x], pa[x + 1], pa[x + 2]);
                yb[q] = luma(k, pb[x], pb[x + 1], pb[x + 2]);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const uint32_t x = min((c >> 1) + (uint32_t)e, cw - 1u) * 6u;
                const int s0 = pa[x] + pa[x + 3] + pb[x] + pb[x + 3];
                const int s1 = pa[x + 1] + pa[x + 4] + pb[x + 1] + pb[x + 4];
                const int s2 = pa[x + 2] + pa[x + 5] + pb[x + 2] + pb[x + 5];
                u[e] = chroma_u(k, s0, s1, s2);
                v[e] = chroma_v(k, s0, s1, s2);
            }
        }
        if (!inside) {                                             // below the picture: luma repeats picture row 0
#pragma unroll
            for (int q = 0; q < 4; ++q) ya[q] = yb[q];
        }
        uint2 la, lb;
        la.x = ((uint32_t)ya[0] & 0xffffu) | ((uint32_t)ya[1] << 16);
        la.y = ((uint32_t)ya[2] & 0xffffu) | ((uint32_t)ya[3] << 16);
        lb.x = ((uint32_t)yb[0] & 0xffffu) | ((uint32_t)yb[1] << 16);
        lb.y = ((uint32_t)yb[2] & 0xffffu) | ((uint32_t)yb[3] << 16);
        *reinterpret_cast<uint2*>(s_y + (2u * i2) * kPitchY + 4u * p) = la;
        *reinterpret_cast<uint2*>(s_y + (2u * i2 + 1u) * kPitchY + 4u * p) = lb;
        *reinterpret_cast<uint32_t*>(s_cb + i2 * kPitchC + 2u * p) = ((uint32_t)u[0] & 0xffffu) | ((uint32_t)u[1] << 16);
        *reinterpret_cast<uint32_t*>(s_cr + i2 * kPitchC + 2u * p) = ((uint32_t)v[0] & 0xffffu) | ((uint32_t)v[1] << 16);
    }
}

template <bool kYuv>
__device__ __forceinline__ void convert_segment(const Source& in, uint32_t f, const FrameGeom& g, uint32_t my, uint32_t m0, uint32_t cnt,
                                                uint32_t lane, int16_t* s_y, int16_t* s_cb, int16_t* s_cr) {
    Patches px;
    if (!kYuv) load_patches(in, f, g, my, m0, cnt, lane, px);
    convert_patches<kYuv>(in, px, f, g, my, m0, cnt, lane, s_y, s_cb, s_cr);
}

// The AC quantiser's multipliers, (1 << 22) / (8 * Q[scan]) (ff_convert_matrix, mpegvideo_enc.c:80-91, qscale 8), in the
// order the column pass delivers coefficients -- entry c * 8 + r belongs to row r of column c -- luma then chroma.  A
// workgroup copies them to LDS once (load_quant_mul) and a lane reads its component's eight per column as two 16-byte
// pieces: round 5 selected each of the 63 between two literals per lane (a compare, a select and a constant to load per
// coefficient, sixty-three registers held for them and the kernel's eight spills).
constexpr int natural_of_scan(int scan) {
    for (int i = 0; i < 64; ++i)
        if (kScanOfNatural[i] == scan) return i;
    return 0;
}
struct QuantMul { uint32_t m[2][64]; };
constexpr QuantMul make_quant_mul() {
    QuantMul q{};
    for (int c = 0; c < 8; ++c)
        for (int r = 0; r < 8; ++r) {
            const int scan = kScanOfNatural[r * 8 + c];
            q.m[0][c * 8 + r] = (1u << 22) / (8u * kQuantLuma[scan]);
            q.m[1][c * 8 + r] = (1u << 22) / (8u * kQuantChroma[scan]);
        }
    return q;
}
__device__ const QuantMul kQuantMul = make_quant_mul();
constexpr uint32_t kQuantMulWords = 128;
__device__ __forceinline__ void load_quant_mul(uint32_t* s_qmul, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = tid; i < kQuantMulWords; i += nthreads) s_qmul[i] = (&kQuantMul.m[0][0])[i];
}

// x * m + c for x, m inside 24 bits: one full-rate instruction, the low 32 bits of the exact result
__device__ __forceinline__ int mad24v(int x, uint32_t m, int c) {
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(m), "v"(c));
    return d;
}
// both 16-bit halves of w: 0 stays 0, anything else becomes 1
__device__ __forceinline__ uint32_t halves_nonzero(uint32_t w) {
    uint32_t d;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(w), "s"(0x00010001u));
    return d;
}
// bits 0..15 of x to the even positions, bits 16..31 to the odd ones (bit i -> 2 i, bit 16 + i -> 2 i + 1)
__device__ __forceinline__ uint32_t interleave_halves(uint32_t x) {
    uint32_t t;
    t = (x ^ (x >> 8)) & 0x0000ff00u; x = x ^ t ^ (t << 8);
    t = (x ^ (x >> 4)) & 0x00f000f0u; x = x ^ t ^ (t << 4);
    t = (x ^ (x >> 2)) & 0x0c0c0c0cu; x = x ^ t ^ (t << 2);
    t = (x ^ (x >> 1)) & 0x22222222u; x = x ^ t ^ (t << 1);
    return x;
}

// Stage 2: lane's block (lane = 6 * MCU in segment + block in MCU) out of the planes, in registers: 8 row passes,
// DCTELEM truncation, 8 column passes, dct_quantize_c.  out: the 64 quantised coefficients, scan order, int16 pairs;
// nz_lo / nz_hi: bit k set where coefficient k (scan order, k >= 1) is not zero.  Pairs are put together
// as the columns come out: holding all 64 values for a pass in scan order cost registers, spills and 8 % of the
// one-kernel encoder's time (and sending them to the LDS line two bytes at a time instead of through out[] cost 3 %).
// The AC quantiser (mpegvideo_enc.c:3702-3712) is sign(x) * ((|x| * m + bias) >> 22) with m = (1 << 22) / (8 * Q)
// (ff_convert_matrix :80-91, qscale 8); for x < 0 that is ceil((x * m - bias) / 2^22) = (x * m + (2^22 - 1 - bias)) >> 22,
// so both signs are one multiply-add and one arithmetic shift: (x * m + (bias ^ (sign & (2^22 - 1)))) >> 22.
// |x * m| stays under 2^31: x is an fdct output of 8-bit samples (|x| <= 2^14), m <= 2^22 / 40.
// The mask is read off the finished pairs (round 6; a compare, a select and an OR per coefficient before): both halves
// of a pair reduced to 0 / 1 by one packed minimum, sixteen pairs shifted into one word -- even scan positions in its low
// half, odd ones in its high half -- and the halves interleaved by four exchange steps.
// s_qmul: the workgroup's copy of kQuantMul (load_quant_mul).
__device__ __forceinline__ void transform_block(const int16_t* s_y, const int16_t* s_cb, const int16_t* s_cr, const uint32_t* s_qmul,
                                                uint32_t lane, uint32_t qbias, uint32_t (&out)[32], uint32_t& nz_lo, uint32_t& nz_hi) {
    const uint32_t m = lane / 6u, k6 = lane - 6u * m;
    const bool is_c = k6 >= 4u;
    const int16_t* in = is_c ? (k6 == 4u ? s_cb : s_cr) + m * 8u
                             : s_y + ((k6 >> 1) * 8u) * kPitchY + m * 16u + (k6 & 1u) * 8u;
    const uint32_t pitch = is_c ? kPitchC : kPitchY;
    const uint4* const qm = reinterpret_cast<const uint4*>(s_qmul + (is_c ? 64u : 0u));
    int d[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {                                  // get_pixels + row_fdct
        const uint4 q = *reinterpret_cast<const uint4*>(in + r * pitch);
        fdct_row_packed(q.x, q.y, q.z, q.w, d[r]);
        // DCTELEM is 16 bit (dsputil.h:38): nothing to truncate -- the planes hold samples of -128..127 (luma / chroma_u /
        // chroma_v, the level-shifted bytes of a YUVJ420P source), for which every row-pass output is inside +-16385
        // (the sum of |weight| * 128 over its eight inputs: 16384 for d[0], d[4], under 15138 for the others), and a
        // column-pass output inside +-8193 (tests/test_oracle_pin.py::test_fdct_outputs_fit_dctelem)
    }
    const int bias = (int)(qbias << 14);   // intra_quant_bias << (QMAT_SHIFT - QUANT_BIAS_SHIFT), :3679
    int held[64];                          // a coefficient waiting for the other half of its pair (the scan's neighbours sit a column apart at most)
#pragma unroll
    for (int c = 0; c < 8; ++c) {                                  // column pass + dct_quantize_c
        int col[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) col[r] = d[r][c];
        const uint4 ma = qm[2 * c], mb = qm[2 * c + 1];            // this column's multipliers (on their way during the pass)
        const uint32_t mul[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
        fdct8<1>(col);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int scan = kScanOfNatural[r * 8 + c];
            const int x = col[r];
            int q;
            if (r == 0 && c == 0) {        // DC: (block[0] + q/2) / q with q = 8 * step, :3670-3676
                constexpr int ql = 8 * kQuantLuma[0], qc = 8 * kQuantChroma[0];
                const int ax = abs(x);
                const int a = is_c ? (ax + (qc >> 1)) / qc : (ax + (ql >> 1)) / ql;
                q = x < 0 ? -a : a;
            } else {
                const int sign = x >> 31;
                q = mad24v(x, mul[r], bias ^ (sign & 0x3fffff)) >> 22;
            }
            // the pair (scan 2 i, 2 i + 1) is put together by one byte permute when its second half arrives (a mask and an
            // OR per coefficient before)
            const int mate = natural_of_scan(scan ^ 1), mr = mate >> 3, mc = mate & 7;
            if (mc < c || (mc == c && mr < r)) out[scan >> 1] = (scan & 1) ? pack16(held[scan ^ 1], q) : pack16(q, held[scan ^ 1]);
            else held[scan] = q;
        }
    }
    uint32_t even_odd_lo = 0u, even_odd_hi = 0u;                  // bit i: coefficient 2 i (32 + 2 i) != 0, bit 16 + i: coefficient 2 i + 1
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        even_odd_lo |= halves_nonzero(out[i]) << i;
        even_odd_hi |= halves_nonzero(out[16 + i]) << i;
    }
    nz_lo = interleave_halves(even_odd_lo) & ~1u;                  // (the DC is not in the mask)
    nz_hi = interleave_halves(even_odd_hi);
}

// coefficient k of lane `lane`'s 128-byte line in an LDS region of 64 lines: 16-byte granules XOR-swizzled by lane, so
// that lanes reading the same granule of their own lines do not meet on banks
__device__ __forceinline__ uint32_t line_offset(uint32_t lane, uint32_t k) {
    return lane * 128u + ((((k >> 3) ^ lane) & 7u) << 4) + ((k & 7u) << 1);
}

}  // namespace enc
}  // namespace amv
