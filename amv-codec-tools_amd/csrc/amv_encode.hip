// amv_encode.hip -- AMV video encode kernels for gfx950 (MI355X).
//
// Reference path: AMVmuxer/ffmpeg (lavc = libavcodec):
//   RGB24 -> YUVJ420P          lavc/imgconvert_template.h:654-, colorspace.h:30-97
//   vertical flip              lavc/mjpegenc.c:454-472 (amv_encode_picture)
//   forward DCT                lavc/jfdctint.c:184-343 (ff_jpeg_fdct_islow)
//   quantise                   lavc/mpegvideo_enc.c:3647-3724 (dct_quantize_c), :70-91, :492-496
//   run/size Huffman           lavc/mjpegenc.c:357-450, code book lavc/mjpeg.c:129-147
//   stuffing, FF escaping, EOI lavc/mjpegenc.c:282-355; SOI only :201-204
// The reference encoder quantises with a matrix no AMV decoder uses; this
// one uses amvlib's fixed tables (AmvJpeg.c:30-61) and a true -128 level shift so that amvlib
// and the patched FFmpeg both decode it.
//
//   amv_forward_kernel  data parallel: one wave per MCU-row segment; colour conversion of 4x2-pixel
//                       patches into LDS planes, then one 8x8 block per lane held in registers:
//                       row passes, column passes, quantise, scan order, one 128-byte line out.
//   amv_pack_kernel     serial inside a frame (DC prediction + variable-length output), so
//                       one lane per frame; a wave stages block b of its 64 frames in LDS with
//                       whole-line loads and every lane codes its own block from there.
//   amv_scan_kernel / amv_gather_kernel   prefix sum of chunk lengths and compaction.
#include "amv_kernels.h"

namespace amv {

namespace {

constexpr int kWave = 64;
constexpr int kSegMcus = 10;

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// One 8-point LL&M pass of jfdctint.c: kPass 0 = row_fdct (:184-258), 1 = column pass (:273-341)
template <int kPass>
__device__ __forceinline__ void fdct8(int (&d)[8]) {
    constexpr int kConstBits = 13, kPass1Bits = 4;
    constexpr int kShift = kPass == 0 ? kConstBits - kPass1Bits : kConstBits + kPass1Bits;
    const int t0 = d[0] + d[7], t7 = d[0] - d[7];
    const int t1 = d[1] + d[6], t6 = d[1] - d[6];
    const int t2 = d[2] + d[5], t5 = d[2] - d[5];
    const int t3 = d[3] + d[4], t4 = d[3] - d[4];
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    if (kPass == 0) {
        d[0] = (t10 + t11) << kPass1Bits;
        d[4] = (t10 - t11) << kPass1Bits;
    } else {
        d[0] = descale(t10 + t11, kPass1Bits);
        d[4] = descale(t10 - t11, kPass1Bits);
    }
    int z1 = (t12 + t13) * 4433;
    d[2] = descale(z1 + t13 * 6270, kShift);
    d[6] = descale(z1 - t12 * 15137, kShift);
    z1 = t4 + t7;
    int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const int z5 = (z3 + z4) * 9633;
    const int u4 = t4 * 2446, u5 = t5 * 16819, u6 = t6 * 25172, u7 = t7 * 12299;
    z1 *= -7373;
    z2 *= -20995;
    z3 = z3 * -16069 + z5;
    z4 = z4 * -3196 + z5;
    d[7] = descale(u4 + z1 + z3, kShift);
    d[5] = descale(u5 + z2 + z4, kShift);
    d[3] = descale(u6 + z2 + z3, kShift);
    d[1] = descale(u7 + z1 + z4, kShift);
}

}  // namespace

namespace {

struct __attribute__((packed, aligned(1))) Px12 { uint32_t w[3]; };   // four RGB pixels, any alignment

// LDS plane pitches in samples: multiples of 8 (16-byte rows for ds_read_b128), padded so that the
// two luma block rows of an MCU do not start on the same bank
constexpr uint32_t kPitchY = kSegMcus * 16 + 8, kPitchC = kSegMcus * 8 + 8;

// RGB_TO_Y / RGB_TO_U / RGB_TO_V of colorspace.h:78-88 with the channel order folded into the weights
struct Weights { int y0, y2, u0, u2, v0, v2; };

__device__ __forceinline__ int luma(const Weights& k, int c0, int c1, int c2) {
    return ((k.y0 * c0 + 601 * c1 + k.y2 * c2 + 512) >> 10) - 128;
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

}  // namespace

// One wave per MCU-row segment of up to kSegMcus MCUs.
//  1. colour conversion: every lane takes 4x2-pixel patches (two unaligned 12-byte loads), writes
//     8 luma and 2+2 chroma samples into the LDS planes.  Bitstream row k is picture row h-1-k
//     (mjpegenc.c:462-467); rows and columns outside the picture repeat the nearest edge sample.
//  2. one lane per 8x8 block, the block in registers: 8 row passes, DCTELEM truncation, 8 column
//     passes, dct_quantize_c, scan order (a compile-time permutation), one 128-byte line out.
// kYuv: the source is planar YUVJ420P, what amv_encoder itself takes (mjpegenc.c:493) -- stage 1 only copies
// samples (level shift 128) instead of converting; everything else, edge repetition included, is the same, so
// that rgb24_to_yuvj420p followed by this form equals the RGB form bit for bit.
template <bool kYuv>
__global__ __launch_bounds__(kWave) void amv_forward_kernel(
    const uint8_t* __restrict__ pix, uint32_t pix_stride, int is_bgr, YuvSource yuv, uint32_t n, FrameGeom g,
    uint32_t nseg, uint32_t per_seg, uint32_t qbias, int16_t* __restrict__ coef) {
    __shared__ __attribute__((aligned(16))) int16_t s_y[16 * kPitchY];
    __shared__ __attribute__((aligned(16))) int16_t s_cb[8 * kPitchC];
    __shared__ __attribute__((aligned(16))) int16_t s_cr[8 * kPitchC];

    const uint32_t lane = threadIdx.x;
    uint32_t bid = blockIdx.x;
    const uint32_t seg = bid % nseg;
    bid /= nseg;
    const uint32_t my = bid % g.mcu_rows;
    const uint32_t f = bid / g.mcu_rows;
    const uint32_t m0 = seg * per_seg;
    if (m0 >= g.mcu_cols) return;          // very wide pictures: the balanced split can leave the last segment empty
    const uint32_t cnt = min(per_seg, g.mcu_cols - m0);
    const uint32_t nb = cnt * 6;
    const uint32_t w = g.width, h = g.height, cw = w >> 1;
    const uint8_t* src = pix + (uint64_t)f * pix_stride * h;
    const Weights k = is_bgr ? Weights{117, 306, 512, -173, -83, 512} : Weights{306, 117, -173, 512, 512, -83};

    const uint32_t d4 = cnt * 4u, inv = (65536u + d4 - 1u) / d4;   // t / d4 == (t * inv) >> 16 for t < 8 * d4 <= 320
    // A lane takes up to kTrips 4x2-pixel patches.  Their pixels are requested all at once (a loop that loaded and
    // converted a patch per trip waited for memory five times in a row); the conversion follows.
    constexpr int kTrips = 5;                                      // 8 * d4 <= 320 = 5 * 64
    Px12 ra[kTrips], rb[kTrips];
    if (!kYuv) {
#pragma unroll
        for (int it = 0; it < kTrips; ++it) {
            const uint32_t t = lane + (uint32_t)it * kWave;
            ra[it] = rb[it] = Px12{{0u, 0u, 0u}};
            if (t < 8u * d4) {
                const uint32_t i2 = (t * inv) >> 16, p = t - i2 * d4;
                const uint32_t k0 = my * 16u + 2u * i2;
                const bool inside = k0 < h;
                const uint32_t row_a = inside ? h - 1u - k0 : 1u, row_b = inside ? h - 2u - k0 : 0u;
                const uint32_t c = m0 * 16u + 4u * p;
                if (c + 3u < w) {
                    ra[it] = *reinterpret_cast<const Px12*>(src + (uint64_t)row_a * pix_stride + c * 3u);
                    rb[it] = *reinterpret_cast<const Px12*>(src + (uint64_t)row_b * pix_stride + c * 3u);
                }
            }
        }
    }
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
            const uint64_t co = (uint64_t)f * yuv.c_frame + (uint64_t)(row_b >> 1) * yuv.c_stride;
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
            }
        } else if (c + 3u < w) {
            int a[12], b[12];
            unpack12(ra[it], a);
            unpack12(rb[it], b);
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
    __syncthreads();
    if (lane >= nb) return;

    const uint32_t m = lane / 6u, k6 = lane - 6u * m;
    const bool is_c = k6 >= 4u;
    const int16_t* in = is_c ? (k6 == 4u ? s_cb : s_cr) + m * 8u
                             : s_y + ((k6 >> 1) * 8u) * kPitchY + m * 16u + (k6 & 1u) * 8u;
    const uint32_t pitch = is_c ? kPitchC : kPitchY;
    int d[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {                                  // get_pixels + row_fdct
        const uint4 q = *reinterpret_cast<const uint4*>(in + r * pitch);
        const uint32_t ws[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int c = 0; c < 8; ++c) d[r][c] = (c & 1) ? ((int)ws[c >> 1] >> 16) : (int)(int16_t)(ws[c >> 1] & 0xffffu);
        fdct8<0>(d[r]);
#pragma unroll
        for (int c = 0; c < 8; ++c) d[r][c] = (int16_t)d[r][c];    // DCTELEM is 16 bit (dsputil.h:38)
    }
    const int bias = (int)(qbias << 14);   // intra_quant_bias << (QMAT_SHIFT - QUANT_BIAS_SHIFT), :3679
    uint32_t out[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) out[i] = 0u;
#pragma unroll
    for (int c = 0; c < 8; ++c) {                                  // column pass + dct_quantize_c
        int col[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) col[r] = d[r][c];
        fdct8<1>(col);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int scan = kScanOfNatural[r * 8 + c];
            const int x = (int16_t)col[r];
            int a;
            if (r == 0 && c == 0) {        // DC: (block[0] + q/2) / q with q = 8 * step, :3670-3676
                constexpr int ql = 8 * kQuantLuma[0], qc = 8 * kQuantChroma[0];
                const int ax = abs(x);
                a = is_c ? (ax + (qc >> 1)) / qc : (ax + (ql >> 1)) / ql;
            } else {                       // AC: (bias + |level|) >> QMAT_SHIFT with ff_convert_matrix's
                                           // (1<<22)/(8*Q) (mpegvideo_enc.c:80-91, qscale 8), :3702-3712
                const int ml = (int)((1u << 22) / (8u * kQuantLuma[scan])), mc = (int)((1u << 22) / (8u * kQuantChroma[scan]));
                const int level = x * (is_c ? mc : ml);
                a = (abs(level) + bias) >> 22;
            }
            const uint32_t val = (uint32_t)(x < 0 ? -a : a) & 0xffffu;
            out[scan >> 1] |= val << (16 * (scan & 1));
        }
    }
    uint4* dst = reinterpret_cast<uint4*>(coef + ((uint64_t)f * g.mcus + (uint64_t)my * g.mcu_cols + m0) * 384u + lane * 64u);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = make_uint4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
}

void launch_forward(const uint8_t* pix, uint32_t pix_stride, int is_bgr, uint32_t n,
                    const FrameGeom& g, uint32_t qbias, int16_t* coef, hipStream_t s) {
    if (n == 0) return;
    const uint32_t nseg = (g.mcu_cols + kSegMcus - 1) / kSegMcus;
    const uint32_t per_seg = (g.mcu_cols + nseg - 1) / nseg;      // balanced: 11 columns -> 6 + 5
    const uint64_t grid = (uint64_t)n * g.mcu_rows * nseg;
    hipLaunchKernelGGL(amv_forward_kernel<false>, dim3((uint32_t)grid), dim3(kWave), 0, s, pix, pix_stride,
                       is_bgr, YuvSource{}, n, g, nseg, per_seg, qbias, coef);
}

void launch_forward_yuv(const YuvSource& src, uint32_t n, const FrameGeom& g, uint32_t qbias, int16_t* coef, hipStream_t s) {
    if (n == 0) return;
    const uint32_t nseg = (g.mcu_cols + kSegMcus - 1) / kSegMcus;
    const uint32_t per_seg = (g.mcu_cols + nseg - 1) / nseg;
    const uint64_t grid = (uint64_t)n * g.mcu_rows * nseg;
    hipLaunchKernelGGL(amv_forward_kernel<true>, dim3((uint32_t)grid), dim3(kWave), 0, s, (const uint8_t*)nullptr, 0u,
                       0, src, n, g, nseg, per_seg, qbias, coef);
}

// ============================================================================================
// entropy coder
// ============================================================================================

namespace {

struct BitWriter {
    uint8_t* out;
    uint32_t pos;
    uint64_t acc;  // pending bits, right aligned
    int nacc;
};

// escape_FF (mjpegenc.c:282-336): every FF of the scan is followed by a 00
__device__ __forceinline__ void emit_byte(BitWriter& w, uint32_t b) {
    w.out[w.pos++] = (uint8_t)b;
    if (b == 0xffu) w.out[w.pos++] = 0;
}

__device__ __forceinline__ void put_bits(BitWriter& w, int nbits, uint32_t value) {
    w.acc = (w.acc << nbits) | value;
    w.nacc += nbits;
    while (w.nacc >= 8) {
        w.nacc -= 8;
        emit_byte(w, (uint32_t)(w.acc >> w.nacc) & 0xffu);
    }
}

// ff_mjpeg_encode_dc / the AC branch of encode_block (mjpegenc.c:357-377, 411-424)
__device__ __forceinline__ void put_coef(BitWriter& w, const uint32_t* book, int run, int val) {
    int mant = val;
    if (val < 0) { val = -val; mant--; }
    const int nb = 32 - __clz(val);
    const uint32_t e = book[(run << 4) | nb];
    put_bits(w, (int)(e >> 16) + nb, ((e & 0xffffu) << nb) | ((uint32_t)mant & ((1u << nb) - 1u)));
}

__device__ __forceinline__ uint32_t slot_offset(uint32_t lane, uint32_t k) {
    return lane * 128u + ((((k >> 3) ^ lane) & 7u) << 4) + ((k & 7u) << 1);
}

}  // namespace

__global__ __launch_bounds__(kWave) void amv_pack_kernel(
    const int16_t* __restrict__ coef, uint32_t n, uint32_t blocks_per_frame,
    const HuffEncodeImage* __restrict__ img, uint8_t* __restrict__ tmp, uint32_t bound,
    uint32_t* __restrict__ lens, const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count) {
    __shared__ uint32_t s_book[4][256];
    __shared__ __attribute__((aligned(16))) uint4 s_slots[kWave * 8];
    __shared__ uint32_t s_frame[kWave];

    // with a list (frames amv_pack_wave_kernel handed back): frames list[0 .. *list_count)
    const uint32_t lane = threadIdx.x;
    const uint32_t f0 = blockIdx.x * kWave;
    if (list) n = *list_count;
    if (f0 >= n) return;
    const uint32_t frame = f0 + lane < n ? (list ? list[f0 + lane] : f0 + lane) : 0xffffffffu;
    s_frame[lane] = frame;
    const bool live = frame != 0xffffffffu;
    for (int i = lane; i < 4 * 256; i += kWave) (&s_book[0][0])[i] = (&img->code[0][0])[i];

    BitWriter w;
    w.out = tmp + (uint64_t)(live ? frame : 0) * bound;
    w.pos = 0;
    w.acc = 0;
    w.nacc = 0;
    if (live) { w.out[0] = 0xff; w.out[1] = 0xd8; w.pos = 2; }   // SOI only, mjpegenc.c:201-204
    int pred0 = 0, pred1 = 0, pred2 = 0;
    const char* slot_bytes = reinterpret_cast<const char*>(s_slots);

    uint32_t k6 = 0;
    for (uint32_t b = 0; b < blocks_per_frame; ++b) {
        __syncthreads();
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) {   // block b of 64 frames: 64 whole lines
            const uint32_t c = i * kWave + lane;
            const uint32_t s = c >> 3, part = c & 7u;
            uint4 v = make_uint4(0, 0, 0, 0);
            const uint32_t fr = s_frame[s];
            if (fr != 0xffffffffu)
                v = reinterpret_cast<const uint4*>(coef + ((uint64_t)fr * blocks_per_frame + b) * 64u)[part];
            s_slots[s * 8u + (part ^ (s & 7u))] = v;
        }
        __syncthreads();
        if (live) {   // encode_block, mjpegenc.c:379-435
            const int cls = k6 < 4 ? 0 : 1;
            const int dc = *reinterpret_cast<const int16_t*>(slot_bytes + slot_offset(lane, 0));
            int diff;
            if (k6 < 4) { diff = dc - pred0; pred0 = dc; }
            else if (k6 == 4) { diff = dc - pred1; pred1 = dc; }
            else { diff = dc - pred2; pred2 = dc; }
            if (diff == 0) put_bits(w, (int)(s_book[cls][0] >> 16), s_book[cls][0] & 0xffffu);
            else put_coef(w, s_book[cls], 0, diff);
            const uint32_t* ac = s_book[2 + cls];
            int run = 0;
            for (uint32_t k = 1; k < 64; ++k) {
                const int v = *reinterpret_cast<const int16_t*>(slot_bytes + slot_offset(lane, k));
                if (v == 0) { ++run; continue; }
                while (run >= 16) { put_bits(w, (int)(ac[0xf0] >> 16), ac[0xf0] & 0xffffu); run -= 16; }
                put_coef(w, ac, run, v);
                run = 0;
            }
            if (run) put_bits(w, (int)(ac[0] >> 16), ac[0] & 0xffffu);   // EOB, :430-431
        }
        if (++k6 == 6) k6 = 0;
    }
    if (live) {
        if (w.nacc) put_bits(w, 8 - w.nacc, (1u << (8 - w.nacc)) - 1u);   // ff_mjpeg_encode_stuffing :338-343
        w.out[w.pos++] = 0xff;                                           // EOI :354
        w.out[w.pos++] = 0xd9;
        lens[frame] = w.pos;
    }
}

void launch_pack(const int16_t* coef, uint32_t n, const FrameGeom& g, const HuffEncodeImage* d_img,
                 uint8_t* tmp, uint32_t bound, uint32_t* lens, const uint32_t* list, const uint32_t* list_count,
                 hipStream_t s) {
    if (n == 0) return;
    const uint32_t grid = (n + kWave - 1) / kWave;   // with a list: upper bound, surplus groups exit at once
    hipLaunchKernelGGL(amv_pack_kernel, dim3(grid), dim3(kWave), 0, s, coef, n, g.blocks, d_img, tmp,
                       bound, lens, list, list_count);
}

// ============================================================================================
// compaction: offs = exclusive scan(lens); blob[offs[i] ..) = tmp[i*bound ..)
// ============================================================================================

__global__ __launch_bounds__(1024) void amv_scan_kernel(const uint32_t* __restrict__ lens, uint32_t n,
                                                        uint64_t* __restrict__ offs, uint64_t cap,
                                                        int32_t* __restrict__ overflow) {
    __shared__ uint64_t s_part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = min(n, t * per), hi = min(n, lo + per);
    uint64_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += lens[i];
    s_part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {   // Hillis-Steele inclusive scan
        const uint64_t add = t >= d ? s_part[t - d] : 0;
        __syncthreads();
        s_part[t] += add;
        __syncthreads();
    }
    uint64_t run = s_part[t] - sum;
    for (uint32_t i = lo; i < hi; ++i) { offs[i] = run; run += lens[i]; }
    if (t == 1023) *overflow = s_part[1023] > cap ? 1 : 0;
}

__global__ __launch_bounds__(256) void amv_gather_kernel(const uint8_t* __restrict__ tmp, uint32_t bound,
                                                         uint32_t* __restrict__ lens,
                                                         const uint64_t* __restrict__ offs,
                                                         uint8_t* __restrict__ blob, uint64_t cap) {
    const uint32_t i = blockIdx.x;
    const uint32_t len = lens[i];
    const uint64_t off = offs[i];
    if (off + len > cap) {   // does not fit the caller's blob: not written, and its length says so
        if (threadIdx.x == 0) lens[i] = 0;
        return;
    }
    const uint8_t* src = tmp + (uint64_t)i * bound;
    for (uint32_t k = threadIdx.x; k < len; k += 256) blob[off + k] = src[k];
}

void launch_compact(const uint8_t* tmp, uint32_t bound, uint32_t* lens, uint32_t n,
                    uint64_t* offs, uint8_t* blob, uint64_t blob_cap, int32_t* overflow, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(amv_scan_kernel, dim3(1), dim3(1024), 0, s, lens, n, offs, blob_cap, overflow);
    hipLaunchKernelGGL(amv_gather_kernel, dim3(n), dim3(256), 0, s, tmp, bound, lens, offs, blob, blob_cap);
}

}  // namespace amv
