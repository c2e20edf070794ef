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
//   amv_forward_kernel  data parallel: one wave per MCU-row segment; colour conversion into
//                       LDS planes, row and column DCT passes one 8-point transform per lane,
//                       quantise + zig-zag, 128-byte lines out.
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

__global__ __launch_bounds__(kWave) void amv_forward_kernel(
    const uint8_t* __restrict__ pix, uint32_t pix_stride, int is_bgr, uint32_t n, FrameGeom g,
    uint32_t nseg, uint32_t qbias, int16_t* __restrict__ coef) {
    __shared__ __attribute__((aligned(16))) int16_t s_y[16 * kSegMcus * 16];
    __shared__ __attribute__((aligned(16))) int16_t s_cb[8 * kSegMcus * 8];
    __shared__ __attribute__((aligned(16))) int16_t s_cr[8 * kSegMcus * 8];
    __shared__ __attribute__((aligned(16))) int16_t s_d[kSegMcus * 6 * 64];   // after the row pass
    __shared__ __attribute__((aligned(16))) int16_t s_o[kSegMcus * 6 * 64];   // quantised, scan order
    __shared__ int s_qmat[2][64];                                              // by scan position

    constexpr uint32_t kPitchY = kSegMcus * 16, kPitchC = kSegMcus * 8;
    const uint32_t lane = threadIdx.x;
    uint32_t bid = blockIdx.x;
    const uint32_t seg = bid % nseg;
    bid /= nseg;
    const uint32_t my = bid % g.mcu_rows;
    const uint32_t f = bid / g.mcu_rows;
    const uint32_t m0 = seg * kSegMcus;
    const uint32_t cnt = min((uint32_t)kSegMcus, g.mcu_cols - m0);
    const uint32_t nb = cnt * 6;
    const uint32_t w = g.width, h = g.height, cw = w >> 1, ch = h >> 1;
    const uint8_t* src = pix + (uint64_t)f * pix_stride * h;
    const int ro = is_bgr ? 2 : 0, bo = is_bgr ? 0 : 2;

    // ff_convert_matrix (mpegvideo_enc.c:80-91) with qscale 8: (1<<22)/(8*Q); [0] unused (DC)
    s_qmat[0][lane] = (int)((1u << 22) / (8u * kQuantLuma[lane]));
    s_qmat[1][lane] = (int)((1u << 22) / (8u * kQuantChroma[lane]));

    // luma plane of this segment.  Bitstream row k is picture row h-1-k (mjpegenc.c:462-467);
    // rows and columns outside the picture repeat the nearest edge sample.
    for (uint32_t t = lane; t < 16 * cnt * 16; t += kWave) {
        const uint32_t i = t / (cnt * 16), j = t % (cnt * 16);
        const uint32_t k = my * 16 + i, c = m0 * 16 + j;
        const uint32_t sy = k < h ? h - 1 - k : 0, sx = c < w ? c : w - 1;
        const uint8_t* p = src + (uint64_t)sy * pix_stride + sx * 3u;
        const int r = p[ro], gg = p[1], b = p[bo];
        s_y[i * kPitchY + j] = (int16_t)(((306 * r + 601 * gg + 117 * b + 512) >> 10) - 128);  // RGB_TO_Y colorspace.h:78-80
    }
    // chroma planes: 2x2 sums (imgconvert_template.h:668-695, RGB_TO_U/V colorspace.h:82-88, shift 2)
    for (uint32_t t = lane; t < 8 * cnt * 8; t += kWave) {
        const uint32_t i = t / (cnt * 8), j = t % (cnt * 8);
        const uint32_t k = my * 8 + i, c = m0 * 8 + j;
        const uint32_t sy = k < ch ? ch - 1 - k : 0, sx = c < cw ? c : cw - 1;
        const uint8_t* p0 = src + (uint64_t)(2 * sy) * pix_stride + (2 * sx) * 3u;
        const uint8_t* p1 = p0 + pix_stride;
        const int r1 = p0[ro] + p0[3 + ro] + p1[ro] + p1[3 + ro];
        const int g1 = p0[1] + p0[4] + p1[1] + p1[4];
        const int b1 = p0[bo] + p0[3 + bo] + p1[bo] + p1[3 + bo];
        s_cb[i * kPitchC + j] = (int16_t)((-173 * r1 - 339 * g1 + 512 * b1 + 2047) >> 12);         // +128 -128
        s_cr[i * kPitchC + j] = (int16_t)((512 * r1 - 429 * g1 - 83 * b1 + 2047) >> 12);
    }
    __syncthreads();

    // row pass (get_pixels + row_fdct): one (block, row) per lane
    for (uint32_t t = lane; t < nb * 8; t += kWave) {
        const uint32_t blk = t >> 3, r = t & 7u, m = blk / 6u, k6 = blk % 6u;
        const int16_t* in = k6 < 4 ? s_y + ((k6 >> 1) * 8u + r) * kPitchY + m * 16u + (k6 & 1u) * 8u
                                   : (k6 == 4 ? s_cb : s_cr) + r * kPitchC + m * 8u;
        int d[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) d[c] = in[c];
        fdct8<0>(d);
        int16_t* o = s_d + blk * 64u + r * 8u;
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = (int16_t)d[c];   // DCTELEM is 16 bit (dsputil.h:38)
    }
    __syncthreads();

    // column pass + dct_quantize_c: one (block, column) per lane
    const int bias = (int)(qbias << 14);   // intra_quant_bias << (QMAT_SHIFT - QUANT_BIAS_SHIFT), :3679
    for (uint32_t t = lane; t < nb * 8; t += kWave) {
        const uint32_t blk = t >> 3, c = t & 7u, cls = (blk % 6u) >= 4u ? 1u : 0u;
        int d[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) d[r] = s_d[blk * 64u + r * 8u + c];
        fdct8<1>(d);
        int16_t* o = s_o + blk * 64u;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t scan = kScanOfNatural[r * 8 + c];
            const int x = (int16_t)d[r];
            int a;
            if (r == 0 && c == 0) {        // DC: (block[0] + q/2) / q with q = 8 * step, :3670-3676
                const int q = cls ? 8 * kQuantChroma[0] : 8 * kQuantLuma[0];
                a = (abs(x) + (q >> 1)) / q;
            } else {                       // AC: (bias + |level|) >> QMAT_SHIFT, :3702-3712
                const int level = x * s_qmat[cls][scan];
                a = (abs(level) + bias) >> 22;
            }
            o[scan] = (int16_t)(x < 0 ? -a : a);
        }
    }
    __syncthreads();

    uint4* dst = reinterpret_cast<uint4*>(coef + ((uint64_t)f * g.mcus + (uint64_t)my * g.mcu_cols + m0) * 384u);
    const uint4* so = reinterpret_cast<const uint4*>(s_o);
    for (uint32_t i = lane; i < nb * 8; i += kWave) dst[i] = so[i];
}

void launch_forward(const uint8_t* pix, uint32_t pix_stride, int is_bgr, uint32_t n,
                    const FrameGeom& g, uint32_t qbias, int16_t* coef, hipStream_t s) {
    if (n == 0) return;
    const uint32_t nseg = (g.mcu_cols + kSegMcus - 1) / kSegMcus;
    const uint64_t grid = (uint64_t)n * g.mcu_rows * nseg;
    hipLaunchKernelGGL(amv_forward_kernel, dim3((uint32_t)grid), dim3(kWave), 0, s, pix, pix_stride,
                       is_bgr, n, g, nseg, qbias, coef);
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
                                                         const uint32_t* __restrict__ lens,
                                                         const uint64_t* __restrict__ offs,
                                                         uint8_t* __restrict__ blob, uint64_t cap) {
    const uint32_t i = blockIdx.x;
    const uint32_t len = lens[i];
    const uint64_t off = offs[i];
    if (off + len > cap) return;
    const uint8_t* src = tmp + (uint64_t)i * bound;
    for (uint32_t k = threadIdx.x; k < len; k += 256) blob[off + k] = src[k];
}

void launch_compact(const uint8_t* tmp, uint32_t bound, const uint32_t* lens, uint32_t n,
                    uint64_t* offs, uint8_t* blob, uint64_t blob_cap, int32_t* overflow, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(amv_scan_kernel, dim3(1), dim3(1024), 0, s, lens, n, offs, blob_cap, overflow);
    hipLaunchKernelGGL(amv_gather_kernel, dim3(n), dim3(256), 0, s, tmp, bound, lens, offs, blob, blob_cap);
}

}  // namespace amv
