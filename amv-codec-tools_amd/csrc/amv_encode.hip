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
#include "amv_encode_common.h"

namespace amv {

using namespace enc;

// One wave per MCU-row segment of up to kSegMcus MCUs: stage 1 (colour conversion into LDS planes) and stage 2 (one
// 8x8 block per lane: fdct, quantise, scan order) of amv_encode_common.h, then one 128-byte line per block out.
// What amvhip_encode_coefs_dev hands out, the serial route (AMVHIP_ENTROPY_SERIAL) and the frames
// amv_encode_frame_kernel hands back go through here; sel: which frames (FrameSel, as in the decoder's rounds).
template <bool kYuv>
__global__ __launch_bounds__(kWave) void amv_forward_kernel(Source in, uint32_t n, FrameSel sel, FrameGeom g, uint32_t nseg, uint32_t per_seg,
                                                            uint32_t qbias, int16_t* __restrict__ coef) {
    __shared__ __attribute__((aligned(16))) int16_t s_planes[kPlaneSamples];
    __shared__ __attribute__((aligned(16))) uint32_t s_qmul[kQuantMulWords];
    load_quant_mul(s_qmul, threadIdx.x, kWave);
    int16_t* const s_y = s_planes;
    int16_t* const s_cb = s_planes + 16 * kPitchY;
    int16_t* const s_cr = s_cb + 8 * kPitchC;

    const uint32_t lane = threadIdx.x;
    uint32_t bid = blockIdx.x;
    const uint32_t seg = bid % nseg;
    bid /= nseg;
    const uint32_t my = bid % g.mcu_rows;
    const uint32_t m0 = seg * per_seg;
    if (m0 >= g.mcu_cols) return;          // very wide pictures: the balanced split can leave the last segment empty
    const uint32_t cnt = min(per_seg, g.mcu_cols - m0);
    const uint32_t nb = cnt * 6;
    // default: one frame per workgroup column.  A round: the launch is small (the round usually has nothing to do and
    // the host cannot know) and its workgroups walk the round's items: item i = frame list[base + i], lines in slot i.
    const uint32_t first = bid / g.mcu_rows, stride = gridDim.x / (g.mcu_rows * nseg);
    uint32_t items = n;
    if (sel.round) {
        const uint32_t have = sel.count ? *sel.count : n;
        items = have > sel.base ? min(sel.round, have - sel.base) : 0u;
    }
    for (uint32_t slot = first; slot < items; slot += stride) {
        const uint32_t f = sel.round ? (sel.list ? sel.list[sel.base + slot] : sel.base + slot) : slot;
        convert_segment<kYuv>(in, f, g, my, m0, cnt, lane, s_y, s_cb, s_cr);
        __syncthreads();
        if (lane < nb) {
            uint32_t out[32], nz_lo, nz_hi;
            transform_block(s_y, s_cb, s_cr, s_qmul, lane, qbias, out, nz_lo, nz_hi);
            uint4* dst = reinterpret_cast<uint4*>(coef + ((uint64_t)slot * g.mcus + (uint64_t)my * g.mcu_cols + m0) * 384u + lane * 64u);
#pragma unroll
            for (int i = 0; i < 8; ++i) dst[i] = make_uint4(out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]);
        }
        __syncthreads();                   // the planes are free again
    }
}

static void launch_forward_any(const Source& in, bool yuv, uint32_t n, const FrameSel& sel, uint32_t items, const FrameGeom& g,
                               uint32_t qbias, int16_t* coef, hipStream_t s) {
    if (items == 0) return;
    const uint32_t nseg = (g.mcu_cols + kSegMcus - 1) / kSegMcus;
    const uint32_t per_seg = (g.mcu_cols + nseg - 1) / nseg;      // balanced: 11 columns -> 6 + 5
    const uint64_t grid = (uint64_t)(sel.round && items > 64u ? 64u : items) * g.mcu_rows * nseg;
    if (yuv)
        hipLaunchKernelGGL(amv_forward_kernel<true>, dim3((uint32_t)grid), dim3(kWave), 0, s, in, n, sel, g, nseg, per_seg, qbias, coef);
    else
        hipLaunchKernelGGL(amv_forward_kernel<false>, dim3((uint32_t)grid), dim3(kWave), 0, s, in, n, sel, g, nseg, per_seg, qbias, coef);
}

void launch_forward(const uint8_t* pix, uint32_t pix_stride, int is_bgr, uint32_t n, const FrameSel& sel, uint32_t items,
                    const FrameGeom& g, uint32_t qbias, int16_t* coef, hipStream_t s) {
    launch_forward_any(Source{pix, pix_stride, is_bgr, YuvSource{}}, false, n, sel, items, g, qbias, coef, s);
}

void launch_forward_yuv(const YuvSource& src, uint32_t n, const FrameSel& sel, uint32_t items, const FrameGeom& g, uint32_t qbias,
                        int16_t* coef, hipStream_t s) {
    launch_forward_any(Source{nullptr, 0u, 0, src}, true, n, sel, items, g, qbias, coef, s);
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

__device__ __forceinline__ uint32_t slot_offset(uint32_t lane, uint32_t k) { return line_offset(lane, k); }

}  // namespace

__global__ __launch_bounds__(kWave) void amv_pack_kernel(
    const int16_t* __restrict__ coef, uint32_t n, FrameSel sel, uint32_t blocks_per_frame,
    const HuffEncodeImage* __restrict__ img, uint8_t* __restrict__ tmp, uint32_t bound,
    uint32_t* __restrict__ lens) {
    __shared__ uint32_t s_book[4][256];
    __shared__ __attribute__((aligned(16))) uint4 s_slots[kWave * 8];
    __shared__ uint32_t s_frame[kWave];

    // default: frames 0 .. n, a frame's lines at its own place.  A round (frames amv_encode_frame_kernel handed back):
    // items base .. base + round, item p = frame list[p] (p < *count), lines in slot p - base.
    const uint32_t lane = threadIdx.x;
    const uint32_t f0 = blockIdx.x * kWave;
    const uint32_t items = sel.round ? min(sel.round, (sel.count ? *sel.count : n) > sel.base ? (sel.count ? *sel.count : n) - sel.base : 0u) : n;
    if (f0 >= items) return;
    const uint32_t slot = f0 + lane;
    const uint32_t frame = slot < items ? (sel.round ? (sel.list ? sel.list[sel.base + slot] : sel.base + slot) : slot) : 0xffffffffu;
    s_frame[lane] = slot < items ? slot : 0xffffffffu;
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

void launch_pack(const int16_t* coef, uint32_t n, const FrameSel& sel, uint32_t items, const FrameGeom& g,
                 const HuffEncodeImage* d_img, uint8_t* tmp, uint32_t bound, uint32_t* lens, hipStream_t s) {
    if (items == 0) return;
    const uint32_t grid = (items + kWave - 1) / kWave;   // a round: upper bound, surplus groups exit at once
    hipLaunchKernelGGL(amv_pack_kernel, dim3(grid), dim3(kWave), 0, s, coef, n, sel, g.blocks, d_img, tmp, bound, lens);
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
    struct __attribute__((packed, aligned(1))) Bytes16 { uint32_t w[4]; };   // a chunk lands at any byte of the blob
    const uint32_t i = blockIdx.x;
    const uint32_t len = lens[i];
    const uint64_t off = offs[i];
    if (off + len > cap) {   // does not fit the caller's blob: not written, and its length says so
        if (threadIdx.x == 0) lens[i] = 0;
        return;
    }
    const uint8_t* src = tmp + (uint64_t)i * bound;
    uint8_t* dst = blob + off;
    const uint32_t whole = len & ~15u;
    for (uint32_t k = threadIdx.x * 16u; k < whole; k += 256u * 16u)
        *reinterpret_cast<Bytes16*>(dst + k) = *reinterpret_cast<const Bytes16*>(src + k);
    if (threadIdx.x < len - whole) dst[whole + threadIdx.x] = src[whole + threadIdx.x];
}

void launch_compact(const uint8_t* tmp, uint32_t bound, uint32_t* lens, uint32_t n,
                    uint64_t* offs, uint8_t* blob, uint64_t blob_cap, int32_t* overflow, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(amv_scan_kernel, dim3(1), dim3(1024), 0, s, lens, n, offs, blob_cap, overflow);
    hipLaunchKernelGGL(amv_gather_kernel, dim3(n), dim3(256), 0, s, tmp, bound, lens, offs, blob, blob_cap);
}

}  // namespace amv
