// amv_reconstruct_ff.hip -- the FFmpeg-compat back half of the decoder (AMVHIP_FLAG_FFMPEG): what the patched
// FFmpeg's amv_decoder computes after its Huffman stage -- decode_block's dequantisation with the sp5x "Q60"
// tables and last_dc = 1024 (libavcodec/mjpegdec.c:376-430,805; sp5xdec.c:40,60-61), the standard zig-zag
// (dsputil.c:50-59), simple_idct_put (simple_idct.c:78-181,183-247,390-398) and the planar YUVJ420P store with
// the AMV flip (mjpegdec.c:672-677,708-716).  No colour conversion: FFmpeg hands out the three planes.
//
// Same work split as amv_reconstruct_kernel: one wave per MCU-row segment (<= 10 MCUs), one 8x8 block per lane
// held in registers through both passes; a lane then owns 8 rows of 8 output bytes and stores them itself.
// DCTELEM is int16 in the reference: every value that it stores into a block (dequantised coefficients, row-pass
// results) is wrapped to 16 bits here too, and the DC-only shortcut of idctRowCondDC -- which is NOT the general
// row formula (8*dc against (16383*dc + 1024) >> 11) -- is taken per row by a select.
#include "amv_block_load.h"

namespace amv {

namespace {

constexpr int kWave = 64;
constexpr int kSegMcus = 10;

__device__ __forceinline__ int s16(int x) { return (int)(int16_t)x; }

// idctRowCondDC, simple_idct.c:78-181
__device__ __forceinline__ void sidct_row(int& r0, int& r1, int& r2, int& r3, int& r4, int& r5, int& r6, int& r7) {
    constexpr int W1 = 22725, W2 = 21407, W3 = 19266, W4 = 16383, W5 = 12873, W6 = 8867, W7 = 4520;
    const bool dc_only = (r1 | r2 | r3 | r4 | r5 | r6 | r7) == 0;
    const int flat = s16(r0 << 3);                                  // :107-117
    int a0 = W4 * r0 + (1 << 10), a1 = a0, a2 = a0, a3 = a0;
    a0 += W2 * r2; a1 += W6 * r2; a2 -= W6 * r2; a3 -= W2 * r2;
    int b0 = W1 * r1 + W3 * r3, b1 = W3 * r1 - W7 * r3, b2 = W5 * r1 - W1 * r3, b3 = W7 * r1 - W5 * r3;
    a0 += W4 * r4 + W6 * r6; a1 += -W4 * r4 - W2 * r6; a2 += -W4 * r4 + W2 * r6; a3 += W4 * r4 - W6 * r6;
    b0 += W5 * r5 + W7 * r7; b1 += -W1 * r5 - W5 * r7; b2 += W7 * r5 + W3 * r7; b3 += W3 * r5 - W1 * r7;
    r0 = dc_only ? flat : s16((a0 + b0) >> 11);
    r7 = dc_only ? flat : s16((a0 - b0) >> 11);
    r1 = dc_only ? flat : s16((a1 + b1) >> 11);
    r6 = dc_only ? flat : s16((a1 - b1) >> 11);
    r2 = dc_only ? flat : s16((a2 + b2) >> 11);
    r5 = dc_only ? flat : s16((a2 - b2) >> 11);
    r3 = dc_only ? flat : s16((a3 + b3) >> 11);
    r4 = dc_only ? flat : s16((a3 - b3) >> 11);
}

// idctSparseColPut, simple_idct.c:183-247 (its `if (col[..])` guards only skip additions of zero); the clip is
// ff_cropTbl (0..255 over -1024..1279; saturation beyond, where the reference reads outside its table)
__device__ __forceinline__ void sidct_col(int& c0, int& c1, int& c2, int& c3, int& c4, int& c5, int& c6, int& c7) {
    constexpr int W1 = 22725, W2 = 21407, W3 = 19266, W4 = 16383, W5 = 12873, W6 = 8867, W7 = 4520;
    int a0 = W4 * (c0 + 32), a1 = a0, a2 = a0, a3 = a0;              // (1 << 19) / W4 = 32
    a0 += W2 * c2; a1 += W6 * c2; a2 -= W6 * c2; a3 -= W2 * c2;
    int b0 = W1 * c1 + W3 * c3, b1 = W3 * c1 - W7 * c3, b2 = W5 * c1 - W1 * c3, b3 = W7 * c1 - W5 * c3;
    a0 += W4 * c4; a1 -= W4 * c4; a2 -= W4 * c4; a3 += W4 * c4;
    b0 += W5 * c5; b1 -= W1 * c5; b2 += W7 * c5; b3 += W3 * c5;
    a0 += W6 * c6; a1 -= W2 * c6; a2 += W2 * c6; a3 -= W6 * c6;
    b0 += W7 * c7; b1 -= W5 * c7; b2 += W3 * c7; b3 -= W1 * c7;
    // (written as an explicit v_med3: hipcc 7.2 folds pairs of `clamp(x >> 20, 0, 255)` into gfx950's
    // v_ashr_pk_u8_i32, whose results differed from the plain arithmetic on MI355X in the parity tests)
    auto clip = [](int x) {
        int d;
        asm("v_med3_i32 %0, %1, 0, %2" : "=v"(d) : "v"(x >> 20), "s"(255));
        return d;
    };
    c0 = clip(a0 + b0); c1 = clip(a1 + b1); c2 = clip(a2 + b2); c3 = clip(a3 + b3);
    c4 = clip(a3 - b3); c5 = clip(a2 - b2); c6 = clip(a1 - b1); c7 = clip(a0 - b0);
}

}  // namespace

// out: per frame, Y plane width*height, then Cb and Cr of ((w+1)/2) x ((h+1)/2), rows tight
// kRound: a round launch (FrameSel::round != 0), whose workgroups walk the items of the round
template <bool kRound>
__global__ __launch_bounds__(kWave) void amv_reconstruct_yuv_kernel(
    SyncSinks in, const uint32_t* __restrict__ nmcu_ok, uint32_t n, FrameSel sel, FrameGeom g, PieceMap pm, uint64_t yuv_frame_bytes,
    uint8_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t s_img[kSegImageBytes + 128];   // + a spare slot per lane (load_segment_blocks)
    const uint32_t lane = threadIdx.x;
    uint32_t item0, my, seg;   // (one MCU row per workgroup: the row group is the row)
    if (!locate_piece(pm, blockIdx.x, item0, my, seg)) return;
    for (uint32_t item = item0;; item += piece_stride(pm)) {
    uint32_t f, slot;
    if (!select_frame(sel, n, item, f, slot)) return;
    const uint32_t m0 = seg * kSegMcus;
    const uint32_t cnt = min((uint32_t)kSegMcus, g.mcu_cols - m0);
    const uint32_t ok = nmcu_ok[f];
    const uint32_t mcu0 = my * g.mcu_cols + m0;

    uint32_t c[32];
    bool skip;
    if (load_segment_blocks(in, f, slot, kRound, g, my * pm.nseg + seg, g.mcu_rows * pm.nseg, mcu0, cnt, ok, lane, s_img, c, skip)) {
    const uint32_t m = lane / 6u, k6 = lane % 6u;
    const bool chroma = k6 >= 4u;
    // decoded: MCUs before the frame's first error -- or, AMVHIP_FLAG_FFMPEG_KEEP (ok counts blocks then), every whole block
    // before it: what mjpeg_decode_scan has put into the picture when decode_block fails (mjpegdec.c:699-716)
    const bool keep = in.ok_in_blocks != 0u;
    const bool decoded = keep ? (mcu0 + m) * 6u + k6 < ok : mcu0 + m < ok;

    // decode_block's dequantisation (mjpegdec.c:388-390,417,424): out[natural] = (DCTELEM)(level * q); the DC
    // arrives as the running sum of differences, FFmpeg keeps 1024 + q0 * that sum (:805) -- equal modulo 2^16,
    // which is all an int16 store keeps
    int v[64];
#pragma unroll
    for (int nat = 0; nat < 64; ++nat) {
        const int scan = kScanOfNatural[nat];
        const int step = chroma ? (int)kQ60Chroma[scan] : (int)kQ60Luma[scan];
        v[nat] = s16(coef_at(c, scan) * step + (nat == 0 ? 1024 : 0));
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
        sidct_row(v[8 * r], v[8 * r + 1], v[8 * r + 2], v[8 * r + 3], v[8 * r + 4], v[8 * r + 5], v[8 * r + 6], v[8 * r + 7]);
#pragma unroll
    for (int col = 0; col < 8; ++col)
        sidct_col(v[col], v[8 + col], v[16 + col], v[24 + col], v[32 + col], v[40 + col], v[48 + col], v[56 + col]);

    // mjpeg_decode_scan's placement (:672-677,708-716): component rows run upwards from
    // start = v * (8 * mb_height - ((height / 2) & 7)) - 1; rows the formula sends outside the plane are dropped
    const uint32_t cw = (g.width + 1u) >> 1, ch = (g.height + 1u) >> 1;
    const uint32_t pw = chroma ? cw : g.width, ph = chroma ? ch : g.height;
    const int vs = chroma ? 1 : 2;
    const int start = vs * (int)(8u * g.mcu_rows - ((g.height >> 1) & 7u)) - 1;
    const uint32_t sx = (chroma ? m0 + m : 2u * (m0 + m) + (k6 & 1u)) * 8u;
    const int sy = (int)((chroma ? my : 2u * my + (k6 >> 1)) * 8u);
    uint8_t* plane = out + (uint64_t)f * yuv_frame_bytes +
                     (chroma ? (uint64_t)g.width * g.height + (k6 == 5u ? (uint64_t)cw * ch : 0ull) : 0ull);
    if (sx < pw) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int p = start - (sy + i);
        if (p < 0 || p >= (int)ph) continue;
        if (keep && !decoded) continue;      // the caller's buffer is left as it was (FFmpeg: whatever the picture held)
        uint32_t lo = 0, hi = 0;
        if (decoded) {   // MCUs at or after a frame's first error stay zero
            lo = (uint32_t)v[8 * i] | ((uint32_t)v[8 * i + 1] << 8) | ((uint32_t)v[8 * i + 2] << 16) | ((uint32_t)v[8 * i + 3] << 24);
            hi = (uint32_t)v[8 * i + 4] | ((uint32_t)v[8 * i + 5] << 8) | ((uint32_t)v[8 * i + 6] << 16) | ((uint32_t)v[8 * i + 7] << 24);
        }
        uint8_t* d = plane + (uint64_t)p * pw + sx;
        if (sx + 8u <= pw && ((uintptr_t)d & 7u) == 0u) {
            *reinterpret_cast<uint2*>(d) = make_uint2(lo, hi);
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 8u; ++j)
                if (sx + j < pw) d[j] = (uint8_t)((j < 4u ? lo >> (8u * j) : hi >> (8u * (j - 4u))) & 0xffu);
        }
    }
    }
    }
    if (!kRound) return;
    seg_sync();   // the image is free again
    }   // next item of the round
}

// true when every row of every plane is reached by mjpegdec.c:672-677's formula (then the kernel writes each
// output byte and no clearing pass is needed); heights of 16k+14 or 16k+15 leave the bottom rows untouched
bool yuv_store_covers_planes(const FrameGeom& g) {
    const int start_y = 2 * (int)(8u * g.mcu_rows - ((g.height >> 1) & 7u)) - 1;
    const int start_c = (int)(8u * g.mcu_rows - ((g.height >> 1) & 7u)) - 1;
    return start_y >= (int)g.height - 1 && start_c >= (int)((g.height + 1u) >> 1) - 1;
}

void launch_reconstruct_yuv(const SyncSinks& sinks, const uint32_t* nmcu_ok, uint32_t n, const FrameSel& sel, uint32_t items,
                            const FrameGeom& g, uint64_t yuv_frame_bytes, uint8_t* out, hipStream_t s) {
    if (items == 0) return;
    const uint32_t nseg = (g.mcu_cols + kSegMcus - 1) / kSegMcus;
    PieceMap pm = make_piece_map(g.mcu_rows, nseg);   // (the launch order of amv_reconstruct_kernel: amv_block_load.h)
    if (sel.round) {
        const uint32_t grid = set_walkers(pm, items > 512u ? 512u : items);
        hipLaunchKernelGGL(amv_reconstruct_yuv_kernel<true>, dim3(grid), dim3(kWave), 0, s, sinks, nmcu_ok, n, sel, g, pm, yuv_frame_bytes, out);
        return;
    }
    const uint32_t most = most_items(pm);
    for (uint32_t base = 0; base < items; base += most) {
        pm.item_base = base;
        const uint32_t grid = set_walkers(pm, items - base < most ? items - base : most);
        hipLaunchKernelGGL(amv_reconstruct_yuv_kernel<false>, dim3(grid), dim3(kWave), 0, s, sinks, nmcu_ok, n, sel, g, pm, yuv_frame_bytes, out);
    }
}

}  // namespace amv
