// amv_reconstruct.hip -- everything after the entropy stage: dequantise, inverse DCT, YCbCr->BGR,
// bottom-up store (IQtIZzBlock / Fast_IDCT / GetYUV / StoreBuffer, AmvJpeg.c:1010-1059, 754-840).
//
// One wave per MCU-row segment (<= 10 MCUs = 60 blocks = 160 pixels of 16 rows):
//   A. lane b loads block b's 64 coefficients into registers: from its dense 128-byte line, or --
//      records form -- after the wave has scattered the segment's (block, index, value) records and
//      DC values into a zeroed LDS image of the 60 blocks (16-byte granules XOR-swizzled by block so
//      that the per-lane 128-byte reads do not collide on banks);
//   B. the whole 8x8 block stays in that lane's registers: de-zig-zag is register renaming (plus
//      one select for amvlib's [3][4] table entry), dequantisation one multiply per coefficient,
//      then 8 row transforms and 8 column transforms with the reference's exact integer
//      arithmetic -- no transposition, no LDS between the passes;
//   C. results go to 16-row Y and 8-row U/V planes in LDS (one 16-byte store per block row; the
//      planes reuse the space of the records' image, which every lane has read by then);
//   D. colour conversion of a 4x2-pixel patch per lane step, two pixels per instruction in packed
//      16-bit arithmetic; the 12 bytes of each patch row go
//      straight to the frame (the lanes of a step cover consecutive 12-byte pieces of a row, so a
//      wave store is one contiguous run; the picture is stored bottom-up, AmvJpeg.c:800).
//
// Pixels of MCUs at or after a frame's first decode error are zero (AMVDec.c:283 + the reference
// stopping at the error).  Compiled with -fwrapv, as the reference's arithmetic wraps.
#include "amv_block_load.h"

namespace amv {

namespace {

struct __attribute__((aligned(4))) Px12 { uint32_t w[3]; };   // four BGR pixels

constexpr int kWave = 64;
constexpr int kSegMcus = 10;   // MCUs per wave: 60 of 64 lanes busy in the transform
// Waves per workgroup: four consecutive MCU rows of a frame.  The waves share nothing (a segment's LDS is its wave's);
// the workgroup only exists so that the chip launches a quarter as many of them -- 1.28 M single-wave workgroups per
// 160 000 frames of 160x120 cost 8 % of the kernel's time in launches (profiles/r02_launch_rate.txt).
constexpr int kRowsPerGroup = 4;   // (five where fours leave too many slots idle: launch_reconstruct)

// 8-point inverse DCT of AmvJpeg.c: idctrow (:1082-1128) when kColumn == false, idctcol
// (:1130-1175, without its final clamp) when true.  The reference's all-AC-zero shortcuts
// (:1087-1092, :1134-1140) are exact special cases of this arithmetic and are not branched on.
// The products are written per input (W1*a4 + W7*a5 instead of W7*(a4+a5) + (W1-W7)*a4): the same numbers
// modulo 2^32, which is what the reference's int arithmetic computes, but every factor is an INPUT of the
// pass and those always fit 24 bits -- a quantised coefficient times a step in the row pass, a value
// shifted right by 8 in the column pass -- so each product is one full-rate 24-bit multiply-add instead
// of a quarter-rate 32-bit multiply (a sum of two inputs can need 25 bits).  Only the two 181* products
// take arbitrary 32-bit operands.
// x * w + c with x and w inside 24 bits: one full-rate instruction, the low 32 bits of the exact result
__device__ __forceinline__ int mad24(int x, int w, int c) {
    int d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(x), "s"(w), "v"(c));
    return d;
}

template <bool kColumn>
__device__ __forceinline__ void idct8(int& v0, int& v1, int& v2, int& v3, int& v4, int& v5, int& v6, int& v7) {
    constexpr int W1 = 2841, W2 = 2676, W3 = 2408, W5 = 1609, W6 = 1108, W7 = 565;
    constexpr int kUp = kColumn ? 256 : 2048;    // <<8 / <<11
    constexpr int kBias = kColumn ? 8192 : 128;
    constexpr int kRound = kColumn ? 4 : 0;
    constexpr int kDown = kColumn ? 3 : 0;
    constexpr int kOut = kColumn ? 14 : 8;
    int a0 = v0 * kUp + kBias, a1 = v4 * kUp;
    const int i2 = v6, i3 = v2, i4 = v1, i5 = v7, i6 = v5, i7 = v3;
    int t;
    int a4 = mad24(i4, W1, mad24(i5, W7, kRound)) >> kDown;     // W7*(x4+x5) + (W1-W7)*x4
    int a5 = mad24(i4, W7, mad24(i5, -W1, kRound)) >> kDown;    // W7*(x4+x5) - (W1+W7)*x5
    int a6 = mad24(i6, W5, mad24(i7, W3, kRound)) >> kDown;     // W3*(x6+x7) - (W3-W5)*x6
    int a7 = mad24(i6, W3, mad24(i7, -W5, kRound)) >> kDown;    // W3*(x6+x7) - (W3+W5)*x7
    t = a0 + a1;
    a0 -= a1;
    int a2 = mad24(i3, W6, mad24(i2, -W2, kRound)) >> kDown;    // W6*(x3+x2) - (W2+W6)*x2
    int a3 = mad24(i3, W2, mad24(i2, W6, kRound)) >> kDown;     // W6*(x3+x2) + (W2-W6)*x3
    a1 = a4 + a6;
    a4 -= a6;
    a6 = a5 + a7;
    a5 -= a7;
    a7 = t + a3;
    t -= a3;
    a3 = a0 + a2;
    a0 -= a2;
    a2 = (181 * (a4 + a5) + 128) >> 8;
    a4 = (181 * (a4 - a5) + 128) >> 8;
    v0 = (a7 + a1) >> kOut;
    v1 = (a3 + a2) >> kOut;
    v2 = (a0 + a4) >> kOut;
    v3 = (t + a6) >> kOut;
    v4 = (t - a6) >> kOut;
    v5 = (a0 - a4) >> kOut;
    v6 = (a3 - a2) >> kOut;
    v7 = (a7 - a1) >> kOut;
}

// the quantiser steps, scan order, four to a dword: [0] luma, [1] chroma.  A lane reads its component's sixteen dwords
// once and multiplies by bytes of them (a select between two literals per coefficient cost 40 instructions a block).
struct QuantWords { uint32_t w[2][16]; };
constexpr QuantWords pack_quant() {
    QuantWords q{};
    for (int i = 0; i < 64; ++i) {
        q.w[0][i >> 2] |= (uint32_t)kQuantLuma[i] << (8 * (i & 3));
        q.w[1][i >> 2] |= (uint32_t)kQuantChroma[i] << (8 * (i & 3));
    }
    return q;
}
__device__ const QuantWords kQuantWords = pack_quant();

// iclp[] of AmvJpeg.c:1073-1080 (table spans -512..511; beyond it the reference reads out of
// bounds, defined here as saturation)
__device__ __forceinline__ int clamp_iclp(int x) { return min(max(x, -256), 255); }

// a.lo * w.lo + a.hi * w.hi + 32768 (16-bit signed halves): the three-operand form, so that the bias stays in its
// register (the builtin becomes the accumulating form plus a move to reload the accumulator)
constexpr uint32_t pair16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ uint32_t dot2_bias(uint32_t a, uint32_t w) {
    uint32_t d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(w), "v"(32768u));
    return d;
}

// two pixels of one channel: (y0 + c, y1 + c) clamped to 0..255, in bytes 0 and 1
__device__ __forceinline__ uint32_t sat_pair(uint32_t yy, uint32_t cc) {
    uint32_t sum, d;
    asm("v_pk_add_i16 %0, %1, %2" : "=v"(sum) : "v"(yy), "v"(cc));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(d) : "v"(sum));
    return d;
}

}  // namespace

// kRound: a round launch (FrameSel::round != 0), whose workgroups walk the items of the round
template <bool kRound, int kRows>
__global__ __launch_bounds__(kWave * kRows) void amv_reconstruct_kernel(
    SyncSinks in, const uint32_t* __restrict__ nmcu_ok, uint32_t n, FrameSel sel,
    FrameGeom g, PieceMap pm, uint32_t flags, uint8_t* __restrict__ out) {
    constexpr uint32_t kPitchY = kSegMcus * 16, kPitchC = kSegMcus * 8;   // int16 units
    // 7 680 bytes (+ the scatter's spare slots): first the records' image of the 60 blocks (128 bytes each), then the three planes
    __shared__ __attribute__((aligned(16))) int16_t s_all[kRows][16 * kPitchY + 2 * 8 * kPitchC + 64];
    static_assert((16 * kPitchY + 2 * 8 * kPitchC) * 2 == kSegImageBytes, "the planes reuse the image");
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t nseg = pm.nseg;
    uint32_t item0, row_group, seg;
    if (!locate_piece(pm, blockIdx.x, item0, row_group, seg)) return;
    const uint32_t my = row_group * kRows + wave;
    if (my >= g.mcu_rows) return;                   // (the whole wave)
    int16_t* const s_mem = s_all[wave];
    int16_t* const s_y = s_mem;
    int16_t* const s_u = s_mem + 16 * kPitchY;
    int16_t* const s_v = s_u + 8 * kPitchC;
    uint8_t* const s_img = reinterpret_cast<uint8_t*>(s_mem);

    uint32_t qw[16];   // this lane's quantiser steps (blocks 4 and 5 of an MCU are chroma), on their way while stage A runs
    {
        const uint4* q4 = reinterpret_cast<const uint4*>(kQuantWords.w[lane % 6u >= 4u ? 1 : 0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 q = q4[i];
            qw[4 * i] = q.x; qw[4 * i + 1] = q.y; qw[4 * i + 2] = q.z; qw[4 * i + 3] = q.w;
        }
    }
    for (uint32_t item = item0;; item += piece_stride(pm)) {
    uint32_t f, slot;
    if (!select_frame(sel, n, item, f, slot)) return;
    const uint32_t m0 = seg * kSegMcus;
    const uint32_t cnt = min((uint32_t)kSegMcus, g.mcu_cols - m0);

    const uint32_t ok = nmcu_ok[f];
    const uint32_t mcu0 = my * g.mcu_cols + m0;                       // first MCU of this segment

    // ---- A + B + C: one block per lane
    uint32_t c[32];
    bool skip;
    if (load_segment_blocks(in, f, slot, kRound, g, my * nseg + seg, g.mcu_rows * nseg, mcu0, cnt, ok, lane, s_img, c, skip)) {
        const uint32_t m = lane / 6u, k6 = lane % 6u;
        const bool chroma = k6 >= 4u;
        // IQtIZzBlock's gather (AmvJpeg.c:1035-1042): out[nat] = coef[scan(nat)] * step[scan(nat)]
        int v[64];
#pragma unroll
        for (int nat = 0; nat < 64; ++nat) {
            const int scan = kScanOfNatural[nat];
            v[nat] = coef_at(c, scan) * (int)((qw[scan >> 2] >> (8 * (scan & 3))) & 255u);
        }
        if (!(flags & kFlagZigzagFixed))     // amvlib's table reads scan position 37 at natural (3,4)
            v[kAmvlibQuirkNatural] = coef_at(c, kAmvlibQuirkScan) * (int)((qw[kAmvlibQuirkScan >> 2] >> (8 * (kAmvlibQuirkScan & 3))) & 255u);
#pragma unroll
        for (int r = 0; r < 8; ++r)
            idct8<false>(v[8 * r], v[8 * r + 1], v[8 * r + 2], v[8 * r + 3], v[8 * r + 4], v[8 * r + 5], v[8 * r + 6], v[8 * r + 7]);
#pragma unroll
        for (int col = 0; col < 8; ++col)
            idct8<true>(v[col], v[8 + col], v[16 + col], v[24 + col], v[32 + col], v[40 + col], v[48 + col], v[56 + col]);
        // GetYUV (AmvJpeg.c:754-787): one 16-byte row at a time.  The +128 of IQtIZzBlock (:1023,1047) joins the
        // chroma terms in stage D (it cannot ride through the column pass on the DC term: the reference's 32-bit
        // arithmetic wraps on absurd coefficients, and the parity tests hold the kernel to that).
        int16_t* dst = chroma ? (k6 == 4u ? s_u : s_v) + m * 8u
                              : s_y + ((k6 >> 1) * 8u) * kPitchY + m * 16u + (k6 & 1u) * 8u;
        const uint32_t pitch = chroma ? kPitchC : kPitchY;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int lo = clamp_iclp(v[8 * r + 2 * q]);
                const int hi = clamp_iclp(v[8 * r + 2 * q + 1]);
                w[q] = __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x05040100u);   // low halves: lo | hi << 16
            }
            *reinterpret_cast<uint4*>(dst + r * pitch) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    if (skip) return;   // (wave-uniform) not this launch's frame
    seg_sync();

    // ---- D: StoreBuffer (AmvJpeg.c:789-840), straight to the frame.  A lane takes a 4x2-pixel patch: the two rows
    // share their chroma samples, whose three products (:808-810) are formed once, with luma's +128 folded in:
    // (256 (y + 128) + c) >> 8 == y + ((c + 32768) >> 8) exactly.  Everything that follows fits 16 bits (|y| <= 256,
    // |c >> 8| <= 440), so two pixels go through each instruction: v_pk_add_i16 adds the pair's luma to the (doubled)
    // chroma term, v_sat_pk_u8_i16 clamps both to 0..255 (:812-827) and packs them into two bytes, and three byte
    // permutes interleave B, G, R.  MCUs that were not decoded get y = -1024, c = 0: every channel clamps to 0.
    const uint32_t vr = min(16u, g.height - my * 16u);                 // rows of this MCU row inside the picture (:798); even
    const uint32_t px = min(cnt * 16u, g.width - m0 * 16u);            // pixels of this segment inside it (:803)
    const uint32_t groups = cnt * 4u;
    const bool all_decoded = mcu0 + cnt <= ok;                         // wave-uniform: the usual case skips the per-patch test
    const uint32_t pairs = (vr + 1u) >> 1;
    // patch t = lane, lane + 64, ...: row pair i2 = t / groups, group gi = t % groups, kept up by addition
    const uint32_t step_i2 = kWave / groups, step_gi = kWave - step_i2 * groups;
    uint32_t i2 = lane / groups, gi = lane - i2 * groups;
    // one base per wave (scalar) + 32-bit byte offsets inside the frame: picture row my*16+i is destination row
    // H-1-(my*16+i) (:800)
    uint8_t* const frame = out + (uint64_t)f * g.frame_bytes;
    const uint32_t row0 = (g.height - 1u - my * 16u) * g.stride + m0 * 48u;
    for (; i2 < pairs; i2 += step_i2, gi += step_gi) {
        if (gi >= groups) { gi -= groups; ++i2; if (i2 >= pairs) break; }
        const uint32_t lc = gi * 4u;
        uint2 ya = *reinterpret_cast<const uint2*>(s_y + (2u * i2) * kPitchY + lc);
        uint2 yb = *reinterpret_cast<const uint2*>(s_y + (2u * i2 + 1u) * kPitchY + lc);
        uint32_t uu = *reinterpret_cast<const uint32_t*>(s_u + i2 * kPitchC + (lc >> 1));
        uint32_t vv = *reinterpret_cast<const uint32_t*>(s_v + i2 * kPitchC + (lc >> 1));
        if (!all_decoded && (mcu0 + (gi >> 2)) >= ok) {
            constexpr uint32_t kDark = 0xfc00fc00u;                    // two int16 of -1024
            ya.x = ya.y = yb.x = yb.y = kDark;
            uu = vv = 0u;
        }
        uint32_t cr[2], cg[2], cb[2];                                  // the chroma term of both pixels of a pair
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // (u, v) of chroma sample e as one pair of int16; each term (:808-810, +128) is one two-element dot product,
            // and bytes 1-2 of it -- the term >> 8 -- go into both halves of a dword with one byte permute
            const uint32_t uv = __builtin_amdgcn_perm(vv, uu, e ? 0x07060302u : 0x05040100u);
            const uint32_t r = dot2_bias(uv, pair16(18, 367));
            const uint32_t gg = dot2_bias(uv, pair16(-159, -220));
            const uint32_t b = dot2_bias(uv, pair16(411, -29));
            cr[e] = __builtin_amdgcn_perm(r, r, 0x02010201u);
            cg[e] = __builtin_amdgcn_perm(gg, gg, 0x02010201u);
            cb[e] = __builtin_amdgcn_perm(b, b, 0x02010201u);
        }
        if (lc >= px) continue;                                        // right of the picture
        uint32_t off = row0 - 2u * i2 * g.stride + lc * 3u;
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const uint2 yy = row ? yb : ya;
            const uint32_t b01 = sat_pair(yy.x, cb[0]), g01 = sat_pair(yy.x, cg[0]), r01 = sat_pair(yy.x, cr[0]);
            const uint32_t b23 = sat_pair(yy.y, cb[1]), g23 = sat_pair(yy.y, cg[1]), r23 = sat_pair(yy.y, cr[1]);
            const uint32_t gr = g01 | (r01 << 16);                         // G0 G1 R0 R1
            const uint32_t bg = b23 | (g23 << 16);                         // B2 B3 G2 G3
            Px12 v;                                                        // :829-831 B,G,R
            v.w[0] = __builtin_amdgcn_perm(gr, b01, 0x01060400u);          // B0 G0 R0 B1
            v.w[1] = __builtin_amdgcn_perm(bg, gr, 0x06040301u);           // G1 R1 B2 G2
            v.w[2] = __builtin_amdgcn_perm(bg, r23, 0x01070500u);          // R2 B3 G3 R3
            if (2u * i2 + (uint32_t)row >= vr) break;                      // odd picture height
            if (lc + 4u <= px) {                                           // rows are 4-byte aligned (AmvJpeg.c:1524), lc*3 is 0 mod 4
                *reinterpret_cast<Px12*>(frame + off) = v;
            } else {                                                       // the picture's last 1-3 pixels
#pragma unroll
                for (uint32_t q = 0; q < 9u; ++q)
                    if (q < (px - lc) * 3u) frame[off + q] = (uint8_t)(v.w[q >> 2] >> (8u * (q & 3u)));
            }
            off -= g.stride;
        }
    }
    if (!kRound) return;
    seg_sync();   // the planes are free again
    }   // next item of the round
}

template <int kRows>
static void launch_rows(const SyncSinks& sinks, const uint32_t* nmcu_ok, uint32_t n, const FrameSel& sel, uint32_t items, const FrameGeom& g,
                        uint32_t flags, uint8_t* out, hipStream_t s) {
    const uint32_t nseg = (g.mcu_cols + kSegMcus - 1) / kSegMcus;
    const uint32_t row_groups = (g.mcu_rows + kRows - 1) / kRows;
    PieceMap pm = make_piece_map(row_groups, nseg);
    if (sel.round) {   // a round launch is small: its workgroups walk
        const uint32_t grid = set_walkers(pm, items > 512u ? 512u : items);
        hipLaunchKernelGGL((amv_reconstruct_kernel<true, kRows>), dim3(grid), dim3(kWave * kRows), 0, s, sinks, nmcu_ok, n, sel, g, pm, flags, out);
        return;
    }
    const uint32_t most = most_items(pm);
    for (uint32_t base = 0; base < items; base += most) {
        pm.item_base = base;
        const uint32_t grid = set_walkers(pm, items - base < most ? items - base : most);
        hipLaunchKernelGGL((amv_reconstruct_kernel<false, kRows>), dim3(grid), dim3(kWave * kRows), 0, s, sinks, nmcu_ok, n, sel, g, pm, flags, out);
    }
}

void launch_reconstruct(const SyncSinks& sinks, const uint32_t* nmcu_ok, uint32_t n, const FrameSel& sel, uint32_t items,
                        const FrameGeom& g, uint32_t flags, uint8_t* out, hipStream_t s) {
    if (items == 0) return;
    // A workgroup's waves are consecutive MCU rows of one frame, and the last workgroup of a frame has idle slots unless
    // the rows divide.  Fives only where fours waste a tenth of the slots more than fives do (176x144, 9 rows: 4.80 ms
    // per 100 000 frames against 5.04): five waves do not spread evenly over a CU's four SIMDs, and 320x240 (15 rows,
    // one slot in sixteen idle in fours) takes 6.29 ms per 64 000 frames in fives against 5.87 in fours.
    const uint32_t slots4 = (g.mcu_rows + 3u) / 4u * 4u, slots5 = (g.mcu_rows + 4u) / 5u * 5u;
    const uint32_t waste4 = slots4 - g.mcu_rows, waste5 = slots5 - g.mcu_rows;
    if (10u * waste4 * slots5 >= 10u * waste5 * slots4 + slots4 * slots5) launch_rows<5>(sinks, nmcu_ok, n, sel, items, g, flags, out, s);
    else launch_rows<kRowsPerGroup>(sinks, nmcu_ok, n, sel, items, g, flags, out, s);
}

}  // namespace amv
