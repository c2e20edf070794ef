// amv_encode_par.hip -- the AMV video encoder as one kernel: pixels in, chunk out, coefficients never leave the chip.
//
// Unlike decoding, encoding has no serial chain that cannot be cut: the DC difference of a block needs only the
// previous block of its component (mjpegenc.c:390-401), and once every block's code length is known the bit position
// of every block is a prefix sum.  One workgroup of four waves per frame; a wave takes an MCU-row segment (<= 10 MCUs =
// 60 blocks, a block per lane) at a time, four segments per round:
//
//   1. colour conversion into the wave's LDS planes, fdct + quantisation of the lane's block in registers
//      (amv_encode_common.h: what amv_forward_kernel runs too); the 64 quantised coefficients go back to LDS as a
//      128-byte line (the planes' space: every lane has read its samples by then), together with a 64-bit mask of the
//      non-zero ones;
//   2. encode_block (mjpegenc.c:379-435), by SYMBOL rather than by block: a block's code is its DC symbol, one run/size
//      symbol per non-zero coefficient and an end-of-block unless coefficient 63 is set, and a segment's blocks differ
//      widely (5.6 non-zero coefficients on average at 320x240, 25 in the fullest of the 60: a lane per block would
//      have the wave walk the fullest block's length, 4x the average).  So the segment's symbols are numbered (prefix
//      sum of the per-block counts), every lane takes an equal run of consecutive numbers -- it finds the block its run
//      starts in, skips the coefficients the lane before it codes, and from there pops the lowest set bit of the
//      non-zero mask per symbol, moving on to the next block's mask when one is used up -- and codes it into its scratch
//      (eight words; a run that needs more is walked again in step 4);
//   3. prefix sum of the runs' lengths over the round (wave scan + partial sums through LDS);
//   4. the scratch is copied to its place in the round's bit string (LDS; neighbouring runs share words, hence atomic OR);
//   5. the string's whole bytes leave for memory with 00 after every FF (escape_FF :282-336); the bits of the last,
//      unfinished byte open the next round's string.
// After the last round the tail is padded with ones (ff_mjpeg_encode_stuffing :338-343) and FF D9 follows (:345-355);
// FF D8 alone goes in front (:201-204).
//
// A frame whose round does not fit the bit-string window (noise: more than ~200 bits a block on average) is handed
// back through a list; amv_forward_kernel + amv_pack_kernel (one lane per frame) take it, a round of dense
// coefficient lines at a time.  Output is byte-identical to that route's and to the CPU oracle's (tests).
#include "amv_encode_common.h"

namespace amv {

using namespace enc;

namespace {

constexpr uint32_t kWaves = 4;                    // waves (segments in flight) per frame
constexpr uint32_t kLanes = kWave * kWaves;
constexpr uint32_t kOwnWords = 8;                 // a lane's scratch: 256 bits (a block takes ~41 on the bench stream)
constexpr uint32_t kWindowWords = 1280;           // the round's bit string: 5 KB
// LDS: four regions of planes / lines (32 KB), the scratch (8 KB), the window (5 KB), the code book (4 KB), the
// symbol numbers (1 KB), the quantiser's multipliers (0.5 KB): 50.5 KB, three workgroups per CU
constexpr uint32_t kRegionBytes = kPlaneSamples * 2;

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t v, uint32_t lane, uint32_t& total) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t y = __shfl_up(x, d);
        if (lane >= (uint32_t)d) x += y;
    }
    total = __shfl(x, kWave - 1);
    return x - v;
}

// exclusive prefix sum over the workgroup's lanes (wave order, lane order); total = the sum of all.  Two barriers.
__device__ __forceinline__ uint32_t group_excl_sum(uint32_t v, uint32_t lane, uint32_t wave, uint32_t& total, uint32_t* s_part) {
    uint32_t wave_total;
    const uint32_t x = wave_excl_sum(v, lane, wave_total);
    __syncthreads();                                   // the previous sum's partials have been read
    if (lane == 0u) s_part[wave] = wave_total;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t i = 0; i < kWaves; ++i) {
        const uint32_t part = s_part[i];
        before += i < wave ? part : 0u;
        all += part;
    }
    total = all;
    return x + before;
}

// Bits appended to a bit string in LDS.  kShared: at an arbitrary bit position of the round's string (neighbouring
// blocks share words, hence the atomic OR).  kOwn: into the lane's own scratch column (word i of lane l at
// words[i * kLanes]: the bank follows the lane; words past kOwnWords are dropped -- the lane then codes its block a
// second time, into the string).
enum { kShared = 1, kOwn = 2 };
struct Emitter {
    uint32_t* words;
    uint32_t wi;        // word being filled
    uint64_t acc;       // pending bits, right aligned
    int nacc;
    uint32_t nbits;     // total length so far
};

template <int kMode>
__device__ __forceinline__ void put(Emitter& e, uint32_t entry, int extra_bits, uint32_t extra) {
    const int len = (int)(entry >> 16) + extra_bits;
    e.nbits += (uint32_t)len;
    e.acc = (e.acc << len) | ((uint64_t)(entry & 0xffffu) << extra_bits) | extra;
    e.nacc += len;
    if (e.nacc >= 32) {
        e.nacc -= 32;
        const uint32_t word = (uint32_t)(e.acc >> e.nacc);
        if (kMode == kShared) atomicOr(&e.words[e.wi], word);
        else if (e.wi < kOwnWords) e.words[e.wi * kLanes] = word;
        ++e.wi;
    }
}

// A lane's place in its segment's symbol sequence: block b (lane number of its owner), what is left of the block's
// non-zero mask, the position coded last, whether the block's DC symbol is still to come, symbols left in the block.
struct Cursor {
    uint32_t b, lo, hi, last, left;
    uint32_t k6, cls;  // block in its MCU; 256 for a chroma block (blocks 4 and 5), else 0: which pair of code books
    int dcv;           // the block's DC difference
    bool at_dc;
};

// symbols of a block: DC, one per non-zero AC coefficient, EOB unless coefficient 63 is coded (mjpegenc.c:430-431)
__device__ __forceinline__ uint32_t symbols_of(uint32_t lo, uint32_t hi) {
    return 1u + (uint32_t)__popc(lo) + (uint32_t)__popc(hi) + ((hi >> 31) ^ 1u);
}

// position of the r-th set bit (r = 0: the lowest) of hi:lo, which has more than r bits set
__device__ __forceinline__ uint32_t nth_set_bit(uint32_t lo, uint32_t hi, uint32_t r) {
    uint32_t c = (uint32_t)__popc(lo);
    const bool upper = r >= c;
    r -= upper ? c : 0u;
    uint32_t w = upper ? hi : lo, base = upper ? 32u : 0u;
#pragma unroll
    for (uint32_t s = 16; s; s >>= 1) {
        c = (uint32_t)__popc(w & ((1u << s) - 1u));
        const bool up = r >= c;
        r -= up ? c : 0u;
        w = up ? w >> s : w;
        base += up ? s : 0u;
    }
    return base;
}

// per-segment tables behind the 60 lines of a region: the blocks' masks in the space of lines 60..63, their first
// symbol's number and DC difference in s_aux
constexpr uint32_t kMaskBase = 60u * 128u;

// (everything about the block that a symbol needs is fetched here, once per block: mask, DC difference, code-book class)
__device__ __forceinline__ void load_block(Cursor& c, const uint8_t* region, const uint32_t* aux) {
    const uint2 m = *reinterpret_cast<const uint2*>(region + kMaskBase + c.b * 8u);
    c.dcv = (int)aux[c.b] >> 16;
    c.lo = m.x; c.hi = m.y; c.last = 0u; c.at_dc = true;
    c.left = symbols_of(m.x, m.y);
}

// One symbol at the cursor -> the emitter; the cursor moves on (to the next block when this one is used up).
// aux[b] = number of block b's first symbol | DC difference << 16.  The three kinds of symbol -- DC difference
// (ff_mjpeg_encode_dc :357-377), run/size of the next non-zero coefficient (:403-428), end of block (:430-431) -- differ
// in where the value comes from and which code book entry is read; the rest is one path (a wave whose lanes are at
// different kinds would otherwise execute all three).
template <int kMode>
__device__ __forceinline__ void code_symbol(Emitter& e, Cursor& c, const uint8_t* region, const uint32_t* aux, const uint32_t* book) {
    const uint32_t cls = c.cls;
    const bool coef = !c.at_dc && (c.lo | c.hi) != 0u;
    // the coefficient's position (0 when the symbol is not a coefficient: the DC's own place, read but not used)
    const bool low = c.lo != 0u;
    const uint32_t half = low ? c.lo : c.hi;
    const uint32_t k = coef ? (uint32_t)__builtin_ctz(half) + (low ? 0u : 32u) : 0u;
    // (selects, not branches: the lanes of a wave are at different kinds of symbol, and every `if` here was a pair of
    // exec-mask saves and restores in the wave's one instruction stream)
    const uint32_t lo_less = c.lo & (c.lo - 1u), hi_less = c.hi & (c.hi - 1u);
    c.lo = coef && low ? lo_less : c.lo;
    c.hi = coef && !low ? hi_less : c.hi;
    const int ac = *reinterpret_cast<const int16_t*>(region + line_offset(c.b, k));   // (the DC's own place when the symbol is no coefficient: read, not used)
    int v = c.at_dc ? c.dcv : (coef ? ac : 0);
    uint32_t run = coef ? k - c.last - 1u : 0u;
    c.last = coef ? k : c.last;
    const uint32_t* acbook = book + 512u + cls;
    while (run >= 16u) { put<kMode>(e, acbook[0xf0], 0, 0u); run -= 16u; }   // ZRL :408-411
    int mant = v;
    if (v < 0) { v = -v; mant--; }
    const int nb = 32 - __clz(v);                      // 0 for a zero DC difference and for end of block
    // DC: table cls, entry nb; coefficient: table 2 + cls, entry run << 4 | nb; end of block: table 2 + cls, entry 0
    const uint32_t entry = book[(c.at_dc ? 0u : 512u) + cls + ((run << 4) | (uint32_t)nb)];
    put<kMode>(e, entry, nb, (uint32_t)mant & ((1u << nb) - 1u));
    c.at_dc = false;
    if (--c.left == 0u) {
        ++c.b;
        c.k6 = c.k6 == 5u ? 0u : c.k6 + 1u;
        c.cls = c.k6 >= 4u ? 256u : 0u;
        load_block(c, region, aux);
    }
}

// the cursor of symbol number j of the segment (j < the segment's symbol count)
__device__ __forceinline__ Cursor seek_symbol(uint32_t j, uint32_t nb, const uint8_t* region, const uint32_t* aux) {
    Cursor c;
    uint32_t b = 0;
#pragma unroll
    for (uint32_t s = 32; s; s >>= 1)                  // the last block that starts at or before j
        if (b + s < nb && (aux[b + s] & 0xffffu) <= j) b += s;
    c.b = b;
    c.k6 = b % 6u;
    c.cls = c.k6 >= 4u ? 256u : 0u;
    load_block(c, region, aux);
    const uint32_t p = j - (aux[b] & 0xffffu);         // symbols of the block that lanes before this one code
    if (p) {
        c.at_dc = false;
        c.left -= p;
        const uint32_t skip = min(p - 1u, (uint32_t)__popc(c.lo) + (uint32_t)__popc(c.hi));   // coefficients among them (the rest: none)
        if (skip) {
            const uint32_t k = nth_set_bit(c.lo, c.hi, skip - 1u);                           // the last of them
            c.last = k;
            if (k < 32u) c.lo &= ~((2u << k) - 1u);       // (k = 31, 63: the shift wraps to 0 and everything goes)
            else { c.lo = 0u; c.hi &= ~((2u << (k - 32u)) - 1u); }
        }
    }
    return c;
}

}  // namespace

template <bool kYuv>
__global__ __launch_bounds__(kLanes, 3) void amv_encode_frame_kernel(
    Source in, uint32_t n, FrameGeom g, uint32_t nseg, uint32_t per_seg, uint32_t qbias, const HuffEncodeImage* __restrict__ img,
    uint8_t* __restrict__ tmp, uint32_t bound, uint32_t* __restrict__ lens, uint32_t* __restrict__ retry_list,
    uint32_t* __restrict__ retry_count) {
    __shared__ __attribute__((aligned(16))) uint8_t s_region[kWaves][kRegionBytes];
    __shared__ uint32_t s_own[kOwnWords * kLanes];
    __shared__ uint32_t s_bits[kWindowWords];
    __shared__ uint32_t s_book[4 * 256];
    __shared__ uint32_t s_part[kWaves];
    __shared__ __attribute__((aligned(16))) uint32_t s_qmul[kQuantMulWords];   // the quantiser's multipliers (load_quant_mul)
    __shared__ uint32_t s_aux[kWaves][kWave + 1];      // per block of a wave's segment: number of its first symbol | DC difference << 16
    __shared__ int s_lastdc[2][kWaves][4];            // DC of the last MCU's Y3, Cb, Cr of the segment a wave took, by round parity

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, tl = threadIdx.x;
    const uint32_t f = blockIdx.x;
    for (uint32_t i = tl; i < 1024u; i += kLanes) s_book[i] = (&img->code[0][0])[i];
    for (uint32_t i = tl; i < kWindowWords; i += kLanes) s_bits[i] = 0u;
    load_quant_mul(s_qmul, tl, kLanes);
    uint8_t* const region = s_region[wave];
    int16_t* const s_y = reinterpret_cast<int16_t*>(region);
    int16_t* const s_cb = s_y + 16 * kPitchY;
    int16_t* const s_cr = s_cb + 8 * kPitchC;
    uint8_t* const out = tmp + (uint64_t)f * bound;

    const uint32_t segs = g.mcu_rows * nseg;
    uint32_t out_pos = 2;                             // bytes of the chunk written so far (FF D8 first)
    __syncthreads();

    // where segment s lies, and whether there is one (very wide pictures: the balanced split can leave the last segment empty)
    auto place = [&](uint32_t s, uint32_t& my, uint32_t& m0, uint32_t& cnt) {
        my = s / nseg;
        m0 = (s - my * nseg) * per_seg;
        const bool has = s < segs && m0 < g.mcu_cols;
        cnt = has ? min(per_seg, g.mcu_cols - m0) : 0u;
        return has;
    };
    // Whole bytes of the window's first `bits` bits leave for memory, 00 after every FF (escape_FF :282-336); after the
    // frame's last round the tail is padded with ones first (ff_mjpeg_encode_stuffing :338-343).  The unfinished byte moves
    // to the head of a cleared window; returns the bits it holds.  The whole workgroup calls it, behind a barrier.
    auto flush = [&](uint32_t bits, bool final_round) -> uint32_t {
        if (final_round && (bits & 7u)) {
            if (tl == 0) atomicOr(&s_bits[bits >> 5], ((1u << (8u - (bits & 7u))) - 1u) << (24u - (bits & 24u)));
            __syncthreads();
        }
        const uint32_t nbytes = final_round ? (bits + 7u) >> 3 : bits >> 3;
        uint32_t ff_before = 0;
        for (uint32_t w0 = 0; w0 * 4u < nbytes; w0 += kLanes) {
            const uint32_t wi = w0 + tl;
            const uint32_t word = wi * 4u < nbytes ? s_bits[wi] : 0u;
            uint32_t ffs = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j)
                ffs += (wi * 4u + j < nbytes && ((word >> (24u - 8u * j)) & 0xffu) == 0xffu) ? 1u : 0u;
            uint32_t tile;
            uint32_t o = out_pos + wi * 4u + ff_before + group_excl_sum(ffs, lane, wave, tile, s_part);
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                if (wi * 4u + j >= nbytes) break;
                const uint32_t byte = (word >> (24u - 8u * j)) & 0xffu;
                out[o++] = (uint8_t)byte;
                if (byte == 0xffu) out[o++] = 0;
            }
            ff_before += tile;
        }
        out_pos += nbytes + ff_before;
        const uint32_t rest = (final_round || !(bits & 7u)) ? 0u : (s_bits[nbytes >> 2] >> (24u - 8u * (nbytes & 3u))) & 0xffu;
        __syncthreads();
        for (uint32_t i = tl; i * 32u < bits + 32u && i < kWindowWords; i += kLanes) s_bits[i] = i == 0u ? rest << 24 : 0u;
        __syncthreads();
        return final_round ? 0u : bits & 7u;
    };
    uint32_t pending = 0;                             // bits in the window
    for (uint32_t s0 = 0, round = 0; s0 < segs; s0 += kWaves, ++round) {
        const uint32_t s = s0 + wave;
        uint32_t my, m0, cnt;
        const bool has_seg = place(s, my, m0, cnt);
        const uint32_t nb = cnt * 6u;
        const bool live = lane < nb;
        const uint32_t k6 = lane % 6u;

        // ---- 1. pixels -> quantised coefficients: a line per block in LDS, its DC and non-zero mask in registers
        uint32_t nz_lo = 0, nz_hi = 0;
        int dc = 0;
        if (has_seg) {
            convert_segment<kYuv>(in, f, g, my, m0, cnt, lane, s_y, s_cb, s_cr);
            wave_sync();
            uint32_t line[32];
            if (live) transform_block(s_y, s_cb, s_cr, s_qmul, lane, qbias, line, nz_lo, nz_hi);
            wave_sync();                                          // every lane has its samples: the planes become the lines
            if (live) {
                dc = (int)(int16_t)(line[0] & 0xffffu);
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i)
                    *reinterpret_cast<uint4*>(region + lane * 128u + ((i ^ (lane & 7u)) << 4)) =
                        make_uint4(line[4 * i], line[4 * i + 1], line[4 * i + 2], line[4 * i + 3]);
                if (lane + 3u >= nb) s_lastdc[round & 1u][wave][lane + 3u - nb] = dc;   // Y3, Cb, Cr of the last MCU
            }
        }
        __syncthreads();

        // ---- 2. the predictor (mjpegenc.c:390-401: the component's previous block); the segment's symbols are numbered
        int prev;
        {
            const int up1 = __shfl_up(dc, 1), up3 = __shfl_up(dc, 3), up6 = __shfl_up(dc, 6);
            int before[3] = {0, 0, 0};                            // the segment before this one; the frame starts from 0
            if (s > 0u && has_seg) {
                const uint32_t pw = wave ? wave - 1u : kWaves - 1u, pr = wave ? round & 1u : (round & 1u) ^ 1u;
                before[0] = s_lastdc[pr][pw][0]; before[1] = s_lastdc[pr][pw][1]; before[2] = s_lastdc[pr][pw][2];
            }
            if (k6 >= 4u) prev = lane >= 6u ? up6 : before[k6 - 3u];
            else if (k6 == 0u) prev = lane >= 6u ? up3 : before[0];
            else prev = up1;
        }
        uint32_t* const aux = s_aux[wave];
        uint32_t symbols;
        {
            const uint32_t mine = live ? symbols_of(nz_lo, nz_hi) : 0u;
            const uint32_t first = wave_excl_sum(mine, lane, symbols);
            aux[lane] = first | ((uint32_t)(dc - prev) << 16);
            *reinterpret_cast<uint2*>(region + kMaskBase + lane * 8u) = make_uint2(nz_lo, nz_hi);   // (zero for a lane without a block)
        }
        wave_sync();
        // a run of consecutive symbols per lane, coded into the scratch
        const uint32_t per = (symbols + kWave - 1u) / kWave;
        const uint32_t j0 = min(symbols, lane * per), j1 = min(symbols, j0 + per);
        Cursor cur{};
        if (j0 < j1) cur = seek_symbol(j0, nb, region, aux);
        const Cursor start = cur;
        Emitter mine{s_own + tl, 0u, 0ull, 0, 0u};
        for (uint32_t t = 0; t < per; ++t)
            if (j0 + t < j1) code_symbol<kOwn>(mine, cur, region, aux, s_book);
        if (mine.nacc && mine.wi < kOwnWords) mine.words[mine.wi * kLanes] = (uint32_t)(mine.acc << (32 - mine.nacc));
        const uint32_t nbits = mine.nbits;

        // ---- 3. where the run's bits go.  The window holds the bits of several rounds (a round of the bench stream is ~10 000
        // bits of its 40 960): bytes leave only when it is more than half full or the frame ends -- one pass of step 5 (its
        // FF count is a workgroup-wide prefix sum: two barriers per tile, a fifth of a round's barriers) per two or three
        // rounds instead of per round: 0.10 ms of the kernel's 1.43 per 8 000 frames of 320x240 were that pass.
        uint32_t round_bits;
        const uint32_t before = group_excl_sum(nbits, lane, wave, round_bits, s_part);
        if (pending + round_bits > kWindowWords * 32u) {          // (the whole workgroup) what is waiting leaves first
            if (pending >= 8u) {
                __syncthreads();                                   // the rounds before have been copied
                pending = flush(pending, false);
            }
            if (pending + round_bits > kWindowWords * 32u) {      // a round that does not fit the window by itself: the one-lane-per-frame route
                if (tl == 0) retry_list[atomicAdd(retry_count, 1u)] = f;
                return;
            }
        }
        const uint32_t pos = pending + before;
        const uint32_t total = pending + round_bits;               // bits in the window after this round

        // ---- 4. ... and the copy: out of the scratch 32 bits at a time, or a second walk for a run that overflowed it
        if (nbits <= kOwnWords * 32u) {
            uint32_t from = 0, at = pos, left = nbits;
            while (left) {
                const uint32_t take = min(left, 32u);
                const uint32_t w0 = mine.words[(from >> 5) * kLanes];   // `take` bits of the scratch from bit `from` on, left-aligned
                const uint32_t piece = w0 & (uint32_t)(0xffffffff00000000ull >> take);
                atomicOr(&s_bits[at >> 5], piece >> (at & 31u));
                if ((at & 31u) + take > 32u) atomicOr(&s_bits[(at >> 5) + 1u], piece << (32u - (at & 31u)));
                from += take; at += take; left -= take;
            }
        } else {
            // the word's earlier bits belong to the run before: zeros here, OR-ed in
            Emitter e{s_bits, pos >> 5, 0ull, (int)(pos & 31u), 0u};
            cur = start;
            for (uint32_t j = j0; j < j1; ++j) code_symbol<kShared>(e, cur, region, aux, s_book);
            if (e.nacc) atomicOr(&e.words[e.wi], (uint32_t)(e.acc << (32 - e.nacc)));
        }

        // ---- 5. whole bytes out (flush, above the loop) once the window is more than half full, and at the frame's end
        const bool final_round = s0 + kWaves >= segs;
        pending = total;
        if (final_round || total > kWindowWords * 16u) {
            __syncthreads();                                       // every run of the round is in the window
            pending = flush(total, final_round);
        }
    }
    if (tl == 0) {
        out[0] = 0xff; out[1] = 0xd8;                              // SOI only, mjpegenc.c:201-204
        out[out_pos] = 0xff; out[out_pos + 1] = 0xd9;              // EOI :354
        lens[f] = out_pos + 2u;
    }
}

void launch_encode_frames(const uint8_t* pix, uint32_t pix_stride, int is_bgr, const YuvSource* yuv, uint32_t n, const FrameGeom& g,
                          uint32_t qbias, const HuffEncodeImage* d_img, uint8_t* tmp, uint32_t bound, uint32_t* lens,
                          uint32_t* retry_list, uint32_t* retry_count, hipStream_t s) {
    if (n == 0) return;
    const uint32_t nseg = (g.mcu_cols + kSegMcus - 1) / kSegMcus;
    const uint32_t per_seg = (g.mcu_cols + nseg - 1) / nseg;      // balanced: 11 columns -> 6 + 5
    if (yuv)
        hipLaunchKernelGGL(amv_encode_frame_kernel<true>, dim3(n), dim3(kLanes), 0, s, Source{nullptr, 0u, 0, *yuv}, n, g, nseg, per_seg,
                           qbias, d_img, tmp, bound, lens, retry_list, retry_count);
    else
        hipLaunchKernelGGL(amv_encode_frame_kernel<false>, dim3(n), dim3(kLanes), 0, s, Source{pix, pix_stride, is_bgr, YuvSource{}}, n, g,
                           nseg, per_seg, qbias, d_img, tmp, bound, lens, retry_list, retry_count);
}

}  // namespace amv
