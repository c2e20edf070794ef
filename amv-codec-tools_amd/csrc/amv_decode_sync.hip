// amv_decode_sync.hip -- entropy stage with parallelism INSIDE a frame (gfx950).
//
// The scan of a frame is one serial chain (AmvJpeg.c:1244-1287: no restart markers, DC predictors
// and bit position carried from the first MCU to the last), which is why amv_huffman_kernel gives
// a frame to a single lane.  That leaves a 10 000-frame stream with 157 waves on a 1 024-SIMD
// chip, each crawling at one dependent table look-up per symbol.  Here the chain is cut with the
// self-synchronisation of Huffman codes:
//
//   amv_unstuff_kernel (one wave per frame)
//      copies the scan into a workspace as big-endian words with the byte after every FF removed
//      (ReadByte, AmvJpeg.c:1061-1071), so that a decoder state is just a bit index.
//
//   amv_huffman_sync2_kernel<L> (records form) and amv_huffman_sync_kernel<L> (dense form: coefficient lines)
//   (independent waves sharing the tables, up to 13 per workgroup, one workgroup per CU -- dense: 10 x 2; L lanes per
//   frame, 64/L frames per wave; waves take tasks of 64/L frames from an atomic queue)
//   0. a lane reads its part of the stream through a 16-word window in LDS that it refills from the
//      workspace (L2) with 16-byte loads whenever any lane of the wave has used its window up, so a
//      wave needs 4 KB of LDS whatever the frame size (dense form: the frames' coefficient lines are
//      zeroed first);
//   1. the bit stream is cut into L equal subsequences; lane i walks subsequence i from a GUESSED
//      state (its first bit, "the DC symbol of block 0 comes next") up to the first symbol boundary
//      past its end and remembers the state it arrives in: (bit, index in block, block in MCU).
//      A guessed start that meets an invalid code or an over-long run does not stop (that would
//      stall every lane to its right until exact states arrive one lane per round): it slips a
//      bit / closes the block and carries on.  The true decoder of a valid stream never takes
//      those branches, and a stream with a real error is caught in pass 4, which is strict;
//   2. the lanes find the state the decoder is in where their share starts.  Lane 0's start is exact; a lane whose left
//      neighbour is final takes that lane's arrival as its own final start.  Dense form (amv_huffman_sync_kernel): every
//      lane takes its left neighbour's arrival state as its start state and walks again if that changed -- after round r
//      lanes 0..r are exact, and in practice wrong starts fall into step with the true decoder with a chance of 0.8 per
//      1 750 bits (the slow part is the luma/chroma phase of the MCU, a 1-in-6 guess), so the loop ends early (worst case
//      L-1 rounds: still correct).  Records form (amv_huffman_sync2_kernel, round 5): the lanes REMEMBER their walks
//      (start -> arrival per share), finality is a prefix scan over the lanes' "which of my walks starts where yours
//      arrived" maps, and lanes with nothing to walk try candidate starts for the shares that are still open: the worst
//      frame of 10 000 takes 6 rounds instead of 9 (the kernel has the details; tools/sim_sync.c is a CPU model of it).
//      These walks only look at symbol lengths and index advances;
//      the number of lanes per frame follows the batch size (huffman_sync_lanes): as many as keep every task resident at
//      once, because the launch lasts as long as one task does;
//   3. a prefix sum of "blocks finished per lane" gives every lane its first block number;
//   3'. records form: a prefix sum of "value-carrying symbols per lane" gives every lane its first record;
//   4. one strict pass decodes values and writes them: records form, one 32-bit word per DC coefficient
//      and per non-zero AC coefficient (index, block modulo 64, value) in stream order, staged per lane
//      in LDS and stored as aligned 32-byte pieces, + where the records of every MCU-row segment lie -- what
//      amv_reconstruct_kernel scatters into LDS; dense form, 2-byte stores into the frame's zeroed
//      coefficient lines.  DC prediction (ycoef/ucoef/vcoef, AmvJpeg.c:1200-1221) is a running sum per
//      component: each lane's sums count from its own start,
//   5. a prefix sum over the lanes' totals gives every lane its three bases: records form, they go into
//      the frame's lane table and the reader adds them; dense form, the lane adds them to the DC values
//      it stored itself.
//   The records form walks in arithmetic on a state chosen for it (fast_skip, fast_stride: "One lane per frame"
//   below says how); the dense form keeps the select-based walks (walk_skip, walk_write).
//
//   amv_huffman_fast_kernel (one lane per frame, records form: what a batch that fills the chip that way gets)
//      passes 1-3 and 5 fall away, pass 4 is fast_stride without a share limit; records leave as whole lines.
//
// Statuses equal the serial kernel's bit for bit (tests): the first error on the true path stops
// the frame, nmcu_ok counts whole MCUs before it, TRUNCATED compares consumed with stored bits.
// Chunks larger than the per-frame workspace window, or with a run of FF bytes longer than the 8-byte
// look-back of the unstuffer (never in a valid stream), are queued for amv_huffman_kernel.
#include <atomic>

#include "amv_kernels.h"

namespace amv {

namespace {

constexpr int kWave = 64;
constexpr uint32_t kRingWords = 16;       // LDS words per lane: the window of its stream a lane works in
// Records are staged per lane in LDS and leave as whole, aligned pieces of eight (a 4-byte store per record from 64
// lanes into 64 different lines was written back to HBM as partial lines several times over: 9.8 GB of writes per
// 160 000 frames for 0.7 GB of records, and those stores were what the walk waited on).  A lane's staging column holds
// two pieces.
constexpr uint32_t kDummyRecord = 0x8000u;    // bit 15: a filler no block owns
constexpr uint32_t kNever = 0xffffffffu;
constexpr uint32_t kTableBytes = (4u << kLut1Bits) * 2u + (4u << kM2Bits) * 2u;   // m1 + m2 of HuffDecodeImage, contiguous

struct State {
    uint32_t p;   // bit index in the unstuffed stream
    uint32_t k;   // next coefficient index in the block, 0 = the DC symbol comes next
    uint32_t k6;  // block inside the MCU, 0..5
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// exclusive prefix sum inside aligned groups of kWidth lanes; total = the group's sum
template <int kWidth>
__device__ __forceinline__ uint32_t seg_excl_sum(uint32_t v, uint32_t sub, uint32_t& total) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < kWidth; d <<= 1) {
        const uint32_t y = __shfl_up(x, d, kWidth);
        if (sub >= (uint32_t)d) x += y;
    }
    total = __shfl(x, kWidth - 1, kWidth);
    return x - v;
}

// inclusive prefix sum over the wave's 64 lanes in six DPP adds (no LDS round trips: the shuffle form above is a
// chain of six ds_bpermute, and the unstuffer does one scan per 256 bytes, each waiting for the one before)
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t x) {
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);   // row_bcast:31 into rows 2 and 3
    return v;
}

// lane i gets lane i - 1's value (lane 0: 0)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, true);   // wave_shr:1
}

}  // namespace

// =============================================================================================
// unstuffing
// =============================================================================================

// A byte is dropped when an odd run of FF bytes precedes it (ReadByte swallows the byte behind every FF it delivers).
// A lane takes 16 bytes of a 1 KB tile: FF flags of its bytes (SWAR: three instructions per word + a multiply that
// gathers them), the neighbour's last flag through a lane shift; unless the wave sees two FF bytes in a row (never in a
// valid stream) "dropped" is simply "the byte before is FF", else the general rule runs with the lane's and its
// neighbour's flags as look-back.  One six-step DPP prefix sum per tile says where a lane's kept bytes go; each word's
// kept bytes are packed with one byte permute (selector by keep mask from a 16-entry LDS table), shifted to their
// place and OR-ed into a ring of words in LDS; whole words then leave as coalesced dword stores.
// (One word per lane and tile, a shuffle-based prefix sum and a byte store per kept byte took 0.47 ms per 160 000
// frames; the same with DPP and LDS words 0.34.)
__global__ __launch_bounds__(256) void amv_unstuff_kernel(
    const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offs,
    const uint32_t* __restrict__ lens, uint32_t n, const uint32_t* __restrict__ ws_line, uint32_t* __restrict__ ws,
    uint32_t* __restrict__ ws_bytes, uint32_t* __restrict__ retry_list, uint32_t* __restrict__ retry_count) {
    constexpr uint32_t kRing = 512;                 // output words being put together, per wave (a tile adds <= 256)
    constexpr uint32_t kTile = kWave * 16u;         // bytes
    __shared__ uint32_t s_words[4][kRing];
    __shared__ uint32_t s_pack[16];                 // keep mask -> selector that packs the kept bytes low, zeros above
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t frame = blockIdx.x * 4u + wave;
    if (threadIdx.x < 16u) {
        uint32_t sel = 0, k = 0;
        for (uint32_t j = 0; j < 4u; ++j)
            if (threadIdx.x & (1u << j)) sel |= j << (8u * k++);
        for (; k < 4u; ++k) sel |= 0x0cu << (8u * k);
        s_pack[threadIdx.x] = sel;
    }
    uint32_t* const ring = s_words[wave];
    for (uint32_t i = lane; i < kRing; i += kWave) ring[i] = 0u;
    __syncthreads();
    if (frame >= n) return;
    uint64_t off = offs[frame];
    uint32_t len = lens[frame];
    if (off > blob_bytes) { off = blob_bytes; len = 0; }
    if ((uint64_t)len > blob_bytes - off) len = (uint32_t)(blob_bytes - off);
    const uint32_t line0 = ws_line[frame];                 // the frame's window: 16-byte pieces ws_line[frame] .. ws_line[frame + 1]
    uint32_t* const out = ws + (uint64_t)line0 * 4u;
    const uint32_t line1 = ws_line[frame + 1u];
    // (the layout is monotone; a window that ends before it starts is still read as empty: the frame goes to the retry
    // list and nothing is written)
    const uint64_t cap_bytes = line1 > line0 ? (uint64_t)(line1 - line0) * 16u : 0u;

    const uint32_t mis = (uint32_t)(off & 3u);
    const uint8_t* base = blob + (off - mis);
    const uint64_t guard = blob_bytes - (off - mis);
    const uint32_t first = mis + 2u, end = mis + len;      // data bytes [first, end), relative to base
    bool retry = len > 2u && (len - 2u) > cap_bytes;        // does not fit its window (chunks that overlap in the blob: the layout ran out)
    uint32_t total = 0, flushed = 0;                        // bytes kept so far; whole words that have left
    if (!retry) {
        uint32_t carry = 0;   // FF flags of the 16 bytes in front of the tile (bit 15: the byte right in front)
        // four tiles' loads go out together, so a frame costs one or two trips to memory
        constexpr uint32_t kBurst = 4;
        for (uint32_t b0 = 0; b0 < end && !retry; b0 += kBurst * kTile) {
            uint4 wv[kBurst];
#pragma unroll
            for (uint32_t q = 0; q < kBurst; ++q) {
                const uint64_t bo = (uint64_t)b0 + q * kTile + lane * 16u;
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                if (bo < end) {
                    if (bo + 16u <= guard) {
                        struct __attribute__((packed, aligned(4))) W4 { uint32_t x[4]; };
                        const W4 v = *reinterpret_cast<const W4*>(base + bo);
                        w[0] = v.x[0]; w[1] = v.x[1]; w[2] = v.x[2]; w[3] = v.x[3];
                    } else {
                        for (uint32_t j = 0; j < 16u; ++j) if (bo + j < guard) w[j >> 2] |= (uint32_t)base[bo + j] << (8u * (j & 3u));
                    }
                }
                wv[q] = make_uint4(w[0], w[1], w[2], w[3]);
            }
#pragma unroll
            for (uint32_t q = 0; q < kBurst; ++q) {
            const uint32_t t0 = b0 + q * kTile;            // the tile's first byte
            if (t0 >= end) break;
            const uint32_t p0 = t0 + lane * 16u;           // this lane's first byte
            const uint32_t w[4] = {wv[q].x, wv[q].y, wv[q].z, wv[q].w};
            // which of this lane's sixteen bytes are FF: bit 7 of a byte of t is set iff the byte is FF; the multiply
            // gathers bits 7, 15, 23, 31 into four neighbouring bits
            uint32_t ff = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                const uint32_t t = ((w[j] & 0x7f7f7f7fu) + 0x01010101u) & w[j] & 0x80808080u;
                ff |= ((((t >> 7) * 0x00204081u) >> 21) & 15u) << (4u * j);
            }
            uint32_t valid = 0xffffu;                                     // bytes inside [first, end)
            if (t0 == 0u || t0 + kTile > end) {                           // (wave-uniform) the chunk's first and last tile
                const uint32_t lo = first > p0 ? min(first - p0, 16u) : 0u, hi = end > p0 ? min(end - p0, 16u) : 0u;
                valid = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                ff &= ~((1u << lo) - 1u);                                 // bytes in front of the data never count
            }
            uint32_t back = wave_shr1(ff);                                // the 16 flags in front of this lane's bytes
            if (lane == 0) back = carry;
            const uint32_t before = ((ff << 1) | (back >> 15)) & 0xffffu; // bit j: the byte in front of byte j is FF
            uint32_t keep;
            // two FF bytes in a row anywhere in the tile (or across its front edge)?
            if (!__any((ff & before) != 0u || (lane == 0 && (carry >> 14) == 3u))) {
                keep = valid & ~before;
            } else {   // the general rule: a byte is dropped iff an odd run of FF bytes precedes it
                const uint32_t hist = back | (ff << 16);                  // byte j of this lane at bit 16 + j
                keep = 0;
                bool deep = false;
#pragma unroll
                for (uint32_t j = 0; j < 16u; ++j) {
                    const uint32_t run = (uint32_t)__builtin_clz(~(hist << (16u - j)) | 1u);   // FF bytes immediately in front of byte j (<= 16 + j)
                    deep = deep || run >= 16u + j;                        // the run may reach past the look-back
                    keep |= (!(run & 1u) ? 1u : 0u) << j;
                }
                keep &= valid;
                if (__any(deep)) { retry = true; break; }
            }
            carry = __builtin_amdgcn_readlane(ff, kWave - 1);
            const uint32_t cnt = (uint32_t)__builtin_popcount(keep);
            const uint32_t upto = wave_incl_sum(cnt);
            const uint32_t tile_total = __builtin_amdgcn_readlane(upto, kWave - 1);
            uint32_t d = total + upto - cnt;                              // where this lane's kept bytes go
            // every word's kept bytes, packed low, in stream order -> their place in output words d / 4 and d / 4 + 1
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                const uint32_t k4 = (keep >> (4u * j)) & 15u;
                const uint32_t packed = __builtin_amdgcn_perm(0u, w[j], s_pack[k4]);
                const uint32_t sh = (d & 3u) * 8u, wd = d >> 2;
                atomicOr(&ring[wd & (kRing - 1u)], packed << sh);
                atomicOr(&ring[(wd + 1u) & (kRing - 1u)], (packed >> 1) >> (31u - sh));
                d += (uint32_t)__builtin_popcount(k4);
            }
            total += tile_total;
            wave_sync();                                                  // every lane's bytes are in the ring
            // whole words leave, big-endian inside the word
            const uint32_t whole = total >> 2;
            for (uint32_t x = flushed + lane; x < whole; x += kWave) {
                const uint32_t v = ring[x & (kRing - 1u)];
                ring[x & (kRing - 1u)] = 0u;
                out[x] = __builtin_amdgcn_perm(0u, v, 0x00010203u);
            }
            flushed = whole;
            wave_sync();                                                  // ... and the words that left are clear again
            }
        }
        // the last, partial word and zeros up to the next 16-byte boundary: the decoder copies whole 16-byte pieces
        if (!retry) {
            wave_sync();
            const uint32_t x = flushed + lane;
            if (x < ((total + 15u) & ~15u) >> 2) out[x] = __builtin_amdgcn_perm(0u, ring[x & (kRing - 1u)], 0x00010203u);
        }
    }
    if (lane == 0) {
        ws_bytes[frame] = retry ? kNever : total;
        if (retry) retry_list[atomicAdd(retry_count, 1u)] = frame;
    }
}

// =============================================================================================
// walks.  Table entries (HuffDecodeImage::m1/m2): bits 0-4 code length + magnitude bits (0 = no
// such code), bits 5-10 how far the coefficient index moves (run + 1; 1 for a DC symbol; 63 for
// end-of-block, which with k >= 1 always reaches 64), bits 11-14 magnitude bits; in m1 bit 15
// means "the code is longer than 9 bits, see m2".
//
// Codes longer than 9 bits sit at the top of the 9-bit prefix space (canonical codes ascend with
// length): in every table they are the prefixes [507, 512) or fewer, i.e. they all begin with six
// one-bits, and a code is at most 16 bits long.  m2 is therefore indexed by the ten bits behind those
// six, 1024 entries per table: its address follows from the window alone, both levels are read
// together and the right one is selected afterwards -- no dependent second LDS round trip, no branch.
// =============================================================================================

namespace {

// A lane's view of its frame's unstuffed words.  The frame stays in the global workspace (L2); the lane
// works in a window of kRingWords = 16 words kept in LDS -- word x of lane l at ring[(x % 16) * 64 + l], so
// the bank depends on the lane only and no two lanes ever collide.  The next eight words are always on
// their way from memory in registers (requested one service earlier, so they have arrived when they are
// needed) and replace the older half of the window once the lane has left it.
// (Keeping whole frames in LDS, as an earlier version did, capped a CU at 40 frames of 160x120 and sent
// frames larger than the LDS window to the serial kernel.)
struct Stream {
    const uint32_t* g;   // the frame's words, 16-byte aligned; words at or past nwords read as zero
    uint32_t nwords;     // multiple of 4 (the unstuffer zeroes the tail of the last 16-byte piece)
    uint32_t* ring;      // this lane's column of the wave's ring
    uint32_t hi;         // the ring holds the lane's words [hi - 16, hi); multiple of 8
    uint4 pf0, pf1;      // words [hi, hi + 8), requested
};

// A walk may run kStride symbols between services: a symbol is at most 27 bits (16 code + 11 magnitude),
// so 10 of them move the read index by at most 9 words, and a service leaves every lane >= 9 words.
constexpr int kStride = 10;
// the writing walk runs 8 symbols between services: at most 8 new records then join at most 7 staged ones
constexpr int kStrideWrite = 8;

__device__ __forceinline__ uint32_t ring_word(const Stream& s, uint32_t x) {
    return s.ring[(x & (kRingWords - 1u)) * kWave];
}

__device__ __forceinline__ uint4 stream_piece(const Stream& s, uint32_t x) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (x < s.nwords) v = *reinterpret_cast<const uint4*>(s.g + x);
    return v;
}

__device__ __forceinline__ void ring_put(const Stream& s, uint32_t x, const uint4& v) {   // x: multiple of 4
    uint32_t* d = s.ring + (x & (kRingWords - 1u)) * kWave;
    d[0] = v.x; d[kWave] = v.y; d[2 * kWave] = v.z; d[3 * kWave] = v.w;
}

__device__ __forceinline__ void stream_request(Stream& s) {
    s.pf0 = stream_piece(s, s.hi);
    s.pf1 = stream_piece(s, s.hi + 4u);
}

// the requested words take the place of the older half of the window; the next eight are requested
__device__ __forceinline__ void stream_advance(Stream& s) {
    ring_put(s, s.hi, s.pf0);
    ring_put(s, s.hi + 4u, s.pf1);
    s.hi += 8u;
    stream_request(s);
}

// start of a walk at word `from`: the window holds words [from & ~7, +16)
__device__ __forceinline__ void stream_open(Stream& s, uint32_t from) {
    const uint32_t lo = from & ~7u;
    const uint4 a = stream_piece(s, lo), b = stream_piece(s, lo + 4u), c = stream_piece(s, lo + 8u), d = stream_piece(s, lo + 12u);
    ring_put(s, lo, a);
    ring_put(s, lo + 4u, b);
    ring_put(s, lo + 8u, c);
    ring_put(s, lo + 12u, d);
    s.hi = lo + kRingWords;
    stream_request(s);
}

// between strides: a lane whose read index is within 9 words of the window's end moves the window on
// (once, with words that arrived long ago; a second time, waiting, only after a run of maximal symbols)
__device__ __forceinline__ void stream_service(Stream& s, uint32_t widx) {
    while (widx + 9u > s.hi) stream_advance(s);
}

// tab = table number << kLut1Bits; v = the next 32 bits of the stream
__device__ __forceinline__ uint32_t lookup(const uint16_t* __restrict__ m1, const uint16_t* __restrict__ m2,
                                           uint32_t tab, uint32_t v) {
    const uint32_t e1 = m1[tab + (v >> (32 - kLut1Bits))];
    const uint32_t e2 = m2[(tab << 1) + ((v >> (32 - 6 - kM2Bits)) & ((1u << kM2Bits) - 1u))];
    return (e1 & 0x8000u) ? e2 : e1;
}

// table of the symbol that comes next: k = 0 -> DC, else AC; blocks 4 and 5 of an MCU are chroma
__device__ __forceinline__ uint32_t table_of(uint32_t k, uint32_t k6) {
    return ((k ? 2u : 0u) + (k6 >= 4u ? 1u : 0u)) << kLut1Bits;
}

// The bit window of a walk: two consecutive stream words hi:lo and bo in [1, 32], the number of bits of hi that are
// used up; the next 32 bits of the stream are ({hi, lo} >> (32 - bo)) -- one v_alignbit_b32.  nextw is the word
// after lo, widx the index of the word after that (fetched from the lane's ring while the look-ups are in flight).
struct Window {
    uint32_t hi, lo, nextw, widx, bo;
};

__device__ __forceinline__ Window window_open(Stream& w, uint32_t p) {
    const uint32_t b = p & 31u, w0 = p >> 5;
    stream_open(w, w0);
    Window x;
    x.bo = b ? b : 32u;                                   // a word boundary: "all of the word before is used up"
    const uint32_t base = b ? w0 : w0 - 1u;               // (then hi is never looked at; any ring word will do)
    x.hi = ring_word(w, base);
    x.lo = ring_word(w, base + 1u);
    x.nextw = ring_word(w, base + 2u);
    x.widx = base + 3u;
    return x;
}

__device__ __forceinline__ uint32_t window_bits(const Window& x) {
    return __builtin_amdgcn_alignbit(x.hi, x.lo, 32u - x.bo);
}

// `used` (<= 27) bits consumed; cand = the ring word at x.widx
__device__ __forceinline__ void window_consume(Window& x, uint32_t used, uint32_t cand) {
    x.bo += used;
    const bool step = x.bo > 32u;
    x.hi = step ? x.lo : x.hi;
    x.lo = step ? x.nextw : x.lo;
    x.nextw = step ? cand : x.nextw;
    x.widx += step ? 1u : 0u;
    x.bo -= step ? 32u : 0u;
}

// Speculative walk from `s` while s.p < limit: where symbols start and how the block position
// moves, nothing else.  Returns the number of blocks finished.  A stride of kStride symbols is straight-line code
// (a lane that is past its limit goes through the motions without moving), so that the scheduler can overlap the
// table look-up of one symbol with the bookkeeping of the one before.
__device__ __forceinline__ uint32_t walk_skip(Stream& w, const uint16_t* __restrict__ m1,
                                              const uint16_t* __restrict__ m2, State& s, uint32_t limit) {
    uint32_t p = s.p, k = s.k, k6 = s.k6, nblk = 0;
    bool active = p < limit;
    if (!__ballot(active)) return 0u;
    Window x = window_open(w, active ? p : 0u);
    uint32_t tab = table_of(k, k6);
    while (__ballot(active) != 0ull) {
        if (active) stream_service(w, x.widx);
#pragma unroll
        for (int it = 0; it < kStride; ++it) {
            const uint32_t cand = ring_word(w, x.widx);        // the word after nextw, in flight with the look-ups
            const uint32_t e = lookup(m1, m2, tab, window_bits(x));
            const uint32_t used = active ? max(e & 31u, 1u) : 0u;   // nonsense under a guessed start: slip one bit
            const uint32_t kn = k + (active ? (e >> 5) & 63u : 0u);
            p += used;
            window_consume(x, used, cand);
            const bool end = kn >= 64u;                        // end of block, a full block, or an over-long run
            k = end ? 0u : kn;
            k6 = end ? (k6 == 5u ? 0u : k6 + 1u) : k6;
            nblk += end ? 1u : 0u;
            tab = table_of(k, k6);
            active = active && p < limit;
        }
    }
    s.p = p; s.k = k; s.k6 = k6;
    return nblk;
}

struct WriteResult {
    uint32_t err;        // kStFormat / kStOverrun when the walk hit a real error
    uint32_t err_blk;    // absolute block the error hit
    uint32_t stop_p;     // bits consumed when the walk ended (FORMAT: incl. the reference's 17-bit give-up)
    bool done;           // the frame's last block was finished here
    uint32_t dc_first;   // first block whose DC symbol this lane decoded, and how many follow
    uint32_t dc_count;
    int sum[3];          // the lane's DC differences added up per component (Y, Cb, Cr)
};

// The MCU-row segments amv_reconstruct_kernel works in: kSegMcus MCUs, the last one of a row shorter.
struct SegGeom {
    uint32_t mcu_cols;    // MCUs per row
    uint32_t per_row;     // segments per row
    uint32_t count;       // segments per frame
};
constexpr uint32_t kSegMcus = 10;   // = amv_reconstruct.hip's

// stage: this lane's column of the wave's staging area; record slot q of lane l lives at dword q * 64 + l (the
// bank depends on the lane only, the address is one shift-and-add)
template <uint32_t kFlush>
__device__ __forceinline__ void stage_put(uint32_t* stage, uint32_t pos, uint32_t word) {
    stage[(pos & (2u * kFlush - 1u)) * kWave] = word;
}

// records [from, from + kFlush) of this lane -> rec (from is a multiple of kFlush; rec + from is aligned to the burst)
template <uint32_t kFlush>
__device__ __forceinline__ void stage_flush(const uint32_t* stage, uint32_t* __restrict__ rec, uint32_t from, uint32_t rec_cap) {
    const uint32_t* p = stage + (from & (2u * kFlush - 1u)) * kWave;
    uint4 v[kFlush / 4u];
#pragma unroll
    for (uint32_t q = 0; q < kFlush / 4u; ++q)
        v[q] = make_uint4(p[(4u * q) * kWave], p[(4u * q + 1u) * kWave], p[(4u * q + 2u) * kWave], p[(4u * q + 3u) * kWave]);
    if (from + kFlush <= rec_cap) {   // never past the frame's record space; an overfull frame is redone densely
        // (plain stores: streaming "nontemporal" ones took 3.96 ms against 2.70 and wrote more, 3.77 GB against 3.35)
#pragma unroll
        for (uint32_t q = 0; q < kFlush / 4u; ++q) *reinterpret_cast<uint4*>(rec + from + 4u * q) = v[q];
    }
}

// The strict, writing walk (HufBlock / DecodeElement, AmvJpeg.c:842-974) from an exact state into the frame's
// (zeroed) coefficient lines: the dense form.  A stride of kStride symbols is straight-line code -- every lane goes
// through every step, a lane that has stopped (end of frame, error, end of its share) without moving -- with
// conditional 2-byte stores.  (The records form has walks of its own: fast_stride, below.)
__device__ __forceinline__ WriteResult walk_write(Stream& w, const uint16_t* __restrict__ m1,
                                                  const uint16_t* __restrict__ m2, State s, uint32_t limit,
                                                  uint32_t blk, uint32_t blocks_per_frame, int16_t* __restrict__ coef) {
    WriteResult r{0u, 0u, 0u, false, 0u, 0u, {0, 0, 0}};
    uint32_t p = s.p, k = s.k, k6 = s.k6;
    r.dc_first = blk + (k ? 1u : 0u);
    int s0 = 0, s1 = 0, s2 = 0;
    bool alive = p < limit;
    uint32_t stop = 0;           // why the lane stopped: 1 invalid code, 2 index past 63, 3 the frame's last block is done
    Window x = window_open(w, p);
    uint32_t tab = table_of(k, k6);
    while (__ballot(alive) != 0ull) {
        if (alive) stream_service(w, x.widx);
#pragma unroll
        for (int it = 0; it < kStride; ++it) {
            const uint32_t cand = ring_word(w, x.widx);
            const uint32_t v = window_bits(x);
            const uint32_t e = lookup(m1, m2, tab, v);
            const uint32_t used = e & 31u, size = (e >> 11) & 15u, adv = (e >> 5) & 63u;
            const bool bad = used == 0u;                         // no code matches: FUNC_FORMAT_ERROR, AmvJpeg.c:887
            const bool isdc = k == 0u;
            const bool iseob = !isdc && adv == 63u;              // end of block (:959-964)
            const uint32_t idx = k + adv - 1u;                   // AC: where the coefficient goes
            const bool over = !bad && !isdc && !iseob && idx > 63u;   // the reference writes out of bounds here (:967-969)
            // magnitude bits -> value (:924-933): the `size` bits behind the code; size 0 gives 0
            const uint32_t mag = __builtin_amdgcn_ubfe(v, 32u - used, size);
            const uint32_t full = (1u << size) - 1u;
            const int val = (int)mag - (mag <= (full >> 1) ? (int)full : 0);
            // consume (a code that matches nothing consumes nothing; the reference has read 17 bits by then)
            const uint32_t eat = (alive && !bad) ? used : 0u;
            p += eat;
            window_consume(x, eat, cand);
            const bool good = alive && !bad && !over;
            const bool dc = isdc && good;                        // DC difference (:945-951), summed per component (:1200-1221)
            const int t = (k6 < 4u ? s0 : (k6 == 4u ? s1 : s2)) + val;
            s0 = (dc && k6 < 4u) ? t : s0;
            s1 = (dc && k6 == 4u) ? t : s1;
            s2 = (dc && k6 == 5u) ? t : s2;
            r.dc_count += dc ? 1u : 0u;
            const bool ac = good && !isdc && !iseob && size != 0u;
            if (dc) coef[(uint64_t)blk * 64u] = (int16_t)t;      // the sum counts from this lane's start; pass 5 adds the base
            if (ac) coef[(uint64_t)blk * 64u + idx] = (int16_t)val;
            const uint32_t newk = isdc ? 1u : idx + 1u;
            const bool block_end = good && (iseob || (!isdc && newk == 64u));
            k = good ? (block_end ? 0u : newk) : k;
            k6 = block_end ? (k6 == 5u ? 0u : k6 + 1u) : k6;
            blk += block_end ? 1u : 0u;
            tab = table_of(k, k6);
            const bool finished = block_end && blk == blocks_per_frame;
            const uint32_t why = bad ? 1u : (over ? 2u : (finished ? 3u : 0u));
            stop = (alive && why) ? why : stop;
            alive = alive && !why && p < limit;
        }
    }
    // the state froze where the lane stopped
    r.err = stop == 1u ? kStFormat : (stop == 2u ? kStOverrun : 0u);
    r.err_blk = blk;
    r.stop_p = stop == 1u ? p + 17u : p;
    r.done = stop == 3u;
    r.sum[0] = s0; r.sum[1] = s1; r.sum[2] = s2;
    return r;
}

}  // namespace

// Outputs of the records form (SyncOut::rec != nullptr), all per frame: its lines of rec (rec_line), seg_start[segs + 1][2]
// {from, to} bounds of each MCU-row segment's records (SyncSinks; entries of segments the decoder never reached, and
// the last one, hold the total twice), lane_tab[L] = {first block whose DC the lane decoded, DC base Y, Cb, Cr} (lanes right of the
// one that met the frame's end or first error: first block ~0), rec_count = total, or ~0 when the frame was handed to
// the serial kernel, whose output is dense coefficient lines.
struct SyncOut {
    int16_t* coef;
    uint32_t* rec;
    const uint32_t* rec_line;
    uint32_t* seg_start;
    uint32_t* lane_tab;
    SegGeom sg;
    uint32_t lane_stride;    // entries per frame in lane_tab (>= the lanes of the launch that writes it)
    uint32_t* rec_count;
    uint32_t* retry_list;    // frames for amv_huffman_kernel
    uint32_t* retry_count;
    uint32_t ok_in_blocks;   // nmcu_ok counts whole blocks instead of whole MCUs (SyncSinks)
};

// The dense form (coefficient lines: amvhip_huffman_decode_dev's output).
// dynamic LDS: [ m1 4 KB | m2 8 KB | per wave: ring of kRingWords words per lane ]
// With a list, the kernel decodes frames list[0 .. *list_count) (surplus waves do nothing).
template <int L>
__global__ __launch_bounds__(kWave* 16) void amv_huffman_sync_kernel(
    const uint32_t* __restrict__ ws, const uint32_t* __restrict__ ws_bytes, uint32_t n,
    const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count,
    uint32_t blocks_per_frame, const uint32_t* __restrict__ ws_line,
    const HuffDecodeImage* __restrict__ img, SyncOut out, int32_t* __restrict__ status,
    uint32_t* __restrict__ nmcu_ok, uint32_t* __restrict__ queue, unsigned long long* __restrict__ stats) {
    constexpr int kFrames = kWave / L;   // frames per wave
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    const uint16_t* m1 = reinterpret_cast<const uint16_t*>(s_mem);
    const uint16_t* m2 = m1 + (4 << kLut1Bits);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t slot = lane / L, sub = lane % L;

    {   // tables, shared by the waves of the workgroup
        const uint4* src = reinterpret_cast<const uint4*>(&img->m1[0][0]);
        uint4* dst = reinterpret_cast<uint4*>(s_mem);
        for (uint32_t i = threadIdx.x; i < kTableBytes / 16u; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();   // the only workgroup-wide barrier; from here the waves are on their own
    uint32_t* ring = reinterpret_cast<uint32_t*>(s_mem + kTableBytes) + wave * (kRingWords * kWave) + lane;
    if (list) n = *list_count;
    const uint32_t ntasks = (n + kFrames - 1) / kFrames;
    // Tasks (kFrames frames each) are handed out through a counter: a wave that finishes early -- the
    // number of synchronisation rounds varies a lot between frames -- takes the next one instead of
    // idling until the slowest wave of the grid is done.  Every wave leaves once the counter passes ntasks.
    for (;;) {
    uint32_t task = 0;
    if (lane == 0) task = atomicAdd(queue, 1u);
    task = __shfl(task, 0);
    if (task >= ntasks) return;

    const bool timing = stats != nullptr && lane == 0;   // optional phase clock (amvhip_entropy_stats)
    unsigned long long tc[6] = {0, 0, 0, 0, 0, 0};
    if (timing) tc[0] = clock64();

    const uint32_t idx = task * kFrames + slot;
    const uint32_t frame = idx < n ? (list ? list[idx] : idx) : kNever;
    const uint32_t total = frame != kNever ? ws_bytes[frame] : kNever;   // kNever: handed to the serial kernel
    const bool live = total != kNever;
    const uint32_t fsafe = live ? frame : 0u;
    int16_t* const coef = out.coef + (uint64_t)fsafe * blocks_per_frame * 64u;
    const uint32_t valid_bits = live ? total * 8u : 0u;
    if (live) {   // the frame's coefficient lines start as zeros
        uint4* z = reinterpret_cast<uint4*>(coef);
        for (uint32_t i = sub; i < blocks_per_frame * 8u; i += L) z[i] = make_uint4(0, 0, 0, 0);
    }
    Stream win{ws + (uint64_t)ws_line[fsafe] * 4u, live ? ((total + 15u) >> 4) * 4u : 0u, ring, 0u, make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};

    // ---- 1/2. speculative walks until every lane's start state equals its neighbour's arrival
    uint32_t S = ((valid_bits + L - 1) / L + 31u) & ~31u;   // bits per lane, a whole number of words
    if (S < 64u) S = 64u;
    // the last lane of a frame has no right neighbour to feed: it only runs in pass 4
    const uint32_t limit = (!live || sub == L - 1) ? 0u : (sub + 1u) * S;
    State entry{sub * S, 0u, 0u}, arrive = entry;
    if (timing) tc[1] = clock64();
    uint32_t my_blocks = walk_skip(win, m1, m2, arrive, limit);
    if (timing) tc[2] = clock64();
    uint32_t rounds = 0;
    const uint64_t seg = L == 64 ? ~0ull : (((1ull << (L & 63)) - 1ull) << (slot * L));
    for (int round = 0; round < L; ++round) {
        State left;
        left.p = __shfl_up(arrive.p, 1, L);
        left.k = __shfl_up(arrive.k, 1, L);
        left.k6 = __shfl_up(arrive.k6, 1, L);
        const bool changed = live && sub != 0 && (left.p != entry.p || left.k != entry.k || left.k6 != entry.k6);
        const uint64_t who = __ballot(changed);
        if (!who) break;
        if (who & seg) ++rounds;
        if (changed) {
            entry = left;
            arrive = left;
            my_blocks = walk_skip(win, m1, m2, arrive, limit);
        }
    }
    if (timing) tc[3] = clock64();

    // ---- 3. first block of every lane (the last lane has not walked: its count is 0, it is last)
    uint32_t all_blocks;
    const uint32_t blk0 = seg_excl_sum<L>(my_blocks, sub, all_blocks);

    // ---- 4. the strict, writing pass.  Lanes left of the frame's end (or first error) are exact;
    // whatever a lane to the right of it does is ignored below.
    __builtin_amdgcn_s_waitcnt(0);   // the zeroing stores have landed before the sparse ones go out
    WriteResult wr{0u, 0u, 0u, false, 0u, 0u, {0, 0, 0}};
    if (live && blk0 < blocks_per_frame)
        wr = walk_write(win, m1, m2, entry, sub == L - 1 ? kNever : limit, blk0, blocks_per_frame, coef);
    if (timing) tc[4] = clock64();
    const uint64_t stop_mask = __ballot(wr.done || wr.err != 0u) & seg;
    uint32_t st = 0, good_blocks = blocks_per_frame;
    if (stop_mask) {
        const int stop_lane = __builtin_ctzll(stop_mask);   // leftmost = the true path
        const uint32_t e = __shfl(wr.err, stop_lane);
        const uint32_t eb = __shfl(wr.err_blk, stop_lane);
        const uint32_t sp = __shfl(wr.stop_p, stop_lane);
        st = e;
        if (e) good_blocks = eb;
        if (sp > valid_bits) st |= kStTruncated;
    } else {
        st = kStFormat; good_blocks = 0;   // unreachable: the last lane runs until the frame ends or fails
    }

    // ---- 5. DC prediction: the sums of the lanes to the left are a lane's base, which it adds to the DCs it stored (a
    // lane reads back only its own stores -- same thread, ordered by the wait -- so no cache is in play).
    uint32_t tot;
    const int by = (int)seg_excl_sum<L>((uint32_t)wr.sum[0], sub, tot);
    const int bu = (int)seg_excl_sum<L>((uint32_t)wr.sum[1], sub, tot);
    const int bv = (int)seg_excl_sum<L>((uint32_t)wr.sum[2], sub, tot);
    __builtin_amdgcn_s_waitcnt(0);
    for (uint32_t j = 0; j < wr.dc_count; ++j) {
        const uint32_t b = wr.dc_first + j, c6 = b % 6u;
        const int base = c6 < 4u ? by : (c6 == 4u ? bu : bv);
        int16_t* q = coef + (uint64_t)b * 64u;
        *q = (int16_t)(*q + base);
    }
    if (timing) {
        tc[5] = clock64();
        for (int q = 0; q < 5; ++q) atomicAdd(&stats[4 + q], tc[q + 1] - tc[q]);
        atomicAdd(&stats[9], 1ull);
    }
    if (live && sub == 0) {
        if (stats) {   // optional: how hard the synchronisation worked (amvhip_entropy_stats)
            atomicAdd(&stats[0], 1ull);
            atomicAdd(&stats[1], (unsigned long long)rounds);
            atomicMax(&stats[2], (unsigned long long)rounds);
        }
        status[frame] = (int32_t)st;
        nmcu_ok[frame] = out.ok_in_blocks ? good_blocks : good_blocks / 6u;
    }
    }   // next task
}

// =============================================================================================
// One lane per frame (a batch that fills the chip that way: no speculation, no lane table).  The symbol step of
// walk_write spends most of its ~90 instructions on selects -- which table, which component's DC sum, is the lane
// alive, did the block end, did a segment start -- and the kernel is bound by VALU issue.  Here the step is
// arithmetic on a state chosen for it, under half as many instructions:
//   * bit position: t = bits consumed - 1.  The window's two words are read from the lane's ring at (t >> 5) (slot 16
//     mirrors slot 0, so the pair is one ds_read2) and the next 32 bits are alignbit(hi, lo, ~t): no window registers
//     to shift along, no "did we cross a word";
//   * tables: 32-bit entries (HuffDecodeImage::fast) in four 8 KB regions; the region's address bits come from the
//     state with shifts and ANDs, both levels are read and OR-ed; the entry's fields sit in bytes of their own, so
//     adding "bits used" to the position or "advance" to the index is one instruction on a byte of the entry;
//   * end of block: bit 6 of (index + advance) -- the end-of-block symbol advances by 192 --, as a mask dc = -bit: the
//     index is ANDed with ~dc, the blocks-to-go counter, the block-in-MCU counter (counting down, wrapping through an
//     unsigned min) and the table choice move by it;
//   * DC prediction: the three running sums live in LDS, the component's slot address follows from the block-in-MCU
//     counter, every step adds (value AND "this is a DC symbol");
//   * a record is written to the next staging slot at every step and counted by adding the entry's "carries a value"
//     bit (which sits at the stride of a staging slot);
//   * the records leave stride-aligned: a stride of eight symbols owns eight record positions (8 * stride + slot), its
//     records in front, fillers behind, and every fourth stride all 64 lanes store their 32 positions as one 128-byte
//     line each (see the loop);
//   * the frame's end: once the blocks-to-go counter reaches 0 the looked-up entry is ANDed to zero: the lane stands
//     still, consuming and emitting nothing, until the wave's slowest lane is done;
//   * errors raise no branch: the entries are OR-ed along (one of their bits says "no such code") and a running unsigned
//     minimum of (index + advance - 65) notices an over-long run; a stride of eight symbols that raised a flag is walked
//     again from its saved start state, one symbol at a time, to find where the walk stops (fast_replay);
//   * where an MCU-row segment starts is looked at once per stride: its entry gets the record counts before and after
//     the stride in which its first DC symbol came (SyncSinks::seg_start: bounds, which the reader's block test tightens).
// LDS addresses are formed with OR where the layout allows (regions aligned to their size).
// =============================================================================================

namespace {

constexpr uint32_t kFastRegion = kFastWords * 4u;            // bytes per table in LDS
constexpr uint32_t kFastTableBytes = 4u * kFastRegion;
constexpr uint32_t kServicePace = 4u;                         // strides between a wave's stream requests (fast_service_paced)
// ... of the speculative walks when a frame has sixteen lanes or more -- launches a wave or two per SIMD deep: nothing
// young is in flight to wait for, and a lane that runs dry asks and waits on the spot -- requests in every stride:
// 0.434 -> 0.426 ms per 10 000 frames of 160x120; with eight lanes per frame (32 000 frames of 320x240, a chip full of waves)
// that costs 7 %, so those keep the pace of the one-lane kernel
template <int L> constexpr uint32_t skip_pace() { return L >= 16 ? 1u : kServicePace; }
constexpr uint32_t kSlot = kWave * 4u;                       // bytes between a lane's consecutive slots
constexpr uint32_t kFastRingBytes = (kRingWords + 1u) * kSlot;   // slot 16 mirrors slot 0
constexpr uint32_t kFastSumBytes = 4u * kSlot;                   // three sums; aligned to its size
constexpr uint32_t fast_stage_bytes(uint32_t flush) { return 2u * flush * kSlot; }
constexpr uint32_t fast_per_wave(uint32_t flush) { return fast_stage_bytes(flush) + kFastSumBytes + kFastRingBytes; }
constexpr uint32_t fast_waves(uint32_t flush) {
    const uint32_t w = (160u * 1024u - kFastTableBytes) / fast_per_wave(flush);
    return w > 16u ? 16u : w;
}

struct FastState {
    uint32_t t;      // bits consumed - 1
    uint32_t k;      // next coefficient index, 0 = the DC symbol comes next
    uint32_t j;      // 5 - block inside the MCU
    uint32_t togo6;  // (block - blocks per frame) << 6: negative until the frame's last block is done (a record's block
                     // field is bits 6-11 of it: the reader knows the frame's block count)
    uint32_t rp8;    // next free record << 8 (a staging slot is 256 bytes)
    uint32_t dc;     // ~0 when the next symbol is a DC symbol, else 0
};

// LDS by byte address: the kernel's dynamic LDS is all the LDS it has, so it starts at 0 (checked on entry) and the
// addresses the walk computes go to the instruction as they are
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_load(uint32_t a) { return *(const lds_u32*)(uintptr_t)a; }
__device__ __forceinline__ void lds_store(uint32_t a, uint32_t v) { *(lds_u32*)(uintptr_t)a = v; }
// (x << 3) + y in one instruction (left alone the compiler forms the shift, an AND and an add)
__device__ __forceinline__ uint32_t shl3_add(uint32_t x, uint32_t y) {
    uint32_t d;
    asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(d) : "v"(x), "v"(y));
    return d;
}

__device__ __forceinline__ void fast_ring_put(const Stream& s, uint32_t x, const uint4& v) {   // x: multiple of 4
    uint32_t* d = s.ring + (x & (kRingWords - 1u)) * kWave;
    d[0] = v.x; d[kWave] = v.y; d[2 * kWave] = v.z; d[3 * kWave] = v.w;
    if ((x & (kRingWords - 1u)) == 0u) s.ring[kRingWords * kWave] = v.x;
}

__device__ __forceinline__ void fast_open(Stream& s) {   // the ring holds words [0, 16)
    const uint4 a = stream_piece(s, 0u), b = stream_piece(s, 4u), c = stream_piece(s, 8u), d = stream_piece(s, 12u);
    fast_ring_put(s, 0u, a);
    fast_ring_put(s, 4u, b);
    fast_ring_put(s, 8u, c);
    fast_ring_put(s, 12u, d);
    s.hi = kRingWords;
    stream_request(s);
}

// Before a stride: eight symbols of <= 27 bits starting in word w read words w .. w + 7 (and the one before w, whose
// bits are all behind the position, at a word boundary: any content will do), so the ring must reach word w + 8.
// A lane puts its requested words into the ring when it needs them (any stride), but ASKS
// for the next eight only in every kPace-th stride.  The wait in front of a put is s_waitcnt vmcnt(0): it is the wave's
// and in order, so it waits for whatever any lane asked for last -- with requests in every stride that is always one
// stride ago, HBM's latency under load; with requests kept to every kPace-th stride most strides find nothing young in
// flight.  have: the requested words are (being) fetched.  A lane that needs words it has not asked for yet (more than
// eight words consumed between two request strides) asks and waits on the spot.  Per 160 000 frames of 160x120 / 128 000
// of 320x240, same box: requests in every stride 1.99 / 4.47 ms, every second 1.90 / 4.33, every fourth 1.82 / 4.10,
// every eighth 1.95 / 4.5 (lanes run dry); two pieces held per lane instead of one: no different; which of the four
// strides: no different.
template <uint32_t kPace>
__device__ __forceinline__ void fast_service_paced(Stream& s, uint32_t w, bool alive, bool& have, uint32_t stride_no) {
    if (alive && w + 9u > s.hi) {
        do {
            if (!have) stream_request(s);
            fast_ring_put(s, s.hi, s.pf0);
            fast_ring_put(s, s.hi + 4u, s.pf1);
            s.hi += 8u;
            have = false;
        } while (w + 9u > s.hi);
    }
    if ((stride_no & (kPace - 1u)) == 0u) {   // (wave-uniform)
        if (alive && !have) {
            stream_request(s);
            have = true;
        }
    }
}

// the next 32 bits
__device__ __forceinline__ uint32_t fast_window(uint32_t ringb, uint32_t t) {
    const uint32_t ra = shl3_add(t & 0x1e0u, ringb);
    return __builtin_amdgcn_alignbit(lds_load(ra), lds_load(ra + kSlot), ~t);
}

// the entry they select: m1 at the region's start, m2 kFastM2Word words in, both read, OR-ed
__device__ __forceinline__ uint32_t fast_lookup(const FastState& s, uint32_t v) {
    // chroma (j = 0, 1) -> + one region; AC -> + two
    const uint32_t toff = (((3u * kFastRegion) >> s.j) & kFastRegion) | (~s.dc & (2u * kFastRegion));
    const uint32_t e1 = lds_load(((v >> 21) & 0x7fcu) | toff);
    const uint32_t x2 = max(v >> 16, kFastLongFirst - 1u);
    const uint32_t e2 = lds_load((x2 << 2) + (toff + (kFastM2Word * 4u - 4u * (kFastLongFirst - 1u))));
    return e1 | e2;
}

// eight symbols, every lane, straight-line.  seen: OR of the entries (kFastInvalid: no such code); worst: the smallest
// (index + advance - 65): < 15 = an AC symbol moved the index past 63.  kLimit: the lane's share of the frame ends at
// bit lim1 + 1 (several lanes per frame): a symbol that would start there or later is not looked at.
template <uint32_t kFlush, bool kLimit>
__device__ __forceinline__ void fast_stride(uint32_t ringb, uint32_t stageb, uint32_t sumb, uint32_t lim1, FastState& s, uint32_t& seen,
                                            uint32_t& worst) {
#pragma unroll
    for (int it = 0; it < kStrideWrite; ++it) {
        // issued together, ahead of the table look-up: the component's DC sum (j = 1 Cb, j = 0 Cr, else Y) and the window
        const uint32_t ca = (((2u * kSlot) >> s.j) & (3u * kSlot)) | sumb;
        const uint32_t sum = lds_load(ca);
        const uint32_t v = fast_window(ringb, s.t);
        __builtin_amdgcn_sched_barrier(0);   // (left alone, the scheduler sinks the sum's read behind the look-up and waits twice)
        uint32_t run = (uint32_t)((int32_t)s.togo6 >> 31);   // 0 once the frame's last block is done
        if (kLimit) run &= (uint32_t)((int32_t)(s.t - lim1) >> 31);
        const uint32_t e = fast_lookup(s, v) & run;
        seen |= e;
        // magnitude bits -> value (AmvJpeg.c:924-933): sign-extended, x >= 0 means "leading 0 bit": value = x - (2^size - 1);
        // x < 0: value = x + 2^size.  Both are x - (full ^ (x >> 31)).  The width operand takes bits 0-4 of e (the size).
        const uint32_t used = e >> 24;
        const int x = __builtin_amdgcn_sbfe((int)v, 0u - used, e);
        const uint32_t full = __builtin_amdgcn_ubfe(0xffffffffu, 0u, e);
        const uint32_t val = (uint32_t)x - (full ^ (uint32_t)(x >> 31));
        s.t += used;
        const uint32_t kn = s.k + ((e >> 16) & 255u);
        worst = min(worst, kn - 65u);
        uint32_t dcn = (uint32_t)__builtin_amdgcn_sbfe((int)kn, 6u, 1u);   // ~0: the block ends with this symbol
        asm("" : "+v"(dcn));   // (knowing where it comes from, the compiler builds kn & ~dcn from a shift, a compare and a select)
        // DC difference (:945-951) joins its component's sum (:1200-1221)
        lds_store(ca, sum + (val & s.dc));
        const uint32_t rv = val + (sum & s.dc);
        lds_store((s.rp8 & ((2u * kFlush - 1u) << 8)) | stageb, (rv << 16) | (s.togo6 & 0xfc0u) | (kn - 1u));
        s.rp8 += e & kFastEmit;
        s.k = kn & ~dcn;
        s.togo6 += kn & 64u;
        s.j = min(s.j + dcn, 5u);   // 0 - 1 wraps to 5
        s.dc = dcn;
    }
}

// A stride that raised a flag, once more from its start state, without writing: where does the walk stop?
// why: 1 no such code (AmvJpeg.c:887), 2 index past 63 (:967-969); the state is the one the strict walk stops in.
template <bool kLimit>
__device__ __forceinline__ uint32_t fast_replay(uint32_t ringb, uint32_t lim1, FastState& s) {
    for (int it = 0; it < kStrideWrite; ++it) {
        if (s.togo6 == 0u) break;
        if (kLimit && (int32_t)(s.t - lim1) >= 0) break;
        const uint32_t e = fast_lookup(s, fast_window(ringb, s.t));
        if (e & kFastInvalid) return 1u;
        const uint32_t kn = s.k + ((e >> 16) & 255u);
        s.t += e >> 24;                      // an over-long run is consumed before the walk gives up
        if (kn - 65u < 15u) return 2u;
        const bool end = (kn & 64u) != 0u;
        s.rp8 += e & kFastEmit;
        s.k = end ? 0u : kn;
        s.togo6 += kn & 64u;
        s.j = end ? (s.j == 0u ? 5u : s.j - 1u) : s.j;
        s.dc = end ? ~0u : 0u;
    }
    return 0u;
}

// the window of a walk that starts in stream word w: the ring holds words [w & ~7, + 16)
__device__ __forceinline__ void fast_open_at(Stream& s, uint32_t w) {
    const uint32_t lo = w & ~7u;
    const uint4 a = stream_piece(s, lo), b = stream_piece(s, lo + 4u), c = stream_piece(s, lo + 8u), d = stream_piece(s, lo + 12u);
    fast_ring_put(s, lo, a);
    fast_ring_put(s, lo + 4u, b);
    fast_ring_put(s, lo + 8u, c);
    fast_ring_put(s, lo + 12u, d);
    s.hi = lo + kRingWords;
    stream_request(s);
}

// The speculative walk in the same arithmetic: from (t, k, j, dc) while fewer than lim1 + 1 bits are consumed; where
// symbols start and how the block position moves, nothing else.  "No such code" (a guessed start) slips one bit and
// leaves the index alone: the entry says so (one bit used, no advance); an over-long run closes the block.
// nblk6: blocks finished << 6; nrec8: symbols that carry a value << 8.
template <uint32_t kPace>
__device__ __forceinline__ void fast_skip(Stream& win, uint32_t ringb, FastState& s, uint32_t lim1, uint32_t& nblk6, uint32_t& nrec8) {
    nblk6 = nrec8 = 0u;
    bool active = (int32_t)(s.t - lim1) < 0;
    if (!__ballot(active)) return;
    fast_open_at(win, active ? (s.t + 1u) >> 5 : 0u);
    bool have = true;            // fast_open_at asked for the next eight words
    uint32_t stride_no = 1u;     // (the first request stride is three strides away: the ring was just filled)
    while (__ballot(active) != 0ull) {
        fast_service_paced<kPace>(win, (s.t + 1u) >> 5, active, have, stride_no++);
#pragma unroll
        for (int it = 0; it < kStrideWrite; ++it) {
            const uint32_t v = fast_window(ringb, s.t);
            const uint32_t e = fast_lookup(s, v) & (uint32_t)((int32_t)(s.t - lim1) >> 31);
            s.t += e >> 24;
            const uint32_t kn = s.k + ((e >> 16) & 255u);
            uint32_t dcn = (uint32_t)__builtin_amdgcn_sbfe((int)kn, 6u, 1u);
            asm("" : "+v"(dcn));
            nrec8 += e & kFastEmit;
            nblk6 += kn & 64u;
            s.k = kn & ~dcn;
            s.j = min(s.j + dcn, 5u);
            s.dc = dcn;
        }
        active = (int32_t)(s.t - lim1) < 0;
    }
}

}  // namespace

// dynamic LDS: [ the four tables, 32 KB | staged records, 2 * kFlush slots per wave (a stride's eight, alternating) |
//                DC sums, 4 slots per wave | rings, 17 slots per wave ]
template <uint32_t kFlush>
__global__ __launch_bounds__(kWave * 16) void amv_huffman_fast_kernel(
    const uint32_t* __restrict__ ws, const uint32_t* __restrict__ ws_bytes, uint32_t n,
    const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count,
    uint32_t blocks_per_frame, const uint32_t* __restrict__ ws_line,
    const HuffDecodeImage* __restrict__ img, SyncOut out, int32_t* __restrict__ status,
    uint32_t* __restrict__ nmcu_ok, uint32_t* __restrict__ queue, unsigned long long* __restrict__ stats) {
    // the stride-aligned record output below is written for eight slots per stride: a 32-record line per four strides
    // (rp8 = stride_no << 11, p[q * kWave] for q < 8, the half of the staging area by (stride_no & 1) * kFlush)
    static_assert(kFlush == 8u, "amv_huffman_fast_kernel stages a stride's eight record slots");
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, nwaves = blockDim.x >> 6;
    {   // tables, shared by the waves of the workgroup
        const uint4* src = reinterpret_cast<const uint4*>(&img->fast[0][0]);
        uint4* dst = reinterpret_cast<uint4*>(s_mem);
        for (uint32_t i = threadIdx.x; i < kFastTableBytes / 16u; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();   // the only workgroup-wide barrier; from here the waves are on their own
    // lds_load takes raw LDS addresses: the dynamic segment must begin at 0.  Should a toolchain ever put something in
    // front of it, the kernel decodes nothing and hands its frames to the serial kernel instead.
    const bool lds_at_zero = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s_mem == 0u;
    const uint32_t stageb = kFastTableBytes + wave * fast_stage_bytes(kFlush) + lane * 4u;
    const uint32_t sumb = kFastTableBytes + nwaves * fast_stage_bytes(kFlush) + wave * kFastSumBytes + lane * 4u;
    const uint32_t ringb = kFastTableBytes + nwaves * (fast_stage_bytes(kFlush) + kFastSumBytes) + wave * kFastRingBytes + lane * 4u;
    uint32_t* const stage = reinterpret_cast<uint32_t*>(s_mem + stageb);
    if (list) n = *list_count;
    const uint32_t ntasks = (n + kWave - 1) / kWave;
    const SegGeom sg = out.sg;
    for (;;) {
        uint32_t task = 0;
        if (lane == 0) task = atomicAdd(queue, 1u);
        task = __shfl(task, 0);
        if (task >= ntasks) return;
        const uint32_t idx = task * kWave + lane;
        const uint32_t frame = idx < n ? (list ? list[idx] : idx) : kNever;
        uint32_t total = frame != kNever ? ws_bytes[frame] : kNever;   // kNever: handed to the serial kernel
        if (!lds_at_zero && total != kNever) {
            out.retry_list[atomicAdd(out.retry_count, 1u)] = frame;
            total = kNever;
        }
        if (frame != kNever && total == kNever) out.rec_count[frame] = kNever;
        if (!lds_at_zero) continue;
        const bool live = total != kNever;
        const uint32_t fsafe = live ? frame : 0u;
        const uint32_t line0 = out.rec_line[fsafe];
        uint32_t* const rec = out.rec + (uint64_t)line0 * 32u;
        const uint32_t line1 = out.rec_line[fsafe + 1u];
        const uint32_t cap_rec = line1 > line0 ? (line1 - line0) * 32u : 0u;   // this frame's record space, a multiple of 32 (monotone layout; read defensively)
        uint2* const seg_out = reinterpret_cast<uint2*>(out.seg_start) + (uint64_t)fsafe * (sg.count + 1u);
        const uint32_t valid_bits = live ? total * 8u : 0u;
        Stream win{ws + (uint64_t)ws_line[fsafe] * 4u, live ? ((total + 15u) >> 4) * 4u : 0u, reinterpret_cast<uint32_t*>(s_mem + ringb), 0u,
                   make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        fast_open(win);
        for (uint32_t c = 0; c < 3u; ++c) lds_store(sumb + c * kSlot, 0u);

        // a lane without a frame stands still from the start
        FastState s{0xffffffffu, 0u, 5u, live ? 0u - (blocks_per_frame << 6) : 0u, 0u, ~0u};
        bool alive = live;
        uint32_t stop = 0;           // 1 no such code, 2 index past 63, 3 the frame's last block is done
        // The lane's records go out stride-aligned: record position 8 * stride + slot.  A stride's records (one per
        // symbol that carries a value: <= 8) fill the front of its eight staging slots, fillers the rest, and after
        // every fourth stride each lane's 32 slots leave as one 128-byte line -- all lanes in the same stride, so the
        // eight store instructions of a line are issued once per four strides with every lane in them, not in every
        // stride with the quarter of the lanes whose line happened to be full (and the registers a line waits in
        // are filled by plain assignment: which quarter is the wave's business, not the lane's).  ~12 % of the
        // stream are fillers (the end-of-block symbols' slots); the reader skips them as it skips the padding.
        // (Records packed densely per lane, lines leaving whenever a lane had 32: 2.10 ms per 160 000 frames, this:
        // 1.94; the stores moved a stride later, behind the next stream service's s_waitcnt: no different -- what that
        // wait waits for is the stream words the OTHER lanes asked for a stride ago, profiles/r03_kernel_phases.txt.)
        uint32_t stride_no = 0;      // (wave-uniform)
        bool have_words = true;      // fast_open asked for words [16, 24)
        uint32_t recpos = 0;         // end of the last line this lane wrote
        uint32_t seg_next = 0, seg_col = 0, seg_blk = 0;   // the next MCU-row segment whose first DC symbol has not come yet
        uint32_t end_blocks = 0;     // whole blocks when the walk stopped
#pragma unroll
        for (uint32_t q = 0; q < 2u * kFlush; ++q) stage_put<kFlush>(stage, q, kDummyRecord);
        if (__ballot(alive) != 0ull) do {
            const bool line_live = alive;   // records of this lane may come in these four strides
            uint4 line[8];
#pragma unroll
            for (int quarter = 0; quarter < 4; ++quarter) {
                fast_service_paced<kServicePace>(win, (s.t + 1u) >> 5, alive, have_words, stride_no);
                const bool running = alive;
                s.rp8 = stride_no << 11;     // this stride's eight slots
                const FastState start = s;
                uint32_t worst = ~0u, seen = 0u;
                fast_stride<kFlush, false>(ringb, stageb, sumb, 0u, s, seen, worst);
                const bool trouble = alive && (worst < 15u || (seen & kFastInvalid) != 0u);
                if (__ballot(trouble) != 0ull) {   // a damaged stream
                    if (trouble) {
                        s = start;
                        stop = fast_replay<false>(ringb, 0u, s);
                        alive = false;
                        for (uint32_t q = s.rp8 >> 8; q < stride_no * 8u + 8u; ++q) stage_put<kFlush>(stage, q, kDummyRecord);   // what the stride staged past the stop
                    }
                }
                if (alive && s.togo6 == 0u) { stop = 3u; alive = false; }
                stage_put<kFlush>(stage, s.rp8 >> 8, kDummyRecord);   // the slot behind the last record (every symbol writes the next slot, counted or not)
                {   // the stride's slots -> a quarter of the line; fillers take their place for the stride after the next
                    uint32_t* p = stage + ((stride_no & 1u) * kFlush) * kWave;
                    line[2 * quarter] = make_uint4(p[0], p[kWave], p[2 * kWave], p[3 * kWave]);
                    line[2 * quarter + 1] = make_uint4(p[4 * kWave], p[5 * kWave], p[6 * kWave], p[7 * kWave]);
#pragma unroll
                    for (uint32_t q = 0; q < 8u; ++q) p[q * kWave] = kDummyRecord;
                }
                bool hit = false;
                uint32_t blocks = 0;
                if (running) {
                    blocks = blocks_per_frame + (uint32_t)((int32_t)s.togo6 >> 6);   // whole blocks so far
                    // blocks whose DC symbol is out: did the next segment's first one come in this stride?  (At most one
                    // does: a segment is >= 6 blocks = 12 symbols.)
                    hit = seg_next < sg.count && blocks + (s.k ? 1u : 0u) > seg_blk;
                    if (!alive) { end_blocks = blocks; s.togo6 = 0u; }   // stands still from here on
                }
                if (__ballot(hit) != 0ull) {
                    if (hit) {
                        seg_out[seg_next++] = make_uint2(start.rp8 >> 8, s.rp8 >> 8);
                        const bool last = seg_col + 1u == sg.per_row;
                        seg_blk += 6u * (last ? sg.mcu_cols - seg_col * kSegMcus : kSegMcus);
                        seg_col = last ? 0u : seg_col + 1u;
                    }
                }
                ++stride_no;
            }
            if (line_live) {
                const uint32_t at = (stride_no - 4u) * 8u;
                if (at + 32u <= cap_rec) {   // never past the frame's record space (a multiple of 32); an overfull frame is redone densely
                    uint4* d = reinterpret_cast<uint4*>(rec + at);
#pragma unroll
                    for (int q = 0; q < 8; ++q) d[q] = line[q];
                }
                recpos = at + 32u;
            }
        } while (__ballot(alive) != 0ull);
        if (live) {
            const uint32_t bits = s.t + 1u + (stop == 1u ? 17u : 0u);   // FORMAT: the reference has read 17 bits by then
            uint32_t st = stop == 1u ? kStFormat : (stop == 2u ? kStOverrun : 0u);
            if (bits > valid_bits) st |= kStTruncated;
            reinterpret_cast<uint4*>(out.lane_tab)[(uint64_t)frame * out.lane_stride] = make_uint4(0u, 0u, 0u, 0u);
            // segments the decoder never started begin (and end) at the total; so does the end of the last one
            for (uint32_t m = seg_next; m <= sg.count; ++m) seg_out[m] = make_uint2(recpos, recpos);
            if (stats) atomicAdd(&stats[0], 1ull);
            if (recpos > cap_rec) {   // more non-zero coefficients than the record space holds
                out.rec_count[frame] = kNever;
                out.retry_list[atomicAdd(out.retry_count, 1u)] = frame;
            } else {
                status[frame] = (int32_t)st;
                nmcu_ok[frame] = out.ok_in_blocks ? end_blocks : end_blocks / 6u;
                out.rec_count[frame] = recpos;
            }
        }
    }   // next task
}

// =============================================================================================
// Several lanes per frame, records form, in the same arithmetic (amv_huffman_sync_kernel<L, true>'s passes 1-5 with
// fast_skip as the speculative walk and fast_stride as the strict one; the dense form keeps the kernel above).
// A lane's share ends at bit `limit`: the strict walk looks at no symbol that starts there or later, so the lanes'
// records partition the frame's; every lane's records start on a piece of kFlush of them (fillers behind its last).
// dynamic LDS as amv_huffman_fast_kernel's.
// =============================================================================================
template <int L>
__global__ __launch_bounds__(kWave * 16) void amv_huffman_sync2_kernel(
    const uint32_t* __restrict__ ws, const uint32_t* __restrict__ ws_bytes, uint32_t n,
    const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count,
    uint32_t blocks_per_frame, const uint32_t* __restrict__ ws_line,
    const HuffDecodeImage* __restrict__ img, SyncOut out, int32_t* __restrict__ status,
    uint32_t* __restrict__ nmcu_ok, uint32_t* __restrict__ queue, unsigned long long* __restrict__ stats) {
    constexpr int kFrames = kWave / L;   // frames per wave
    constexpr uint32_t kFlush = 8u;
    constexpr uint32_t kNoLimit = 0x7ffffffeu;   // as lim1: no position reaches it
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u, nwaves = blockDim.x >> 6;
    const uint32_t slot = lane / L, sub = lane % L;
    // (the heavy frames' launch of a batch that has none: gone before it has touched the tables -- its workgroups hold the
    // LDS the light frames' launch behind it is waiting for)
    if (list && *list_count == 0u) return;
    {   // tables, shared by the waves of the workgroup
        const uint4* src = reinterpret_cast<const uint4*>(&img->fast[0][0]);
        uint4* dst = reinterpret_cast<uint4*>(s_mem);
        for (uint32_t i = threadIdx.x; i < kFastTableBytes / 16u; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();   // the only workgroup-wide barrier; from here the waves are on their own
    // lds_load takes raw LDS addresses: the dynamic segment must begin at 0.  Should a toolchain ever put something in
    // front of it, the kernel decodes nothing and hands its frames to the serial kernel instead.
    const bool lds_at_zero = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)s_mem == 0u;
    const uint32_t stageb = kFastTableBytes + wave * fast_stage_bytes(kFlush) + lane * 4u;
    const uint32_t sumb = kFastTableBytes + nwaves * fast_stage_bytes(kFlush) + wave * kFastSumBytes + lane * 4u;
    const uint32_t ringb = kFastTableBytes + nwaves * (fast_stage_bytes(kFlush) + kFastSumBytes) + wave * kFastRingBytes + lane * 4u;
    uint32_t* const stage = reinterpret_cast<uint32_t*>(s_mem + stageb);
    if (list) n = *list_count;
    const uint32_t ntasks = (n + kFrames - 1) / kFrames;
    const SegGeom sg = out.sg;
    for (;;) {
        uint32_t task = 0;
        if (lane == 0) task = atomicAdd(queue, 1u);
        task = __shfl(task, 0);
        if (task >= ntasks) return;

        const bool timing = stats != nullptr && lane == 0;   // optional phase clock (amvhip_entropy_stats)
        unsigned long long tc[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long wall0 = 0;
        if (timing) { tc[0] = tc[1] = clock64(); wall0 = wall_clock64(); }

        const uint32_t idx = task * kFrames + slot;
        const uint32_t frame = idx < n ? (list ? list[idx] : idx) : kNever;
        uint32_t total = frame != kNever ? ws_bytes[frame] : kNever;   // kNever: handed to the serial kernel
        if (!lds_at_zero && total != kNever) {
            if (sub == 0) out.retry_list[atomicAdd(out.retry_count, 1u)] = frame;
            total = kNever;
        }
        if (frame != kNever && total == kNever && sub == 0) out.rec_count[frame] = kNever;
        if (!lds_at_zero) continue;
        const bool live = total != kNever;
        const uint32_t fsafe = live ? frame : 0u;
        const uint32_t line0 = out.rec_line[fsafe];
        uint32_t* const rec = out.rec + (uint64_t)line0 * 32u;
        const uint32_t line1 = out.rec_line[fsafe + 1u];
        const uint32_t cap_rec = line1 > line0 ? (line1 - line0) * 32u : 0u;   // this frame's record space, a multiple of 32 (monotone layout; read defensively)
        uint2* const seg_out = reinterpret_cast<uint2*>(out.seg_start) + (uint64_t)fsafe * (sg.count + 1u);
        const uint32_t valid_bits = live ? total * 8u : 0u;
        Stream win{ws + (uint64_t)ws_line[fsafe] * 4u, live ? ((total + 15u) >> 4) * 4u : 0u, reinterpret_cast<uint32_t*>(s_mem + ringb), 0u,
                   make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};

        // ---- 1/2. speculative walks until every lane knows the state the decoder is in where its share starts.
        // Round 4's form: every lane walks its share from a guess, then takes its left neighbour's arrival as its start and
        // walks again while that changes.  A 10 000-frame launch then lasts as long as its unluckiest frame: a wrong
        // trajectory that does not fall in step with the true one for k shares in a row (one in five per share, more where
        // blocks are long) costs k rounds, one lane of the frame walking and the others watching -- 2.5 rounds on
        // average, 9 for the worst of 10 000 frames (tools/sim_sync.c is a CPU model of this loop).  Now (round 5):
        //  * a lane REMEMBERS what it has walked: its share's memo, (start -> arrival, blocks, records) of up to kMemo walks,
        //    in the staging and DC-sum words of its LDS column, which are free until pass 4;
        //  * FINALITY runs ahead of the walks: lane 0's start is exact; a lane whose left neighbour is final takes that
        //    lane's arrival as its own final start, and when its share has been walked from that state before -- by
        //    anybody -- it is final too, at once, and so on to the right (a loop of shuffles, no walk);
        //  * lanes with nothing to walk HELP: every arrival somebody has found at the end of share x - 1 is a candidate
        //    start of share x, and where candidates run out the guess "a block starts here" is tried with the other five
        //    positions in the MCU (the 1-in-6 part of the guess) -- so that when finality reaches a share, the true start
        //    has, with luck, been walked from already.  One task per lane and round, the frame's leftmost shares first.
        //  The first non-final lane of a frame always walks from its final start, so a frame is done after at most L - 1
        //  rounds whatever the helpers found; on the synthetic stream the worst frame of 4 000 takes 4 rounds instead of 9.
        uint32_t S = ((valid_bits + L - 1) / L + 31u) & ~31u;   // bits per lane, a whole number of words
        if (S < 64u) S = 64u;
        constexpr uint32_t kMemo = 5u, kPend = 4u;      // words 0..14 of a column: the memo; 15..18: this round's tasks; 19: entries held
        constexpr uint32_t kGuess = 5u << 11;            // state code: (t - (x * S - 1)) | k << 5 | j << 11; the guess: DC of block 0 comes next
        const uint32_t stage0 = stageb - lane * 4u, sum0 = sumb - lane * 4u, col0 = slot * L;
        const uint64_t seg = L == 64 ? ~0ull : (((1ull << (L & 63)) - 1ull) << (slot * L));
        auto col_addr = [&](uint32_t col, uint32_t w) { return (w < 16u ? stage0 + w * kSlot : sum0 + (w - 16u) * kSlot) + col * 4u; };
        // the last lane of a frame has no right neighbour to feed: its share is only walked in pass 4
        const bool walks = live && sub != L - 1;
        uint32_t my_blocks6 = 0u, my_recs8 = 0u, seeded = 0u, rounds = 0u;
        lds_store(col_addr(lane, 19u), 0u);
        {   // round 0: every lane from its guess (lane 0: the exact state)
            FastState st{sub * S - 1u, 0u, 5u, 0u, 0u, ~0u};
            const uint32_t lim1 = walks ? (sub + 1u) * S - 1u : 0u;
            if (!walks) st.t = lim1;
            uint32_t b6, r8;
            fast_skip<skip_pace<L>()>(win, ringb, st, lim1, b6, r8);
            if (walks) {
                lds_store(col_addr(lane, 0u), kGuess | (((st.t - lim1) | (st.k << 5) | (st.j << 11)) << 16));
                lds_store(col_addr(lane, 1u), b6);
                lds_store(col_addr(lane, 2u), r8);
                lds_store(col_addr(lane, 19u), 1u);
            }
        }
        if (timing) tc[2] = clock64();
        // which of its memo's walks is a lane's FINAL one (7: none yet), and the start it belongs to
        constexpr uint32_t kNone = 7u;
        uint32_t final_at = kNone, final_code = kGuess;
        bool settled = !live;
        for (int round = 0; round < L + 1; ++round) {
            wave_sync();
            // what the lane's own share and the share to its left have been walked from / arrived at, into registers
            const uint32_t held = lds_load(col_addr(lane, 19u));
            const uint32_t left_n = sub ? lds_load(col_addr(lane - 1u, 19u)) : 0u;
            uint32_t mine[kMemo], cand[kMemo];
#pragma unroll
            for (uint32_t q = 0; q < kMemo; ++q) {
                const uint32_t a = lds_load(col_addr(lane, 3u * q)), b = lds_load(col_addr(sub ? lane - 1u : lane, 3u * q));
                mine[q] = q < held ? a : kNever;                     // entry code | arrival code << 16
                cand[q] = q < left_n ? b >> 16 : kNever;             // arrival codes at the end of the share before
            }
            // Finality.  A lane's link: "if walk q' of the share to my left is that share's final one, which of MY walks starts
            // where it arrived?" -- eight 3-bit fields (q' -> q, 7 = none; 7 -> 7).  Lane 0's answer is walk 0 whatever comes
            // in (the guess IS its start); the frame's last lane, which is never walked ahead, is final as soon as its left
            // neighbour is.  Links compose, so a prefix scan over the frame's lanes gives every lane its final walk at once
            // (a loop that moved finality one lane per trip cost a fifth of a round with 64 lanes per frame).
            uint32_t link = (kNone << 15) | (kNone << 18) | (kNone << 21);
#pragma unroll
            for (uint32_t qp = 0; qp < kMemo; ++qp) {
                uint32_t to = kNone;
#pragma unroll
                for (uint32_t q = 0; q < kMemo; ++q) to = (mine[q] != kNever && (mine[q] & 0xffffu) == cand[qp]) ? q : to;
                if (sub == L - 1) to = cand[qp] != kNever ? 0u : kNone;
                link |= to << (3u * qp);
            }
            if (sub == 0) link = 0u;
            if (!live) link = 0u;
#pragma unroll
            for (int d = 1; d < L; d <<= 1) {
                const uint32_t before = __shfl_up(link, d, L);       // the links of the d lanes to the left, composed
                uint32_t both = 0u;
#pragma unroll
                for (uint32_t f = 0; f < 8u; ++f) both |= ((link >> (3u * ((before >> (3u * f)) & 7u))) & 7u) << (3u * f);
                if (sub >= (uint32_t)d) link = both;
            }
            final_at = link & 7u;                                    // (every field holds the same: lane 0's link ignores its input)
            const uint32_t left_at = wave_shr1(final_at);
            settled = !live || final_at != kNone;
            const bool front = live && !settled && sub != 0 && left_at != kNone;   // the frame's first open lane: its start is final
            if (sub == 0) final_code = kGuess;
            else {
#pragma unroll
                for (uint32_t q = 0; q < kMemo; ++q) if (left_at == q) final_code = cand[q];
            }
            const uint64_t open = __ballot(!settled);
            if (!open) break;
            if (open & seg) ++rounds;
            // this round's tasks of the lane's own share: its final start, or else the arrivals found at the end of the share
            // before that it has not walked from (newest first), then guesses with another position in the MCU
            uint32_t p = 0u;
            if (walks && front) {
                lds_store(col_addr(lane, 15u), final_code);
                p = 1u;
            } else if (walks && !settled) {
                const uint32_t most = min(kMemo - min(held, kMemo), kPend);
                uint32_t taken[kPend] = {kNever, kNever, kNever, kNever};
#pragma unroll
                for (uint32_t qq = 0; qq < kMemo; ++qq) {
                    const uint32_t c = cand[kMemo - 1u - qq];
                    bool known = c == kNever;
#pragma unroll
                    for (uint32_t m = 0; m < kMemo; ++m) known = known || (mine[m] & 0xffffu) == c;
#pragma unroll
                    for (uint32_t m = 0; m < kPend; ++m) known = known || taken[m] == c;
                    if (!known && p < most) {
#pragma unroll
                        for (uint32_t m = 0; m < kPend; ++m) if (m == p) taken[m] = c;
                        lds_store(col_addr(lane, 15u + p), c);
                        ++p;
                    }
                }
                while (seeded < 5u && p + 1u < most) {     // (one place stays free for what finality brings)
                    ++seeded;
                    lds_store(col_addr(lane, 15u + p), (5u - seeded) << 11);
                    ++p;
                }
            }
            // one task per lane: a share's first task is its own lane's; the others go to the frame's lanes that have none,
            // in lane order (a frame's leftmost shares first)
            const uint32_t extras = p ? p - 1u : 0u;
            const bool idle = live && p == 0u;
            uint32_t all_extras, all_idle;
            const uint32_t ex_at = seg_excl_sum<L>(extras, sub, all_extras);
            const uint32_t id_at = seg_excl_sum<L>(idle ? 1u : 0u, sub, all_idle);
            wave_sync();
            uint32_t owner = sub, rank = 0u;            // whose share, and which of its tasks
            bool work = p != 0u;
            if (__ballot(all_extras != 0u) != 0ull) {
                // the lane whose extras hold number id_at: the last one whose first extra is numbered <= id_at (of several
                // lanes with the same number the last is the one that has extras)
                uint32_t lo = 0u;
#pragma unroll
                for (int d = L >> 1; d >= 1; d >>= 1) {
                    const uint32_t probe = __shfl(ex_at, (int)(lo + (uint32_t)d), L);
                    if (lo + (uint32_t)d < (uint32_t)L && probe <= id_at) lo += (uint32_t)d;
                }
                const uint32_t lo_at = __shfl(ex_at, (int)lo, L), lo_n = __shfl(extras, (int)lo, L);
                if (idle && id_at < all_extras && id_at - lo_at < lo_n) { owner = lo; rank = id_at - lo_at + 1u; work = true; }
            }
            FastState st{0u, 0u, 5u, 0u, 0u, ~0u};
            uint32_t lim1 = 0u, code = 0u;
            if (work) {
                code = lds_load(col_addr(col0 + owner, 15u + rank));
                lim1 = (owner + 1u) * S - 1u;
                st.t = owner * S - 1u + (code & 31u);
                st.k = (code >> 5) & 63u;
                st.j = code >> 11;
                st.dc = st.k ? 0u : ~0u;
            }
            uint32_t b6 = 0u, r8 = 0u;
            fast_skip<skip_pace<L>()>(win, ringb, st, lim1, b6, r8);
            if (work) {
                const uint32_t arr = (st.t - lim1) | (st.k << 5) | (st.j << 11);
                // the walk from a final start always lands in the memo: in its last place when the memo is full
                uint32_t at = lds_load(col_addr(col0 + owner, 19u)) + rank;
                if (rank == 0u && front) at = min(at, kMemo - 1u);
                if (at < kMemo) {
                    lds_store(col_addr(col0 + owner, 3u * at), code | (arr << 16));
                    lds_store(col_addr(col0 + owner, 3u * at + 1u), b6);
                    lds_store(col_addr(col0 + owner, 3u * at + 2u), r8);
                }
            }
            wave_sync();
            if (p) {    // how many of this share's tasks found a lane: its own + the extras numbered below the idle lanes' count
                const uint32_t done = 1u + min(extras, all_idle > ex_at ? all_idle - ex_at : 0u);
                lds_store(col_addr(lane, 19u), min(held + done, kMemo));
            }
        }
        // the final walk's counts, for EVERY lane that has one -- also when the loop ended on its bound with a frame of the
        // wave still open (that frame goes to the serial kernel below; the wave's other frames are settled, and their
        // block and record prefix sums need their counts all the same)
        if (walks && final_at != kNone) {
            my_blocks6 = lds_load(col_addr(lane, 3u * final_at + 1u));
            my_recs8 = lds_load(col_addr(lane, 3u * final_at + 2u));
        }
        if (timing) tc[3] = clock64();
        // the state every lane starts pass 4 in: the bit offset inside the share's first word takes the code's five low bits --
        // a symbol is at most 16 + 11 = 27 bits long, so a walk that stops at the first symbol STARTING at or behind the
        // share's end arrives 0 .. 26 bits behind it
        static_assert(16 + 11 < 32, "an arrival's offset behind the share's end must fit the state code's five bits");
        FastState entry{sub * S - 1u + (final_code & 31u), (final_code >> 5) & 63u, final_code >> 11, 0u, 0u, 0u};
        entry.dc = entry.k ? 0u : ~0u;
        if (sub == L - 1 || !live) { my_blocks6 = 0u; my_recs8 = 0u; }
        // (cannot happen: the first open lane of a frame walks from its final start in every round.  Should a lane be left
        // without one all the same, the frame goes to the serial kernel rather than out with a guessed state in it.)
        const bool unsettled = (__ballot(!settled) & seg) != 0ull;

        // ---- 3. first block and first record of every lane (the last lane has not walked: its counts are 0, it is last)
        uint32_t all_blocks, all_recs;
        const uint32_t blk0 = seg_excl_sum<L>(my_blocks6 >> 6, sub, all_blocks);
        const uint32_t rec0 = seg_excl_sum<L>(((my_recs8 >> 8) + kFlush - 1u) & ~(kFlush - 1u), sub, all_recs);

        // ---- 4. the strict, writing pass.  Lanes left of the frame's end (or first error) are exact; whatever a lane to
        // the right of it does is ignored below.
        for (uint32_t c = 0; c < 3u; ++c) lds_store(sumb + c * kSlot, 0u);
        FastState s = entry;
        s.togo6 = (blk0 - blocks_per_frame) << 6;
        s.rp8 = rec0 << 8;
        bool alive = live && blk0 < blocks_per_frame;
        if (!alive) s.togo6 = 0u;                       // stands still from the start
        const uint32_t wlim1 = sub == L - 1 ? kNoLimit : (sub + 1u) * S - 1u;
        const uint32_t dc_first = blk0 + (s.k ? 1u : 0u);
        uint32_t stop = 0;           // 1 no such code, 2 index past 63, 3 the frame's last block is done
        uint32_t flushed = rec0;     // records before this one have left for memory (a multiple of kFlush)
        uint32_t end_blocks = 0;     // whole blocks when the walk stopped
        // the first segment start this lane can meet: the first MCU at or after its first DC block
        uint32_t seg_next, seg_col, seg_blk;
        {
            const uint32_t m_first = (dc_first + 5u) / 6u;
            uint32_t row = m_first / sg.mcu_cols;
            const uint32_t col = m_first - row * sg.mcu_cols;
            seg_col = (col + kSegMcus - 1u) / kSegMcus;
            if (seg_col >= sg.per_row) { ++row; seg_col = 0u; }
            seg_next = row * sg.per_row + seg_col;
            seg_blk = (row * sg.mcu_cols + seg_col * kSegMcus) * 6u;
        }
        if (__ballot(alive) != 0ull) {
        fast_open_at(win, alive ? (s.t + 1u) >> 5 : 0u);
        bool have_words = true;      // fast_open_at asked for the next eight words
        uint32_t stride_no = 1u;
        do {   // (bottom-tested, as in amv_huffman_fast_kernel)
            fast_service_paced<kServicePace>(win, (s.t + 1u) >> 5, alive, have_words, stride_no++);
            if ((s.rp8 >> 8) - flushed >= kFlush) {
                stage_flush<kFlush>(stage, rec, flushed, cap_rec);
                flushed += kFlush;
            }
            const bool running = alive;
            const FastState start = s;
            uint32_t worst = ~0u, seen = 0u;
            fast_stride<kFlush, true>(ringb, stageb, sumb, wlim1, s, seen, worst);
            const bool trouble = alive && (worst < 15u || (seen & kFastInvalid) != 0u);
            if (__ballot(trouble) != 0ull) {   // a damaged stream (or a lane right of the true path's end)
                if (trouble) {
                    s = start;
                    stop = fast_replay<true>(ringb, wlim1, s);
                    alive = false;
                }
            }
            if (alive && s.togo6 == 0u) { stop = 3u; alive = false; }
            if (alive && (int32_t)(s.t - wlim1) >= 0) alive = false;   // the end of the lane's share
            bool hit = false;
            if (running) {
                const uint32_t blocks = blocks_per_frame + (uint32_t)((int32_t)s.togo6 >> 6);   // whole blocks so far
                hit = seg_next < sg.count && blocks + (s.k ? 1u : 0u) > seg_blk;
                if (!alive) { end_blocks = blocks; s.togo6 = 0u; }   // stands still from here on
            }
            if (__ballot(hit) != 0ull) {
                if (hit) {
                    seg_out[seg_next++] = make_uint2(start.rp8 >> 8, s.rp8 >> 8);
                    const bool last = seg_col + 1u == sg.per_row;
                    seg_blk += 6u * (last ? sg.mcu_cols - seg_col * kSegMcus : kSegMcus);
                    seg_col = last ? 0u : seg_col + 1u;
                }
            }
        } while (__ballot(alive) != 0ull);
        }
        const uint32_t recpos = s.rp8 >> 8;
        {   // what is still staged leaves padded to a whole piece with records no block owns
            const uint32_t end = (recpos + kFlush - 1u) & ~(kFlush - 1u);
            for (uint32_t q = recpos; q < end; ++q) stage_put<kFlush>(stage, q, kDummyRecord);
            while (flushed < end) {
                stage_flush<kFlush>(stage, rec, flushed, cap_rec);
                flushed += kFlush;
            }
        }
        if (timing) tc[4] = clock64();
        const uint32_t err = stop == 1u ? kStFormat : (stop == 2u ? kStOverrun : 0u);
        const uint32_t stop_p = s.t + 1u + (stop == 1u ? 17u : 0u);   // FORMAT: the reference has read 17 bits by then
        const uint64_t stop_mask = __ballot(stop != 0u) & seg;
        uint32_t st = 0, good_blocks = blocks_per_frame, rec_total = 0, seg_seen = 0;
        int stop_lane = 0;
        if (stop_mask) {
            stop_lane = __builtin_ctzll(stop_mask);   // leftmost = the true path
            const uint32_t e = __shfl(err, stop_lane);
            const uint32_t eb = __shfl(end_blocks, stop_lane);
            const uint32_t sp = __shfl(stop_p, stop_lane);
            rec_total = __shfl(recpos, stop_lane);
            seg_seen = __shfl(seg_next, stop_lane);
            st = e;
            if (e) good_blocks = eb;
            if (sp > valid_bits) st |= kStTruncated;
        } else {
            st = kStFormat; good_blocks = 0;   // unreachable: the last lane runs until the frame ends or fails
        }

        // ---- 5. DC prediction: the sums of the lanes to the left are a lane's base; it goes into the frame's lane table
        // and the reader adds it
        uint32_t tot;
        const uint32_t by = seg_excl_sum<L>(lds_load(sumb), sub, tot);
        const uint32_t bu = seg_excl_sum<L>(lds_load(sumb + kSlot), sub, tot);
        const uint32_t bv = seg_excl_sum<L>(lds_load(sumb + 2u * kSlot), sub, tot);
        if (live) {
            // lanes right of the one that met the end (or the first error) walked from states no decoder reaches
            const bool real = (int)lane <= stop_lane && blk0 < blocks_per_frame;
            reinterpret_cast<uint4*>(out.lane_tab)[(uint64_t)frame * out.lane_stride + sub] = make_uint4(real ? dc_first : kNever, by, bu, bv);
            // segments the decoder never started begin (and end) at the total; so does the end of the last one
            for (uint32_t m = seg_seen + sub; m <= sg.count; m += L) seg_out[m] = make_uint2(rec_total, rec_total);
        }
        uint32_t wave_rounds = rounds;                        // the wave's frame that needed most
        if (stats) {
#pragma unroll
            for (int off = 32; off; off >>= 1) wave_rounds = max(wave_rounds, (uint32_t)__shfl_xor((int)wave_rounds, off));
        }
        if (timing) {
            tc[5] = clock64();
            for (int q = 0; q < 5; ++q) atomicAdd(&stats[4 + q], tc[q + 1] - tc[q]);
            atomicAdd(&stats[9], 1ull);
            if (task < kTraceTasks) {   // the task's own line (amvhip_entropy_trace): where the launch's slowest waves spend their time
                unsigned long long* tr = stats + kTraceBase + (unsigned long long)task * 8ull;
                tr[0] = wall0;                              // constant-rate clock, 100 MHz: when the task began ...
                tr[1] = wall_clock64();                     // ... and ended
                for (int q = 0; q < 4; ++q) tr[2 + q] = tc[q + 2] - tc[q + 1];   // first walk, rounds, strict pass, DC pass (shader clocks)
                tr[6] = wave_rounds | (unsigned long long)S << 32;
                tr[7] = (unsigned long long)blockIdx.x << 32 | (unsigned long long)wave << 16 | (unsigned long long)L;
            }
        }
        if (live && sub == 0) {
            if (stats) {   // optional: how hard the synchronisation worked (amvhip_entropy_stats)
                atomicAdd(&stats[0], 1ull);
                atomicAdd(&stats[1], (unsigned long long)rounds);
                atomicMax(&stats[2], (unsigned long long)rounds);
            }
            if (rec_total > cap_rec || unsettled) {   // more non-zero coefficients than the record space holds
                out.rec_count[frame] = kNever;
                out.retry_list[atomicAdd(out.retry_count, 1u)] = frame;
            } else {
                status[frame] = (int32_t)st;
                nmcu_ok[frame] = out.ok_in_blocks ? good_blocks : good_blocks / 6u;
                out.rec_count[frame] = rec_total | ((uint32_t)(L - 1) << 24);   // (a frame holds < 2^21 records: amvhip_api.hip)
            }
        }
    }   // next task
}

namespace {

template <int L>
void launch_sync(const uint32_t* ws, const uint32_t* ws_bytes, uint32_t n, const uint32_t* list,
                 const uint32_t* list_count, const FrameGeom& g, const uint32_t* ws_line,
                 const HuffDecodeImage* d_img, const SyncOut& out, int32_t* status, uint32_t* nmcu_ok,
                 uint32_t* queue, unsigned long long* stats, uint32_t cus, hipStream_t s) {
    constexpr uint32_t kMaxWaves = 10u;
    constexpr uint32_t kPerWave = kRingWords * kWave * 4u;
    // the attribute belongs to the device's copy of the function: once per device and instantiation
    static std::atomic<uint64_t> raised{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(amv_huffman_sync_kernel<L>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTableBytes + kMaxWaves * kPerWave));
        raised.fetch_or(bit, std::memory_order_relaxed);
    }
    // Workgroups: as many as the chip holds (the rest of the tasks come from the queue); a batch that does not fill
    // them gets smaller workgroups, so that its waves spread over all the compute units instead of filling a few.
    const uint32_t groups = cus * 2u;
    const uint32_t tasks = (n + (uint32_t)(kWave / L) - 1u) / (uint32_t)(kWave / L);
    uint32_t waves = (tasks + groups - 1u) / groups;
    if (waves < 4u) waves = 4u;
    if (waves > kMaxWaves) waves = kMaxWaves;
    uint32_t grid = (tasks + waves - 1u) / waves;
    if (grid > groups) grid = groups;
    hipLaunchKernelGGL((amv_huffman_sync_kernel<L>), dim3(grid), dim3(kWave * waves), kTableBytes + waves * kPerWave, s, ws,
                       ws_bytes, n, list, list_count, g.blocks, ws_line, d_img, out, status, nmcu_ok, queue, stats);
}

template <uint32_t kFlush>
void launch_fast(const uint32_t* ws, const uint32_t* ws_bytes, uint32_t n, const uint32_t* list,
                 const uint32_t* list_count, const FrameGeom& g, const uint32_t* ws_line,
                 const HuffDecodeImage* d_img, const SyncOut& out, int32_t* status, uint32_t* nmcu_ok,
                 uint32_t* queue, unsigned long long* stats, uint32_t cus, hipStream_t s) {
    constexpr uint32_t kMaxWaves = fast_waves(kFlush);
    static std::atomic<uint64_t> raised{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(amv_huffman_fast_kernel<kFlush>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kFastTableBytes + kMaxWaves * fast_per_wave(kFlush)));
        raised.fetch_or(bit, std::memory_order_relaxed);
    }
    const uint32_t tasks = (n + (uint32_t)kWave - 1u) / (uint32_t)kWave;
    uint32_t waves = (tasks + cus - 1u) / cus;
    if (waves < 4u) waves = 4u;
    if (waves > kMaxWaves) waves = kMaxWaves;
    // a workgroup on EVERY compute unit the tasks can reach (round 6: 2 500 tasks in workgroups of ten were 250 workgroups
    // on a chip of 256 units; the waves that find the queue empty leave at once)
    const uint32_t grid = tasks < cus ? tasks : cus;
    hipLaunchKernelGGL((amv_huffman_fast_kernel<kFlush>), dim3(grid), dim3(kWave * waves), kFastTableBytes + waves * fast_per_wave(kFlush), s,
                       ws, ws_bytes, n, list, list_count, g.blocks, ws_line, d_img, out, status, nmcu_ok, queue, stats);
}

template <int L>
void launch_sync2(const uint32_t* ws, const uint32_t* ws_bytes, uint32_t n, const uint32_t* list,
                  const uint32_t* list_count, const FrameGeom& g, const uint32_t* ws_line,
                  const HuffDecodeImage* d_img, const SyncOut& out, int32_t* status, uint32_t* nmcu_ok,
                  uint32_t* queue, unsigned long long* stats, uint32_t cus, hipStream_t s) {
    constexpr uint32_t kMaxWaves = fast_waves(8u);
    static std::atomic<uint64_t> raised{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(amv_huffman_sync2_kernel<L>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kFastTableBytes + kMaxWaves * fast_per_wave(8u)));
        raised.fetch_or(bit, std::memory_order_relaxed);
    }
    // as launch_sync: as many workgroups as the chip holds, smaller ones for a batch that does not fill them
    const uint32_t tasks = (n + (uint32_t)(kWave / L) - 1u) / (uint32_t)(kWave / L);
    uint32_t waves = (tasks + cus - 1u) / cus;
    if (waves < 4u) waves = 4u;
    if (waves > kMaxWaves) waves = kMaxWaves;
    const uint32_t grid = tasks < cus ? tasks : cus;   // (every compute unit the tasks can reach: launch_fast)
    hipLaunchKernelGGL((amv_huffman_sync2_kernel<L>), dim3(grid), dim3(kWave * waves), kFastTableBytes + waves * fast_per_wave(8u), s, ws,
                       ws_bytes, n, list, list_count, g.blocks, ws_line, d_img, out, status, nmcu_ok, queue, stats);
}

}  // namespace

// Lanes per frame for a batch of n frames of `pixels` pixels on a device with `cus` compute units.
// More lanes per frame mean shorter walks (a shorter launch when the chip is not full) but more
// speculative work per frame: every lane needs ~4 400 bits to fall in step whatever its share of the
// frame.  Measured on MI355X (entropy kernel, ms; 160x120 unless said): a batch that fills the chip with ONE lane per
// frame does best with exactly that -- no speculative work at all: 160 000 frames 2.11 with one lane, 2.70 with two,
// 3.15 with four; 120 000: 1.65 / 2.20; 80 000: 1.50 / 1.53; 60 000: 1.21 / 1.19 -- so one lane from ~230 frames per CU,
// two from ~150 (40 000: 0.97 with two, 1.01 with eight).  Below that the best share is about the synchronisation
// length and as many lanes as keep every task resident: 20 000: 0.67 with eight; 10 000: 0.49 with sixteen (0.52 / 0.63
// with eight / thirty-two); 320x240: 32 000 frames 1.90 with eight (1.84 with sixteen), 8 000: 0.68 with sixteen,
// 2 000: 0.32 with sixty-four.  `wanted` (a power of two up to 64) overrides.
int huffman_sync_lanes(uint32_t n, uint32_t cus, int wanted, uint64_t pixels, bool records) {
    if (wanted == 1 || wanted == 2 || wanted == 4 || wanted == 8 || wanted == 16 || wanted == 32 || wanted == 64) return wanted;
    if (n >= cus * 230u) return 1;
    if (n >= cus * 150u) return 2;
    // One generation of waves holds 13 tasks per compute unit (fast_waves: the LDS a task needs), and a launch whose tasks are
    // all resident lasts as long as one task does.  Round 6, sweeps on one box (tools/sweep_lanes.sh, profiles/r06_lanes_sweep.txt;
    // entropy kernel, ms): the step between "all resident" and "a second generation" is where the lane count should halve --
    // 160x120, sixteen lanes / eight: 12 000 frames (3 000 tasks) 0.46 / 0.53, 14 000 (3 500 tasks) 0.58 / 0.56; 320x240:
    // 11 000 frames 0.85 / 0.99, 13 000 1.02 / 1.04, 14 000 1.11 / 1.03 -- so the table counts in generations of 13 waves per
    // unit (rounds 4 - 5 had counted ten, the workgroup the launch happened to get).
    // (records: the records form, amv_huffman_sync2_kernel.  The dense form's kernel -- amvhip_huffman_decode_dev, nothing on the
    // decode path -- holds ten waves per unit and was not swept: it keeps the table it had.)
    const uint64_t resident = (uint64_t)cus * (records ? fast_waves(8u) : 10u);
    // Short frames (160x120: 3.5 kB) beyond what eight lanes keep resident: two lanes.  Their launch stays one generation
    // deep up to 106 000 frames and lasts 0.80 ms from 10 000 frames to 30 000 (one task per SIMD, half a frame per lane), where
    // eight lanes in a second generation need 0.88 at 28 000 frames, 0.91 at 30 000 and 0.96 at 34 000 against 0.81 / 0.81 /
    // 0.96.  Long frames do not follow: 320x240, 32 000 frames 2.32 with two lanes against 1.86 with eight (a lane's fixed
    // cost of falling in step is a smaller part of a 14-kB frame's share, eight lanes stay efficient).
    if (records && pixels <= 30000u && (uint64_t)n * 8u > resident * 64u) return 2;
    int full = 8;                                  // chip full: 8 lanes up to 320x240 (32 000 frames: 1.90 ms against 1.84 with 16)
    while (full < 64 && (uint64_t)full * 25000u <= pixels) full *= 2;
    int fill = 8;                                  // small batch: as many lanes as keep every task resident ...
    if (n <= resident) fill = 64;
    else if (n <= 2u * resident) fill = 32;
    else if (n <= 4u * resident) fill = 16;
    // ... but no share much shorter than the synchronisation length, or the re-walk rounds take over.  Round 5 (lanes that
    // remember their walks, finality by prefix scan, idle lanes walking candidates), 160x120, lanes 8 / 16 / 32 / 64:
    // 1 250 frames 0.45 / 0.31 / 0.24 / 0.22 ms with 1.2 / 2.2 / 3.5 / 5.3 rounds, 2 500: 0.45 / 0.33 / 0.25 / 0.29,
    // 5 000: 0.46 / 0.36 / 0.31 / 0.43, 10 000: 0.53 / 0.45 / 0.45 / 0.70, 20 000: 0.68 / 0.69 / 0.83 / 1.20;
    // 320x240, lanes 16 / 32 / 64: 500 frames 0.53 / 0.36 / 0.26, 2 000: 0.54 / 0.38 / 0.33, 8 000: 0.72 / 0.77 / 0.83
    // (round 4's kernel: 1 250 frames 0.32 / 0.27 / 0.33 with 16 / 32 / 64, 2 500: 0.37 / 0.34 / 0.46, 5 000: 0.38 / 0.41 / 0.56)
    int cap = 8;
    while (cap < 64 && (uint64_t)cap * 2u * 500u <= pixels) cap *= 2;
    if (n <= cus * 5u) cap = 64;                  // half of ten waves per unit or less: the shortest shares still win
    if (fill > cap) fill = cap;
    return fill > full ? fill : full;
}

// Heavy and light frames of a batch (launch_split_by_weight): thread per frame, a wave's frames of a class appended with one
// atomic add.  The order inside a list is whatever the waves' turns make it; nothing but the order of work depends on it.
__global__ __launch_bounds__(256) void amv_split_kernel(const uint32_t* __restrict__ lens, uint32_t n, const uint32_t* __restrict__ ws_line,
                                                        uint32_t* __restrict__ heavy, uint32_t* __restrict__ light, uint32_t* __restrict__ count) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u;
    const bool in = i < n;
    // a frame's scan has (length + 47) / 16 pieces of the workspace (entropy_front's layout); twice the mean of that is the line
    const uint64_t all = ws_line[n];
    // (the frame's pieces as the layout counts them: 64-bit, so that a length of 0xffffffd1 or more does not wrap into a light frame)
    const bool is_heavy = in && (((uint64_t)lens[in ? i : 0u] + 47u) >> 4) * n > 2u * all;
    const uint64_t hm = __ballot(is_heavy), lm = __ballot(in && !is_heavy);
    uint32_t hb = 0u, lb = 0u;
    if (lane == 0u) {
        if (hm) hb = atomicAdd(&count[0], (uint32_t)__builtin_popcountll(hm));
        if (lm) lb = atomicAdd(&count[1], (uint32_t)__builtin_popcountll(lm));
    }
    hb = (uint32_t)__shfl((int)hb, 0);
    lb = (uint32_t)__shfl((int)lb, 0);
    const uint64_t below = (1ull << lane) - 1ull;
    if (is_heavy) heavy[hb + (uint32_t)__builtin_popcountll(hm & below)] = i;
    else if (in) light[lb + (uint32_t)__builtin_popcountll(lm & below)] = i;
}

void launch_split_by_weight(const uint32_t* lens, uint32_t n, const uint32_t* ws_line, uint32_t* heavy, uint32_t* light, uint32_t* count,
                            hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(amv_split_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, lens, n, ws_line, heavy, light, count);
}

void launch_unstuff(const uint8_t* blob, uint64_t blob_bytes, const uint64_t* offs, const uint32_t* lens, uint32_t n,
                    const uint32_t* ws_line, uint32_t* ws, uint32_t* ws_bytes, uint32_t* retry_list, uint32_t* retry_count,
                    hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(amv_unstuff_kernel, dim3((n + 3u) / 4u), dim3(256), 0, s, blob, blob_bytes, offs, lens, n,
                       ws_line, ws, ws_bytes, retry_list, retry_count);
}

void launch_huffman_sync(const uint32_t* ws, const uint32_t* ws_bytes, uint32_t n, const uint32_t* list,
                         const uint32_t* list_count, const FrameGeom& g, const uint32_t* ws_line, int lanes_per_frame,
                         const HuffDecodeImage* d_img, const SyncSinks& sinks, int32_t* status, uint32_t* nmcu_ok,
                         uint32_t* queue, unsigned long long* stats, uint32_t cus, hipStream_t s) {
    if (n == 0) return;
    const uint32_t per_row = (g.mcu_cols + kSegMcus - 1u) / kSegMcus;
    SyncOut out{sinks.coef, sinks.rec, sinks.rec_line, sinks.seg_start, sinks.lane_tab, SegGeom{g.mcu_cols, per_row, per_row * g.mcu_rows},
                sinks.lanes, sinks.rec_count, sinks.retry_list, sinks.retry_count, sinks.ok_in_blocks};
#define AMV_SYNC_ARGS ws, ws_bytes, n, list, list_count, g, ws_line, d_img, out, status, nmcu_ok, queue, stats, cus, s
    if (sinks.rec) {
        switch (lanes_per_frame) {
            case 64: launch_sync2<64>(AMV_SYNC_ARGS); break;
            case 32: launch_sync2<32>(AMV_SYNC_ARGS); break;
            case 8: launch_sync2<8>(AMV_SYNC_ARGS); break;
            case 4: launch_sync2<4>(AMV_SYNC_ARGS); break;
            case 2: launch_sync2<2>(AMV_SYNC_ARGS); break;
            case 1: launch_fast<8>(AMV_SYNC_ARGS); break;
            default: launch_sync2<16>(AMV_SYNC_ARGS); break;
        }
    } else {
        switch (lanes_per_frame) {
            case 64: launch_sync<64>(AMV_SYNC_ARGS); break;
            case 32: launch_sync<32>(AMV_SYNC_ARGS); break;
            case 8: launch_sync<8>(AMV_SYNC_ARGS); break;
            case 4: launch_sync<4>(AMV_SYNC_ARGS); break;
            case 2: launch_sync<2>(AMV_SYNC_ARGS); break;
            case 1: launch_sync<1>(AMV_SYNC_ARGS); break;
            default: launch_sync<16>(AMV_SYNC_ARGS); break;
        }
    }
#undef AMV_SYNC_ARGS
}

// Space per frame in the two workspaces between the decode stages (round 4; one stride for every frame before -- the
// record space sized from the batch's MEAN chunk, the scan windows from the picture size: a heavy frame among light ones
// went to the serial kernel): three small launches lay both out from the chunk lengths -- per-workgroup sums, a scan of
// the sums in one workgroup, and the offsets.
namespace {
constexpr uint32_t kLayoutBlock = 256;

__device__ __forceinline__ uint32_t layout_lines(uint32_t len, const LayoutSpec& sp) {
    const uint64_t want = ((uint64_t)len * sp.per_byte_x2 >> 1) + sp.add;          // (a chunk length is whatever the caller wrote there)
    return (uint32_t)((min(want, (uint64_t)sp.hi) + ((1u << sp.unit_shift) - 1u)) >> sp.unit_shift);
}

// exclusive scan of two values over the workgroup's 256 threads; returns the workgroup's totals in tot.  The sums are
// 64 bits wide: a frame may claim up to 0x0fffffff pieces (a chunk length is whatever the caller wrote there), and 17 of
// those in one workgroup would wrap a 32-bit prefix -- a frame's window would then START BELOW its predecessor's.
struct Sum2 { uint64_t x, y; };
__device__ __forceinline__ Sum2 block_scan2(uint2 v, Sum2* s_wave, Sum2& tot) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    Sum2 inc{v.x, v.y};
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t ax = (uint64_t)__shfl_up((unsigned long long)inc.x, d), ay = (uint64_t)__shfl_up((unsigned long long)inc.y, d);
        if (lane >= (uint32_t)d) { inc.x += ax; inc.y += ay; }
    }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    Sum2 base{0u, 0u};
    tot = Sum2{0u, 0u};
    for (uint32_t k = 0; k < kLayoutBlock / 64u; ++k) {
        const Sum2 w = s_wave[k];
        if (k < wave) { base.x += w.x; base.y += w.y; }
        tot.x += w.x; tot.y += w.y;
    }
    return Sum2{base.x + inc.x - v.x, base.y + inc.y - v.y};
}
__device__ __forceinline__ uint32_t sat32(uint64_t v) { return (uint32_t)min(v, (uint64_t)0xffffffffu); }
}  // namespace

__global__ __launch_bounds__(kLayoutBlock) void amv_layout_sums_kernel(const uint32_t* __restrict__ lens, uint32_t n, LayoutSpec a, LayoutSpec b,
                                                                      uint2* __restrict__ sums) {
    __shared__ Sum2 s_wave[kLayoutBlock / 64u];
    const uint32_t i = blockIdx.x * kLayoutBlock + threadIdx.x;
    const uint32_t len = i < n ? lens[i] : 0u;
    Sum2 tot;
    (void)block_scan2(i < n ? make_uint2(layout_lines(len, a), b.line ? layout_lines(len, b) : 0u) : make_uint2(0u, 0u), s_wave, tot);
    if (threadIdx.x == 0u) sums[blockIdx.x] = make_uint2(sat32(tot.x), sat32(tot.y));   // (both capacities are below 2^32)
}

// sums[0 .. nb) -> exclusive prefix, saturating at the two capacities; sums[nb] = the totals
__global__ __launch_bounds__(1024) void amv_layout_scan_kernel(uint2* __restrict__ sums, uint32_t nb, uint32_t cap_a, uint32_t cap_b) {
    __shared__ uint64_t s_x[1024], s_y[1024];
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t lo = min(threadIdx.x * per, nb), up = min(lo + per, nb);
    uint64_t mx = 0, my = 0;
    for (uint32_t i = lo; i < up; ++i) { mx += sums[i].x; my += sums[i].y; }
    s_x[threadIdx.x] = mx; s_y[threadIdx.x] = my;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint64_t lx = threadIdx.x >= d ? s_x[threadIdx.x - d] : 0u, ly = threadIdx.x >= d ? s_y[threadIdx.x - d] : 0u;
        __syncthreads();
        s_x[threadIdx.x] += lx; s_y[threadIdx.x] += ly;
        __syncthreads();
    }
    uint64_t ax = threadIdx.x ? s_x[threadIdx.x - 1u] : 0u, ay = threadIdx.x ? s_y[threadIdx.x - 1u] : 0u;
    for (uint32_t i = lo; i < up; ++i) {
        const uint2 v = sums[i];
        sums[i] = make_uint2((uint32_t)min(ax, (uint64_t)cap_a), (uint32_t)min(ay, (uint64_t)cap_b));
        ax += v.x; ay += v.y;
    }
    if (threadIdx.x == 0u) sums[nb] = make_uint2((uint32_t)min(s_x[1023], (uint64_t)cap_a), (uint32_t)min(s_y[1023], (uint64_t)cap_b));
}

__global__ __launch_bounds__(kLayoutBlock) void amv_layout_write_kernel(const uint32_t* __restrict__ lens, uint32_t n, LayoutSpec a, LayoutSpec b,
                                                                       const uint2* __restrict__ sums, uint32_t nb) {
    __shared__ Sum2 s_wave[kLayoutBlock / 64u];
    const uint32_t i = blockIdx.x * kLayoutBlock + threadIdx.x;
    const uint32_t len = i < n ? lens[i] : 0u;
    Sum2 tot;
    const Sum2 at = block_scan2(i < n ? make_uint2(layout_lines(len, a), b.line ? layout_lines(len, b) : 0u) : make_uint2(0u, 0u), s_wave, tot);
    const uint2 base = sums[blockIdx.x];
    if (i < n) {                                               // monotone: base and the prefix never decrease, the minimum keeps that
        a.line[i] = (uint32_t)min((uint64_t)base.x + at.x, (uint64_t)a.cap_lines);
        if (b.line) b.line[i] = (uint32_t)min((uint64_t)base.y + at.y, (uint64_t)b.cap_lines);
    }
    if (i == 0u) {                                             // the end of the last frame's space
        a.line[n] = sums[nb].x;
        if (b.line) b.line[n] = sums[nb].y;
    }
}

// The three launches above as ONE workgroup, for batches of up to kLayoutSmall frames (round 6): a 10 000-frame stream's
// decode is nine launches of which these three and the memset of the retry counters behind them do next to nothing --
// 5 us each of a 0.7-ms step, and of the 0.3 ms of a rank's 1 250-frame share of it.  Sixteen threads take a block of 256
// frames, sixteen frames each; the arithmetic is the three kernels' own, saturation for saturation (a block's total
// through sat32, the blocks' prefix clamped at the capacity, a frame's line = min(prefix + its place in the block, capacity)),
// so the lines are the same numbers whatever route made them (tests/test_gpu_parity.py::test_decode_lengths_that_overflow_the_layout runs both).
// zero[0 .. nzero) is cleared on the way: the retry counter and the task queues of the kernels behind.
constexpr uint32_t kLayoutSmall = 16384;
__global__ __launch_bounds__(1024) void amv_layout_small_kernel(const uint32_t* __restrict__ lens, uint32_t n, LayoutSpec a, LayoutSpec b,
                                                                uint32_t* __restrict__ zero, uint32_t nzero) {
    __shared__ uint64_t s_tx[kLayoutSmall / kLayoutBlock], s_ty[kLayoutSmall / kLayoutBlock];   // per block: total, then prefix
    const uint32_t t = threadIdx.x, blk = t >> 4, sub = t & 15u;
    const uint32_t nb = (n + kLayoutBlock - 1u) / kLayoutBlock;
    if (t < nzero) zero[t] = 0u;
    const uint32_t first = blk * kLayoutBlock + sub * 16u;
    uint32_t va[16], vb[16];
    uint64_t mx = 0, my = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) {
        const uint32_t i = first + k;
        const uint32_t len = i < n ? lens[i] : 0u;
        va[k] = i < n ? layout_lines(len, a) : 0u;
        vb[k] = i < n && b.line ? layout_lines(len, b) : 0u;
        mx += va[k]; my += vb[k];
    }
    // the sixteen threads of a block: inclusive scan of their sums (they are sixteen consecutive lanes of one wave)
    uint64_t ix = mx, iy = my;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        const uint64_t ax = (uint64_t)__shfl_up((unsigned long long)ix, d, 16), ay = (uint64_t)__shfl_up((unsigned long long)iy, d, 16);
        if (sub >= (uint32_t)d) { ix += ax; iy += ay; }
    }
    if (sub == 15u) { s_tx[blk] = sat32(ix); s_ty[blk] = sat32(iy); }   // amv_layout_sums_kernel: the block's total
    __syncthreads();
    if (t < 64u) {   // amv_layout_scan_kernel: exclusive prefix of the blocks' totals, clamped at the capacities
        const uint32_t cap_a = a.cap_lines, cap_b = b.line ? b.cap_lines : 0u;
        const uint64_t vx = t < nb ? s_tx[t] : 0u, vy = t < nb ? s_ty[t] : 0u;
        uint64_t px = vx, py = vy;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t ax = (uint64_t)__shfl_up((unsigned long long)px, d), ay = (uint64_t)__shfl_up((unsigned long long)py, d);
            if (t >= (uint32_t)d) { px += ax; py += ay; }
        }
        const uint64_t allx = (uint64_t)__shfl((unsigned long long)px, 63), ally = (uint64_t)__shfl((unsigned long long)py, 63);
        s_tx[t] = min(px - vx, (uint64_t)cap_a);
        s_ty[t] = min(py - vy, (uint64_t)cap_b);
        if (t == 0u) {                                         // the end of the last frame's space
            a.line[n] = (uint32_t)min(allx, (uint64_t)cap_a);
            if (b.line) b.line[n] = (uint32_t)min(ally, (uint64_t)cap_b);
        }
    }
    __syncthreads();
    // amv_layout_write_kernel: base of the block + the frame's place in it
    uint64_t atx = s_tx[blk] + (ix - mx), aty = s_ty[blk] + (iy - my);
    uint32_t la[16], lb[16];
#pragma unroll
    for (uint32_t k = 0; k < 16u; ++k) {
        la[k] = (uint32_t)min(atx, (uint64_t)a.cap_lines);
        lb[k] = (uint32_t)min(aty, (uint64_t)b.cap_lines);
        atx += va[k]; aty += vb[k];
    }
    if (first + 16u <= n) {   // the thread's sixteen lines as four 16-byte stores (the arrays are the context's own: aligned)
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            reinterpret_cast<uint4*>(a.line + first)[q] = make_uint4(la[4 * q], la[4 * q + 1], la[4 * q + 2], la[4 * q + 3]);
            if (b.line) reinterpret_cast<uint4*>(b.line + first)[q] = make_uint4(lb[4 * q], lb[4 * q + 1], lb[4 * q + 2], lb[4 * q + 3]);
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) {
            if (first + k < n) {
                a.line[first + k] = la[k];
                if (b.line) b.line[first + k] = lb[k];
            }
        }
    }
}

uint64_t layout_workspace(uint32_t n) { return ((uint64_t)(n + kLayoutBlock - 1u) / kLayoutBlock + 1u) * sizeof(uint2); }

// zero[0 .. nzero): words the launches behind the layout expect cleared (nzero <= 64); force_large: the three-launch route
// whatever the batch size (the test that holds the two routes to the same numbers)
void launch_layout(const uint32_t* lens, uint32_t n, const LayoutSpec& a, const LayoutSpec& b, void* work, uint32_t* zero, uint32_t nzero,
                   bool force_large, hipStream_t s) {
    if (n == 0) return;
    if (n <= kLayoutSmall && !force_large) {
        // a thread per sixteen frames, whole waves (the blocks' scan wants the first one whole): 1 250 frames are two waves
        const uint32_t threads = (((n + 15u) / 16u) + 63u) & ~63u;
        hipLaunchKernelGGL(amv_layout_small_kernel, dim3(1), dim3(threads), 0, s, lens, n, a, b, zero, nzero);
        return;
    }
    if (nzero) (void)hipMemsetAsync(zero, 0, (size_t)nzero * 4u, s);
    const uint32_t nb = (n + kLayoutBlock - 1u) / kLayoutBlock;
    uint2* sums = static_cast<uint2*>(work);
    hipLaunchKernelGGL(amv_layout_sums_kernel, dim3(nb), dim3(kLayoutBlock), 0, s, lens, n, a, b, sums);
    hipLaunchKernelGGL(amv_layout_scan_kernel, dim3(1), dim3(1024), 0, s, sums, nb, a.cap_lines, b.line ? b.cap_lines : 0u);
    hipLaunchKernelGGL(amv_layout_write_kernel, dim3(nb), dim3(kLayoutBlock), 0, s, lens, n, a, b, (const uint2*)sums, nb);
}

}  // namespace amv
