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
//   amv_huffman_sync_kernel<L> (L lanes per frame, 64/L frames per wave)
//   1. the bit stream is cut into L equal subsequences; lane i walks subsequence i from a GUESSED
//      state (its first bit, "the DC symbol of block 0 comes next") up to the first symbol boundary
//      past its end and remembers the state it arrives in: (bit, index in block, block in MCU).
//      A guessed start that meets an invalid code or an over-long run does not stop (that would
//      stall every lane to its right until exact states arrive one lane per round): it slips a
//      bit / closes the block and carries on.  The true decoder of a valid stream never takes
//      those branches, and a stream with a real error is caught in pass 4, which is strict;
//   2. every lane takes its left neighbour's arrival state as its start state and walks again if
//      that changed.  Lane 0's start is exact, so after round r lanes 0..r are exact; in practice
//      wrong starts fall into step with the true decoder after a few MCUs (the slow part is the
//      luma/chroma phase of the MCU) and the loop ends early (worst case L-1 rounds: still correct).
//      These walks use "skip" tables (symbol length, index advance) and touch no coefficient;
//   3. a prefix sum of "blocks finished per lane" gives every lane its first block number;
//   4. one strict pass decodes values and writes them (2-byte stores into the frame's zeroed
//      coefficient lines), parking DC differences in LDS;
//   5. the three DC predictors (ycoef/ucoef/vcoef, AmvJpeg.c:1200-1221) become a prefix sum over
//      the parked differences.
//
// Statuses equal the serial kernel's bit for bit (tests): the first error on the true path stops
// the frame, nmcu_ok counts whole MCUs before it, TRUNCATED compares consumed with stored bits.
// Chunks larger than the per-frame workspace window, or with a run of FF bytes longer than the
// 7-byte look-back of the unstuffer (never in a valid stream), are queued for amv_huffman_kernel.
#include "amv_kernels.h"

namespace amv {

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerGroup = 4;
constexpr uint32_t kNever = 0xffffffffu;

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

__device__ __forceinline__ uint32_t load_word(const uint32_t* __restrict__ words, uint32_t nwords, uint32_t i) {
    return i < nwords ? words[i] : 0u;   // past the stored bits the stream reads as zeros
}

}  // namespace

// =============================================================================================
// unstuffing
// =============================================================================================

__global__ __launch_bounds__(kWave* kWavesPerGroup) void amv_unstuff_kernel(
    const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offs,
    const uint32_t* __restrict__ lens, uint32_t n, uint32_t cap_words, uint32_t* __restrict__ ws,
    uint32_t* __restrict__ ws_bytes, uint32_t* __restrict__ retry_list, uint32_t* __restrict__ retry_count) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t frame = blockIdx.x * kWavesPerGroup + wave;
    if (frame >= n) return;
    uint64_t off = offs[frame];
    uint32_t len = lens[frame];
    if (off > blob_bytes) { off = blob_bytes; len = 0; }
    if ((uint64_t)len > blob_bytes - off) len = (uint32_t)(blob_bytes - off);
    uint8_t* out = reinterpret_cast<uint8_t*>(ws + (uint64_t)frame * cap_words);

    const uint32_t mis = (uint32_t)(off & 3u);
    const uint8_t* base = blob + (off - mis);
    const uint64_t guard = blob_bytes - (off - mis);
    const uint32_t first = mis + 2u, end = mis + len;      // data bytes [first, end), relative to base
    bool retry = len > 2u && (len - 2u) > cap_words * 4u;   // does not fit its window
    uint32_t total = 0;
    if (!retry) {
        uint32_t prev_word = 0;
        for (uint32_t t0 = 0; t0 * 4u < end; t0 += kWave) {
            const uint32_t wi = t0 + lane;
            uint32_t w = 0;
            if (wi * 4u < end) {
                const uint64_t bo = (uint64_t)wi * 4u;
                if (bo + 4u <= guard) w = *reinterpret_cast<const uint32_t*>(base + bo);
                else for (uint32_t q = 0; q < 4u; ++q) if (bo + q < guard) w |= (uint32_t)base[bo + q] << (8u * q);
            }
            uint32_t q = __shfl_up(w, 1);
            if (lane == 0) q = prev_word;
            prev_word = __shfl(w, kWave - 1);
            const uint64_t both = ((uint64_t)w << 32) | q;   // byte j of w sits at bit 32 + 8j
            uint32_t keep = 0, cnt = 0;
            bool deep = false;
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                const uint32_t pos = wi * 4u + j;
                uint32_t run = 0;   // FF bytes immediately before pos, never counting bytes before `first`
#pragma unroll
                for (uint32_t back = 1; back <= 4u + j; ++back) {
                    const bool ff = pos >= first + back && ((both >> (32u + 8u * j - 8u * back)) & 0xffu) == 0xffu;
                    if (ff && run == back - 1u) run = back;
                }
                if (run == 4u + j && pos > first + run) deep = true;   // the run reaches past the look-back
                const bool kept = pos >= first && pos < end && !(run & 1u);   // dropped iff an odd run of FF precedes it
                keep |= (kept ? 1u : 0u) << j;
                cnt += kept ? 1u : 0u;
            }
            if (__any(deep)) { retry = true; break; }
            uint32_t tile_total;
            uint32_t d = total + seg_excl_sum<kWave>(cnt, lane, tile_total);
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j)
                if (keep & (1u << j)) { out[d ^ 3u] = (uint8_t)(w >> (8u * j)); ++d; }   // big-endian inside the word
            total += tile_total;
        }
        // zero the tail of the last word
        if (!retry && lane < ((4u - (total & 3u)) & 3u)) out[(total + lane) ^ 3u] = 0;
    }
    if (lane == 0) {
        ws_bytes[frame] = retry ? kNever : total;
        if (retry) retry_list[atomicAdd(retry_count, 1u)] = frame;
    }
}

// =============================================================================================
// walks
// =============================================================================================

namespace {

// Speculative walk from `s` while s.p < limit: where symbols start and how the block position
// moves, nothing else.  Returns the number of blocks finished.  The next stream word is always
// one load ahead of its use.
__device__ __forceinline__ uint32_t walk_skip(const uint32_t* __restrict__ words, uint32_t nwords,
                                              const uint16_t* __restrict__ s1, const uint16_t* __restrict__ s2,
                                              State& s, uint32_t limit) {
    uint32_t p = s.p, k = s.k, k6 = s.k6, nblk = 0;
    if (p >= limit) return 0u;
    uint32_t widx = p >> 5;
    const uint32_t bo = p & 31u;
    uint64_t acc = (((uint64_t)load_word(words, nwords, widx) << 32) | load_word(words, nwords, widx + 1u)) << bo;
    int nb = 64 - (int)bo;
    widx += 2u;
    uint32_t nextw = load_word(words, nwords, widx);
    uint32_t tab = ((k ? 2u : 0u) + (k6 >= 4u ? 1u : 0u)) << kLut1Bits;
    do {
        const uint32_t v = (uint32_t)(acc >> 32);
        uint32_t e = s1[tab + (v >> (32 - kLut1Bits))];
        if (e & 0x8000u) e = s2[((e & 0xffu) << kLut2Bits) | ((v >> (32 - kLut1Bits - kLut2Bits)) & ((1u << kLut2Bits) - 1u))];
        uint32_t used = e & 31u;
        used = used ? used : 1u;                 // nonsense under a guessed start: slip one bit
        const uint32_t kn = k + ((e >> 8) & 127u);
        acc <<= used;
        nb -= (int)used;
        p += used;
        if (nb <= 32) {
            acc |= (uint64_t)nextw << (32 - nb);
            nb += 32;
            ++widx;
            nextw = load_word(words, nwords, widx);
        }
        const bool end = kn >= 64u;              // end of block, a full block, or an over-long run
        k = end ? 0u : kn;
        k6 = end ? (k6 == 5u ? 0u : k6 + 1u) : k6;
        nblk += end ? 1u : 0u;
        tab = ((k ? 2u : 0u) + (k6 >= 4u ? 1u : 0u)) << kLut1Bits;
    } while (p < limit);
    s.p = p; s.k = k; s.k6 = k6;
    return nblk;
}

struct WriteResult {
    uint32_t err;       // kStFormat / kStOverrun when the walk hit a real error
    uint32_t err_blk;   // absolute block the error hit
    uint32_t stop_p;    // bits consumed when the walk ended (FORMAT: incl. the reference's 17-bit give-up)
    bool done;          // the frame's last block was finished here
};

// The strict, writing walk (HufBlock / DecodeElement, AmvJpeg.c:842-974) from an exact state.
__device__ __forceinline__ WriteResult walk_write(const uint32_t* __restrict__ words, uint32_t nwords,
                                                  const uint16_t* __restrict__ l1, const uint16_t* __restrict__ l2,
                                                  State s, uint32_t limit, uint32_t blk, uint32_t blocks_per_frame,
                                                  int16_t* __restrict__ coef, int16_t* __restrict__ dc) {
    WriteResult r{0u, 0u, 0u, false};
    uint32_t p = s.p, k = s.k, k6 = s.k6;
    uint32_t widx = p >> 5;
    const uint32_t bo = p & 31u;
    uint64_t acc = (((uint64_t)load_word(words, nwords, widx) << 32) | load_word(words, nwords, widx + 1u)) << bo;
    int nb = 64 - (int)bo;
    widx += 2u;
    uint32_t nextw = load_word(words, nwords, widx);
    while (p < limit) {
        const uint32_t v = (uint32_t)(acc >> 32);
        const uint32_t tab = (k == 0u ? 0u : 2u) + (k6 >= 4u ? 1u : 0u);
        uint32_t e = l1[(tab << kLut1Bits) + (v >> (32 - kLut1Bits))];
        if (e & 0x8000u) e = l2[((e & 0xffu) << kLut2Bits) | ((v >> (32 - kLut1Bits - kLut2Bits)) & ((1u << kLut2Bits) - 1u))];
        const uint32_t len = (e >> 8) & 31u;
        if (len == 0u) {                         // no code matches: FUNC_FORMAT_ERROR, AmvJpeg.c:887
            r.err = kStFormat; r.err_blk = blk; r.stop_p = p + 17u;
            break;
        }
        const uint32_t sym = e & 0xffu, size = sym & 15u;
        int val = 0;
        if (size) {
            const uint32_t mag = (v << len) >> (32u - size);
            val = (int)mag;
            if (mag < (1u << (size - 1u))) val -= (1 << size) - 1;   // :924-933
        }
        const uint32_t used = len + size;
        acc <<= used;
        nb -= (int)used;
        p += used;
        if (nb <= 32) {
            acc |= (uint64_t)nextw << (32 - nb);
            nb += 32;
            ++widx;
            nextw = load_word(words, nwords, widx);
        }
        bool block_end = false;
        if (k == 0u) {                           // DC difference (:945-951)
            dc[blk] = (int16_t)val;
            k = 1u;
        } else if (sym == 0u) {                  // end of block (:959-964)
            block_end = true;
        } else {
            k += sym >> 4;
            if (k > 63u) {                       // the reference writes out of bounds here (:967-969)
                r.err = kStOverrun; r.err_blk = blk; r.stop_p = p;
                break;
            }
            if (size) coef[(uint64_t)blk * 64u + k] = (int16_t)val;
            block_end = ++k == 64u;
        }
        if (block_end) {
            k = 0u;
            k6 = k6 == 5u ? 0u : k6 + 1u;
            if (++blk == blocks_per_frame) { r.done = true; r.stop_p = p; break; }
        }
    }
    return r;
}

}  // namespace

// dynamic LDS: [ HuffDecodeImage 16 KB | per wave: dc[64/L][dc_cap] ]
template <int L>
__global__ __launch_bounds__(kWave* kWavesPerGroup) void amv_huffman_sync_kernel(
    const uint32_t* __restrict__ ws, const uint32_t* __restrict__ ws_bytes, uint32_t n,
    uint32_t blocks_per_frame, uint32_t cap_words, uint32_t dc_cap, const HuffDecodeImage* __restrict__ img,
    int16_t* __restrict__ coef, int32_t* __restrict__ status, uint32_t* __restrict__ nmcu_ok,
    unsigned long long* __restrict__ stats) {
    constexpr int kFrames = kWave / L;   // frames per wave
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    const uint16_t* s_l1 = reinterpret_cast<const uint16_t*>(s_mem);
    const uint16_t* s_l2 = s_l1 + (4 << kLut1Bits);
    const uint16_t* s_s1 = s_l2 + (kLut2Pages << kLut2Bits);
    const uint16_t* s_s2 = s_s1 + (4 << kLut1Bits);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t slot = lane / L, sub = lane % L;

    {
        const uint4* src = reinterpret_cast<const uint4*>(img);
        uint4* dst = reinterpret_cast<uint4*>(s_mem);
        for (uint32_t i = threadIdx.x; i < (uint32_t)sizeof(HuffDecodeImage) / 16u; i += kWave * kWavesPerGroup) dst[i] = src[i];
    }
    __syncthreads();   // the only workgroup-wide barrier

    const uint32_t frame = (blockIdx.x * kWavesPerGroup + wave) * kFrames + slot;
    const uint32_t total = frame < n ? ws_bytes[frame] : kNever;
    const bool live = total != kNever;            // not: past the batch, or handed to the serial kernel
    const uint32_t* words = ws + (uint64_t)(live ? frame : 0) * cap_words;
    int16_t* fcoef = coef + (uint64_t)(live ? frame : 0) * blocks_per_frame * 64u;
    int16_t* dc = reinterpret_cast<int16_t*>(s_mem + sizeof(HuffDecodeImage)) + (wave * kFrames + slot) * dc_cap;
    const uint32_t valid_bits = live ? total * 8u : 0u;
    const uint32_t nwords = live ? (total + 3u) >> 2 : 0u;

    const bool timing = stats != nullptr && lane == 0;   // optional phase clock (amvhip_entropy_stats)
    unsigned long long tc[6] = {0, 0, 0, 0, 0, 0};
    if (timing) tc[0] = clock64();

    // ---- 0. zero this frame's coefficient lines
    if (live) {
        uint4* z = reinterpret_cast<uint4*>(fcoef);
        for (uint32_t i = sub; i < blocks_per_frame * 8u; i += L) z[i] = make_uint4(0, 0, 0, 0);
    }

    // ---- 1/2. speculative walks until every lane's start state equals its neighbour's arrival
    uint32_t S = ((valid_bits + L - 1) / L + 31u) & ~31u;   // bits per lane, a whole number of words
    if (S < 64u) S = 64u;
    // the last lane of a frame has no right neighbour to feed: it only runs in pass 4
    const uint32_t limit = (!live || sub == L - 1) ? 0u : (sub + 1u) * S;
    State entry{sub * S, 0u, 0u}, arrive = entry;
    if (timing) tc[1] = clock64();
    uint32_t my_blocks = walk_skip(words, nwords, s_s1, s_s2, arrive, limit);
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
            my_blocks = walk_skip(words, nwords, s_s1, s_s2, arrive, limit);
        }
    }

    if (timing) tc[3] = clock64();
    // ---- 3. first block of every lane
    uint32_t all_blocks;
    const uint32_t blk0 = seg_excl_sum<L>(my_blocks, sub, all_blocks);

    // ---- 4. the strict, writing pass.  Lanes left of the frame's end (or first error) are exact;
    // whatever a lane to the right of it does is ignored below.
    __builtin_amdgcn_s_waitcnt(0);   // the zeroing stores have landed before the sparse ones go out
    WriteResult wr{0u, 0u, 0u, false};
    if (live && blk0 < blocks_per_frame)
        wr = walk_write(words, nwords, s_l1, s_l2, entry, sub == L - 1 ? kNever : limit, blk0, blocks_per_frame, fcoef, dc);
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
    const uint32_t mcus = good_blocks / 6u;

    // ---- 5. DC prediction over whole MCUs: value = running sum of the component's differences
    wave_sync();
    const uint32_t per = (mcus + L - 1) / L;
    const uint32_t m_lo = live ? min(mcus, sub * per) : 0u, m_hi = live ? min(mcus, m_lo + per) : 0u;
    uint32_t sy = 0, su = 0, sv = 0;
    for (uint32_t m = m_lo; m < m_hi; ++m) {
        const int16_t* d = dc + m * 6u;
        sy += (uint32_t)(d[0] + d[1] + d[2] + d[3]);
        su += (uint32_t)d[4];
        sv += (uint32_t)d[5];
    }
    uint32_t tot;
    uint32_t py = seg_excl_sum<L>(sy, sub, tot), pu = seg_excl_sum<L>(su, sub, tot), pv = seg_excl_sum<L>(sv, sub, tot);
    for (uint32_t m = m_lo; m < m_hi; ++m) {
        const int16_t* d = dc + m * 6u;
        int16_t* o = fcoef + (uint64_t)m * 384u;
#pragma unroll
        for (int q = 0; q < 4; ++q) { py += (uint32_t)d[q]; o[q * 64] = (int16_t)py; }
        pu += (uint32_t)d[4]; o[256] = (int16_t)pu;
        pv += (uint32_t)d[5]; o[320] = (int16_t)pv;
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
        nmcu_ok[frame] = mcus;
    }
}

namespace {

template <int L>
void launch_sync(const uint32_t* ws, const uint32_t* ws_bytes, uint32_t n, const FrameGeom& g, uint32_t cap_words,
                 const HuffDecodeImage* d_img, int16_t* coef, int32_t* status, uint32_t* nmcu_ok,
                 unsigned long long* stats, hipStream_t s) {
    constexpr int kFrames = kWave / L;
    const uint32_t dc_cap = (g.blocks + 7u) & ~7u;
    const uint32_t lds = (uint32_t)sizeof(HuffDecodeImage) + kWavesPerGroup * kFrames * dc_cap * 2u;
    static bool raised = false;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(amv_huffman_sync_kernel<L>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    const uint32_t per_group = kWavesPerGroup * kFrames;
    hipLaunchKernelGGL(amv_huffman_sync_kernel<L>, dim3((n + per_group - 1) / per_group), dim3(kWave * kWavesPerGroup),
                       lds, s, ws, ws_bytes, n, g.blocks, cap_words, dc_cap, d_img, coef, status, nmcu_ok, stats);
}

}  // namespace

bool huffman_sync_fits(const FrameGeom& g, int lanes_per_frame) {
    const uint32_t dc_cap = (g.blocks + 7u) & ~7u;
    return sizeof(HuffDecodeImage) + (size_t)kWavesPerGroup * (kWave / lanes_per_frame) * dc_cap * 2u <= 150u * 1024u;
}

void launch_unstuff(const uint8_t* blob, uint64_t blob_bytes, const uint64_t* offs, const uint32_t* lens, uint32_t n,
                    uint32_t cap_words, uint32_t* ws, uint32_t* ws_bytes, uint32_t* retry_list, uint32_t* retry_count,
                    hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(amv_unstuff_kernel, dim3((n + kWavesPerGroup - 1) / kWavesPerGroup), dim3(kWave * kWavesPerGroup), 0,
                       s, blob, blob_bytes, offs, lens, n, cap_words, ws, ws_bytes, retry_list, retry_count);
}

void launch_huffman_sync(const uint32_t* ws, const uint32_t* ws_bytes, uint32_t n, const FrameGeom& g, uint32_t cap_words,
                         int lanes_per_frame, const HuffDecodeImage* d_img, int16_t* coef, int32_t* status,
                         uint32_t* nmcu_ok, unsigned long long* stats, hipStream_t s) {
    if (n == 0) return;
    switch (lanes_per_frame) {
        case 64: launch_sync<64>(ws, ws_bytes, n, g, cap_words, d_img, coef, status, nmcu_ok, stats, s); break;
        case 32: launch_sync<32>(ws, ws_bytes, n, g, cap_words, d_img, coef, status, nmcu_ok, stats, s); break;
        case 8: launch_sync<8>(ws, ws_bytes, n, g, cap_words, d_img, coef, status, nmcu_ok, stats, s); break;
        default: launch_sync<16>(ws, ws_bytes, n, g, cap_words, d_img, coef, status, nmcu_ok, stats, s); break;
    }
}

}  // namespace amv
