// amv_encode_par.hip -- the entropy coder with parallelism inside a frame (gfx950).
//
// Unlike decoding, encoding has no serial chain that cannot be cut: the DC difference of a block
// needs only the previous block of its component (mjpegenc.c:390-401), and once every block's code
// length is known the bit position of every block is a prefix sum.  One team of lanes per frame (one
// wave for small frames, four waves from 256 blocks, eight from 1 024 blocks up -- the whole workgroup):
//
//   1. each lane takes blocks lane, lane+team size, ...: loads the block's 64 quantised coefficients (one
//      128-byte line) into registers and adds up the length of its code (encode_block,
//      mjpegenc.c:379-435, without writing);
//   2. a prefix sum over the block lengths (wave scan, plus partial sums through LDS across the
//      waves of a team) gives every block's first bit;
//   3. each lane codes its blocks again, this time OR-ing the bits into the frame's bit string in
//      LDS (32-bit big-endian words; neighbouring blocks share words, hence the atomic OR);
//   4. the tail is padded with ones (ff_mjpeg_encode_stuffing :338-343), FF bytes are counted and
//      the string is written out as FF D8, bytes with 00 after every FF (escape_FF :282-336), FF D9.
//
// Frames whose bit string does not fit the LDS window are left to amv_pack_kernel (one lane per
// frame), through the same kind of hand-back list the decoder uses.  Output is byte-identical to
// that kernel's and to the CPU oracle's (tests).
#include "amv_kernels.h"

namespace amv {

namespace {

constexpr int kWave = 64;
constexpr int kMaxWaves = 8;   // frames per workgroup (they share the code book); fewer when the window is large

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

__device__ __forceinline__ int coef_at(const uint32_t (&c)[32], int i) {
    return (i & 1) ? ((int)c[i >> 1] >> 16) : (int)(int16_t)(c[i >> 1] & 0xffffu);
}

// Bits appended to a bit string in LDS.  kCount: lengths only.  kShared: at an arbitrary bit position of the frame's
// string (neighbouring blocks share words, hence the atomic OR).  kOwn: into the lane's own scratch column (word i of
// lane l at words[i * stride]; words past `cap` are dropped: the lane then codes its blocks a second time, into the string).
enum { kCount = 0, kShared = 1, kOwn = 2 };
struct Emitter {
    uint32_t* words;
    uint32_t wi;        // word being filled
    uint64_t acc;       // pending bits, right aligned
    int nacc;
    uint32_t nbits;     // total length so far (every mode)
    uint32_t stride, cap;   // kOwn
};

template <int kMode>
__device__ __forceinline__ void put(Emitter& e, uint32_t entry, int extra_bits, uint32_t extra) {
    const int len = (int)(entry >> 16) + extra_bits;
    e.nbits += (uint32_t)len;
    if (kMode == kCount) return;
    e.acc = (e.acc << len) | ((uint64_t)(entry & 0xffffu) << extra_bits) | extra;
    e.nacc += len;
    if (e.nacc >= 32) {
        e.nacc -= 32;
        const uint32_t word = (uint32_t)(e.acc >> e.nacc);
        if (kMode == kShared) atomicOr(&e.words[e.wi], word);
        else if (e.wi < e.cap) e.words[e.wi * e.stride] = word;
        ++e.wi;
    }
}

// encode_block (mjpegenc.c:379-435) over a block held in registers; prev_dc = the component's predictor
template <int kEmit>
__device__ __forceinline__ void code_block(Emitter& e, const uint32_t (&c)[32], int prev_dc, const uint32_t* dcbook,
                                           const uint32_t* acbook) {
    int diff = coef_at(c, 0) - prev_dc;
    {   // ff_mjpeg_encode_dc :357-377
        int mant = diff;
        if (diff < 0) { diff = -diff; mant--; }
        const int nb = 32 - __clz(diff);   // 0 for diff == 0
        put<kEmit>(e, dcbook[nb], nb, (uint32_t)mant & ((1u << nb) - 1u));
    }
    int run = 0;
#pragma unroll
    for (int k = 1; k < 64; ++k) {
        int v = coef_at(c, k);
        if (v == 0) { ++run; continue; }
        while (run >= 16) { put<kEmit>(e, acbook[0xf0], 0, 0u); run -= 16; }   // ZRL :408-411
        int mant = v;
        if (v < 0) { v = -v; mant--; }
        const int nb = 32 - __clz(v);
        put<kEmit>(e, acbook[(run << 4) | nb], nb, (uint32_t)mant & ((1u << nb) - 1u));
        run = 0;
    }
    if (run) put<kEmit>(e, acbook[0], 0, 0u);   // EOB :430-431
}

// the block whose DC is this block's predictor (same component), or -1 for the first MCU
__device__ __forceinline__ int pred_block(uint32_t b) {
    const uint32_t k6 = b % 6u;
    if (k6 >= 4u) return (int)b - 6;
    return k6 == 0u ? (int)b - 3 : (int)b - 1;
}

}  // namespace

// A team = the lanes that share one frame: one wave (kTeam == 1, several frames per workgroup) or the
// whole workgroup of kTeam waves (large frames: more lanes per frame, more frames resident per CU).
template <int kTeam>
__device__ __forceinline__ void team_sync() {
    if (kTeam == 1) wave_sync();
    else __syncthreads();
}

template <int kTeam>
__device__ __forceinline__ uint32_t team_excl_sum(uint32_t v, uint32_t tl, uint32_t& total, uint32_t* s_part) {
    uint32_t wave_total;
    const uint32_t x = wave_excl_sum(v, tl & 63u, wave_total);
    if (kTeam == 1) { total = wave_total; return x; }
    const uint32_t w = tl >> 6;
    __syncthreads();                                   // the previous sum's partials have been read
    if ((tl & 63u) == 0u) s_part[w] = wave_total;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int i = 0; i < kTeam; ++i) {
        const uint32_t part = s_part[i];
        before += (uint32_t)i < w ? part : 0u;
        all += part;
    }
    total = all;
    return x + before;
}

// dynamic LDS: [ code book 4 KB | per team: bits[cap_words] | block lengths[blocks_cap] | 8 words of partial sums | 16 words of scratch per lane ]
template <int kTeam>
__global__ __launch_bounds__(kWave* kMaxWaves) void amv_pack_wave_kernel(
    const int16_t* __restrict__ coef, uint32_t n, uint32_t blocks_per_frame, uint32_t blocks_cap,
    uint32_t cap_words, const HuffEncodeImage* __restrict__ img, uint8_t* __restrict__ tmp, uint32_t bound,
    uint32_t* __restrict__ lens, uint32_t* __restrict__ retry_list, uint32_t* __restrict__ retry_count) {
    constexpr uint32_t kLanes = kWave * kTeam;         // lanes per frame
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    uint32_t* book = reinterpret_cast<uint32_t*>(s_mem);
    const uint32_t team = kTeam == 1 ? threadIdx.x >> 6 : 0u;
    const uint32_t tl = kTeam == 1 ? threadIdx.x & 63u : threadIdx.x;   // lane inside the team
    for (uint32_t i = threadIdx.x; i < 1024u; i += blockDim.x) book[i] = (&img->code[0][0])[i];
    __syncthreads();
    const uint32_t frame = kTeam == 1 ? blockIdx.x * (blockDim.x >> 6) + team : blockIdx.x;
    if (frame >= n) return;                            // kTeam > 1: the whole workgroup leaves

    constexpr uint32_t kOwnWords = 16;                 // a lane's scratch: 512 bits for its ~4 blocks (41 bits a block on the bench stream)
    uint32_t* bits = reinterpret_cast<uint32_t*>(s_mem + 4096u + team * (cap_words + blocks_cap + 8u + kLanes * kOwnWords) * 4u);
    uint32_t* blen = bits + cap_words;
    uint32_t* s_part = blen + blocks_cap;
    uint32_t* own = s_part + 8u + tl;                  // word i of this lane's scratch at own[i * kLanes]: the bank follows the lane
    const int16_t* fcoef = coef + (uint64_t)frame * blocks_per_frame * 64u;
    for (uint32_t i = tl; i < cap_words; i += kLanes) bits[i] = 0u;

    // ---- 1. every block is coded once, into its lane's scratch (blocks lane, lane + team size, ... one behind the
    // other), and its length noted.  (The second walk over the coefficients that put the bits in place once the
    // positions were known cost as much as this one: 0.42 of the kernel's 0.92 ms.)
    Emitter mine{own, 0u, 0ull, 0, 0u, kLanes, kOwnWords};
    for (uint32_t b = tl; b < blocks_per_frame; b += kLanes) {
        uint32_t c[32];
        const uint4* src = reinterpret_cast<const uint4*>(fcoef + (uint64_t)b * 64u);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const uint4 q = src[i]; c[4 * i] = q.x; c[4 * i + 1] = q.y; c[4 * i + 2] = q.z; c[4 * i + 3] = q.w; }
        const int pb = pred_block(b);
        const int prev = pb >= 0 ? (int)fcoef[(uint64_t)pb * 64u] : 0;
        const uint32_t cls = (b % 6u) >= 4u ? 1u : 0u;
        const uint32_t before = mine.nbits;
        code_block<kOwn>(mine, c, prev, book + cls * 256u, book + (2u + cls) * 256u);
        blen[b] = mine.nbits - before;
    }
    if (mine.nacc && mine.wi < kOwnWords) own[mine.wi * kLanes] = (uint32_t)(mine.acc << (32 - mine.nacc));
    const bool spilled = mine.nbits > kOwnWords * 32u;   // more bits than the scratch holds: this lane walks its blocks again
    team_sync<kTeam>();

    // ---- 2. first bit of every block: each lane owns a run of consecutive blocks for the scan
    const uint32_t per = (blocks_per_frame + kLanes - 1) / kLanes;
    const uint32_t lo = min(blocks_per_frame, tl * per), hi = min(blocks_per_frame, lo + per);
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; ++b) sum += blen[b];
    uint32_t total_bits;
    uint32_t run_pos = team_excl_sum<kTeam>(sum, tl, total_bits, s_part);
    for (uint32_t b = lo; b < hi; ++b) { const uint32_t l = blen[b]; blen[b] = run_pos; run_pos += l; }
    team_sync<kTeam>();
    const uint32_t nbytes = (total_bits + 7u) >> 3;
    if (nbytes + 8u > cap_words * 4u) {   // does not fit the window: the one-lane-per-frame kernel takes it
        if (tl == 0) retry_list[atomicAdd(retry_count, 1u)] = frame;
        return;
    }

    // ---- 3. the bits go to their place: a copy out of the scratch, 32 bits at a time ...
    if (!spilled) {
        uint32_t from = 0;                             // bit cursor in the scratch
        for (uint32_t b = tl; b < blocks_per_frame; b += kLanes) {
            uint32_t pos = blen[b];
            uint32_t left = (b + 1u < blocks_per_frame ? blen[b + 1u] : total_bits) - pos;
            while (left) {
                const uint32_t take = min(left, 32u);
                // `take` bits of the scratch from bit `from` on, left-aligned
                const uint32_t w0 = own[(from >> 5) * kLanes], w1 = (from >> 5) + 1u < kOwnWords ? own[((from >> 5) + 1u) * kLanes] : 0u;
                const uint32_t piece = (uint32_t)((((uint64_t)w0 << 32) | w1) >> (32u - (from & 31u))) & (uint32_t)(0xffffffff00000000ull >> take);
                // into the string at bit `pos`
                atomicOr(&bits[pos >> 5], piece >> (pos & 31u));
                if ((pos & 31u) + take > 32u) atomicOr(&bits[(pos >> 5) + 1u], piece << (32u - (pos & 31u)));
                from += take; pos += take; left -= take;
            }
        }
    } else {   // ... or, for a lane whose blocks did not fit its scratch, a second walk
        for (uint32_t b = tl; b < blocks_per_frame; b += kLanes) {
            uint32_t c[32];
            const uint4* src = reinterpret_cast<const uint4*>(fcoef + (uint64_t)b * 64u);
#pragma unroll
            for (int i = 0; i < 8; ++i) { const uint4 q = src[i]; c[4 * i] = q.x; c[4 * i + 1] = q.y; c[4 * i + 2] = q.z; c[4 * i + 3] = q.w; }
            const int pb = pred_block(b);
            const int prev = pb >= 0 ? (int)fcoef[(uint64_t)pb * 64u] : 0;
            const uint32_t cls = (b % 6u) >= 4u ? 1u : 0u;
            const uint32_t pos = blen[b];
            Emitter e{bits, pos >> 5, 0ull, (int)(pos & 31u), 0u, 1u, 0u};   // the word's earlier bits belong to the previous block: zeros here, OR-ed in
            code_block<kShared>(e, c, prev, book + cls * 256u, book + (2u + cls) * 256u);
            if (e.nacc) atomicOr(&e.words[e.wi], (uint32_t)(e.acc << (32 - e.nacc)));
        }
    }
    team_sync<kTeam>();
    if (tl == 0 && (total_bits & 7u))   // ff_mjpeg_encode_stuffing: ones up to the byte boundary
        atomicOr(&bits[total_bits >> 5], ((1u << (8u - (total_bits & 7u))) - 1u) << (24u - (total_bits & 24u)));
    team_sync<kTeam>();

    // ---- 4. FF D8, the bytes with 00 after every FF, FF D9
    uint8_t* out = tmp + (uint64_t)frame * bound;
    uint32_t ff_before = 0;
    for (uint32_t w0 = 0; w0 * 4u < nbytes; w0 += kLanes) {
        const uint32_t wi = w0 + tl;
        const uint32_t word = wi * 4u < nbytes ? bits[wi] : 0u;
        uint32_t cnt = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j)
            cnt += (wi * 4u + j < nbytes && ((word >> (24u - 8u * j)) & 0xffu) == 0xffu) ? 1u : 0u;
        uint32_t tile;
        uint32_t o = 2u + wi * 4u + ff_before + team_excl_sum<kTeam>(cnt, tl, tile, s_part);
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            if (wi * 4u + j >= nbytes) break;
            const uint32_t byte = (word >> (24u - 8u * j)) & 0xffu;
            out[o++] = (uint8_t)byte;
            if (byte == 0xffu) out[o++] = 0;
        }
        ff_before += tile;
    }
    if (tl == 0) {
        out[0] = 0xff; out[1] = 0xd8;                          // SOI only, mjpegenc.c:201-204
        const uint32_t end = 2u + nbytes + ff_before;
        out[end] = 0xff; out[end + 1] = 0xd9;                  // EOI :354
        lens[frame] = end + 2u;
    }
}

bool launch_pack_wave(const int16_t* coef, uint32_t n, const FrameGeom& g, const HuffEncodeImage* d_img, uint8_t* tmp,
                      uint32_t bound, uint32_t* lens, uint32_t* retry_list, uint32_t* retry_count, hipStream_t s) {
    // bit-string window: ~2x the 0.2 bytes per pixel AMV runs at
    uint32_t cap_bytes = ((g.width * g.height * 2u / 5u) + 1023u) & ~1023u;
    if (cap_bytes < 2048u) cap_bytes = 2048u;
    const uint32_t blocks_cap = (g.blocks + 3u) & ~3u;
    // per team: the bit string, block lengths, partial sums, and per lane a scratch of 16 words (kOwnWords)
    const auto per_team_of = [&](uint32_t lanes) { return cap_bytes + blocks_cap * 4u + 32u + lanes * 64u; };
    static bool raised = false;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(amv_pack_wave_kernel<1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(amv_pack_wave_kernel<4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(amv_pack_wave_kernel<8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    if (g.blocks >= 1024u && 4096u + per_team_of(512u) <= 150u * 1024u) {
        const uint32_t lds_team = 4096u + per_team_of(512u);   // large frames: eight waves per frame, one frame per workgroup (measured: 4 -> 1.17 ms, 8 -> 0.89, 16 -> 1.40 per 8 000 frames of 320x240)
        hipLaunchKernelGGL(amv_pack_wave_kernel<8>, dim3(n), dim3(kWave * 8), lds_team, s, coef, n, g.blocks, blocks_cap,
                           cap_bytes / 4u, d_img, tmp, bound, lens, retry_list, retry_count);
        return true;
    }
    if (g.blocks >= 256u && 4096u + per_team_of(256u) <= 150u * 1024u) {
        const uint32_t lds_team = 4096u + per_team_of(256u);    // medium frames: four waves per frame (160x120: 2 -> 1.36 ms, 4 -> 1.15 per 40 000 frames)
        hipLaunchKernelGGL(amv_pack_wave_kernel<4>, dim3(n), dim3(kWave * 4), lds_team, s, coef, n, g.blocks, blocks_cap,
                           cap_bytes / 4u, d_img, tmp, bound, lens, retry_list, retry_count);
        return true;
    }
    const uint32_t per_team = per_team_of(64u);
    uint32_t waves = 4;
    while (waves > 1u && 4096u + waves * per_team > 79u * 1024u) waves >>= 1;   // aim at two workgroups per CU
    const uint32_t lds = 4096u + waves * per_team;
    if (lds > 150u * 1024u) return false;
    hipLaunchKernelGGL(amv_pack_wave_kernel<1>, dim3((n + waves - 1) / waves), dim3(kWave * waves), lds, s, coef, n,
                       g.blocks, blocks_cap, cap_bytes / 4u, d_img, tmp, bound, lens, retry_list, retry_count);
    return true;
}

}  // namespace amv
