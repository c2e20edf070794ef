// amv_adpcm.hip -- IMA ADPCM (AMV chunk layout) and the synthetic-source generators, gfx950.
//
// Reference: decode  C-AMVDecoder/amvlib/AMVDec.c:312-320 (chunk header) and AdpcmIma.c:170-242
//            (AdpcmImaExpandNibble / AdpcmImaDecodeFrame, mono: high nibble first);
//            encode  AMVmuxer/ffmpeg/libavcodec/adpcm.c:219-227 (adpcm_ima_compress_sample)
//            and :461-498 (AMV framing: le16 first sample, le16 step index, le32 sample count).
//
// The predictor loop is a serial chain inside a chunk; chunks are independent on decode (each
// carries predictor + step index), so one lane owns one chunk: it reads 16 chunk bytes at a time
// (one unaligned 16-byte load), walks the 32 nibbles with the index chain running ahead of the
// step-table lookups (LDS) and those ahead of the predictor chain, and stores the 32 samples as
// one 64-byte run.  On encode the reference carries step_index from chunk to chunk, and the end
// index of a chunk does depend on where it started (measured on the synthetic audio: 69 % of the
// chunks, 8.6 distinct encoder states still alive at the end of a chunk), so guessing does not
// work.  The chain is cut with the fact that step_index has only 89 values:
// amv_adpcm_map_kernel runs every chunk from all 89 starts (state only, no output, one lane per
// (chunk, start) pair), the amv_adpcm_chain_* kernels compose the 89-entry maps (256 chunks per
// workgroup through LDS, then the workgroup maps, then back down), and the real encode runs one
// lane per chunk from its now-known start.
#include <atomic>

#include "amv_kernels.h"

namespace amv {

namespace {

__device__ __forceinline__ int clip16(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ int clip_index(int v) { return min(max(v, 0), 88); }

// under-aligned wide accesses: gfx950 under HSA serves them in hardware, one instruction each
struct __attribute__((packed, aligned(1))) Bytes16 { uint32_t w[4]; };
struct __attribute__((packed, aligned(1))) Bytes8 { uint32_t w[2]; };
struct __attribute__((packed, aligned(2))) Pcm8 { uint32_t w[4]; };
struct __attribute__((packed, aligned(2))) Pcm32 { uint32_t w[16]; };

// the step table in LDS (indexed per lane on the critical path; a constant-memory table would be a
// dependent global load per sample)
__device__ __forceinline__ void load_steps(uint32_t* s_step) {
    for (uint32_t i = threadIdx.x; i < 89u; i += blockDim.x) s_step[i] = (uint32_t)kImaStep[i];
    __syncthreads();
}

// kImaIndexAdjust[nibble] = {-1,-1,-1,-1,2,4,6,8} on the magnitude bits
__device__ __forceinline__ int index_adjust(uint32_t mag3) {
    const int t = 2 * (int)mag3 - 6;
    return t > 0 ? t : -1;
}

// AdpcmImaExpandNibble, AdpcmIma.c:170-204 with shift 3
__device__ __forceinline__ int expand(int& predictor, int& index, uint32_t nibble, const uint32_t* s_step) {
    const int step = (int)s_step[index];
    index = clip_index(index + index_adjust(nibble & 7u));
    const int diff = ((2 * (int)(nibble & 7u) + 1) * step) >> 3;
    predictor = clip16((nibble & 8u) ? predictor - diff : predictor + diff);
    return predictor;
}

// adpcm_ima_compress_sample, adpcm.c:219-227.  min(7, |delta|*4/step) is taken bit by bit (the
// classic IMA quantiser ladder: identical quotient, no integer division on the chain).
__device__ __forceinline__ uint32_t compress(int& prev, int& index, int sample, const uint32_t* s_step) {
    const int delta = sample - prev;
    const uint32_t step = s_step[index];
    uint32_t d4 = (uint32_t)abs(delta) << 2;
    uint32_t q = 0;
    if (d4 >= step * 4u) { q = 4u; d4 -= step * 4u; }
    if (d4 >= step * 2u) { q |= 2u; d4 -= step * 2u; }
    if (d4 >= step) q |= 1u;
    const int mag = (int)((step * (2u * q + 1u)) >> 3);   // step * yamaha_difflookup[nibble] / 8
    prev = clip16(delta < 0 ? prev - mag : prev + mag);
    index = clip_index(index + index_adjust(q));
    return q + (delta < 0 ? 8u : 0u);
}

// a chunk's samples from (prev, index): 16 samples per 32-byte load pair, 8 bytes out.  m is even.
template <bool kWrite>
__device__ __forceinline__ void encode_run(const int16_t* __restrict__ x, uint32_t m, int& prev, int& index,
                                           uint8_t* __restrict__ d, const uint32_t* s_step) {
    uint32_t k = 0;
    for (; k + 16u <= m; k += 16u) {
        const Pcm8 a = *reinterpret_cast<const Pcm8*>(x + k);
        const Pcm8 b = *reinterpret_cast<const Pcm8*>(x + k + 8u);
        Bytes8 o;
        o.w[0] = 0u;
        o.w[1] = 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t word = j < 8 ? a.w[j >> 1] : b.w[(j - 8) >> 1];
            const int sample = (j & 1) ? ((int)word >> 16) : (int)(int16_t)(word & 0xffffu);
            const uint32_t nib = compress(prev, index, sample, s_step);
            o.w[j >> 3] |= nib << (8 * ((j >> 1) & 3) + ((j & 1) ? 0 : 4));   // :489-493 high nibble = earlier sample
        }
        if (kWrite) *reinterpret_cast<Bytes8*>(d + (k >> 1)) = o;
    }
    for (; k < m; k += 2u) {
        const uint32_t hi = compress(prev, index, x[k], s_step);
        const uint32_t lo = compress(prev, index, x[k + 1u], s_step);
        if (kWrite) d[k >> 1] = (uint8_t)((hi << 4) | lo);
    }
}

constexpr uint32_t kChainBlock = 256;   // chunks whose maps one workgroup composes through LDS (24 KB)

}  // namespace

__global__ __launch_bounds__(64) void amv_adpcm_decode_kernel(
    const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offs,
    const uint32_t* __restrict__ lens, uint32_t n, int16_t* __restrict__ pcm,
    const uint64_t* __restrict__ pcm_offs, int32_t* __restrict__ final_state) {
    __shared__ uint32_t s_step[96];
    load_steps(s_step);
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    const uint64_t off = offs[i];
    const uint32_t len = lens[i];
    if (len <= 8 || off > blob_bytes || len > blob_bytes - off) return;
    const uint8_t* c = blob + off;
    int predictor = (int16_t)(c[0] | (c[1] << 8));   // AMVDec.c:312
    int index = clip_index(c[2]);                    // AMVDec.c:313 (the reference indexes its table unchecked)
    int16_t* o = pcm + pcm_offs[i];
    const uint8_t* p = c + 8;
    const uint32_t nb = len - 8u;
    uint32_t k = 0;
    for (; k + 16u <= nb; k += 16u) {                // AdpcmIma.c:225-237, 32 samples per trip
        const Bytes16 in = *reinterpret_cast<const Bytes16*>(p + k);
        Pcm32 out;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t nib[8];
            int st[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {            // the index chain needs only the nibbles
                nib[j] = (in.w[w] >> (8 * (j >> 1) + ((j & 1) ? 0 : 4))) & 15u;
                st[j] = index;
                index = clip_index(index + index_adjust(nib[j] & 7u));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) st[j] = (int)s_step[st[j]];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int diff = ((2 * (int)(nib[j] & 7u) + 1) * st[j]) >> 3;
                predictor = clip16((nib[j] & 8u) ? predictor - diff : predictor + diff);
                if (j & 1) out.w[4 * w + (j >> 1)] |= (uint32_t)predictor << 16;
                else out.w[4 * w + (j >> 1)] = (uint32_t)predictor & 0xffffu;
            }
        }
        *reinterpret_cast<Pcm32*>(o + 2u * k) = out;
    }
    for (; k < nb; ++k) {
        const uint32_t byte = p[k];
        o[2u * k] = (int16_t)expand(predictor, index, byte >> 4, s_step);
        o[2u * k + 1u] = (int16_t)expand(predictor, index, byte & 15u, s_step);
    }
    if (final_state) { final_state[2 * i] = predictor; final_state[2 * i + 1] = index; }
}

// amvlib's own encoder, AdpcmIma.c:43-160: IMA-WAV block layout (4-byte header, low nibble =
// earlier sample), and a compressor that differs from FFmpeg's: the quotient passes through an
// unsigned char before it is limited to 7, and the predicted delta uses the UPDATED step.
// Nothing in the reference calls it; it is here because AdpcmImaEncodeFrame is exported.
__global__ void amv_adpcm_wav_encode_kernel(const int16_t* __restrict__ x, int groups,
                                            int32_t* __restrict__ state, uint8_t* __restrict__ frame) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int prev = x[0];                               // :106
    int index = state[1];
    frame[0] = (uint8_t)(prev & 0xff);             // :108-111
    frame[1] = (uint8_t)((prev >> 8) & 0xff);
    frame[2] = (uint8_t)index;
    frame[3] = 0;
    auto comp = [&](int sample) -> uint32_t {      // AdpcmImaCompressSample :43-89
        int delta = sample - prev;
        const uint32_t sign = delta < 0 ? 1u : 0u;
        if (delta < 0) delta = -delta;
        uint32_t nib = (uint32_t)((delta << 2) / kImaStep[clip_index(index)]) & 0xffu;   // unsigned char
        if (nib > 7u) nib = 7u;
        index = clip_index(index + kImaIndexAdjust[nib]);
        const int pd = (kImaStep[index] * (int)nib) / 4 + kImaStep[index] / 8;
        prev = clip16(sign ? prev - pd : prev + pd);
        return nib + (sign << 3);
    };
    const int16_t* s = x + 1;                      // :112
    uint8_t* d = frame + 4;
    for (int g = 0; g < groups; ++g, s += 8)       // :125-155, mono
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = comp(s[2 * k]) & 0x0fu;
            const uint32_t hi = comp(s[2 * k + 1]);
            *d++ = (uint8_t)(lo | ((hi << 4) & 0xf0u));
        }
    state[0] = prev;
    state[1] = index;
}

// state-only run of chunk i from start index s: where does step_index end up?  One lane per
// (chunk, start) pair, pairs packed densely into waves.
__global__ __launch_bounds__(64) void amv_adpcm_map_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs,
    const uint32_t* __restrict__ nsamp, uint32_t n, uint8_t* __restrict__ map /* [n][96] */) {
    __shared__ uint32_t s_step[96];
    load_steps(s_step);
    const uint64_t pair = (uint64_t)blockIdx.x * 64u + threadIdx.x;
    const uint32_t i = (uint32_t)(pair / 89u), s = (uint32_t)(pair % 89u);
    if (i >= n) return;
    const int16_t* x = pcm + pcm_offs[i];
    const uint32_t m = nsamp[i] & ~1u;
    int prev = m ? x[0] : 0, index = (int)s;
    encode_run<false>(x, m, prev, index, nullptr, s_step);
    map[(uint64_t)i * 96u + s] = (uint8_t)index;
}

// composition of the maps of kChainBlock consecutive chunks: bmap[b][s] = where start s ends up
__global__ __launch_bounds__(128) void amv_adpcm_chain_block_kernel(const uint8_t* __restrict__ map, uint32_t n,
                                                                    uint8_t* __restrict__ bmap) {
    __shared__ uint32_t s_map[kChainBlock * 24u];
    const uint32_t c0 = blockIdx.x * kChainBlock, cnt = min(kChainBlock, n - c0);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(map + (uint64_t)c0 * 96u);
    for (uint32_t i = threadIdx.x; i < cnt * 24u; i += 128u) s_map[i] = src[i];
    __syncthreads();
    if (threadIdx.x >= 89u) return;
    const uint8_t* m8 = reinterpret_cast<const uint8_t*>(s_map);
    uint32_t v = threadIdx.x;
    for (uint32_t c = 0; c < cnt; ++c) v = m8[c * 96u + v];
    bmap[(uint64_t)blockIdx.x * 96u + threadIdx.x] = (uint8_t)v;
}

// the serial walk over the workgroup maps (n / 256 steps through LDS): start index of every block
__global__ __launch_bounds__(128) void amv_adpcm_chain_top_kernel(const uint8_t* __restrict__ bmap, uint32_t nb,
                                                                  int32_t* __restrict__ bstart) {
    __shared__ uint32_t s_map[kChainBlock * 24u];
    uint32_t v = 0;   // the encoder context starts zeroed
    for (uint32_t t0 = 0; t0 < nb; t0 += kChainBlock) {
        const uint32_t cnt = min(kChainBlock, nb - t0);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(bmap + (uint64_t)t0 * 96u);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cnt * 24u; i += 128u) s_map[i] = src[i];
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint8_t* m8 = reinterpret_cast<const uint8_t*>(s_map);
            for (uint32_t c = 0; c < cnt; ++c) {
                bstart[t0 + c] = (int32_t)v;
                v = m8[c * 96u + v];
            }
        }
    }
}

// back down: start index of every chunk of a block from the block's start
__global__ __launch_bounds__(128) void amv_adpcm_chain_fill_kernel(const uint8_t* __restrict__ map, uint32_t n,
                                                                   const int32_t* __restrict__ bstart,
                                                                   int32_t* __restrict__ start) {
    __shared__ uint32_t s_map[kChainBlock * 24u];
    __shared__ int32_t s_start[kChainBlock];
    const uint32_t c0 = blockIdx.x * kChainBlock, cnt = min(kChainBlock, n - c0);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(map + (uint64_t)c0 * 96u);
    for (uint32_t i = threadIdx.x; i < cnt * 24u; i += 128u) s_map[i] = src[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint8_t* m8 = reinterpret_cast<const uint8_t*>(s_map);
        uint32_t v = (uint32_t)bstart[blockIdx.x];
        for (uint32_t c = 0; c < cnt; ++c) {
            s_start[c] = (int32_t)v;
            v = m8[c * 96u + v];
        }
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < cnt; c += 128u) start[c0 + c] = s_start[c];
}

__global__ __launch_bounds__(64) void amv_adpcm_encode_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs,
    const uint32_t* __restrict__ nsamp, uint32_t n, const int32_t* __restrict__ step_in,
    uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs) {
    __shared__ uint32_t s_step[96];
    load_steps(s_step);
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    const int16_t* x = pcm + pcm_offs[i];
    const uint32_t pairs = nsamp[i] >> 1;
    uint8_t* d = blob + offs[i];
    int prev = pairs ? x[0] : 0;                    // adpcm.c:464
    int index = clip_index(step_in[i]);
    d[0] = (uint8_t)(prev & 0xff);                  // :465 le16 first sample
    d[1] = (uint8_t)((prev >> 8) & 0xff);
    d[2] = (uint8_t)index;                          // :466 le16 step index
    d[3] = 0;
    const uint32_t cnt = pairs << 1;                // :479 le32 sample count
    d[4] = (uint8_t)cnt; d[5] = (uint8_t)(cnt >> 8); d[6] = (uint8_t)(cnt >> 16); d[7] = (uint8_t)(cnt >> 24);
    encode_run<true>(x, cnt, prev, index, d + 8, s_step);
}

// The reference's trellis search (adpcm_compress_trellis, adpcm.c:287-443, IMA branch; `-trellis N`): a beam of the
// 2^N best decoder states (sorted by squared error, at most one per decoded sample value), three candidate nibbles
// around the plain quantiser's choice for the better half of the beam and one for the rest (:333,373-385), the best
// path frozen into the output every 128 samples (:405-417).  One lane per chunk; the beam lives in LDS
// ([field][buffer][slot][lane]: a lane's accesses never meet another lane's bank), the back-pointers
// (nibble | previous path << 4, 16 bits) in a workspace laid out [64 chunks][path][lane].
// Chunks are independent: the step index comes in per chunk and goes out per chunk.
__global__ __launch_bounds__(64) void amv_adpcm_trellis_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs, const uint32_t* __restrict__ nsamp, uint32_t n,
    const int32_t* __restrict__ step_in, uint32_t trellis, uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs,
    int32_t* __restrict__ step_out, uint16_t* __restrict__ paths) {
    extern __shared__ uint32_t s_trellis[];
    __shared__ uint32_t s_step[96];
    load_steps(s_step);
    const uint32_t F = 1u << trellis, lane = threadIdx.x;
    const uint32_t i_chunk = blockIdx.x * 64u + lane;
    const bool live = i_chunk < n;
    // node fields: [field 0..3][buffer 0..1][slot][lane]; order of the two frontiers: [buffer][rank][lane]
    uint32_t* const f_ssd = s_trellis + lane;
    uint32_t* const f_smp = f_ssd + 2u * F * 64u;
    uint32_t* const f_stp = f_smp + 2u * F * 64u;
    uint32_t* const f_pth = f_stp + 2u * F * 64u;
    uint32_t* const f_ord = f_pth + 2u * F * 64u;
    auto at = [&](uint32_t* field, uint32_t buf, uint32_t slot) -> uint32_t& { return field[(buf * F + slot) * 64u]; };
    constexpr uint32_t kNone = 0xffffffffu;
    uint16_t* const my_paths = paths + (uint64_t)blockIdx.x * (F * 128u) * 64u + lane;   // entry e at my_paths[e * 64]

    const int16_t* x = live ? pcm + pcm_offs[i_chunk] : pcm;
    const uint32_t cnt = live ? (nsamp[i_chunk] & ~1u) : 0u;
    uint8_t* d = live ? blob + offs[i_chunk] : blob;
    const int first = cnt ? x[0] : 0;
    const int index0 = live ? clip_index(step_in[i_chunk]) : 0;
    if (live) {
        d[0] = (uint8_t)(first & 0xff); d[1] = (uint8_t)((first >> 8) & 0xff);       // adpcm.c:465-466,479
        d[2] = (uint8_t)index0; d[3] = 0;
        d[4] = (uint8_t)cnt; d[5] = (uint8_t)(cnt >> 8); d[6] = (uint8_t)(cnt >> 16); d[7] = (uint8_t)(cnt >> 24);
    }
    // nodes[0] = {ssd 0, path 0, step, sample1 = the chunk's first sample} in buffer 1 (:309-316)
    at(f_ssd, 1, 0) = 0u; at(f_smp, 1, 0) = (uint32_t)first; at(f_stp, 1, 0) = (uint32_t)index0; at(f_pth, 1, 0) = 0u;
    for (uint32_t k = 0; k < F; ++k) { at(f_ord, 0, k) = k ? kNone : 0u; at(f_ord, 1, k) = kNone; }
    uint32_t cur = 0;          // which order array holds the current frontier (its nodes live in buffer (i & 1) ^ 1)
    uint32_t pathn = 0;
    int froze = -1;
    auto put_nibble = [&](uint32_t k, uint32_t nib) {   // sample k's nibble: high half of its byte first (:485-486)
        uint8_t* b = d + 8u + (k >> 1);
        *b = (k & 1u) ? (uint8_t)((*b & 0xf0u) | nib) : (uint8_t)((*b & 0x0fu) | (nib << 4));
    };
    for (uint32_t i = 0; i < cnt; ++i) {
        const uint32_t nb = i & 1u, ob = nb ^ 1u, nxt = cur ^ 1u;
        const int sample = x[i];
        uint32_t made = 0, nn = 0;     // nodes allocated in buffer nb; entries of the next frontier
        for (uint32_t k = 0; k < F; ++k) at(f_ord, nxt, k) = kNone;
        for (uint32_t j = 0; j < F; ++j) {
            const uint32_t src = at(f_ord, cur, j);
            if (src == kNone) break;
            const int range = j < F / 2u ? 1 : 0;                                   // :333
            const int step = (int)at(f_stp, ob, src), st = (int)s_step[step];
            const int predictor = (int)at(f_smp, ob, src);
            const uint32_t base_ssd = at(f_ssd, ob, src), src_path = at(f_pth, ob, src);
            const int div = (sample - predictor) * 4 / st;                         // :376
            int nmin = min(max(div - range, -7), 6), nmax = min(max(div + range, -6), 7);
            if (nmin <= 0) --nmin;                                                   // distinguish -0 from +0
            if (nmax < 0) --nmax;
            for (int nidx = nmin; nidx <= nmax; ++nidx) {
                const uint32_t nibble = (uint32_t)(nidx < 0 ? 7 - nidx : nidx);
                const int look = (nibble & 8u) ? -(int)(2u * (nibble & 7u) + 1u) : (int)(2u * (nibble & 7u) + 1u);
                const int dec = clip16(predictor + (st * look) / 8);
                const int diff = sample - dec;
                const uint32_t ssd = base_ssd + (uint32_t)(diff * diff);
                if (nn == F && ssd >= at(f_ssd, nb, at(f_ord, nxt, F - 1u))) continue;   // :342
                bool dup = false;                                                    // one state per decoded value, :347-352
                for (uint32_t k = 0; k < nn; ++k) dup = dup || (int)at(f_smp, nb, at(f_ord, nxt, k)) == dec;
                if (dup) continue;
                uint32_t k = 0;
                while (k < nn && ssd >= at(f_ssd, nb, at(f_ord, nxt, k))) ++k;       // first rank it beats (:353-354)
                uint32_t u;
                if (nn == F) {
                    u = at(f_ord, nxt, F - 1u);                                      // the worst one makes room, its path id stays
                } else {
                    u = made++;
                    at(f_pth, nb, u) = pathn++;
                    ++nn;
                }
                at(f_ssd, nb, u) = ssd;
                at(f_stp, nb, u) = (uint32_t)clip_index(step + kImaIndexAdjust[nibble]);
                at(f_smp, nb, u) = (uint32_t)dec;
                my_paths[(uint64_t)at(f_pth, nb, u) * 64u] = (uint16_t)(nibble | (src_path << 4));
                for (uint32_t m = nn - 1u; m > k; --m) at(f_ord, nxt, m) = at(f_ord, nxt, m - 1u);   // memmove, :365
                at(f_ord, nxt, k) = u;
            }
        }
        cur = nxt;
        const uint32_t best = at(f_ord, cur, 0);
        if (at(f_ssd, nb, best) > (1u << 28)) {                                     // :398-402
            const uint32_t off = at(f_ssd, nb, best);
            for (uint32_t j = 1; j < F; ++j) {
                const uint32_t q = at(f_ord, cur, j);
                if (q == kNone) break;
                at(f_ssd, nb, q) -= off;
            }
            at(f_ssd, nb, best) = 0u;
        }
        if ((int)i == froze + 128) {                                                // :405-417
            uint32_t p = at(f_pth, nb, best);
            for (int k = (int)i; k > froze; --k) {
                const uint32_t e = my_paths[(uint64_t)p * 64u];
                put_nibble((uint32_t)k, e & 15u);
                p = e >> 4;
            }
            froze = (int)i;
            pathn = 0;
            for (uint32_t j = 1; j < F; ++j) at(f_ord, cur, j) = kNone;
        }
    }
    if (cnt) {
        const uint32_t nb = (cnt - 1u) & 1u, best = at(f_ord, cur, 0);
        uint32_t p = at(f_pth, nb, best);
        for (int k = (int)cnt - 1; k > froze; --k) {
            const uint32_t e = my_paths[(uint64_t)p * 64u];
            put_nibble((uint32_t)k, e & 15u);
            p = e >> 4;
        }
        if (step_out) step_out[i_chunk] = (int32_t)at(f_stp, nb, best);              // :429
    } else if (live && step_out) {
        step_out[i_chunk] = index0;
    }
}

// workspace of launch_adpcm_trellis: bytes for n chunks
uint64_t adpcm_trellis_workspace(uint32_t n, uint32_t trellis) {
    return (uint64_t)((n + 63u) / 64u) * 64u * ((1u << trellis) * 128u) * sizeof(uint16_t);
}

bool launch_adpcm_trellis(const int16_t* pcm, const uint64_t* pcm_offs, const uint32_t* nsamp, uint32_t n, const int32_t* step_in,
                          uint32_t trellis, uint8_t* blob, const uint64_t* offs, int32_t* step_out, uint16_t* paths, hipStream_t s) {
    if (n == 0) return true;
    const uint32_t lds = 5u * 2u * (1u << trellis) * 64u * 4u;
    static std::atomic<uint64_t> raised{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!(raised.load() & (1ull << (dev & 63)))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(amv_adpcm_trellis_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                5 * 2 * 32 * 64 * 4) != hipSuccess)
            return false;
        raised.fetch_or(1ull << (dev & 63));
    }
    hipLaunchKernelGGL(amv_adpcm_trellis_kernel, dim3((n + 63) / 64), dim3(64), lds, s, pcm, pcm_offs, nsamp, n, step_in, trellis, blob,
                       offs, step_out, paths);
    return true;
}

void launch_adpcm_decode(const uint8_t* blob, uint64_t blob_bytes, const uint64_t* offs,
                         const uint32_t* lens, uint32_t n, int16_t* pcm, const uint64_t* pcm_offs,
                         int32_t* final_state, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(amv_adpcm_decode_kernel, dim3((n + 63) / 64), dim3(64), 0, s, blob, blob_bytes,
                       offs, lens, n, pcm, pcm_offs, final_state);
}

void launch_adpcm_wav_encode(const int16_t* samples, int groups, int32_t* state, uint8_t* frame, hipStream_t s) {
    hipLaunchKernelGGL(amv_adpcm_wav_encode_kernel, dim3(1), dim3(64), 0, s, samples, groups, state, frame);
}

void launch_adpcm_map(const int16_t* pcm, const uint64_t* pcm_offs, const uint32_t* nsamp, uint32_t n,
                      uint8_t* map, int32_t* start, hipStream_t s) {
    // map: (n + nb) * 96 bytes, start: n + nb words, nb = adpcm_chain_blocks(n)
    if (n == 0) return;
    const uint32_t nb = adpcm_chain_blocks(n);
    uint8_t* bmap = map + (uint64_t)n * 96u;
    int32_t* bstart = start + n;
    const uint64_t pairs = (uint64_t)n * 89u;
    hipLaunchKernelGGL(amv_adpcm_map_kernel, dim3((uint32_t)((pairs + 63u) / 64u)), dim3(64), 0, s, pcm, pcm_offs, nsamp,
                       n, map);
    hipLaunchKernelGGL(amv_adpcm_chain_block_kernel, dim3(nb), dim3(128), 0, s, map, n, bmap);
    hipLaunchKernelGGL(amv_adpcm_chain_top_kernel, dim3(1), dim3(128), 0, s, bmap, nb, bstart);
    hipLaunchKernelGGL(amv_adpcm_chain_fill_kernel, dim3(nb), dim3(128), 0, s, map, n, bstart, start);
}

uint32_t adpcm_chain_blocks(uint32_t n) { return (n + kChainBlock - 1u) / kChainBlock; }

void launch_adpcm_encode(const int16_t* pcm, const uint64_t* pcm_offs, const uint32_t* nsamp,
                         uint32_t n, const int32_t* step_in, uint8_t* blob, const uint64_t* offs,
                         hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(amv_adpcm_encode_kernel, dim3((n + 63) / 64), dim3(64), 0, s, pcm, pcm_offs,
                       nsamp, n, step_in, blob, offs);
}

// ============================================================================================
// synthetic sources: integer only, byte-identical to the CPU generator
// the parity tests use
// ============================================================================================

namespace {

__device__ __forceinline__ int isin(uint32_t a) {
    a &= 255u;
    const uint32_t q = a & 63u;
    switch (a >> 6) {
        case 0: return kSinQ14[q];
        case 1: return kSinQ14[64 - q];
        case 2: return -kSinQ14[q];
        default: return -kSinQ14[64 - q];
    }
}
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ int tri(uint32_t v) { v &= 511u; return (int)(v < 256u ? v : 511u - v); }
__device__ __forceinline__ uint8_t clip8(int v) { return (uint8_t)min(max(v, 0), 255); }

}  // namespace

__global__ __launch_bounds__(256) void amv_synth_frames_kernel(uint32_t seed, uint32_t first, uint32_t n,
                                                               uint32_t w, uint32_t h, uint8_t* __restrict__ rgb) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t per = (uint64_t)w * h;
    if (idx >= per * n) return;
    const uint32_t t = first + (uint32_t)(idx / per);
    const uint32_t pix = (uint32_t)(idx % per), y = pix / w, x = pix % w;
    const int cx = (int)w / 2, cy = (int)h / 2, rad = (int)h * 3 / 8;
    const int cell = (int)w / 10 > 0 ? (int)w / 10 : 1;
    const int sn = isin(t * 2u), cs = isin(t * 2u + 64u);
    const uint32_t gx = 512u * 256u / w, gy = 512u * 256u / h;
    int r = 48 + ((tri(((x * gx) >> 8) + t * 3u) * 5) >> 3);
    int g = 48 + ((tri(((y * gy) >> 8) + t * 2u) * 5) >> 3);
    int b = 48 + ((tri((((x * gx) + (y * gy)) >> 9) + t * 5u) * 5) >> 3);
    const int dx = (int)x - cx, dy = (int)y - cy;
    if (dx * dx + dy * dy < rad * rad) {
        const int u = (dx * cs + dy * sn) >> 14, v = (dy * cs - dx * sn) >> 14;
        const int chk = (((u + 4096) / cell) ^ ((v + 4096) / cell)) & 1;
        r = chk ? 230 - (r >> 3) : 25 + (r >> 3);
        g = chk ? 230 - (g >> 3) : 25 + (g >> 3);
        b = chk ? 230 - (b >> 3) : 25 + (b >> 3);
    }
    const uint32_t nz = mix32(seed ^ mix32(t * 0x9e3779b9u + y * 65537u + x));
    uint8_t* p = rgb + idx * 3u;
    p[0] = clip8(r + (int)(nz % 25u) - 12);
    p[1] = clip8(g + (int)((nz >> 8) % 25u) - 12);
    p[2] = clip8(b + (int)((nz >> 16) % 25u) - 12);
}

__global__ __launch_bounds__(256) void amv_synth_audio_kernel(uint32_t seed, uint64_t first, uint64_t n,
                                                              int16_t* __restrict__ pcm) {
    const uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (k >= n) return;
    const uint64_t i = first + k;
    int v = (6000 * isin((uint32_t)((i * 1301u) >> 8)) + 3000 * isin((uint32_t)((i * 3907u) >> 8)) +
             1500 * isin((uint32_t)((i * 9973u) >> 8))) >> 14;
    const uint32_t r = mix32(seed ^ mix32((uint32_t)i * 0x85ebca6bu + (uint32_t)(i >> 32)));
    v += (int)(r % 401u) - 200;
    pcm[k] = (int16_t)clip16(v);
}

void launch_synth_frames(uint32_t seed, uint32_t first, uint32_t n, uint32_t w, uint32_t h,
                         uint8_t* rgb, hipStream_t s) {
    const uint64_t total = (uint64_t)w * h * n;
    if (total == 0) return;
    hipLaunchKernelGGL(amv_synth_frames_kernel, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s,
                       seed, first, n, w, h, rgb);
}

void launch_synth_audio(uint32_t seed, uint64_t first, uint64_t n, int16_t* pcm, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(amv_synth_audio_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, seed,
                       first, n, pcm);
}

}  // namespace amv
