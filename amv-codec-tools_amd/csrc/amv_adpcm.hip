// amv_adpcm.hip -- IMA ADPCM (AMV chunk layout), gfx950.
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
// chunks), so the chunks form one chain.  It is cut the way the entropy stage cuts a frame's bit
// stream: every chunk is coded from a GUESSED start (the index a short state-only run over the
// tail of the chunk before arrives at: right for 59 % of the synthetic chunks) and notes the index
// it ends with; a chunk whose predecessor ended elsewhere than it assumed is coded again, and so
// on while ends keep changing (amv_adpcm_guess_kernel, amv_adpcm_sweep_kernel: the lists shrink
// by 3.7x per sweep on the synthetic audio, 1.7 encodes of work in all instead of 89).  Chunk 0
// starts from the true index, so after k sweeps the first k chunks are final whatever the data:
// a stream whose chain does not settle within the sweeps given (amv_adpcm_settle_kernel raises a
// flag on the device) takes the exhaustive route instead, which needs no guess because
// step_index has only 89 values: amv_adpcm_map_kernel runs every chunk from all 89 starts (state
// only, one lane per (chunk, start) pair), the amv_adpcm_chain_* kernels compose the 89-entry
// maps (256 chunks per workgroup through LDS, then the workgroup maps, then back down), and the
// encode runs one lane per chunk from its now-known start.  Those kernels are always queued
// (the stream is never waited for) and leave at once when the flag is down.
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

// adpcm_ima_compress_sample, adpcm.c:219-227, shaped for a lane that is alone on its SIMD (the sweeps of the index chain
// are as long as one chunk's serial chain, so what counts is the length of the dependency chain per sample, not the
// instruction count):
//   * samples and predictor are kept biased by 32768, so |delta| is one v_sad_u32 and the clip is a med3 to 0..65535;
//   * min(7, |delta| * 4 / step) is one float multiply: trunc(float(|delta|) * r) with r = 4 / step nudged up by 2^-20.
//     Exact for every |delta| < 65536 and every step of the table: the nudge outweighs the two roundings (2^-23 each) so
//     exact multiples do not fall short, and 7 * (2^-20 + 2^-22) < 1 / 32767, the closest a quotient below 8 comes to the
//     next integer from underneath (tests/test_abi_and_host.py checks all 89 x 65536 cases against the integer division
//     with the table amvhip_adpcm_quotient_table hands out);
//   * the table look-up for the NEXT step leaves the chain: the index moves by -1, +2, +4, +6 or +8, so the five steps it
//     can arrive at and their reciprocals (a 32-byte row per index: AdpcmRow) are requested as soon as the index is
//     known, a sample ahead of their use, and the quotient picks among them (one byte permute + selects).
struct AdpcmRow {
    uint32_t s12, s34, s0;   // the step after a move of +2 | +4 << 16, of +6 | +8 << 16, of -1
    float r0, r1, r2, r3, r4;  // reciprocals (4 / step, nudged) in the order -1, +2, +4, +6, +8
};
static_assert(sizeof(AdpcmRow) == 32, "two 16-byte LDS reads per row");

struct AdpcmTables {
    AdpcmRow row[89];
    float rcp[89];
};

constexpr float quotient_factor(int step) { return (float)((4.0 / step) * (1.0 + 1.0 / 1048576.0)); }
constexpr int clip_index_c(int v) { return v < 0 ? 0 : (v > 88 ? 88 : v); }
constexpr AdpcmTables make_adpcm_tables() {
    AdpcmTables t{};
    for (int i = 0; i < 89; ++i) {
        const int to[5] = {clip_index_c(i - 1), clip_index_c(i + 2), clip_index_c(i + 4), clip_index_c(i + 6), clip_index_c(i + 8)};
        t.row[i].s0 = (uint32_t)kImaStep[to[0]];
        t.row[i].s12 = (uint32_t)kImaStep[to[1]] | ((uint32_t)kImaStep[to[2]] << 16);
        t.row[i].s34 = (uint32_t)kImaStep[to[3]] | ((uint32_t)kImaStep[to[4]] << 16);
        t.row[i].r0 = quotient_factor(kImaStep[to[0]]);
        t.row[i].r1 = quotient_factor(kImaStep[to[1]]);
        t.row[i].r2 = quotient_factor(kImaStep[to[2]]);
        t.row[i].r3 = quotient_factor(kImaStep[to[3]]);
        t.row[i].r4 = quotient_factor(kImaStep[to[4]]);
        t.rcp[i] = quotient_factor(kImaStep[i]);
    }
    return t;
}
static constexpr AdpcmTables kAdpcmHost = make_adpcm_tables();
__device__ const AdpcmTables kAdpcmTables = make_adpcm_tables();

// LDS image of a workgroup that encodes: the rows, the reciprocals and the steps themselves.  Lanes look up rows of
// their own, so a plain [89][32 bytes] table would put lanes whose indices differ by 8 on the same banks (a row is
// an eighth of the 256-byte bank row).  Every half row exists four times instead, and lane l reads copy l & 3: lanes
// that ds_read_b128 serves together and that sit on different step indices meet on a bank only when their copies are
// the same AND their indices differ by a multiple of four.  (Sixteen copies -- no conflict whatever the indices --
// measured the same to the percent, alone and beside the video kernels: the walk waits for the look-up's latency, not
// for its banks; 45 KB of LDS per workgroup kept a CU from taking the entropy kernel's workgroup next to this one.)
constexpr uint32_t kEncodeBlock = 256;
constexpr uint32_t kRowCopies = 4;       // 11 KB of tables per workgroup
struct EncodeLds {
    uint4 row[2][89][kRowCopies];
    float rcp[96];
    uint32_t step[96];
};

__device__ __forceinline__ void load_encode_tables(EncodeLds& l) {
    const uint4* src = reinterpret_cast<const uint4*>(kAdpcmTables.row);
    for (uint32_t i = threadIdx.x; i < 2u * 89u * kRowCopies; i += blockDim.x) {
        const uint32_t half = i / (89u * kRowCopies), row = (i / kRowCopies) % 89u;
        (&l.row[0][0][0])[i] = src[2u * row + half];
    }
    for (uint32_t i = threadIdx.x; i < 89u; i += blockDim.x) {
        l.rcp[i] = kAdpcmTables.rcp[i];
        l.step[i] = (uint32_t)kImaStep[i];
    }
    __syncthreads();
}

struct EncodeState {
    uint32_t prev;   // predictor + 32768
    uint32_t index;
    uint32_t step;   // kImaStep[index]
    float rcp;       // its quotient factor
};

__device__ __forceinline__ EncodeState encode_state(int prev, int index, const EncodeLds& l) {
    return EncodeState{(uint32_t)(prev + 32768), (uint32_t)index, l.step[index], l.rcp[index]};
}

// one sample (biased by 32768) -> its nibble
__device__ __forceinline__ uint32_t compress(EncodeState& s, uint32_t sample, const EncodeLds& l) {
    const uint32_t slot = threadIdx.x & (kRowCopies - 1u);
    const uint4 ra = l.row[0][s.index][slot], rb = l.row[1][s.index][slot];
    uint32_t ad;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(ad) : "v"(sample), "v"(s.prev));
    const uint32_t sign = sample < s.prev ? 1u : 0u;
    const uint32_t q = min((uint32_t)((float)ad * s.rcp), 7u);
    const uint32_t mag = __umul24(s.step, 2u * q + 1u) >> 3;                      // step * yamaha_difflookup[nibble] / 8
    const int moved = (int)(s.prev + (mag ^ (0u - sign)) + sign);                 // prev -+ mag
    s.prev = (uint32_t)min(max(moved, 0), 65535);
    const uint32_t up = __builtin_amdgcn_ubfe(0x97530000u, q * 4u, 4u);           // kImaIndexAdjust[q] + 1
    s.index = (uint32_t)min(max((int)(s.index + up) - 1, 0), 88);
    const bool low = q < 4u, odd = (q & 1u) != 0u;
    const uint32_t hop = __builtin_amdgcn_perm(ra.y, ra.x, __umul24(q, 0x0202u) + 0x0c0bf8f8u);   // halves 0..3 of {s12, s34} for q = 4..7
    s.step = low ? ra.z : hop;
    const float r12 = odd ? __uint_as_float(rb.y) : __uint_as_float(rb.x);
    const float r34 = odd ? __uint_as_float(rb.w) : __uint_as_float(rb.z);
    s.rcp = low ? __uint_as_float(ra.w) : (q < 6u ? r12 : r34);
    return q | (sign << 3);
}

// sixteen samples (eight words, biased here) -> eight bytes
__device__ __forceinline__ Bytes8 compress16(EncodeState& s, const uint32_t* w, const EncodeLds& l) {
    Bytes8 o;
    o.w[0] = 0u;
    o.w[1] = 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t word = w[j >> 1] ^ 0x80008000u;
        const uint32_t nib = compress(s, (j & 1) ? word >> 16 : word & 0xffffu, l);
        o.w[j >> 3] |= nib << (8 * ((j >> 1) & 3) + ((j & 1) ? 0 : 4));   // :489-493 high nibble = earlier sample
    }
    return o;
}

// A chunk's samples from state s (m is even).  One lane owns the chunk, so what it waits for is its own memory
// latency: the samples come a 128-byte line (64 samples) at a time -- one exposed round trip per 64 samples instead of
// one per sixteen (the sweeps of the index chain run a wave per SIMD or less: 130 us per sweep before, the chain itself
// takes most of that).  No look-ahead: with the next line requested a line ahead (and really in flight during the
// arithmetic: requested without a branch, so that the compiler waits by count) the chain was 10 % slower -- 32 more
// registers and their moves inside the dependent chain's code cost more than half a microsecond of waiting per line.
template <bool kWrite>
__device__ __forceinline__ void encode_run(const int16_t* __restrict__ x, uint32_t m, EncodeState& s,
                                           uint8_t* __restrict__ d, const EncodeLds& l) {
    uint32_t k = 0;
    for (; k + 64u <= m; k += 64u) {   // a 128-byte line (64 samples) at a time
        Pcm8 cur[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] = *reinterpret_cast<const Pcm8*>(x + k + 8 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = cur[2 * i + (j >> 2)].w[j & 3];
            const Bytes8 o = compress16(s, w, l);
            if (kWrite) *reinterpret_cast<Bytes8*>(d + (k >> 1) + 8 * i) = o;
        }
    }
    for (; k + 16u <= m; k += 16u) {
        const Pcm8 a = *reinterpret_cast<const Pcm8*>(x + k);
        const Pcm8 b = *reinterpret_cast<const Pcm8*>(x + k + 8u);
        const uint32_t w[8] = {a.w[0], a.w[1], a.w[2], a.w[3], b.w[0], b.w[1], b.w[2], b.w[3]};
        const Bytes8 o = compress16(s, w, l);
        if (kWrite) *reinterpret_cast<Bytes8*>(d + (k >> 1)) = o;
    }
    for (; k < m; k += 2u) {
        const uint32_t hi = compress(s, (uint32_t)(x[k] + 32768), l);
        const uint32_t lo = compress(s, (uint32_t)(x[k + 1u] + 32768), l);
        if (kWrite) d[k >> 1] = (uint8_t)((hi << 4) | lo);
    }
}

constexpr uint32_t kChainBlock = 256;   // chunks whose maps one workgroup composes through LDS (24 KB)

}  // namespace

// Decode, a wave per chunk.  Both chains of AdpcmImaExpandNibble (AdpcmIma.c:170-204) are compositions of "add a constant,
// clamp": the step index moves by index_table[nibble] inside 0..88 whatever the predictor does, and once the indices are
// known the predictor moves by a known +-diff inside -32768..32767.  x -> clamp(x + a, lo, hi) composed with another of its
// kind is of its kind again ((a1, lo1, hi1) then (a2, lo2, hi2) = (a1 + a2, clamp(lo1 + a2, lo2, hi2), clamp(hi1 + a2,
// lo2, hi2))), so both chains are prefix scans: every lane takes a run of consecutive samples (22 of a 1378-sample chunk),
// folds its run's index moves into one such triple, the wave scans the triples (six shuffle steps), every lane now knows the
// index its run starts at and looks its steps up; the same again for the predictor.  2.5x the arithmetic of the serial walk,
// but the chunk is read and written as the contiguous bytes it is (one lane per chunk read 16 bytes per 32 samples of its
// own at a 697-byte stride and waited for each) and a chunk is ~1 100 instructions deep instead of 16 000.  For a
// chip-filling batch that comes out even (0.42 ms per 200 000 chunks against 0.40: the kernel is bound by its 1 140
// vector instructions per wave); a single chunk -- AmvAudioDecode, the AVCodec plugin -- is decoded in microseconds.
namespace {

constexpr uint32_t kPer = 12;                // nibble bytes (pairs of samples) per lane and tile: a compile-time run length
constexpr uint32_t kTileBytes = 64 * kPer;   // nibble bytes a wave takes at a time (a chunk of 1378 samples has 689)
constexpr uint32_t kDecodeWaves = 4;         // chunks per workgroup

struct ClampAdd { int a, lo, hi; };          // x -> min(max(x + a, lo), hi)

__device__ __forceinline__ ClampAdd then(const ClampAdd& f, const ClampAdd& g, int lo, int hi) {   // g after f
    (void)lo; (void)hi;
    return ClampAdd{f.a + g.a, min(max(f.lo + g.a, g.lo), g.hi), min(max(f.hi + g.a, g.lo), g.hi)};
}

// exclusive scan over the wave's lanes of the composition (lane 0 gets the identity on lo..hi), and the whole wave's
template <int kLo, int kHi>
__device__ __forceinline__ ClampAdd wave_compose_before(ClampAdd mine, uint32_t lane, ClampAdd& all) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const ClampAdd left{__shfl_up(mine.a, d), __shfl_up(mine.lo, d), __shfl_up(mine.hi, d)};
        if (lane >= (uint32_t)d) mine = then(left, mine, kLo, kHi);
    }
    all = ClampAdd{__shfl(mine.a, 63), __shfl(mine.lo, 63), __shfl(mine.hi, 63)};
    ClampAdd before{__shfl_up(mine.a, 1), __shfl_up(mine.lo, 1), __shfl_up(mine.hi, 1)};
    if (lane == 0u) before = ClampAdd{0, kLo, kHi};
    return before;
}

__device__ __forceinline__ int apply(const ClampAdd& f, int x) { return min(max(x + f.a, f.lo), f.hi); }

}  // namespace

__global__ __launch_bounds__(64 * kDecodeWaves) void amv_adpcm_decode_kernel(
    const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offs,
    const uint32_t* __restrict__ lens, uint32_t n, int16_t* __restrict__ pcm,
    const uint64_t* __restrict__ pcm_offs, int32_t* __restrict__ final_state) {
    __shared__ uint32_t s_step[96];
    __shared__ __attribute__((aligned(16))) uint8_t s_in[kDecodeWaves][kTileBytes + 16];
    __shared__ __attribute__((aligned(16))) int16_t s_out[kDecodeWaves][2 * kTileBytes + 8];
    load_steps(s_step);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t i = blockIdx.x * kDecodeWaves + wave;
    if (i >= n) return;                               // (the whole wave; no workgroup barrier follows)
    const uint64_t off = offs[i];
    const uint32_t len = lens[i];
    if (len <= 8 || off > blob_bytes || len > blob_bytes - off) return;
    const uint8_t* c = blob + off;
    int predictor = (int16_t)(c[0] | (c[1] << 8));   // AMVDec.c:312
    int index = clip_index(c[2]);                    // AMVDec.c:313 (the reference indexes its table unchecked)
    int16_t* o = pcm + pcm_offs[i];
    const uint8_t* p = c + 8;
    const uint32_t nb = len - 8u;
    uint8_t* const in = s_in[wave];
    int16_t* const outs = s_out[wave];
    for (uint32_t t0 = 0; t0 < nb; t0 += kTileBytes) {
        const uint32_t tile = min(kTileBytes, nb - t0);
        // the tile's bytes, 16 per lane, as they lie in memory
        if (lane * 16u + 16u <= tile) {
            *reinterpret_cast<Bytes16*>(in + lane * 16u) = *reinterpret_cast<const Bytes16*>(p + t0 + lane * 16u);
        } else if (lane * 16u < tile) {               // the chunk's last piece: not a byte past its end is read
            for (uint32_t x = lane * 16u; x < tile; ++x) in[x] = p[t0 + x];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // this lane's run: kPer bytes = 2 * kPer samples, high nibble first (AdpcmIma.c:231-234); past the tile's end: none
        const uint32_t b0 = lane * kPer;
        uint32_t nib[2 * kPer];
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            const uint32_t byte = b0 + k < tile ? in[b0 + k] : 0x100u;    // 0x100: no sample here
            nib[2 * k] = byte >> 4;
            nib[2 * k + 1] = byte & 0x10fu;
        }
        // the run's index moves as one clamp-add
        ClampAdd mine{0, 0, 88};
#pragma unroll
        for (uint32_t k = 0; k < 2u * kPer; ++k) {
            const int adj = nib[k] < 16u ? index_adjust(nib[k] & 7u) : 0;
            mine = ClampAdd{mine.a + adj, clip_index(mine.lo + adj), clip_index(mine.hi + adj)};
        }
        ClampAdd all;
        const ClampAdd before = wave_compose_before<0, 88>(mine, lane, all);
        int idx = apply(before, index);
        index = apply(all, index);                   // where the tile ends: the next tile's (or the caller's) start
        // the steps, hence the signed differences, hence the run's predictor moves as one clamp-add
        int diff[2 * kPer];
        ClampAdd mine_p{0, -32768, 32767};
#pragma unroll
        for (uint32_t k = 0; k < 2u * kPer; ++k) {
            const bool there = nib[k] < 16u;
            const int step = (int)s_step[idx];
            const int mag = ((2 * (int)(nib[k] & 7u) + 1) * step) >> 3;
            diff[k] = there ? ((nib[k] & 8u) ? -mag : mag) : 0;
            idx = there ? clip_index(idx + index_adjust(nib[k] & 7u)) : idx;
            mine_p = ClampAdd{mine_p.a + diff[k], clip16(mine_p.lo + diff[k]), clip16(mine_p.hi + diff[k])};
        }
        ClampAdd all_p;
        const ClampAdd before_p = wave_compose_before<-32768, 32767>(mine_p, lane, all_p);
        int pr = apply(before_p, predictor);
        predictor = apply(all_p, predictor);
        // the samples, to LDS in pairs, then out as the contiguous bytes they are
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) {
            const int s0 = clip16(pr + diff[2 * k]);
            pr = clip16(s0 + diff[2 * k + 1]);
            if (b0 + k < tile) *reinterpret_cast<uint32_t*>(outs + 2u * (b0 + k)) = ((uint32_t)s0 & 0xffffu) | ((uint32_t)pr << 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int16_t* dst = o + 2u * t0;
        const uint32_t out_bytes = 4u * tile;
        for (uint32_t x = lane * 16u; x + 16u <= out_bytes; x += 64u * 16u)
            *reinterpret_cast<Pcm8*>(reinterpret_cast<uint8_t*>(dst) + x) = *reinterpret_cast<const Pcm8*>(reinterpret_cast<const uint8_t*>(outs) + x);
        if (lane < (out_bytes & 15u) / 2u)          // the tile's last 4, 8 or 12 bytes
            dst[(out_bytes & ~15u) / 2u + lane] = outs[(out_bytes & ~15u) / 2u + lane];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();            // the tile's LDS is free again
    }
    if (final_state && lane == 0u) { final_state[2 * i] = predictor; final_state[2 * i + 1] = index; }
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

// ---- the guessed-start route ----------------------------------------------------------------------------------
// device words with agent scope: what one lane stores another lane of the same launch may read (never torn, possibly
// the value before -- the sweeps are written for that)
__device__ __forceinline__ uint32_t peek(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void poke(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int encode_chunk(const int16_t* __restrict__ x, uint32_t nsamp, int start, uint8_t* __restrict__ d,
                                            const EncodeLds& l) {
    const uint32_t pairs = nsamp >> 1;
    const int prev = pairs ? x[0] : 0;              // adpcm.c:464
    d[0] = (uint8_t)(prev & 0xff);                  // :465 le16 first sample
    d[1] = (uint8_t)((prev >> 8) & 0xff);
    d[2] = (uint8_t)start;                          // :466 le16 step index
    d[3] = 0;
    const uint32_t cnt = pairs << 1;                // :479 le32 sample count
    d[4] = (uint8_t)cnt; d[5] = (uint8_t)(cnt >> 8); d[6] = (uint8_t)(cnt >> 16); d[7] = (uint8_t)(cnt >> 24);
    EncodeState s = encode_state(prev, start, l);
    encode_run<true>(x, cnt, s, d + 8, l);
    return (int)s.index;
}

constexpr uint32_t kGuessTail = 128;    // samples of the chunk before that the guess is run over
constexpr uint32_t kSettleMost = 1024;  // entries the one-workgroup kernel takes on
constexpr uint32_t kSettleRounds = 48;

// every chunk from a guessed start; state[i] = {start used, end reached}
__global__ __launch_bounds__(kEncodeBlock) void amv_adpcm_guess_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs, const uint32_t* __restrict__ nsamp, uint32_t n,
    uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs, uint2* __restrict__ state) {
    __shared__ EncodeLds s_tab;
    load_encode_tables(s_tab);
    const uint32_t i = blockIdx.x * kEncodeBlock + threadIdx.x;
    if (i >= n) return;
    int start = 0;                                  // chunk 0: the encoder context starts zeroed
    if (i) {
        const uint32_t mp = nsamp[i - 1u] & ~1u, tail = min(mp, kGuessTail);
        const int16_t* t = pcm + pcm_offs[i - 1u] + (mp - tail);
        EncodeState s = encode_state(tail ? t[0] : 0, 0, s_tab);
        encode_run<false>(t, tail, s, nullptr, s_tab);
        start = (int)s.index;
    }
    const int end = encode_chunk(pcm + pcm_offs[i], nsamp[i], start, blob + offs[i], s_tab);
    state[i] = make_uint2((uint32_t)start, (uint32_t)end);
}

// the chunks whose predecessor ended elsewhere than they assumed
__global__ __launch_bounds__(256) void amv_adpcm_mismatch_kernel(const uint2* __restrict__ state, uint32_t n, uint32_t* __restrict__ list,
                                                                 uint32_t* __restrict__ count) {
    __shared__ uint32_t s_count, s_base;
    if (threadIdx.x == 0) s_count = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x + 1u;
    const bool wrong = i < n && state[i].x != state[i - 1u].y;
    uint32_t slot = 0;
    if (wrong) slot = atomicAdd(&s_count, 1u);                  // one update of the global counter per workgroup
    __syncthreads();
    if (threadIdx.x == 0 && s_count) s_base = atomicAdd(count, s_count);
    __syncthreads();
    if (wrong) list[s_base + slot] = i;
}

// One sweep over a list: a listed chunk whose predecessor's end is not the start it used is coded again from there; if
// its own end moves, its successor is listed for the next sweep.  A chunk is listed by its predecessor only, so no list
// holds it twice.  Predecessor and successor may be in the same list: whichever of the predecessor's ends the successor
// reads, it is listed again when that end moved, and skips the work then if it had read the new one already.
__device__ __forceinline__ void sweep_one(const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs,
                                          const uint32_t* __restrict__ nsamp, uint32_t n, uint8_t* __restrict__ blob,
                                          const uint64_t* __restrict__ offs, uint2* __restrict__ state, uint32_t i,
                                          uint32_t* __restrict__ list_out, uint32_t* __restrict__ count_out, const EncodeLds& s_tab) {
    uint32_t* st = reinterpret_cast<uint32_t*>(state);
    const uint32_t start = peek(st + 2u * (i - 1u) + 1u);
    if (start == peek(st + 2u * i)) return;
    const uint32_t end = (uint32_t)encode_chunk(pcm + pcm_offs[i], nsamp[i], (int)start, blob + offs[i], s_tab);
    poke(st + 2u * i, start);
    if (end == peek(st + 2u * i + 1u)) return;
    poke(st + 2u * i + 1u, end);
    if (i + 1u < n) list_out[atomicAdd(count_out, 1u)] = i + 1u;
}

__global__ __launch_bounds__(kEncodeBlock) void amv_adpcm_sweep_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs, const uint32_t* __restrict__ nsamp, uint32_t n,
    uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs, uint2* __restrict__ state, const uint32_t* __restrict__ list_in,
    const uint32_t* __restrict__ count_in, uint32_t* __restrict__ list_out, uint32_t* __restrict__ count_out) {
    __shared__ EncodeLds s_tab;
    const uint32_t count = *count_in;
    if (blockIdx.x * kEncodeBlock >= count) return;
    // a sweep is as long as one chunk's serial chain and occupies a wave per SIMD or less: beside another stream's
    // kernels (the co-resident video decode) its waves should issue whenever they can
    __builtin_amdgcn_s_setprio(3);
    load_encode_tables(s_tab);
    for (uint32_t k = blockIdx.x * kEncodeBlock + threadIdx.x; k < count; k += gridDim.x * kEncodeBlock)
        sweep_one(pcm, pcm_offs, nsamp, n, blob, offs, state, list_in[k], list_out, count_out, s_tab);
}

// What the sweeps left, in one workgroup: rounds over the list until it is empty.  If it holds too much or does not
// empty within the rounds given, *need_map = 1 sends the stream down the exhaustive route.
__global__ __launch_bounds__(256) void amv_adpcm_settle_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs, const uint32_t* __restrict__ nsamp, uint32_t n,
    uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs, uint2* __restrict__ state, uint32_t* __restrict__ list_a,
    uint32_t* __restrict__ count_a, uint32_t* __restrict__ list_b, uint32_t* __restrict__ count_b, uint32_t* __restrict__ need_map) {
    __shared__ EncodeLds s_tab;
    __builtin_amdgcn_s_setprio(3);
    load_encode_tables(s_tab);
    for (uint32_t round = 0;; ++round) {
        const uint32_t count = peek(count_a);
        if (count == 0u) return;
        if (count > kSettleMost || round == kSettleRounds) {
            if (threadIdx.x == 0) *need_map = 1u;
            return;
        }
        for (uint32_t k = threadIdx.x; k < count; k += 256u)
            sweep_one(pcm, pcm_offs, nsamp, n, blob, offs, state, peek(list_a + k), list_b, count_b, s_tab);
        __threadfence();
        __syncthreads();                            // everyone has read count_a and finished its appends
        if (threadIdx.x == 0) poke(count_a, 0u);
        __threadfence();
        __syncthreads();
        uint32_t* t = list_a; list_a = list_b; list_b = t;
        t = count_a; count_a = count_b; count_b = t;
    }
}

// ---- the exhaustive route (queued behind the other; every kernel of it leaves at once unless *need says otherwise) ---
// state-only run of chunk i from start index s: where does step_index end up?  One lane per
// (chunk, start) pair, pairs packed densely into waves.
__global__ __launch_bounds__(kEncodeBlock) void amv_adpcm_map_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs,
    const uint32_t* __restrict__ nsamp, uint32_t n, uint8_t* __restrict__ map /* [n][96] */, const uint32_t* __restrict__ need) {
    __shared__ EncodeLds s_tab;
    if (need && *need == 0u) return;
    load_encode_tables(s_tab);
    const uint64_t pairs = (uint64_t)n * 89u;
    for (uint64_t pair = (uint64_t)blockIdx.x * kEncodeBlock + threadIdx.x; pair < pairs; pair += (uint64_t)gridDim.x * kEncodeBlock) {
        const uint32_t i = (uint32_t)(pair / 89u);
        const int16_t* x = pcm + pcm_offs[i];
        const uint32_t m = nsamp[i] & ~1u;
        EncodeState s = encode_state(m ? x[0] : 0, (int)(pair % 89u), s_tab);
        encode_run<false>(x, m, s, nullptr, s_tab);
        map[(uint64_t)i * 96u + (uint32_t)(pair % 89u)] = (uint8_t)s.index;
    }
}

// composition of the maps of kChainBlock consecutive chunks: bmap[b][s] = where start s ends up
__global__ __launch_bounds__(128) void amv_adpcm_chain_block_kernel(const uint8_t* __restrict__ map, uint32_t n,
                                                                    uint8_t* __restrict__ bmap, const uint32_t* __restrict__ need) {
    __shared__ uint32_t s_map[kChainBlock * 24u];
    if (need && *need == 0u) return;
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
                                                                  int32_t* __restrict__ bstart, const uint32_t* __restrict__ need) {
    __shared__ uint32_t s_map[kChainBlock * 24u];
    if (need && *need == 0u) return;
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
                                                                   int32_t* __restrict__ start, const uint32_t* __restrict__ need) {
    __shared__ uint32_t s_map[kChainBlock * 24u];
    __shared__ int32_t s_start[kChainBlock];
    if (need && *need == 0u) return;
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

__global__ __launch_bounds__(kEncodeBlock) void amv_adpcm_encode_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs,
    const uint32_t* __restrict__ nsamp, uint32_t n, const int32_t* __restrict__ step_in,
    uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs, const uint32_t* __restrict__ need) {
    __shared__ EncodeLds s_tab;
    if (need && *need == 0u) return;
    load_encode_tables(s_tab);
    const uint32_t i = blockIdx.x * kEncodeBlock + threadIdx.x;
    if (i >= n) return;
    encode_chunk(pcm + pcm_offs[i], nsamp[i], clip_index(step_in[i]), blob + offs[i], s_tab);
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
    hipLaunchKernelGGL(amv_adpcm_decode_kernel, dim3((n + kDecodeWaves - 1u) / kDecodeWaves), dim3(64 * kDecodeWaves), 0, s, blob, blob_bytes,
                       offs, lens, n, pcm, pcm_offs, final_state);
}

void launch_adpcm_wav_encode(const int16_t* samples, int groups, int32_t* state, uint8_t* frame, hipStream_t s) {
    hipLaunchKernelGGL(amv_adpcm_wav_encode_kernel, dim3(1), dim3(64), 0, s, samples, groups, state, frame);
}

void launch_adpcm_map(const int16_t* pcm, const uint64_t* pcm_offs, const uint32_t* nsamp, uint32_t n,
                      uint8_t* map, int32_t* start, const uint32_t* need, hipStream_t s) {
    // map: (n + nb) * 96 bytes, start: n + nb words, nb = adpcm_chain_blocks(n)
    if (n == 0) return;
    const uint32_t nb = adpcm_chain_blocks(n);
    uint8_t* bmap = map + (uint64_t)n * 96u;
    int32_t* bstart = start + n;
    const uint64_t groups = ((uint64_t)n * 89u + kEncodeBlock - 1u) / kEncodeBlock;
    hipLaunchKernelGGL(amv_adpcm_map_kernel, dim3((uint32_t)(groups < 4096u ? groups : 4096u)), dim3(kEncodeBlock), 0, s, pcm, pcm_offs,
                       nsamp, n, map, need);
    hipLaunchKernelGGL(amv_adpcm_chain_block_kernel, dim3(nb), dim3(128), 0, s, map, n, bmap, need);
    hipLaunchKernelGGL(amv_adpcm_chain_top_kernel, dim3(1), dim3(128), 0, s, bmap, nb, bstart, need);
    hipLaunchKernelGGL(amv_adpcm_chain_fill_kernel, dim3(nb), dim3(128), 0, s, map, n, bstart, start, need);
}

uint32_t adpcm_chain_blocks(uint32_t n) { return (n + kChainBlock - 1u) / kChainBlock; }

void adpcm_quotient_table(float out[89]) {
    for (int i = 0; i < 89; ++i) out[i] = kAdpcmHost.rcp[i];
}

void launch_adpcm_encode(const int16_t* pcm, const uint64_t* pcm_offs, const uint32_t* nsamp,
                         uint32_t n, const int32_t* step_in, uint8_t* blob, const uint64_t* offs, const uint32_t* need,
                         hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(amv_adpcm_encode_kernel, dim3((n + kEncodeBlock - 1u) / kEncodeBlock), dim3(kEncodeBlock), 0, s, pcm, pcm_offs,
                       nsamp, n, step_in, blob, offs, need);
}

// The guessed-start route.  work: adpcm_chain_workspace(n) bytes = state[n] (uint2), two lists of n words, 64 words of
// counters (zeroed here; word 63 is the flag launch_adpcm_map / launch_adpcm_encode are given as `need`).
uint64_t adpcm_chain_workspace(uint32_t n) { return (uint64_t)n * 16u + 256u; }

const uint32_t* launch_adpcm_chain(const int16_t* pcm, const uint64_t* pcm_offs, const uint32_t* nsamp, uint32_t n, uint8_t* blob,
                                   const uint64_t* offs, void* work, uint32_t sweeps, hipStream_t s) {
    uint2* state = static_cast<uint2*>(work);
    uint32_t* list[2] = {reinterpret_cast<uint32_t*>(state + n), reinterpret_cast<uint32_t*>(state + n) + n};
    uint32_t* count = list[1] + n;                   // [0 .. sweeps + 1]: one per list generation; [63]: the flag
    if (sweeps > 60u) sweeps = 60u;
    (void)hipMemsetAsync(count, 0, 256, s);
    hipLaunchKernelGGL(amv_adpcm_guess_kernel, dim3((n + kEncodeBlock - 1u) / kEncodeBlock), dim3(kEncodeBlock), 0, s, pcm, pcm_offs, nsamp, n,
                       blob, offs, state);
    if (n > 1u) {
        hipLaunchKernelGGL(amv_adpcm_mismatch_kernel, dim3((n + 254u) / 256u), dim3(256), 0, s, state, n, list[0], count);
        // sweep k's list is a fraction of the one before; the grid is sized for the first and strides if it must
        uint32_t grid = (n + kEncodeBlock - 1u) / kEncodeBlock;
        for (uint32_t k = 0; k < sweeps; ++k) {
            hipLaunchKernelGGL(amv_adpcm_sweep_kernel, dim3(grid), dim3(kEncodeBlock), 0, s, pcm, pcm_offs, nsamp, n, blob, offs, state,
                               list[k & 1u], count + k, list[(k + 1u) & 1u], count + k + 1u);
            grid = grid > 256u ? (grid + 1u) / 2u : grid;
        }
        hipLaunchKernelGGL(amv_adpcm_settle_kernel, dim3(1), dim3(256), 0, s, pcm, pcm_offs, nsamp, n, blob, offs, state,
                           list[sweeps & 1u], count + sweeps, list[(sweeps + 1u) & 1u], count + sweeps + 1u, count + 63);
    }
    return count + 63;
}

}  // namespace amv
