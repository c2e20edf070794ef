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

// adpcm_ima_compress_sample, adpcm.c:219-227, shaped for a lane that is alone on its SIMD.  The sweeps of the index chain are
// as long as ONE chunk's serial chain, and a lone wave on gfx950 pays 9 cycles per dependent vector instruction, 5.4 - 6.3
// per independent one and ~60 per LDS read (tools/microbench_lone_wave.hip, profiles/r04_lone_wave.txt): what counts is the
// number of instructions per sample AND the length of the two loops that run through them (predictor -> predictor,
// quantiser factor -> nibble -> next factor).  Everything is float arithmetic on exactly representable values:
//   * d = sample - predictor; min(7, |d| * 4 / step) + 8 = min(15, trunc(fma(|d|, r, 8))) with r = 4 / step nudged up by 2^-20:
//     exact for every |d| < 65536 and every step of the table -- the nudge (<= 2^-17 on a quotient below 8) plus the roundings
//     of r and of the fma (2^-21 each) stay under 1 / 32767, the closest a quotient comes to the next integer from underneath,
//     and an exact multiple k >= 1 lands at k + 0.9 * 2^-20 - 2^-21 > k (tests/test_abi_and_host.py checks all 89 x 65536
//     cases against the integer division with the table amvhip_adpcm_quotient_table hands out, product and fma form);
//   * the + 8 is there for its bit pattern: a float in [8, 16) is 0x41000000 | q << 20, so byte 2 of it is 16 * q -- the offset
//     of cell q in a row of 16-byte cells (one SDWA add makes the LDS address, no conversion) and the nibble's magnitude in
//     place for the output byte (one SDWA move);
//   * (step * (2 q + 1)) >> 3 = trunc(fma(q + 8, step / 4, step / 8 - 2 step)): a multiple of 1/8 below 2^16, exact;
//   * the table is indexed by (step index, q) and holds the state AFTER that move -- step / 4, step / 8 - 2 step, r and the row
//     of the new index (AdpcmCell, 89 x 8 cells of 16 bytes): index_table, the clamp to 0..88 and the step look-up are one
//     ds_read_b128 whose address is row + byte 2 of (q + 8).
// 15 vector instructions per sample with output, 12 without (27 with the integer form of round 3); the predictor loop is nine
// instructions long, the factor loop four + the LDS read.
struct AdpcmCell {
    float s4, s8m, rcp;   // step / 4, step / 8 - 2 * step, quotient factor of the index this cell leads to
    uint32_t row;         // byte offset of that index's row of cells
};
static_assert(sizeof(AdpcmCell) == 16, "one ds_read_b128 per sample");

struct AdpcmTables {
    AdpcmCell cell[89][8];
    float rcp[89];
};

constexpr float quotient_factor(int step) { return (float)((4.0 / step) * (1.0 + 1.0 / 1048576.0)); }
constexpr int clip_index_c(int v) { return v < 0 ? 0 : (v > 88 ? 88 : v); }
constexpr AdpcmTables make_adpcm_tables() {
    AdpcmTables t{};
    for (int i = 0; i < 89; ++i) {
        for (int q = 0; q < 8; ++q) {
            const int to = clip_index_c(i + (q < 4 ? -1 : 2 * q - 6));       // kImaIndexAdjust
            const int step = kImaStep[to];
            t.cell[i][q].s4 = (float)(step / 4.0);
            t.cell[i][q].s8m = (float)(step / 8.0 - 2.0 * step);
            t.cell[i][q].rcp = quotient_factor(step);
            t.cell[i][q].row = (uint32_t)to * 128u;
        }
        t.rcp[i] = quotient_factor(kImaStep[i]);
    }
    return t;
}
static constexpr AdpcmTables kAdpcmHost = make_adpcm_tables();
__device__ const AdpcmTables kAdpcmTables = make_adpcm_tables();

// LDS image of a workgroup that encodes: the cells (11.4 KB)
constexpr uint32_t kEncodeBlock = 128;   // two waves share the cells: 19.6 KB of LDS, eight workgroups per CU
struct EncodeLds {
    uint4 cell[89 * 8];
};

__device__ __forceinline__ void load_encode_tables(EncodeLds& l) {
    const uint4* src = reinterpret_cast<const uint4*>(&kAdpcmTables.cell[0][0]);
    for (uint32_t i = threadIdx.x; i < 89u * 8u; i += blockDim.x) l.cell[i] = src[i];
    __syncthreads();
}

struct EncodeState {
    float prev;            // predictor
    float s4, s8m, rcp;    // of the current step index
    uint32_t row;          // 128 * step index
};

__device__ __forceinline__ void take_cell(EncodeState& s, const uint4& c) {
    s.s4 = __uint_as_float(c.x);
    s.s8m = __uint_as_float(c.y);
    s.rcp = __uint_as_float(c.z);
    s.row = c.w;
}

__device__ __forceinline__ EncodeState encode_state(int prev, int index, const EncodeLds& l) {
    EncodeState s;
    s.prev = (float)prev;
    take_cell(s, index < 88 ? l.cell[(index + 1) * 8] : l.cell[88 * 8 + 4]);   // a cell that leads to `index`
    return s;
}

__device__ __forceinline__ int state_index(const EncodeState& s) { return (int)(s.row >> 7); }

// One sample.  kSlot < 4 (kWrite): the nibble goes to byte kSlot of the pair (qb, sb) as magnitude << 4 and sign << 7.
template <bool kWrite, int kSlot>
__device__ __forceinline__ void compress(EncodeState& s, float sample, const EncodeLds& l, uint32_t& qb, uint32_t& sb, uint32_t& tick) {
    const float d = sample - s.prev;
    const float q8 = fminf(truncf(__builtin_fmaf(__builtin_fabsf(d), s.rcp, 8.0f)), 15.0f);
    // the next cell is asked for the moment its address exists -- the factor loop (cell -> quotient -> address -> cell) is
    // the longer of the two, and the scheduler would put four independent instructions in front of the read
    uint32_t addr;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2"
        : "=v"(addr) : "v"(s.row), "v"(q8));
    const float s4 = s.s4, s8m = s.s8m;
    // ... as TWO 8-byte reads: a ds_read_b128 with a row of its own per lane occupies the CU's LDS for 27 cycles, a
    // ds_read_b64 for 6 - 7 (tools/microbench_lds_lookup.hip, profiles/r04_lds_lookup.txt), and with three waves per SIMD
    // asking once per sample that was what the guess pass was bound by.  The half the factor loop waits for goes first;
    // the address is hidden from the compiler in between, or it makes the two reads one again.
    const uint8_t* cells = reinterpret_cast<const uint8_t*>(l.cell);
    const uint2 hi = *reinterpret_cast<const uint2*>(cells + addr + 8u);
    asm("" : "+v"(addr));
    const uint2 lo = *reinterpret_cast<const uint2*>(cells + addr);
    take_cell(s, make_uint4(lo.x, lo.y, hi.x, hi.y));
    __builtin_amdgcn_sched_barrier(0);
    const float mag = truncf(__builtin_fmaf(q8, s4, s8m));                          // step * yamaha_difflookup[nibble] / 8
    s.prev = __builtin_amdgcn_fmed3f(s.prev + __builtin_copysignf(mag, d), -32768.0f, 32767.0f);
    tick = __float_as_uint(q8);     // known before the cell is asked for: what the next sample's conversion is tied behind
    if (kWrite) {
        const uint32_t c80 = 0x80u;
        if (kSlot == 0) {
            asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(qb) : "v"(q8));
            asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(sb) : "v"(d), "v"(c80));
        } else if (kSlot == 1) {
            asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(qb) : "v"(q8));
            asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(sb) : "v"(d), "v"(c80));
        } else if (kSlot == 2) {
            asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(qb) : "v"(q8));
            asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(sb) : "v"(d), "v"(c80));
        } else {
            asm("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(qb) : "v"(q8));
            asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(sb) : "v"(d), "v"(c80));
        }
    }
}

// the two samples of a word as floats (one SDWA conversion each).  `after` is not read: it ties the conversion behind the
// state it names, or the scheduler converts a whole line's 64 samples ahead and the kernel needs 140 registers
__device__ __forceinline__ float sample_lo(uint32_t w, uint32_t after) {
    float f;
    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(f) : "v"(w), "v"(after));
    return f;
}
__device__ __forceinline__ float sample_hi(uint32_t w, uint32_t after) {
    float f;
    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f) : "v"(w), "v"(after));
    return f;
}

// eight samples (four words) -> four bytes: every byte of (qb | sb) holds one nibble << 4; :489-493 high nibble = earlier sample
template <bool kWrite>
__device__ __forceinline__ uint32_t compress8(EncodeState& s, const uint32_t* w, const EncodeLds& l) {
    uint32_t q0 = 0u, s0 = 0u, q1 = 0u, s1 = 0u, tick = s.row;
    compress<kWrite, 0>(s, sample_lo(w[0], tick), l, q0, s0, tick);
    compress<kWrite, 1>(s, sample_hi(w[0], tick), l, q0, s0, tick);
    compress<kWrite, 2>(s, sample_lo(w[1], tick), l, q0, s0, tick);
    compress<kWrite, 3>(s, sample_hi(w[1], tick), l, q0, s0, tick);
    compress<kWrite, 0>(s, sample_lo(w[2], tick), l, q1, s1, tick);
    compress<kWrite, 1>(s, sample_hi(w[2], tick), l, q1, s1, tick);
    compress<kWrite, 2>(s, sample_lo(w[3], tick), l, q1, s1, tick);
    compress<kWrite, 3>(s, sample_hi(w[3], tick), l, q1, s1, tick);
    if (!kWrite) return 0u;
    const uint32_t a = q0 | s0, b = q1 | s1;
    const uint32_t even = __builtin_amdgcn_perm(b, a, 0x06040200u), odd = __builtin_amdgcn_perm(b, a, 0x07050301u);
    return even | (odd >> 4);
}

// sixteen samples (eight words) -> eight bytes
template <bool kWrite>
__device__ __forceinline__ Bytes8 compress16(EncodeState& s, const uint32_t* w, const EncodeLds& l) {
    Bytes8 o;
    o.w[0] = compress8<kWrite>(s, w, l);
    o.w[1] = compress8<kWrite>(s, w + 4, l);
    return o;
}

// A chunk's samples from state s (m is even).  One lane owns the chunk, so what it waits for is its own memory
// latency: the samples come a 128-byte line (64 samples) at a time -- one exposed round trip per 64 samples instead of
// one per sixteen.  No look-ahead: with the next line requested a line ahead (and really in flight during the
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
            const Bytes8 o = compress16<kWrite>(s, w, l);
            if (kWrite) *reinterpret_cast<Bytes8*>(d + (k >> 1) + 8 * i) = o;
        }
    }
    for (; k + 16u <= m; k += 16u) {
        const Pcm8 a = *reinterpret_cast<const Pcm8*>(x + k);
        const Pcm8 b = *reinterpret_cast<const Pcm8*>(x + k + 8u);
        const uint32_t w[8] = {a.w[0], a.w[1], a.w[2], a.w[3], b.w[0], b.w[1], b.w[2], b.w[3]};
        const Bytes8 o = compress16<kWrite>(s, w, l);
        if (kWrite) *reinterpret_cast<Bytes8*>(d + (k >> 1)) = o;
    }
    for (; k < m; k += 2u) {
        uint32_t qb = 0u, sb = 0u, tick;
        compress<kWrite, 1>(s, (float)x[k], l, qb, sb, tick);
        compress<kWrite, 0>(s, (float)x[k + 1u], l, qb, sb, tick);
        if (kWrite) d[k >> 1] = (uint8_t)((((qb | sb) >> 8) & 0xf0u) | (((qb | sb) >> 4) & 0x0fu));
    }
}

// ---- a wave's 64 chunks at a time, samples and bytes staged through LDS ---------------------------------------------
// One lane per chunk makes every lane read and write lines of its own: 64 lines per memory instruction, 16 bytes of each
// read and 8 written at a time.  Measured on the guess pass over 200 000 chunks (variant builds, tools/time_adpcm.py):
// 572 us as that, 249 with the stores left out, 210 with the loads aimed at one line, 154 with neither -- the arithmetic
// is a quarter of the kernel, the rest is the shape of its memory accesses.  Here the WAVE moves the data: a tile of 32
// samples per row is fetched as 64-byte pieces (four lanes to a row, sixteen rows to an instruction) a tile ahead of
// the arithmetic, into sixteen registers that go to LDS when the tile's turn comes; every lane then reads its own row.
// The bytes a lane makes collect in registers for four tiles and leave through the same LDS tile as 64-byte pieces,
// again four lanes to a row.  (16-byte pieces of a row sit XOR-ed with bits of the row number: the lanes one ds pass
// serves meet on a bank two at a time.)  All 64 lanes of a wave call encode_rows together; a lane without work passes
// m = 0.  LDS-DMA for the fetch (global_load_lds_dwordx4 into a second buffer, no registers) was slower: a wave that is
// alone on its SIMD -- the sweeps -- pays 100+ cycles to issue each piece, and hipcc follows the builtin with
// s_waitcnt vmcnt(0) as soon as any LDS read comes behind it.
constexpr uint32_t kTile = 32;               // samples per row and tile (64 bytes in, 16 bytes out)
struct StageLds {
    uint4 buf[64 * 4];                       // 64 rows of 64 bytes: 4 KB per wave
};

// LDS hand-over between the lanes of ONE wave: its DS instructions execute in issue order, so all it takes is that the
// compiler keeps them in program order (a fence would also drain the vector-memory counter: the fetch ahead, the stores)
__device__ __forceinline__ void wave_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// 16 bytes at any even (load) or any (store) address, as GLOBAL accesses: through a generic pointer they would be flat_*
// instructions, which count on both memory counters and complete out of order -- every wait would be for everything
typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
typedef U32x4 U32x4Align2 __attribute__((aligned(2)));
typedef U32x4 U32x4Align1 __attribute__((aligned(1)));
__device__ __forceinline__ Pcm8 global_load_16(uint64_t p) {
    const U32x4 v = *(const __attribute__((address_space(1))) U32x4Align2*)p;
    return Pcm8{{v.x, v.y, v.z, v.w}};
}
__device__ __forceinline__ void global_store_16(uint64_t p, const uint4& v) {
    U32x4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *(__attribute__((address_space(1))) U32x4Align1*)p = t;
}

__device__ __forceinline__ uint64_t shfl64(uint64_t v, uint32_t from) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, (int)from), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)from);
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t swz(uint32_t row) { return (row >> 1) & 3u; }

template <bool kWrite>
__device__ __forceinline__ void encode_rows(const int16_t* __restrict__ x, uint32_t m, EncodeState& s, uint8_t* __restrict__ d,
                                            const EncodeLds& l, StageLds& st) {
    const uint32_t lane = threadIdx.x & 63u, piece = lane & 3u, r0 = lane >> 2;
    const uint32_t mine = m / kTile;
    // the wave's longest row: its tile count bounds the loop, its start is the address rows without a tile read
    uint32_t key = (mine << 6) | lane;
#pragma unroll
    for (int off = 32; off; off >>= 1) key = max(key, (uint32_t)__shfl_xor((int)key, off));
    key = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
    const uint32_t most = key >> 6;
    if (most) {
        const uint64_t safe = shfl64(reinterpret_cast<uint64_t>(x), key & 63u);
        // the four rows this lane fetches for (and stores for): r0 + 16 j
        uint64_t rp[4], rd[4];
        uint32_t rt[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            rp[j] = shfl64(reinterpret_cast<uint64_t>(x), r0 + 16u * j);
            rd[j] = kWrite ? shfl64(reinterpret_cast<uint64_t>(d), r0 + 16u * j) : 0u;
            rt[j] = (uint32_t)__shfl((int)mine, (int)(r0 + 16u * j));
        }
        auto fetch = [&](uint32_t t, Pcm8* r) {      // a row that has no tile t reads `safe`: no branch, never a byte past anybody's samples
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j)
                r[j] = global_load_16((t < rt[j] ? rp[j] + (uint64_t)t * (2u * kTile) : safe) + (piece << 4));
        };
        uint4 acc[4];                                 // the lane's bytes of four tiles
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = make_uint4(0u, 0u, 0u, 0u);
        auto flush = [&](uint32_t g) {                // tiles 4 g .. 4 g + 3 of every row that has all four: 64 bytes per row
            wave_sync();
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) st.buf[lane * 4u + (k ^ swz(lane))] = acc[k];
            wave_sync();
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                const uint32_t row = r0 + 16u * j;
                const uint4 v = st.buf[row * 4u + (piece ^ swz(row))];
                if (4u * g + 4u <= rt[j])
                    global_store_16(rd[j] + (uint64_t)g * 64u + (piece << 4), v);
            }
        };
        Pcm8 next[4];
        fetch(0u, next);
        for (uint32_t t = 0; t < most; ++t) {
            wave_sync();                              // the tile before has been read
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                const uint32_t row = r0 + 16u * j;
                st.buf[row * 4u + (piece ^ swz(row))] = make_uint4(next[j].w[0], next[j].w[1], next[j].w[2], next[j].w[3]);
            }
            wave_sync();
            if (t + 1u < most) fetch(t + 1u, next);
            uint4 cur[4];
#pragma unroll
            for (uint32_t c = 0; c < 4u; ++c) cur[c] = st.buf[lane * 4u + (c ^ swz(lane))];
            // the four tiles before leave now, through the tile's LDS (free until the next trip), ahead of this tile's
            // arithmetic: the next wait on the vector-memory counter finds these stores a tile old
            if (kWrite && t && (t & 3u) == 0u) flush(t / 4u - 1u);
            if (t < mine) {
                const uint32_t w0[8] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w, cur[1].x, cur[1].y, cur[1].z, cur[1].w};
                const uint32_t w1[8] = {cur[2].x, cur[2].y, cur[2].z, cur[2].w, cur[3].x, cur[3].y, cur[3].z, cur[3].w};
                const Bytes8 a = compress16<kWrite>(s, w0, l), b = compress16<kWrite>(s, w1, l);
                const uint4 o = make_uint4(a.w[0], a.w[1], b.w[0], b.w[1]);
                if ((t & 3u) == 0u) acc[0] = o;       // (t is the wave's: these are not per-lane selects)
                else if ((t & 3u) == 1u) acc[1] = o;
                else if ((t & 3u) == 2u) acc[2] = o;
                else acc[3] = o;
            }
        }
        if (kWrite) {
            if ((most & 3u) == 0u) flush(most / 4u - 1u);
            // the tiles behind a row's last full group of four: the lane's own 16-byte stores
            const uint32_t lo = mine & ~3u;
#pragma unroll
            for (uint32_t k = 0; k < 3u; ++k)
                if (k < (mine & 3u)) *reinterpret_cast<Bytes16*>(d + (uint64_t)(lo + k) * 16u) = Bytes16{{acc[k].x, acc[k].y, acc[k].z, acc[k].w}};
        }
        wave_sync();
    }
    const uint32_t done = mine * kTile;               // what is left of the row: fewer than 32 samples, read directly
    encode_run<kWrite>(x + done, m - done, s, kWrite ? d + (done >> 1) : d, l);
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

// state[i] = {start the chunk's bytes were coded from, end reached from it}, replaced as ONE 64-bit word so that no reader
// and no second writer ever sees one chunk's start beside another coding's end; returns the end that was there.  Whoever
// changes a chunk's end lists its successor: that rule, and a listed chunk being coded again whenever its start is not its
// predecessor's end, is all the sweeps rest on.  kPredicted in the start: the end is where the chunk WILL end from that
// start (the front sweep's look-ahead), its bytes are not coded yet -- no real start equals it, so the chunk is coded
// when its turn comes.
constexpr uint32_t kPredicted = 0x100u;
__device__ __forceinline__ uint32_t swap_state(uint2* p, uint32_t start, uint32_t end) {
    const uint64_t old = __hip_atomic_exchange(reinterpret_cast<uint64_t*>(p), (uint64_t)start | (uint64_t)end << 32, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
    return (uint32_t)(old >> 32);
}

// the chunks of a wave's 64 lanes (live: this lane has one), each from its start index; returns the end index
__device__ __forceinline__ int encode_chunk(const int16_t* __restrict__ x, uint32_t nsamp, int start, uint8_t* __restrict__ d,
                                            bool live, const EncodeLds& l, StageLds& st) {
    const uint32_t pairs = live ? nsamp >> 1 : 0u;
    const int prev = pairs ? x[0] : 0;              // adpcm.c:464
    const uint32_t cnt = pairs << 1;                // :479 le32 sample count
    if (live) {
        // :465 le16 first sample, :466 le16 step index, :479 le32 sample count -- the eight bytes as one store
        *reinterpret_cast<Bytes8*>(d) = Bytes8{{((uint32_t)prev & 0xffffu) | ((uint32_t)start & 0xffu) << 16, cnt}};
    }
    EncodeState s = encode_state(prev, start, l);
    encode_rows<true>(x, cnt, s, d + 8, l, st);
    return state_index(s);
}

// ... and one lane's chunk by itself (the sweeps over short lists: a wave with a handful of live lanes is as long as one
// chunk's serial chain, and staging costs that chain a fifth more than reading the lane's own lines as they come)
__device__ __forceinline__ int encode_chunk_alone(const int16_t* __restrict__ x, uint32_t nsamp, int start, uint8_t* __restrict__ d,
                                                  const EncodeLds& l) {
    const uint32_t pairs = nsamp >> 1;
    const int prev = pairs ? x[0] : 0;
    const uint32_t cnt = pairs << 1;
    *reinterpret_cast<Bytes8*>(d) = Bytes8{{((uint32_t)prev & 0xffffu) | ((uint32_t)start & 0xffu) << 16, cnt}};
    EncodeState s = encode_state(prev, start, l);
    encode_run<true>(x, cnt, s, d + 8, l);
    return state_index(s);
}

constexpr uint32_t kGuessTail = 128;    // samples of the chunk before that the guess is run over
constexpr uint32_t kSettleMost = 2048;  // entries the one-workgroup kernel takes on (the front sweep may leave five per head)
constexpr uint32_t kSettleRounds = 48;

// every chunk from a guessed start; state[i] = {start used, end reached}
__global__ __launch_bounds__(kEncodeBlock) void amv_adpcm_guess_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs, const uint32_t* __restrict__ nsamp, uint32_t n,
    uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs, uint2* __restrict__ state, uint32_t* __restrict__ list,
    uint32_t* __restrict__ count) {
    __shared__ EncodeLds s_tab;
    __shared__ StageLds s_stage[kEncodeBlock / 64u];
    __shared__ uint32_t s_end[kEncodeBlock / 64u];
    load_encode_tables(s_tab);
    StageLds& st = s_stage[threadIdx.x >> 6];
    const uint32_t i = blockIdx.x * kEncodeBlock + threadIdx.x;
    const bool live = i < n;
    int start = 0;                                  // chunk 0: the encoder context starts zeroed
    {
        const bool has = live && i > 0u;
        const uint32_t mp = has ? nsamp[i - 1u] & ~1u : 0u, tail = min(mp, kGuessTail);
        const int16_t* t = has ? pcm + pcm_offs[i - 1u] + (mp - tail) : pcm;
        EncodeState s = encode_state(tail ? t[0] : 0, 0, s_tab);
        encode_rows<false>(t, tail, s, nullptr, s_tab, st);
        if (has) start = state_index(s);
    }
    const int end = encode_chunk(live ? pcm + pcm_offs[i] : pcm, live ? nsamp[i] : 0u, start, live ? blob + offs[i] : blob, live, s_tab, st);
    if (live) state[i] = make_uint2((uint32_t)start, (uint32_t)end);
    // The first list: the chunks whose predecessor ended elsewhere than they assumed.  The predecessor is the lane before
    // (the wave before, through LDS); a workgroup's first chunk cannot know yet and is listed whatever it assumed -- the
    // first sweep looks before it codes.  (A kernel of its own for this was 11 us between the guess pass and the sweeps.)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 63u) s_end[wave] = (uint32_t)end;
    __syncthreads();
    const uint32_t left = (uint32_t)__shfl_up(end, 1);          // (by every lane: lane 1 reads lane 0's)
    const uint32_t before = lane ? left : (wave ? s_end[wave - 1u] : ~0u);
    const bool wrong = live && i > 0u && (uint32_t)start != before;
    const uint64_t mask = __ballot(wrong);
    if (mask) {
        uint32_t base = 0;
        if (lane == 0u) base = atomicAdd(count, (uint32_t)__popcll(mask));
        base = (uint32_t)__shfl((int)base, 0);
        if (wrong) list[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = i;
    }
}

// One sweep over a list: a listed chunk whose predecessor's end is not the start it used is coded again from there; if
// its own end moves, its successor is listed for the next sweep.  A chunk is listed by its predecessor only, so no list
// holds it twice.  Predecessor and successor may be in the same list: whichever of the predecessor's ends the successor
// reads, it is listed again when that end moved, and skips the work then if it had read the new one already.
template <bool kStaged>
__device__ __forceinline__ void sweep_one(const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs,
                                          const uint32_t* __restrict__ nsamp, uint32_t n, uint8_t* __restrict__ blob,
                                          const uint64_t* __restrict__ offs, uint2* __restrict__ state, uint32_t i, bool listed,
                                          uint32_t* __restrict__ list_out, uint32_t* __restrict__ count_out, const EncodeLds& s_tab,
                                          StageLds* st) {
    // (kStaged: all 64 lanes of the wave come here together, st = the wave's staging tile; `listed`: this lane has an entry i
    // of the list)
    uint32_t* sw = reinterpret_cast<uint32_t*>(state);
    uint32_t start = 0u;
    bool todo = false;
    if (listed) {
        start = peek(sw + 2u * (i - 1u) + 1u);
        todo = start != peek(sw + 2u * i);
    }
    uint32_t end;
    if (kStaged) {
        end = (uint32_t)encode_chunk(todo ? pcm + pcm_offs[i] : pcm, todo ? nsamp[i] : 0u, (int)start, todo ? blob + offs[i] : blob, todo,
                                     s_tab, *st);
        if (!todo) return;
    } else {
        if (!todo) return;
        end = (uint32_t)encode_chunk_alone(pcm + pcm_offs[i], nsamp[i], (int)start, blob + offs[i], s_tab);
    }
    if (swap_state(state + i, start, end) != end && i + 1u < n) list_out[atomicAdd(count_out, 1u)] = i + 1u;
}

constexpr uint32_t kStagedAbove = 32768;   // list entries from which a sweep is bound by its memory accesses, not by one chunk's chain

__global__ __launch_bounds__(kEncodeBlock) void amv_adpcm_sweep_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs, const uint32_t* __restrict__ nsamp, uint32_t n,
    uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs, uint2* __restrict__ state, const uint32_t* __restrict__ list_in,
    const uint32_t* __restrict__ count_in, uint32_t* __restrict__ list_out, uint32_t* __restrict__ count_out) {
    __shared__ EncodeLds s_tab;
    __shared__ StageLds s_stage[kEncodeBlock / 64u];
    const uint32_t count = *count_in;
    if (blockIdx.x * kEncodeBlock >= count) return;
    // a sweep is as long as one chunk's serial chain and occupies a wave per SIMD or less: beside another stream's
    // kernels (the co-resident video decode) its waves should issue whenever they can
    __builtin_amdgcn_s_setprio(3);
    load_encode_tables(s_tab);
    StageLds& st = s_stage[threadIdx.x >> 6];
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t base = blockIdx.x * kEncodeBlock + (threadIdx.x & ~63u); base < count; base += gridDim.x * kEncodeBlock) {
        const bool listed = base + lane < count;
        const uint32_t i = listed ? list_in[base + lane] : 1u;
        if (count > kStagedAbove) sweep_one<true>(pcm, pcm_offs, nsamp, n, blob, offs, state, i, listed, list_out, count_out, s_tab, &st);
        else sweep_one<false>(pcm, pcm_offs, nsamp, n, blob, offs, state, i, listed, list_out, count_out, s_tab, nullptr);
    }
}

// A sweep that looks ahead (round 4).  A listed chunk is the head of a FRONT: if its end moves, its successor must be coded
// again, and so on -- one chunk per sweep, each sweep one chunk's serial chain long.  Once the list is short there are
// lanes to spare: a workgroup per head codes the head again (one lane) and, beside it, runs the next kFrontAhead chunks
// from ALL 89 start indices (state only, a lane per (chunk, start)): when the head's new end is known, the maps say at once
// where each of those chunks ends.  Their predicted ends are written to `state` with the start they follow from, marked
// kPredicted (the chunks' bytes are stale until the next round codes them again -- from starts that are now all known, in
// ONE round), and every chunk whose predecessor's end moved is listed, as the plain sweep does: once, whichever of two
// fronts reaches it.  The sweeps' rule that a listed chunk is coded whenever its start is not its predecessor's end makes
// whatever two fronts write about the same chunk converge.  With more than kFrontMost heads the kernel is a plain sweep.
constexpr uint32_t kFrontAhead = 4;
constexpr uint32_t kFrontMost = 320;        // heads: seven waves each, so about two waves per SIMD at most
constexpr uint32_t kFrontHead = 384;        // waves 0-5: 4 x 89 map lanes; wave 6, lane 0: the head
constexpr uint32_t kFrontThreads = 448;
static_assert(kFrontAhead * 89u <= kFrontHead && kFrontHead + 64u == kFrontThreads, "map lanes, then the head's wave");

__global__ __launch_bounds__(kFrontThreads) void amv_adpcm_front_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs, const uint32_t* __restrict__ nsamp, uint32_t n,
    uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs, uint2* __restrict__ state, const uint32_t* __restrict__ list_in,
    const uint32_t* __restrict__ count_in, uint32_t* __restrict__ list_out, uint32_t* __restrict__ count_out,
    uint32_t* __restrict__ listed, uint32_t* __restrict__ need_map) {
    __shared__ EncodeLds s_tab;
    __shared__ uint8_t s_map[kFrontAhead][96];
    __shared__ uint32_t s_head[4];               // todo, start, end of the head
    const uint32_t count = *count_in;
    if (blockIdx.x >= count) return;
    __builtin_amdgcn_s_setprio(3);
    load_encode_tables(s_tab);
    const bool ahead = count <= kFrontMost && (kFrontAhead + 1u) * count + 8u <= n;     // (a list holds n entries: room for every append)
    uint32_t* sw = reinterpret_cast<uint32_t*>(state);
    if (!ahead) {                                    // the list is still long (or the stream tiny): a plain sweep, a lane per entry
        for (uint32_t k = blockIdx.x * kFrontThreads + threadIdx.x; k < count; k += gridDim.x * kFrontThreads)
            sweep_one<false>(pcm, pcm_offs, nsamp, n, blob, offs, state, list_in[k], true, list_out, count_out, s_tab, nullptr);
        return;
    }
    auto append = [&](uint32_t x) {
        // once per chunk (`listed`: a bit per chunk, zero when the chain starts): a chunk two lanes of the next round code at
        // the same time, from two different readings of its predecessor's end, would be left with bytes of both
        if (atomicOr(listed + (x >> 5), 1u << (x & 31u)) >> (x & 31u) & 1u) return;
        const uint32_t slot = atomicAdd(count_out, 1u);
        if (slot < n) list_out[slot] = x;
        else *need_map = 1u;                         // (cannot happen with the room checked above; the exhaustive route is always right)
    };
    for (uint32_t h = blockIdx.x; h < count; h += gridDim.x) {
        const uint32_t i = list_in[h];
        if (threadIdx.x == kFrontHead) {
            const uint32_t start = peek(sw + 2u * (i - 1u) + 1u);
            uint32_t what = start != peek(sw + 2u * i) ? 1u : 0u;
            if (what) {
                // a run of chunks whose starts are all about to move has ONE head, the first; the kFrontAhead chunks behind
                // it are in its look-ahead, and coding them here from an end that is about to move would only write ends
                // (and predictions) for the next rounds to take back.  Chunk j's end is about to move when the start its
                // end belongs to is not its predecessor's end.
                bool moving[kFrontAhead + 1u];
#pragma unroll
                for (uint32_t d = 1; d <= kFrontAhead + 1u; ++d)
                    moving[d - 1u] = i > d && (peek(sw + 2u * (i - d)) & 0xffu) != peek(sw + 2u * (i - d - 1u) + 1u);
                uint32_t d = 0;
                while (d <= kFrontAhead && moving[d]) ++d;                 // chunk i-1-d is the nearest one that stays
                if (d >= 1u && d <= kFrontAhead) what = 2u;               // the head is i-d, and i is in its look-ahead
            }
            s_head[0] = what;
            s_head[1] = start;
        }
        __syncthreads();
        const uint32_t what = s_head[0];
        if (what == 1u) {
            if (threadIdx.x == kFrontHead) {
                s_head[2] = (uint32_t)encode_chunk_alone(pcm + pcm_offs[i], nsamp[i], (int)s_head[1], blob + offs[i], s_tab);
            } else if (threadIdx.x < kFrontAhead * 89u) {
                const uint32_t k = threadIdx.x / 89u, st = threadIdx.x - k * 89u, x = i + 1u + k;
                if (x < n) {
                    const int16_t* p = pcm + pcm_offs[x];
                    const uint32_t m = nsamp[x] & ~1u;
                    EncodeState e = encode_state(m ? p[0] : 0, (int)st, s_tab);
                    encode_run<false>(p, m, e, nullptr, s_tab);
                    s_map[k][st] = (uint8_t)state_index(e);
                }
            }
        }
        __syncthreads();
        if (what == 2u && threadIdx.x == 0u) append(i);     // coded when its start is known: next round, from the head's prediction
        if (what == 1u && threadIdx.x == 0u) {
            uint32_t at = s_head[2];
            if (swap_state(state + i, s_head[1], at) != at) {
                for (uint32_t x = i + 1u, k = 0; x < n; ++x, ++k) {
                    // its predecessor's end moved: validated next round, whatever its state says now -- x may be a head
                    // itself, about to replace what is read here by what it coded from the end that has just moved
                    append(x);
                    if ((peek(sw + 2u * x) & 0xffu) == at) break;    // x's end is the end from `at` already
                    if (k == kFrontAhead) break;     // ... and past the maps, that round finds where it ends
                    const uint32_t pred = s_map[k][at];
                    if (swap_state(state + x, at | kPredicted, pred) == pred) break;   // it ends where it ended: nothing behind it moves
                    at = pred;
                }
            }
        }
        __syncthreads();                             // s_head / s_map are free again
    }
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
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t round = 0;; ++round) {
        const uint32_t count = peek(count_a);
        if (count == 0u) return;
        if (count > kSettleMost || round == kSettleRounds) {
            if (threadIdx.x == 0) *need_map = 1u;
            return;
        }
        for (uint32_t base = threadIdx.x & ~63u; base < count; base += 256u) {
            const bool listed = base + lane < count;
            sweep_one<false>(pcm, pcm_offs, nsamp, n, blob, offs, state, listed ? peek(list_a + base + lane) : 1u, listed, list_b, count_b, s_tab, nullptr);
        }
        __threadfence();
        __syncthreads();                            // everyone has read count_a and finished its appends
        if (threadIdx.x == 0) poke(count_a, 0u);
        __threadfence();
        __syncthreads();
        uint32_t* t = list_a; list_a = list_b; list_b = t;
        t = count_a; count_a = count_b; count_b = t;
    }
}

// The chain's own check, behind the last round: every chunk's bytes were coded from the start its state names (states are
// replaced whole), so the stream is the sequential encoder's if and only if every chunk's start is its predecessor's end.
// A chunk for which that does not hold -- none, unless the list handling above has a hole -- sends the stream down the
// exhaustive route instead of out of the door.  And the BYTES are looked at, not only the states (round 5): the step index
// in a chunk's header (adpcm.c:466, byte 2) is the start its nibbles were coded from by whichever lane wrote it last -- a chunk
// two lanes coded at once from different readings of its predecessor's end (what the front sweep's `listed` bits are there
// to prevent) can carry a header that is not its state's start, and a state still marked kPredicted has no bytes at all.
// (every chunk has a header, also one of no samples: encode_chunk writes the eight bytes whenever the lane has a chunk)
__global__ __launch_bounds__(256) void amv_adpcm_check_kernel(const uint2* __restrict__ state, uint32_t n, const uint8_t* __restrict__ blob,
                                                             const uint64_t* __restrict__ offs, uint32_t* __restrict__ need_map) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t start = state[i].x;
    bool bad = i ? start != state[i - 1u].y : start != 0u;
    bad = bad || (uint32_t)blob[offs[i] + 2u] != start;          // (start > 88 or kPredicted never equals a byte it wrote)
    if (bad) *need_map = 1u;
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
        map[(uint64_t)i * 96u + (uint32_t)(pair % 89u)] = (uint8_t)state_index(s);
    }
}

// Composition of the maps of kChainBlock consecutive chunks (bmap[b][s] = where start s ends up) and -- by the workgroup
// that finishes last -- the serial walk over those (n / 256 steps through LDS): the start index of every block.
// *done counts the workgroups that have written their map (zero when the kernel starts).
__global__ __launch_bounds__(128) void amv_adpcm_chain_kernel(const uint8_t* __restrict__ map, uint32_t n, uint8_t* __restrict__ bmap,
                                                              int32_t* __restrict__ bstart, uint32_t* __restrict__ done,
                                                              const uint32_t* __restrict__ need) {
    __shared__ uint32_t s_map[kChainBlock * 24u];
    __shared__ uint32_t s_last;
    if (need && *need == 0u) return;
    const uint32_t nb = gridDim.x;
    {
        const uint32_t c0 = blockIdx.x * kChainBlock, cnt = min(kChainBlock, n - c0);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(map + (uint64_t)c0 * 96u);
        for (uint32_t i = threadIdx.x; i < cnt * 24u; i += 128u) s_map[i] = src[i];
        __syncthreads();
        if (threadIdx.x < 89u) {
            const uint8_t* m8 = reinterpret_cast<const uint8_t*>(s_map);
            uint32_t v = threadIdx.x;
            for (uint32_t c = 0; c < cnt; ++c) v = m8[c * 96u + v];
            bmap[(uint64_t)blockIdx.x * 96u + threadIdx.x] = (uint8_t)v;
        }
    }
    __threadfence();                                // this workgroup's map is out before it counts itself
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(done, 1u) + 1u == nb ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    uint32_t v = 0;   // the encoder context starts zeroed
    for (uint32_t t0 = 0; t0 < nb; t0 += kChainBlock) {
        const uint32_t cnt = min(kChainBlock, nb - t0);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(bmap + (uint64_t)t0 * 96u);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cnt * 24u; i += 128u) s_map[i] = peek(src + i);   // (other workgroups' stores: past this CU's L1)
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

// Every chunk from the start index the maps give it: a workgroup walks from its block's start through the maps of the
// chunks before its own (one lane, <= 256 steps through LDS -- the space the cells and the staging tiles take afterwards).
__global__ __launch_bounds__(kEncodeBlock) void amv_adpcm_encode_mapped_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs, const uint32_t* __restrict__ nsamp, uint32_t n,
    const uint8_t* __restrict__ map, const int32_t* __restrict__ bstart, uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs,
    const uint32_t* __restrict__ need) {
    constexpr uint32_t kRaw = (uint32_t)(sizeof(EncodeLds) + sizeof(StageLds) * (kEncodeBlock / 64u));
    static_assert(kRaw >= kEncodeBlock * 96u && kChainBlock % kEncodeBlock == 0u, "a workgroup's worth of maps fits where the cells go");
    __shared__ __attribute__((aligned(16))) uint8_t s_raw[kRaw];
    __shared__ uint8_t s_start[kEncodeBlock];
    if (need && *need == 0u) return;
    const uint32_t c0 = blockIdx.x * kEncodeBlock, block = c0 / kChainBlock;
    uint32_t v = (uint32_t)bstart[block];            // (thread 0's copy is the one that walks)
    for (uint32_t t0 = block * kChainBlock; t0 <= c0; t0 += kEncodeBlock) {
        const uint32_t cnt = min(kEncodeBlock, n - t0);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(map + (uint64_t)t0 * 96u);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_raw);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cnt * 24u; i += kEncodeBlock) dst[i] = src[i];
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint32_t c = 0; c < cnt; ++c) {
                if (t0 == c0) s_start[c] = (uint8_t)v;
                v = s_raw[c * 96u + v];
            }
        }
    }
    __syncthreads();
    const uint32_t i = c0 + threadIdx.x;
    const bool live = i < n;
    const int start = live ? (int)s_start[threadIdx.x] : 0;
    __syncthreads();                                 // the maps have been read: their space becomes cells and staging tiles
    EncodeLds& s_tab = *reinterpret_cast<EncodeLds*>(s_raw);
    StageLds* s_stage = reinterpret_cast<StageLds*>(s_raw + sizeof(EncodeLds));
    load_encode_tables(s_tab);
    encode_chunk(live ? pcm + pcm_offs[i] : pcm, live ? nsamp[i] : 0u, start, live ? blob + offs[i] : blob, live, s_tab,
                 s_stage[threadIdx.x >> 6]);
}

__global__ __launch_bounds__(kEncodeBlock) void amv_adpcm_encode_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs,
    const uint32_t* __restrict__ nsamp, uint32_t n, const int32_t* __restrict__ step_in,
    uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs, const uint32_t* __restrict__ need) {
    __shared__ EncodeLds s_tab;
    __shared__ StageLds s_stage[kEncodeBlock / 64u];
    if (need && *need == 0u) return;
    load_encode_tables(s_tab);
    const uint32_t i = blockIdx.x * kEncodeBlock + threadIdx.x;
    const bool live = i < n;
    encode_chunk(live ? pcm + pcm_offs[i] : pcm, live ? nsamp[i] : 0u, live ? clip_index(step_in[i]) : 0, live ? blob + offs[i] : blob, live,
                 s_tab, s_stage[threadIdx.x >> 6]);
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
                      uint8_t* map, int32_t* bstart, uint32_t* done, const uint32_t* need, hipStream_t s) {
    // map: (n + nb) * 96 bytes, bstart: nb words, nb = adpcm_chain_blocks(n); *done == 0
    if (n == 0) return;
    const uint32_t nb = adpcm_chain_blocks(n);
    uint8_t* bmap = map + (uint64_t)n * 96u;
    const uint64_t groups = ((uint64_t)n * 89u + kEncodeBlock - 1u) / kEncodeBlock;
    hipLaunchKernelGGL(amv_adpcm_map_kernel, dim3((uint32_t)(groups < 4096u ? groups : 4096u)), dim3(kEncodeBlock), 0, s, pcm, pcm_offs,
                       nsamp, n, map, need);
    hipLaunchKernelGGL(amv_adpcm_chain_kernel, dim3(nb), dim3(128), 0, s, map, n, bmap, bstart, done, need);
}

void launch_adpcm_encode_mapped(const int16_t* pcm, const uint64_t* pcm_offs, const uint32_t* nsamp, uint32_t n, const uint8_t* map,
                                const int32_t* bstart, uint8_t* blob, const uint64_t* offs, const uint32_t* need, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(amv_adpcm_encode_mapped_kernel, dim3((n + kEncodeBlock - 1u) / kEncodeBlock), dim3(kEncodeBlock), 0, s, pcm,
                       pcm_offs, nsamp, n, map, bstart, blob, offs, need);
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
// counters (word 63 is the flag launch_adpcm_map / launch_adpcm_encode are given as `need`) and a bit per chunk for the
// front sweep; counters and bits are zeroed here.
static uint64_t chain_zeroed_bytes(uint32_t n) { return 256u + (((uint64_t)n + 31u) / 32u) * 4u; }
uint64_t adpcm_chain_workspace(uint32_t n) { return (uint64_t)n * 16u + chain_zeroed_bytes(n); }

const uint32_t* launch_adpcm_chain(const int16_t* pcm, const uint64_t* pcm_offs, const uint32_t* nsamp, uint32_t n, uint8_t* blob,
                                   const uint64_t* offs, void* work, uint32_t sweeps, bool settle, hipStream_t s) {
    uint2* state = static_cast<uint2*>(work);
    uint32_t* list[2] = {reinterpret_cast<uint32_t*>(state + n), reinterpret_cast<uint32_t*>(state + n) + n};
    uint32_t* count = list[1] + n;                   // [0 .. sweeps + 2]: one per list generation; [62]: launch_adpcm_map's `done`; [63]: the flag
    if (sweeps > 59u) sweeps = 59u;
    if (hipMemsetAsync(count, 0, chain_zeroed_bytes(n), s) != hipSuccess) return nullptr;   // (the caller reports it: nothing has been queued)
    hipLaunchKernelGGL(amv_adpcm_guess_kernel, dim3((n + kEncodeBlock - 1u) / kEncodeBlock), dim3(kEncodeBlock), 0, s, pcm, pcm_offs, nsamp, n,
                       blob, offs, state, list[0], count);
    if (n > 1u) {
        // sweep k's list is a fraction of the one before; the grid is sized for the first and strides if it must
        uint32_t grid = (n + kEncodeBlock - 1u) / kEncodeBlock;
        for (uint32_t k = 0; k < sweeps; ++k) {
            hipLaunchKernelGGL(amv_adpcm_sweep_kernel, dim3(grid), dim3(kEncodeBlock), 0, s, pcm, pcm_offs, nsamp, n, blob, offs, state,
                               list[k & 1u], count + k, list[(k + 1u) & 1u], count + k + 1u);
            grid = grid > 256u ? (grid + 1u) / 2u : grid;
        }
        // one sweep that looks four chunks ahead of every head (a plain sweep while the list is still long), then the rest
        if (settle) {
            hipLaunchKernelGGL(amv_adpcm_front_kernel, dim3(kFrontMost), dim3(kFrontThreads), 0, s, pcm, pcm_offs, nsamp, n, blob, offs,
                               state, list[sweeps & 1u], count + sweeps, list[(sweeps + 1u) & 1u], count + sweeps + 1u, count + 64,
                               count + 63);
            hipLaunchKernelGGL(amv_adpcm_settle_kernel, dim3(1), dim3(256), 0, s, pcm, pcm_offs, nsamp, n, blob, offs, state,
                               list[(sweeps + 1u) & 1u], count + sweeps + 1u, list[sweeps & 1u], count + sweeps + 2u, count + 63);
        }
        // (settle == false: a test knob -- the chain is left where its launched sweeps got it, and the check has to notice)
        hipLaunchKernelGGL(amv_adpcm_check_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, state, n, blob, offs, count + 63);
    }
    return count + 63;
}

}  // namespace amv
