// amv_decode.hip -- AMV video decode kernels for gfx950 (MI355X).
//
// What the reference computes (C-AMVDecoder/amvlib/AmvJpeg.c, cited per function below) is
// one serial loop per frame: Huffman-decode an MCU, dequantise, IDCT, convert, store.  Here
// the frame is cut at the only place the data allows:
//
//   amv_huffman_kernel      entropy stage.  Serial inside a frame (no restart markers, DC
//                           predictors chain through the whole scan: AmvJpeg.c:1200-1221,1406),
//                           so one LANE owns one frame and a wave walks 64 frames block by
//                           block.  Each lane decodes into a private 128-byte LDS slot; after
//                           every block the wave writes the 64 slots out as 64 whole 128-byte
//                           lines (8 dwordx4 stores per lane) and clears them, so HBM sees
//                           only full-line coalesced writes and no separate memset.
//   (amv_decode_sync.hip holds the faster entropy kernel with parallelism inside a frame, which
//   hands oversize or pathological chunks back to this one; amv_reconstruct.hip holds the stage
//   after it.)
//
// Compiled with -fwrapv: the integer pipeline relies on two's-complement wrap exactly as the
// reference's compiler output does.
#include "amv_kernels.h"

namespace amv {

// ============================================================================================
// entropy stage
// ============================================================================================

namespace {

constexpr int kWave = 64;

struct BitReader {
    const uint8_t* base;  // 4-byte aligned address at or before the chunk
    uint64_t guard;       // bytes readable from base
    uint32_t p;           // next byte, relative to base
    uint32_t end;         // one past the chunk's last byte, relative to base
    uint64_t acc;         // unread bits, MSB first
    int nbits;            // valid bits in acc
    int pad;              // zero bits appended past the end of the chunk
};

__device__ __forceinline__ uint32_t load_word(const BitReader& r, uint32_t p) {
    if (p >= r.end) return 0u;
    const uint64_t off = (uint64_t)(p & ~3u);
    if (off + 4 <= r.guard) return *reinterpret_cast<const uint32_t*>(r.base + off);
    uint32_t w = 0;
    for (int k = 0; k < 4; ++k)
        if (off + k < r.guard) w |= (uint32_t)r.base[off + k] << (8 * k);
    return w;
}

// ReadByte (AmvJpeg.c:1061-1071): every byte is data, and the byte after an FF is dropped
// without being looked at.  Four bytes at a time when the aligned word holds no FF.
__device__ __forceinline__ void refill(BitReader& r) {
    while (r.nbits <= 32) {
        const uint32_t p = r.p;
        const uint32_t w = load_word(r, p);
        const bool any_ff = ((~w - 0x01010101u) & w & 0x80808080u) != 0u;
        if ((p & 3u) == 0u && p + 4u <= r.end && !any_ff) {
            r.acc |= (uint64_t)__builtin_bswap32(w) << (32 - r.nbits);
            r.nbits += 32;
            r.p = p + 4u;
        } else {
            const bool in = p < r.end;
            const uint32_t b = in ? ((w >> ((p & 3u) * 8u)) & 0xffu) : 0u;
            r.pad += in ? 0 : 8;
            r.acc |= (uint64_t)b << (56 - r.nbits);
            r.nbits += 8;
            r.p = p + 1u + (b == 0xffu ? 1u : 0u);
        }
    }
}

// One (run, size, value) element: DecodeElement, AmvJpeg.c:842-936, as two table lookups on the
// next 16 bits instead of the reference's bit-by-bit walk.  Returns the symbol, or -1 when no
// code of <= 16 bits matches (FUNC_FORMAT_ERROR, :887).
__device__ __forceinline__ int decode_element(BitReader& r, const uint16_t* l1, const uint16_t* l2,
                                              int& value) {
    refill(r);
    const uint32_t v = (uint32_t)(r.acc >> 32);
    uint32_t e = l1[v >> (32 - kLut1Bits)];
    if (e & 0x8000u) e = l2[((e & 0xffu) << kLut2Bits) | ((v >> (32 - kLut1Bits - kLut2Bits)) & ((1u << kLut2Bits) - 1u))];
    const int len = (int)((e >> 8) & 31u);
    if (len == 0) {
        r.nbits -= 17;  // the reference has read 17 bits when it gives up
        value = 0;
        return -1;
    }
    const int sym = (int)(e & 0xffu);
    const int size = sym & 15;
    int val = 0;
    if (size) {
        const uint32_t mag = (v << len) >> (32 - size);
        val = (int)mag;
        if (mag < (1u << (size - 1))) val -= (1 << size) - 1;  // :924-933
    }
    r.acc <<= (len + size);
    r.nbits -= (len + size);
    value = val;
    return sym;
}

// byte offset of coefficient k inside a lane's 128-byte slot; 16-byte granules are XOR-ed with
// the lane so that lanes writing the same k land in different banks
__device__ __forceinline__ uint32_t slot_offset(uint32_t lane, uint32_t k) {
    return lane * 128u + ((((k >> 3) ^ lane) & 7u) << 4) + ((k & 7u) << 1);
}

}  // namespace

__global__ __launch_bounds__(kWave) void amv_huffman_kernel(
    const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offs,
    const uint32_t* __restrict__ lens, uint32_t n, uint32_t blocks_per_frame,
    const HuffDecodeImage* __restrict__ img, int16_t* __restrict__ coef,
    int32_t* __restrict__ status, uint32_t* __restrict__ nmcu_ok,
    const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_count, uint32_t base, uint32_t by_slot) {
    __shared__ __attribute__((aligned(16))) uint16_t s_l1[4 << kLut1Bits];
    __shared__ __attribute__((aligned(16))) uint16_t s_l2[kLut2Pages << kLut2Bits];
    __shared__ __attribute__((aligned(16))) uint4 s_slots[kWave * 8];
    __shared__ uint32_t s_frame[kWave];

    // Work items base .. : with a list (frames the synchronising kernel handed back) item p is frame list[p],
    // p < *list_count; without one, item p is frame p, p < n.  by_slot: the coefficient lines of item p go to slot
    // p - base of a workspace that holds one round of items (amvhip_api.hip), else to the frame's own place.
    const uint32_t lane = threadIdx.x;
    const uint32_t f0 = base + blockIdx.x * kWave;
    const bool ok_in_blocks = (by_slot & 2u) != 0u;   // AMVHIP_FLAG_FFMPEG_KEEP: nmcu_ok counts whole blocks (amv_kernels.h)
    by_slot &= 1u;
    if (list) n = *list_count;
    if (f0 >= n) return;
    const uint32_t frame = f0 + lane < n ? (list ? list[f0 + lane] : f0 + lane) : 0xffffffffu;
    s_frame[lane] = frame == 0xffffffffu ? frame : (by_slot ? f0 + lane - base : frame);

    {   // table image -> LDS, slots cleared
        const uint4* src = reinterpret_cast<const uint4*>(img);
        uint4* d1 = reinterpret_cast<uint4*>(s_l1);
        uint4* d2 = reinterpret_cast<uint4*>(s_l2);
        constexpr int n1 = (int)(sizeof(s_l1) / 16), n2 = (int)(sizeof(s_l2) / 16);
        for (int i = lane; i < n1; i += kWave) d1[i] = src[i];
        for (int i = lane; i < n2; i += kWave) d2[i] = src[n1 + i];
        for (int i = lane; i < kWave * 8; i += kWave) s_slots[i] = make_uint4(0, 0, 0, 0);
    }

    BitReader r;
    bool live = frame != 0xffffffffu;
    uint32_t st = 0, mcu_done = 0, blocks_done = 0;
    {
        uint64_t off = live ? offs[frame] : 0;
        uint32_t len = live ? lens[frame] : 0;
        if (off > blob_bytes) { off = blob_bytes; len = 0; }            // never read outside the blob
        if ((uint64_t)len > blob_bytes - off) len = (uint32_t)(blob_bytes - off);
        const uint32_t mis = (uint32_t)(off & 3u);
        r.base = blob + (off - mis);
        r.guard = blob_bytes - (off - mis);
        r.p = mis + 2u;  // skip FF D8 (AmvJpeg.c:1527)
        r.end = mis + len;
        r.acc = 0;
        r.nbits = 0;
        r.pad = 0;
    }
    int pred0 = 0, pred1 = 0, pred2 = 0;  // ycoef/ucoef/vcoef, AmvJpeg.c:1511
    char* slot_bytes = reinterpret_cast<char*>(s_slots);
    __syncthreads();

    uint32_t k6 = 0;  // block index inside the MCU
    for (uint32_t b = 0; b < blocks_per_frame; ++b) {
        if (live) {  // HufBlock, AmvJpeg.c:939-974, + DC prediction of DecodeMCUBlock :1200-1221
            const int cls = k6 < 4 ? 0 : 1;
            int val;
            int sym = decode_element(r, s_l1 + (cls << kLut1Bits), s_l2, val);
            if (sym < 0) {
                st |= kStFormat;
                live = false;
            } else {
                int dc;
                if (k6 < 4) { pred0 = (int16_t)(pred0 + val); dc = pred0; }
                else if (k6 == 4) { pred1 = (int16_t)(pred1 + val); dc = pred1; }
                else { pred2 = (int16_t)(pred2 + val); dc = pred2; }
                *reinterpret_cast<int16_t*>(slot_bytes + slot_offset(lane, 0)) = (int16_t)dc;
                const uint16_t* ac = s_l1 + ((2 + cls) << kLut1Bits);
                uint32_t k = 1;
                while (k < 64) {
                    sym = decode_element(r, ac, s_l2, val);
                    if (sym < 0) { st |= kStFormat; live = false; break; }
                    if (sym == 0) break;  // end of block, :959-964
                    k += (uint32_t)(sym >> 4);
                    if (k > 63) { st |= kStOverrun; live = false; break; }  // reference: out-of-bounds write
                    *reinterpret_cast<int16_t*>(slot_bytes + slot_offset(lane, k)) = (int16_t)val;
                    ++k;
                }
            }
            if (live) ++blocks_done;
        }
        __syncthreads();
        // 64 slots -> 64 lines of the coefficient array, then clear
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i) {
            const uint32_t c = i * kWave + lane;
            const uint32_t s = c >> 3, part = c & 7u;
            const uint32_t idx = s * 8u + (part ^ (s & 7u));
            const uint4 v = s_slots[idx];
            s_slots[idx] = make_uint4(0, 0, 0, 0);
            const uint32_t fr = s_frame[s];
            if (fr != 0xffffffffu) {
                uint4* dst = reinterpret_cast<uint4*>(coef + ((uint64_t)fr * blocks_per_frame + b) * 64u);
                dst[part] = v;
            }
        }
        __syncthreads();
        if (++k6 == 6) {
            k6 = 0;
            if (live) ++mcu_done;
        }
    }
    if (frame != 0xffffffffu) {
        if (r.pad > r.nbits) st |= kStTruncated;  // consumed bits that the chunk does not hold
        status[frame] = (int32_t)st;
        nmcu_ok[frame] = ok_in_blocks ? blocks_done : mcu_done;
    }
}

void launch_huffman(const uint8_t* blob, uint64_t blob_bytes, const uint64_t* offs,
                    const uint32_t* lens, uint32_t n, const FrameGeom& g,
                    const HuffDecodeImage* d_img, int16_t* coef, int32_t* status,
                    uint32_t* nmcu_ok, const uint32_t* list, const uint32_t* list_count, uint32_t base, uint32_t items,
                    bool by_slot, bool ok_in_blocks, hipStream_t s) {
    if (items == 0) return;
    const uint32_t grid = (items + kWave - 1) / kWave;   // an upper bound: groups past the end of the work exit at once
    hipLaunchKernelGGL(amv_huffman_kernel, dim3(grid), dim3(kWave), 0, s, blob, blob_bytes, offs,
                       lens, n, g.blocks, d_img, coef, status, nmcu_ok, list, list_count, base, (by_slot ? 1u : 0u) | (ok_in_blocks ? 2u : 0u));
}

}  // namespace amv
