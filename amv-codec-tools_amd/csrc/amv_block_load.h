// amv_block_load.h -- stage A of the reconstruction kernels: one 8x8 block of quantised coefficients per lane.
//
// A wave owns an MCU-row segment of `cnt` MCUs (<= 10, i.e. <= 60 blocks).  Lane b ends up with block b's 64
// coefficients (scan order, int16 pairs in 32 dwords): from its dense 128-byte line, or -- records form -- after
// the wave has scattered the segment's (block, index, value) records into a zeroed LDS image of the blocks (16-byte
// granules XOR-swizzled by block so that the per-lane 128-byte reads do not collide on banks) and added the DC base
// of the entropy lane that decoded the block (SyncSinks::lane_tab).
#pragma once
#include <cstdlib>

#include "amv_kernels.h"
#include "amv_piece_map.h"

namespace amv {

constexpr uint32_t kSegImageBytes = 60u * 128u;   // the LDS image of a segment's <= 60 blocks; 128 spare bytes follow it
constexpr uint32_t kDummyRecordWord = 0x8000u;   // bit 15: a filler no block owns (amv_decode_sync.hip's kDummyRecord)

// A segment belongs to one wave and so does its LDS: what one lane wrote another lane of the same wave may read once
// the wave has passed this point (a wave's LDS operations are served in order; the fence keeps the compiler from
// moving them across).  No workgroup barrier: the waves of a workgroup work on segments of their own.
__device__ __forceinline__ void seg_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// s_img: kSegImageBytes + 128 bytes of LDS, 16-byte aligned; the caller may reuse it after a seg_sync().
// segidx: this segment's number in the frame (mcu_row * segments_per_row + segment).
// Returns true when this lane holds a block (lane < cnt * 6).
// Which frame this workgroup works on: work item `item` of the launch (FrameSel); false when there is none (past
// the end of a round's work).  slot: where the frame's dense lines live, if it has any.  A default launch has one
// item per workgroup (blockIdx.x); a round launch is small and its workgroups walk the round's items.
__device__ __forceinline__ bool select_frame(const FrameSel& sel, uint32_t n, uint32_t item, uint32_t& f, uint32_t& slot) {
    f = slot = item;
    if (!sel.round) return true;
    const uint32_t p = sel.base + item;
    if (item >= sel.round || p >= (sel.count ? *sel.count : n)) return false;
    f = sel.list ? sel.list[p] : p;
    return true;
}

// dense_only: a round launch -- the frame's lines are in slot `slot` whatever rec_count says.  A default launch over
// records leaves frames that went to the serial kernel alone: skip = true (for the whole workgroup), nothing loaded.
__device__ __forceinline__ bool load_segment_blocks(const SyncSinks& in, uint32_t f, uint32_t slot, bool dense_only,
                                                    const FrameGeom& g, uint32_t segidx,
                                                    uint32_t nsegs, uint32_t mcu0, uint32_t cnt, uint32_t ok, uint32_t lane,
                                                    uint8_t* s_img, uint32_t (&c)[32], bool& skip) {
    constexpr uint32_t kWave = 64;
    const uint32_t nb = cnt * 6u;
    // everything the wave must know before it can ask for its records is requested at once (one round trip to
    // memory, not three in a row): form of the frame, the segment's record range
    uint32_t rc = 0xffffffffu, r0 = 0u, r1 = 0u;
    if (in.rec != nullptr && !dense_only) {
        // a segment's records lie in [from of its own entry, to of the next one) -- bounds, not exact positions: the range
        // may begin with records of the (<= 4) blocks before the segment and end with records of the (<= 4) blocks behind
        // it, which the block test below drops
        const uint32_t* ss = in.seg_start + ((uint64_t)f * (nsegs + 1u) + segidx) * 2u;
        rc = in.rec_count[f];
        r0 = ss[0];
        r1 = ss[3];
    }
    const bool records = rc != 0xffffffffu;
    skip = !records && in.rec != nullptr && !dense_only;   // a round launch reconstructs this frame
    if (skip) return false;
    int dc_base = 0;
    if (records) {   // records -> dense image of the segment's blocks in LDS
        // blocks of this segment that were decoded: whole MCUs' (ok counts MCUs), or -- AMVHIP_FLAG_FFMPEG_KEEP -- every
        // whole block before the frame's first error (ok counts blocks then: SyncSinks::ok_in_blocks)
        const uint32_t ok_blocks = in.ok_in_blocks ? ok : ok * 6u;
        const uint32_t blocks_ok_here = ok_blocks > mcu0 * 6u ? min(nb, ok_blocks - mcu0 * 6u) : 0u;
        if (!blocks_ok_here) r1 = r0;
        // The segment's records are asked for all at once, four consecutive ones per lane and instruction (a loop of
        // one 4-byte load per lane and trip waited for memory eight times in a row: 1.6 of the kernel's 4.4 ms); the
        // image is zeroed while they are on their way.
        struct __attribute__((packed, aligned(4))) Rec4 { uint32_t w[4]; };
        constexpr uint32_t kAhead = 3;                                   // 768 records per trip; a segment has ~500
        const uint32_t* rec = in.rec + (uint64_t)in.rec_line[f] * 32u + r0;
        const uint32_t nrec = r1 - r0;
        Rec4 q[kAhead];
#pragma unroll
        for (uint32_t j = 0; j < kAhead; ++j) {
            const uint32_t i = j * 4u * kWave + lane * 4u;
            q[j] = Rec4{{kDummyRecordWord, kDummyRecordWord, kDummyRecordWord, kDummyRecordWord}};
            if (i < nrec) q[j] = *reinterpret_cast<const Rec4*>(rec + i);   // (may read up to 3 words past r1: see ensure())
        }
        uint4* img16 = reinterpret_cast<uint4*>(s_img);
        for (uint32_t i = lane; i < nb * 8u; i += kWave) img16[i] = make_uint4(0, 0, 0, 0);
        seg_sync();
        // Scatter, without a branch per record: the word shifted left by one holds 2 * index in bits 1-6, the block field
        // in bits 7-12 and the filler flag in bit 16.  Adding (64 - first block) << 7 turns the field into the block's
        // number in the segment (modulo 64, carry into bit 13); with bit 16 kept, one unsigned compare against
        // "blocks decoded << 7" rejects fillers and other segments' blocks alike, and a record past the range's end
        // compares against 0.  A rejected record goes to the lane's spare slot behind the image.
        const uint32_t first7 = (64u - ((mcu0 * 6u - g.blocks) & 63u)) << 7;   // the block field counts from the frame's end
        const uint32_t ok7 = blocks_ok_here << 7;
        const uint32_t spare = kSegImageBytes + lane * 2u;
        for (uint32_t base = 0;;) {
#pragma unroll
            for (uint32_t j = 0; j < kAhead; ++j) {
                if (base + j * 4u * kWave >= nrec) break;                   // (wave-uniform) nothing of this piece is in range
                const int32_t left = (int32_t)(nrec - base - j * 4u * kWave) - (int32_t)(lane * 4u);   // records from this lane's first on
#pragma unroll
                for (uint32_t e = 0; e < 4u; ++e) {
                    const uint32_t u = q[j].w[e] << 1;
                    const uint32_t t = u + first7;
                    const uint32_t b7 = t & 0x11f80u;                                    // block in segment << 7, filler flag
                    const uint32_t at = b7 | ((u & 0x7eu) ^ ((t >> 3) & 0x70u));         // 16-byte granule XOR block, as the reader expects
                    const bool take = b7 < (left > (int32_t)e ? ok7 : 0u);
                    *reinterpret_cast<int16_t*>(s_img + (take ? at : spare)) = (int16_t)(q[j].w[e] >> 16);
                }
            }
            base += kAhead * 4u * kWave;
            if (base >= nrec) break;
#pragma unroll
            for (uint32_t j = 0; j < kAhead; ++j) {   // a segment with more records than one trip holds
                const uint32_t i = base + j * 4u * kWave + lane * 4u;
                if (i < nrec) q[j] = *reinterpret_cast<const Rec4*>(rec + i);
            }
        }
        // a DC record counts from its lane's first block: the base of the lane that decoded this block's DC
        // (how many lanes decoded THIS frame is in its rec_count: a chip-filling batch gives its heavy frames several and
        // the others one, and a one-lane frame has no row in the table)
        const uint32_t frame_lanes = ((rc >> 24) & 63u) + 1u;
        if (frame_lanes > 1u) {
            // the entropy lane that decoded this block's DC symbol: the last one whose first block is not behind it (first blocks
            // ascend along the row; lanes right of the frame's end hold ~0).  A binary search by lane shuffles: as a loop of
            // readlanes over the row it cost a heavy frame's segment 13 000 cycles with 16 lanes to the frame (0.4 ms per
            // 10 000 such frames).
            const uint32_t babs = mcu0 * 6u + lane, k6 = lane % 6u;
            const uint4 ent = lane < frame_lanes ? reinterpret_cast<const uint4*>(in.lane_tab)[(uint64_t)f * in.lanes + lane]
                                                 : make_uint4(0xffffffffu, 0u, 0u, 0u);
            uint32_t at = 0u;                           // lane 0 starts the frame: base 0
#pragma unroll
            for (uint32_t step = 32u; step >= 1u; step >>= 1) {
                const uint32_t probe = at + step;
                const uint32_t first = (uint32_t)__shfl((int)ent.x, (int)(probe & 63u));
                if (probe < frame_lanes && first <= babs) at = probe;
            }
            const int by = __shfl((int)ent.y, (int)at), bu = __shfl((int)ent.z, (int)at), bv = __shfl((int)ent.w, (int)at);
            dc_base = at ? (k6 < 4u ? by : (k6 == 4u ? bu : bv)) : 0;
        }
        seg_sync();
    }
    if (lane >= nb) return false;
    if (records) {
        const uint4* src = reinterpret_cast<const uint4*>(s_img) + lane * 8u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 q = src[(uint32_t)i ^ (lane & 7u)];
            c[4 * i] = q.x; c[4 * i + 1] = q.y; c[4 * i + 2] = q.z; c[4 * i + 3] = q.w;
        }
        c[0] = (c[0] & 0xffff0000u) | ((c[0] + (uint32_t)dc_base) & 0xffffu);   // int16 arithmetic, as the predictors wrap
    } else {
        const uint4* src = reinterpret_cast<const uint4*>(in.coef + (((uint64_t)slot * g.mcus + mcu0) * 6u + lane) * 64u);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 q = src[i];
            c[4 * i] = q.x; c[4 * i + 1] = q.y; c[4 * i + 2] = q.z; c[4 * i + 3] = q.w;
        }
    }
    return true;
}

// int16 number `i` of a block held as 32 dwords
__device__ __forceinline__ int coef_at(const uint32_t (&c)[32], int i) {
    return (i & 1) ? ((int)c[i >> 1] >> 16) : (int)(int16_t)(c[i >> 1] & 0xffffu);
}

}  // namespace amv
