// amv_kernels.h -- launch wrappers of the HIP kernels (internal to libamvhip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "amv_tables.h"

namespace amv {

// status bits, same values as AMVHIP_ST_* in include/amvhip.h
enum : uint32_t { kStFormat = 1u, kStOverrun = 2u, kStTruncated = 4u };
enum : uint32_t { kFlagZigzagFixed = 1u, kFlagFfmpeg = 2u, kFlagFfmpegKeep = 4u };

// ---- decode -------------------------------------------------------------------------------
// entropy stage: one lane per frame, coefficients staged per block in LDS and written out as
// whole 128-byte lines.  coef: [..][blocks][64] int16, scan order, DC already predicted.
// Work items base .. base + items: item p is frame list[p] (p < *list_count) or, without a list, frame p (p < n);
// its lines go to coef slot p - base (by_slot) or to the frame's own place.
void launch_huffman(const uint8_t* blob, uint64_t blob_bytes, const uint64_t* offs,
                    const uint32_t* lens, uint32_t n, const FrameGeom& g,
                    const HuffDecodeImage* d_img, int16_t* coef, int32_t* status,
                    uint32_t* nmcu_ok, const uint32_t* list, const uint32_t* list_count, uint32_t base, uint32_t items,
                    bool by_slot, bool ok_in_blocks, hipStream_t s);
// entropy stage with parallelism inside a frame (amv_decode_sync.hip): unstuff into a workspace
// (frame i in the 16-byte pieces ws_line[i] .. ws_line[i + 1] of ws, ws_bytes[i] = unstuffed length or ~0 when the frame is handed to
// launch_huffman through retry_list / *retry_count), then L lanes per frame synchronise and decode.
void launch_unstuff(const uint8_t* blob, uint64_t blob_bytes, const uint64_t* offs, const uint32_t* lens, uint32_t n,
                    const uint32_t* ws_line, uint32_t* ws, uint32_t* ws_bytes, uint32_t* retry_list, uint32_t* retry_count,
                    hipStream_t s);
// lanes per frame for a batch of n frames of `pixels` pixels on a device with `cus` compute units (amv_decode_sync.hip)
int huffman_sync_lanes(uint32_t n, uint32_t cus, int wanted, uint64_t pixels, bool records);
// the statistics buffer of the synchronising entropy kernel (amvhip_entropy_stats / _trace): 16 words of counters, then
// one line of eight 64-bit words per task (wave) for the first kTraceTasks tasks of a launch
constexpr uint32_t kTraceBase = 16, kTraceTasks = 16384;
constexpr size_t kStatsBytes = (size_t)(kTraceBase + 8u * kTraceTasks) * 8u;
// Where the entropy stage puts its result.  rec == nullptr: dense coefficient lines in coef
// ([n][blocks][64] int16).  Otherwise the records form, per frame: its lines of rec (one word per DC coefficient and
// per non-zero AC coefficient, stream order: bits 0-5 index in block (0 = DC), 6-11 (block - blocks per frame) modulo 64, bit 15 filler,
// 16-31 value; a DC value counts from the decoding lane's first block), seg_start[segs + 1][2] = {from, to} for each
// MCU-row segment of kSegMcus MCUs (what one wave of the reconstruction takes): the segment's first record is at or
// after `from` with only records of the <= 4 blocks before it in between, and the records of the segment before end
// ahead of `to`, with only records of this segment's first <= 4 blocks in between (the exact position twice, or the
// record counts around the eight symbols in which the segment began), lane_tab[lanes] = {first block,
// DC base Y, Cb, Cr} of each of the lanes that decoded the frame (a row of `lanes` entries per frame; how many of them a
// frame used is in its rec_count), rec_count (bits 0-23 total, bits 24-29 lanes that decoded the frame - 1; or ~0 = this
// frame is in dense form in coef because it went through amv_huffman_kernel).
struct SyncSinks {
    int16_t* coef;
    uint32_t* rec;
    const uint32_t* rec_line;   // [n + 1]: frame i's records live in 128-byte lines rec_line[i] .. rec_line[i + 1] of rec (launch_layout)
    uint32_t* seg_start;
    uint32_t* lane_tab;
    uint32_t lanes;
    uint32_t* rec_count;
    uint32_t* retry_list;
    uint32_t* retry_count;
    // AMVHIP_FLAG_FFMPEG_KEEP: nmcu_ok[] counts whole BLOCKS decoded before a frame's first error, not whole MCUs (the
    // entropy kernels write it so, the reconstruction reads it so) -- an internal array then, never a caller's
    uint32_t ok_in_blocks;
};
// Space per frame from the frame's own chunk length, for the unstuffed scans (a) and, when b.line != nullptr, for the record
// hand-over (b) alike: min(hi, per_byte_x2 / 2 * lens[i] + add) units, rounded up to whole lines of 1 << unit_shift units;
// line[0 .. n] = where each frame's lines begin (an exclusive scan), never past cap_lines (a frame that gets less than it
// needs is decoded by the serial kernel).  work: layout_workspace(n) bytes.
struct LayoutSpec {
    uint32_t per_byte_x2, add, hi, unit_shift, cap_lines;
    uint32_t* line;
};
uint64_t layout_workspace(uint32_t n);
void launch_layout(const uint32_t* lens, uint32_t n, const LayoutSpec& a, const LayoutSpec& b, void* work, uint32_t* zero, uint32_t nzero,
                   bool force_large, hipStream_t s);
// Frames whose chunk is more than twice the batch's mean chunk (by the pieces the layout gave their scans: ws_line) ->
// heavy[0 .. count[0]), the others -> light[0 .. count[1]); count[0..1] zeroed by the caller.  A chip-filling batch decodes
// the light ones one lane per frame and the heavy ones with several lanes each (amvhip_api.hip: entropy_front).
void launch_split_by_weight(const uint32_t* lens, uint32_t n, const uint32_t* ws_line, uint32_t* heavy, uint32_t* light, uint32_t* count,
                            hipStream_t s);
// list/list_count: optional frame list; *queue: a zeroed task counter per launch
void launch_huffman_sync(const uint32_t* ws, const uint32_t* ws_bytes, uint32_t n, const uint32_t* list,
                         const uint32_t* list_count, const FrameGeom& g, const uint32_t* ws_line, int lanes_per_frame,
                         const HuffDecodeImage* d_img, const SyncSinks& sinks, int32_t* status, uint32_t* nmcu_ok,
                         uint32_t* queue, unsigned long long* stats, uint32_t cus, hipStream_t s);
// Which frames a reconstruction launch takes.  Default: frame = blockIdx.x for all n frames, dense lines (if any) at
// the frame's own place; frames of a records launch that went to the serial kernel (rec_count ~0) are skipped --
// a round launch picks them up.  Round: work items base .. base + round, item p = frame list[p] (p < *count) or
// frame p (p < n), dense lines in slot p - base; the launch is small (its workgroups walk the items) because the
// round usually has nothing to do and the host cannot know.
struct FrameSel {
    const uint32_t* list;
    const uint32_t* count;
    uint32_t base;
    uint32_t round;   // items of this round; 0 = default launch
};
// dequantise + IDCT + YCbCr->BGR + flipped store: one wave per MCU-row segment
// sinks.rec == nullptr: every frame is dense in sinks.coef; otherwise per frame as rec_count says
void launch_reconstruct(const SyncSinks& sinks, const uint32_t* nmcu_ok, uint32_t n, const FrameSel& sel, uint32_t items,
                        const FrameGeom& g, uint32_t flags, uint8_t* out, hipStream_t s);

// FFmpeg-compat back half (amv_reconstruct_ff.hip): Q60 dequantisation, simple_idct_put, YUVJ420P planes
// (Y, Cb, Cr; tight rows) flipped as mjpegdec.c:672-677 does.  yuv_store_covers_planes: false when that formula
// leaves plane rows unwritten (the caller clears the output first).
void launch_reconstruct_yuv(const SyncSinks& sinks, const uint32_t* nmcu_ok, uint32_t n, const FrameSel& sel, uint32_t items,
                            const FrameGeom& g, uint64_t yuv_frame_bytes, uint8_t* out, hipStream_t s);
bool yuv_store_covers_planes(const FrameGeom& g);

// ---- encode -------------------------------------------------------------------------------
// planar YUVJ420P source (what the reference's amv_encoder takes, mjpegenc.c:493): frame i's planes at
// y + i*y_frame, cb/cr + i*c_frame (bytes); rows y_stride / c_stride apart.  c_rows422: the chroma planes have a row
// per luma row (YUVJ422P, the other entry of pix_fmts); the two rows over a 4:2:0 sample are averaged, rounding up.
struct YuvSource {
    const uint8_t* y;
    const uint8_t* cb;
    const uint8_t* cr;
    uint32_t y_stride, c_stride;
    uint64_t y_frame, c_frame;
    uint32_t c_rows422;
};
// The whole encoder, one workgroup per frame (amv_encode_par.hip): pixels (RGB24/BGR24, or *yuv when not null) ->
// FF D8 + escaped scan + FF D9 in tmp[i*bound..], lens[i].  Frames whose bits do not fit the kernel's LDS window are
// appended to retry_list / *retry_count for launch_forward + launch_pack (rounds).
void launch_encode_frames(const uint8_t* pix, uint32_t pix_stride, int is_bgr, const YuvSource* yuv, uint32_t n, const FrameGeom& g,
                          uint32_t qbias, const HuffEncodeImage* d_img, uint8_t* tmp, uint32_t bound, uint32_t* lens,
                          uint32_t* retry_list, uint32_t* retry_count, hipStream_t s);
// The two-stage route (amv_encode.hip).  colour conversion + level shift + forward DCT + quantise -> dense lines
// coef [..][blocks][64] int16 scan order; entropy coder, one lane per frame -> tmp[frame*bound..], lens[frame].
// sel: default (round == 0) = frames 0 .. n with lines at the frame's own place; round = items base .. base + round,
// item p = frame list[p] (p < *count), lines in slot p - base.  items: upper bound of the work (grid size).
void launch_forward(const uint8_t* pix, uint32_t pix_stride, int is_bgr, uint32_t n, const FrameSel& sel, uint32_t items,
                    const FrameGeom& g, uint32_t qbias, int16_t* coef, hipStream_t s);
void launch_forward_yuv(const YuvSource& src, uint32_t n, const FrameSel& sel, uint32_t items, const FrameGeom& g, uint32_t qbias,
                        int16_t* coef, hipStream_t s);
void launch_pack(const int16_t* coef, uint32_t n, const FrameSel& sel, uint32_t items, const FrameGeom& g,
                 const HuffEncodeImage* d_img, uint8_t* tmp, uint32_t bound, uint32_t* lens, hipStream_t s);
// exclusive scan of lens -> offs (single workgroup), then gather tmp -> blob.  A chunk that would end past
// blob_cap is not copied and its lens entry becomes 0; *overflow = 1 when the total exceeds blob_cap.
void launch_compact(const uint8_t* tmp, uint32_t bound, uint32_t* lens, uint32_t n,
                    uint64_t* offs, uint8_t* blob, uint64_t blob_cap, int32_t* overflow, hipStream_t s);

// ---- picture rescale (amv_resample.hip): img_resample of libavcodec/imgresample.c ------------------------
struct ResamplePlanes {     // YUV420P frames: frame i's planes at y + i*y_frame, cb/cr + i*c_frame; chroma (w>>1) x (h>>1)
    uint8_t* y;
    uint8_t* cb;
    uint8_t* cr;
    uint32_t y_stride, c_stride;
    uint64_t y_frame, c_frame;
    uint32_t width, height;
};
struct ResampleFilters {    // img_resample_full_init (:425-472): 16 phases x 4 taps each way, 16.16 increments
    int16_t h[64], v[64];
    int h_incr, v_incr;
};
void launch_resample(const ResamplePlanes& src, const ResamplePlanes& dst, const ResampleFilters& f, uint32_t n, hipStream_t s);

// ---- ADPCM --------------------------------------------------------------------------------
void launch_adpcm_decode(const uint8_t* blob, uint64_t blob_bytes, const uint64_t* offs,
                         const uint32_t* lens, uint32_t n, int16_t* pcm, const uint64_t* pcm_offs,
                         int32_t* final_state, hipStream_t s);
// The reference's step_index carry (adpcm.c:461-498) without a serial pass over the stream.  launch_adpcm_chain: every
// chunk coded from a guessed start, then `sweeps` launches + one settling workgroup that code again what started wrong;
// returns the device word that is 1 when the stream did not settle (the word before it is a zeroed counter for
// launch_adpcm_map's `done`).  launch_adpcm_map (89-way state map of every chunk + composition of the maps -> the start of
// every block of 256 chunks) and launch_adpcm_encode_mapped (every chunk coded from the start the maps give it) leave at once
// when *need == 0 (need == nullptr: run).
uint64_t adpcm_chain_workspace(uint32_t n);
const uint32_t* launch_adpcm_chain(const int16_t* pcm, const uint64_t* pcm_offs, const uint32_t* nsamp, uint32_t n, uint8_t* blob,
                                   const uint64_t* offs, void* work, uint32_t sweeps, bool settle, hipStream_t s);
void adpcm_quotient_table(float out[89]);   // the encoder's quotient factors (see amv_adpcm.hip: compress)
uint32_t adpcm_chain_blocks(uint32_t n);   // launch_adpcm_map needs map[(n + blocks) * 96] and bstart[blocks]
void launch_adpcm_map(const int16_t* pcm, const uint64_t* pcm_offs, const uint32_t* nsamp, uint32_t n,
                      uint8_t* map, int32_t* bstart, uint32_t* done, const uint32_t* need, hipStream_t s);
void launch_adpcm_encode_mapped(const int16_t* pcm, const uint64_t* pcm_offs, const uint32_t* nsamp, uint32_t n, const uint8_t* map,
                                const int32_t* bstart, uint8_t* blob, const uint64_t* offs, const uint32_t* need, hipStream_t s);
// amvlib's IMA-WAV-layout frame encoder (AdpcmIma.c:43-160), one lane
void launch_adpcm_wav_encode(const int16_t* samples, int groups, int32_t* state, uint8_t* frame, hipStream_t s);
void launch_adpcm_encode(const int16_t* pcm, const uint64_t* pcm_offs, const uint32_t* nsamp,
                         uint32_t n, const int32_t* step_in, uint8_t* blob, const uint64_t* offs, const uint32_t* need,
                         hipStream_t s);

// the reference's trellis search (adpcm.c:287-443), one lane per independent chunk; paths: adpcm_trellis_workspace bytes.
// false: the device refused the kernel's LDS size
uint64_t adpcm_trellis_workspace(uint32_t n, uint32_t trellis);
bool launch_adpcm_trellis(const int16_t* pcm, const uint64_t* pcm_offs, const uint32_t* nsamp, uint32_t n, const int32_t* step_in,
                          uint32_t trellis, uint8_t* blob, const uint64_t* offs, int32_t* step_out, uint16_t* paths, hipStream_t s);

// ---- synthetic sources ----------------------------------------------------------------------
void launch_synth_frames(uint32_t seed, uint32_t first, uint32_t n, uint32_t w, uint32_t h,
                         uint8_t* rgb, hipStream_t s);
void launch_synth_audio(uint32_t seed, uint64_t first, uint64_t n, int16_t* pcm, hipStream_t s);

}  // namespace amv
