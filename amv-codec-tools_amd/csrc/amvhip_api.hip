// amvhip_api.hip -- host side of libamvhip.so: the batch C ABI of include/amvhip.h.
//
// Owns the context (device tables, grow-only workspace, event-based kernel timing) and turns
// each C entry point into kernel launches on the caller's stream.  Nothing here computes codec
// results on the CPU: if the device is missing the calls fail.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/amvhip.h"
#include "amv_kernels.h"

using namespace amv;

namespace {

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct ProfRec {
    int kernel;
    hipEvent_t a, b;
};

}  // namespace

struct amvhip_ctx {
    int device = 0;
    std::string err;
    HuffDecodeImage* d_dec = nullptr;
    HuffEncodeImage* d_enc = nullptr;
    // workspace
    DevBuf coef, status, nmcu, tmp, lens, offs, flag, map, start, retry, enc_retry, stats, ws, ws_line, layout, ws_bytes, rec, rec_line, seg_start, lane_tab, rec_count, scaled, trellis_ws, chain, split;
    // amvhip_decode_submit_dev / _collect_dev: what the entropy stage hands to the reconstruction exists twice, so that
    // the entropy stage of one batch can run (stream `front`) beside the reconstruction of the batch before (`back`)
    struct DecodeSet { DevBuf nmcu, retry, rec, rec_line, seg_start, lane_tab, rec_count; } second;
    hipStream_t front = nullptr, back = nullptr;
    hipEvent_t ev_in = nullptr, ev_front = nullptr, ev_done[2] = {nullptr, nullptr};
    uint64_t submitted = 0, collected = 0;
    DevBuf* last_decode_retry = nullptr;   // whose first word counts the frames the LAST decode call handed to the serial kernel
    int sync_lanes = 0;   // AMVHIP_SYNC_LANES: 8/16/32/64 lanes per frame; 0 = by batch size (huffman_sync_lanes)
    bool split_heavy = true;   // AMVHIP_SPLIT=0: a one-lane-per-frame batch keeps its heavy frames on one lane too
    uint32_t heavy_lanes = 16; // AMVHIP_SPLIT=n: lanes a heavy frame gets (1, 2, 4 ... 64)
    bool last_split = false;   // the last decode call made the two lists (amvhip_decode_split_stats)
    uint32_t cus = 256;   // compute units of the device
    bool want_stats = false;
    bool layout_large = false;   // AMVHIP_LAYOUT=large: the three-launch layout whatever the batch size (test knob)
    double ws_bytes_per_frame = 0.0;
    int entropy_mode = AMVHIP_ENTROPY_AUTO;
    int adpcm_sweeps = 0;            // AMVHIP_ADPCM_SWEEPS: sweeps of the guessed-start route (-1: exhaustive route only)
    bool adpcm_sweeps_set = false;   // false: by stream length
    bool adpcm_settle = true;        // "nosettle": the chain stops after its launched sweeps (test knob: its check must notice)
    uint32_t chain_n = 0;            // chunks of the last chained ADPCM encode (where its counters are in `chain`)
    // host-pointer staging (one in-order stream of the context's own carries every host-buffer entry point)
    DevBuf h_in, h_offs, h_lens, h_out, h_status, h_aux, a_in, a_tab, a_out;
    hipStream_t hstream = nullptr;
    // amvhip_decode_batch_async: decoded frames go back to the host on a stream of their own, out of two staging buffers used
    // in turn, so that the copy of one call runs beside the upload and the kernels of the next (a window of the amvlib reader
    // is 59 MB of frames back for 3.6 MB of chunks in)
    hipStream_t dstream = nullptr;
    DevBuf v_out[2], v_status[2];
    hipEvent_t ev_decoded = nullptr, ev_copied[2] = {nullptr, nullptr};
    uint64_t async_calls = 0;
    // timing
    bool prof = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    uint64_t launches[AMVHIP_K_COUNT] = {};
    double total_ms[AMVHIP_K_COUNT] = {};
    std::mutex mu;    // the kernels' workspace: one _dev call enqueues at a time
    std::mutex hmu;   // the host-buffer staging (h_*, a_*): one host-buffer call at a time; taken before mu, never after
};

namespace {

int fail(amvhip_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((ctx), e_ == hipErrorOutOfMemory ? AMVHIP_ERR_NOMEM : AMVHIP_ERR_DEVICE, \
                        "%s: %s", #expr, hipGetErrorString(e_));                               \
    } while (0)

int ensure(amvhip_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return AMVHIP_OK;
    if (c->front) {   // a submitted batch may still be using the buffer
        HIP_TRY(c, hipStreamSynchronize(c->front));
        HIP_TRY(c, hipStreamSynchronize(c->back));
    }
    if (b.p) HIP_TRY(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(c, hipMalloc(&b.p, want));
    b.cap = want;
    return AMVHIP_OK;
}

// ---- table images ---------------------------------------------------------------------------

const uint8_t* symbols_of(int t) {
    return t < 2 ? kHuffDcSymbols : (t == 2 ? kHuffAcLumaSymbols : kHuffAcChromaSymbols);
}

// canonical code assignment of JPEG Annex C (what AmvJpeg.c:1454-1481 and mjpeg.c:129-147 both
// derive): codes of each length are consecutive, and the first code of length l+1 is
// (last code of length l + 1) << 1
void build_images(HuffDecodeImage& dec, HuffEncodeImage& enc) {
    memset(&dec, 0, sizeof dec);
    memset(&enc, 0, sizeof enc);
    int pages = 0;
    for (int t = 0; t < 4; ++t) {
        const uint8_t* syms = symbols_of(t);
        uint32_t code = 0;
        int k = 0;
        for (int len = 1; len <= 16; ++len) {
            for (int j = 0; j < kHuffCount[t][len - 1]; ++j, ++code) {
                const uint32_t sym = syms[k++];
                enc.code[t][sym] = code | ((uint32_t)len << 16);
                const uint16_t entry = (uint16_t)(sym | ((uint32_t)len << 8));
                if (len <= kLut1Bits) {
                    const uint32_t lo = code << (kLut1Bits - len);
                    for (uint32_t x = 0; x < (1u << (kLut1Bits - len)); ++x) dec.l1[t][lo + x] = entry;
                } else {
                    const int rest = len - kLut1Bits;  // 1..7
                    const uint32_t prefix = code >> rest;
                    if (!(dec.l1[t][prefix] & 0x8000u)) dec.l1[t][prefix] = (uint16_t)(0x8000u | (uint32_t)pages++);
                    const uint32_t page = dec.l1[t][prefix] & 0xffu;
                    const uint32_t lo = (code & ((1u << rest) - 1u)) << (kLut2Bits - rest);
                    for (uint32_t x = 0; x < (1u << (kLut2Bits - rest)); ++x) dec.l2[page][lo + x] = entry;
                }
            }
            code <<= 1;
        }
    }
    if (pages > kLut2Pages) abort();  // static property of the K.3 tables (11 pages)
    // the synchronising kernel's form of the same tables
    auto merged = [](uint16_t e, int t) -> uint16_t {
        const uint32_t len = (e >> 8) & 31u, sym = e & 0xffu, size = sym & 15u;
        if (len == 0) return 0;
        const uint32_t adv = t < 2 ? 1u : (sym == 0 ? 63u : (sym >> 4) + 1u);
        return (uint16_t)((len + size) | (adv << 5) | (size << 11));
    };
    const int first = (1 << kLut1Bits) - kLut2PagesPerTable;
    for (int t = 0; t < 4; ++t) {
        for (int i = 0; i < (1 << kLut1Bits); ++i) {
            const uint16_t e = dec.l1[t][i];
            if (!(e & 0x8000u)) { dec.m1[t][i] = merged(e, t); continue; }
            if (i < first) abort();   // static property of the K.3 tables: long codes live in the last 5 prefixes
            dec.m1[t][i] = 0x8000u;
        }
        // m2: the 16-bit window 111111 xxxxxxxxxx -> entry of the code it starts with (0 where none does)
        for (int x = 0; x < (1 << kM2Bits); ++x) {
            const uint32_t w16 = 0xfc00u | (uint32_t)x;
            const uint16_t e1 = dec.l1[t][w16 >> (16 - kLut1Bits)];
            if (e1 & 0x8000u) dec.m2[t][x] = merged(dec.l2[e1 & 0xffu][(w16 >> (16 - kLut1Bits - kLut2Bits)) & ((1u << kLut2Bits) - 1u)], t);
        }
    }
    // the one-lane-per-frame walk's form (amv_tables.h)
    auto fast = [](uint16_t e, int t) -> uint32_t {
        const uint32_t len = (e >> 8) & 31u, sym = e & 0xffu, size = sym & 15u;
        if (len == 0) return kFastInvalid | (1u << 24);   // one bit used, no advance: what a guessed start does with it
        const bool dc = t < 2;
        const uint32_t adv = dc ? 1u : (sym == 0 ? kFastEobAdvance : (sym >> 4) + 1u);
        return size | ((dc || size) ? kFastEmit : 0u) | (adv << 16) | ((len + size) << 24);
    };
    for (int t = 0; t < 4; ++t) {
        for (int i = 0; i < (1 << kLut1Bits); ++i) {
            const uint16_t e = dec.l1[t][i];
            dec.fast[t][i] = (e & 0x8000u) ? 0u : fast(e, t);
        }
        for (uint32_t w16 = kFastLongFirst; w16 < 0x10000u; ++w16) {
            const uint16_t e1 = dec.l1[t][w16 >> (16 - kLut1Bits)];
            if (e1 & 0x8000u)
                dec.fast[t][kFastM2Word + 1u + (w16 - kFastLongFirst)] =
                    fast(dec.l2[e1 & 0xffu][(w16 >> (16 - kLut1Bits - kLut2Bits)) & ((1u << kLut2Bits) - 1u)], t);
        }
    }
}

// ---- timing ---------------------------------------------------------------------------------

struct Timed {
    amvhip_ctx* c;
    hipStream_t s;
    ProfRec r{};
    bool on;
    Timed(amvhip_ctx* ctx, int kernel, hipStream_t st) : c(ctx), s(st), on(ctx->prof) {
        if (!on) return;
        r.kernel = kernel;
        for (hipEvent_t* e : {&r.a, &r.b}) {
            if (!c->pool.empty()) { *e = c->pool.back(); c->pool.pop_back(); }
            else if (hipEventCreate(e) != hipSuccess) { on = false; return; }
        }
        (void)hipEventRecord(r.a, s);
    }
    ~Timed() {
        if (!on) return;
        (void)hipEventRecord(r.b, s);
        c->recs.push_back(r);
    }
};

void drain(amvhip_ctx* c) {
    for (ProfRec& r : c->recs) {
        float ms = 0;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            c->launches[r.kernel]++;
            c->total_ms[r.kernel] += ms;
        }
        c->pool.push_back(r.a);
        c->pool.push_back(r.b);
    }
    c->recs.clear();
}

int check_launch(amvhip_ctx* c, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, AMVHIP_ERR_DEVICE, "%s launch: %s", what, hipGetErrorString(e));
    return AMVHIP_OK;
}

int select_device(amvhip_ctx* c) {
    HIP_TRY(c, hipSetDevice(c->device));
    return AMVHIP_OK;
}

// Every entry point but amvhip_decode_submit_dev / _collect_dev: batches submitted earlier share the context's
// workspace with what is about to be queued, so they finish first (nothing to wait for when none is in flight).
int use_device(amvhip_ctx* c) {
    if (int r = select_device(c)) return r;
    if (c->front && c->submitted != 0) {
        HIP_TRY(c, hipStreamSynchronize(c->front));
        HIP_TRY(c, hipStreamSynchronize(c->back));
    }
    return AMVHIP_OK;
}

// the stream of the host-buffer entry points: created on first use, non-blocking (it does not order itself against
// the caller's other streams)
int host_stream(amvhip_ctx* c, hipStream_t* out) {
    if (!c->hstream) HIP_TRY(c, hipStreamCreateWithFlags(&c->hstream, hipStreamNonBlocking));
    *out = c->hstream;
    return AMVHIP_OK;
}

}  // namespace

// =============================================================================================
// context
// =============================================================================================

extern "C" int amvhip_create(amvhip_ctx** out, int device) {
    if (!out) return AMVHIP_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
        return AMVHIP_ERR_DEVICE;
    amvhip_ctx* c = new amvhip_ctx;
    c->device = device;
    static HuffDecodeImage dec;
    static HuffEncodeImage enc;
    static std::once_flag once;
    std::call_once(once, [] { build_images(dec, enc); });
    auto die = [&](int code) { amvhip_destroy(c); return code; };
    if (hipSetDevice(device) != hipSuccess) return die(AMVHIP_ERR_DEVICE);
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->cus = (uint32_t)cus;
    if (const char* e = getenv("AMVHIP_SYNC_LANES")) {   // tuning knob: lanes per frame of the entropy kernel
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) c->sync_lanes = v;
    }
    if (const char* e = getenv("AMVHIP_SPLIT")) {   // tuning / test knob: 0 = no split, a power of two = lanes per heavy frame
        const int v = atoi(e);
        c->split_heavy = v != 0;
        if (v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) c->heavy_lanes = (uint32_t)v;
        if (v == -1) c->heavy_lanes = 1;    // two lists, one lane per frame in both (measurements)
    }
    if (const char* e = getenv("AMVHIP_LAYOUT")) c->layout_large = strcmp(e, "large") == 0;   // test knob: the three-launch layout for small batches too
    if (const char* e = getenv("AMVHIP_ADPCM_SWEEPS")) {   // tuning / test knob: "map" = exhaustive route only, or a sweep count
        if (strcmp(e, "nosettle") == 0) {
            c->adpcm_settle = false;   // sweeps by stream length, then nothing: the chain's check sends the stream down the exhaustive route
        } else {
            c->adpcm_sweeps_set = true;
            c->adpcm_sweeps = strcmp(e, "map") == 0 ? -1 : (atoi(e) < 0 ? 0 : (atoi(e) > 60 ? 60 : atoi(e)));
        }
    }
    if (hipMalloc((void**)&c->d_dec, sizeof dec) != hipSuccess) return die(AMVHIP_ERR_NOMEM);
    if (hipMalloc((void**)&c->d_enc, sizeof enc) != hipSuccess) return die(AMVHIP_ERR_NOMEM);
    if (hipMemcpy(c->d_dec, &dec, sizeof dec, hipMemcpyHostToDevice) != hipSuccess) return die(AMVHIP_ERR_DEVICE);
    if (hipMemcpy(c->d_enc, &enc, sizeof enc, hipMemcpyHostToDevice) != hipSuccess) return die(AMVHIP_ERR_DEVICE);
    *out = c;
    return AMVHIP_OK;
}

extern "C" void amvhip_destroy(amvhip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    drain(c);
    for (hipEvent_t e : c->pool) (void)hipEventDestroy(e);
    if (c->hstream) { (void)hipStreamSynchronize(c->hstream); (void)hipStreamDestroy(c->hstream); }
    if (c->dstream) { (void)hipStreamSynchronize(c->dstream); (void)hipStreamDestroy(c->dstream); }
    for (hipEvent_t e : {c->ev_decoded, c->ev_copied[0], c->ev_copied[1]})
        if (e) (void)hipEventDestroy(e);
    for (DevBuf* b : {&c->v_out[0], &c->v_out[1], &c->v_status[0], &c->v_status[1]})
        if (b->p) (void)hipFree(b->p);
    for (hipStream_t q : {c->front, c->back})
        if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
    for (hipEvent_t e : {c->ev_in, c->ev_front, c->ev_done[0], c->ev_done[1]})
        if (e) (void)hipEventDestroy(e);
    for (DevBuf* b : {&c->coef, &c->status, &c->nmcu, &c->tmp, &c->lens, &c->offs, &c->flag, &c->map,
                      &c->start, &c->retry, &c->enc_retry, &c->stats, &c->ws, &c->ws_line, &c->layout, &c->ws_bytes, &c->rec, &c->rec_line, &c->seg_start, &c->lane_tab, &c->rec_count, &c->scaled, &c->trellis_ws, &c->chain, &c->split, &c->h_in, &c->h_offs, &c->h_lens, &c->h_out, &c->h_status, &c->h_aux, &c->a_in, &c->a_tab, &c->a_out,
                      &c->second.nmcu, &c->second.retry, &c->second.rec, &c->second.rec_line, &c->second.seg_start, &c->second.lane_tab, &c->second.rec_count})
        if (b->p) (void)hipFree(b->p);
    if (c->d_dec) (void)hipFree(c->d_dec);
    if (c->d_enc) (void)hipFree(c->d_enc);
    delete c;
}

extern "C" const char* amvhip_last_error(const amvhip_ctx* c) { return c ? c->err.c_str() : "null context"; }
extern "C" int amvhip_device(const amvhip_ctx* c) { return c ? c->device : -1; }

extern "C" uint32_t amvhip_stride(uint32_t w) { return (w * 24 + 31) / 32 * 4; }
extern "C" uint64_t amvhip_frame_bytes(uint32_t w, uint32_t h) { return (uint64_t)amvhip_stride(w) * h; }
extern "C" uint64_t amvhip_yuv420_frame_bytes(uint32_t w, uint32_t h) {
    return (uint64_t)w * h + 2ull * ((w + 1) / 2) * ((h + 1) / 2);
}
extern "C" uint32_t amvhip_encode_bound(uint32_t w, uint32_t h) {
    // per coefficient at most a 16-bit code + 11 magnitude bits (< 4 bytes), doubled by FF escaping
    return 4 + ((w + 15) / 16) * ((h + 15) / 16) * 6 * 64 * 4 * 2;
}

// =============================================================================================
// video decode
// =============================================================================================

static int size_ok(uint32_t w, uint32_t h) { return w > 0 && h > 0 && w <= AMVHIP_MAX_DIM && h <= AMVHIP_MAX_DIM; }

// Frames the synchronising kernel does not decode go through amv_huffman_kernel, whose output is dense coefficient
// lines: all of them in AMVHIP_ENTROPY_SERIAL mode (and for pictures of >= 16384 blocks), else the few it hands back
// (oversize chunks, long FF runs, more records than the record space holds).  `items` bounds the work; with a list the
// real count sits on the device.
struct Fallback {
    const uint32_t* list;
    const uint32_t* count;
    uint32_t items;
};

// unstuffing + the synchronising kernel into `sinks` (dense when sinks.rec == nullptr, records otherwise); fb says
// what is left for the serial kernel
// lanes: lanes per frame of the synchronising kernel; heavy_lanes != 0 (records form, a batch that gets ONE lane per frame):
// frames whose chunk is over twice the batch's mean get that many lanes instead -- a wave's 64 frames finish together, and one
// noise frame among 63 quiet ones kept them all waiting for six times their own length (a stream with every 16th frame noise
// spent 4.3 ms per 160 000 frames in the entropy kernel for 1.3 times the uniform stream's symbols).  The split is made on the
// device (the lengths are there): two frame lists, two launches that take their frames from them.
static int entropy_front(amvhip_ctx* c, const uint8_t* d_blob, uint64_t blob_bytes, const uint64_t* d_offs,
                         const uint32_t* d_lens, uint32_t n, const FrameGeom& g, SyncSinks sinks, int32_t* d_status,
                         uint32_t* d_nmcu_ok, DevBuf& retry, hipStream_t st, Fallback& fb, int lanes, int heavy_lanes = 0,
                         const LayoutSpec& rec_layout = LayoutSpec{0u, 0u, 0u, 0u, 0u, nullptr}) {
    // Window per frame for the unstuffed scan in the global workspace: the frame's own chunk length + the zeroed tail of its
    // last 16-byte piece + a piece of slack, in 16-byte pieces laid out on the device (round 4; 5/16 byte per pixel for every
    // frame before: a chunk over 1.6x the usual size went to the serial kernel, the others used 60 % of their window).  The
    // total is bounded by what the chunks occupy; chunks that overlap in the blob make the layout run out, and the frames
    // past its end take the serial kernel.
    uint64_t ws_lines = (blob_bytes + (uint64_t)n * 48u) / 16u + 4u;
    if (ws_lines > 0xffffffffull) ws_lines = 0xffffffffull;
    if (c->entropy_mode == AMVHIP_ENTROPY_SERIAL || g.blocks >= 16384u) {
        if (sinks.rec) HIP_TRY(c, hipMemsetAsync(sinks.rec_count, 0xff, (size_t)n * 4, st));   // every frame dense
        fb = Fallback{nullptr, nullptr, n};
        c->last_split = false;
        return AMVHIP_OK;
    }
    if (int r = ensure(c, retry, ((size_t)n + 8) * 4)) return r;   // [retry count, task counter, 6 spare | retry list n]
    if (int r = ensure(c, c->ws, (size_t)ws_lines * 16 + 64)) return r;
    if (int r = ensure(c, c->ws_line, ((size_t)n + 1) * 4)) return r;
    if (int r = ensure(c, c->ws_bytes, (size_t)n * 4)) return r;
    if (int r = ensure(c, c->layout, layout_workspace(n))) return r;
    uint32_t* retry_count = (uint32_t*)retry.p;
    uint32_t* retry_list = retry_count + 8;
    sinks.retry_list = retry_list;
    sinks.retry_count = retry_count;
    // (the layout launch also clears the retry counter and the task queues behind it: retry_count[0 .. 8))
    launch_layout(d_lens, n, LayoutSpec{2u, 32u, 0xffffffe0u, 4u, (uint32_t)ws_lines, (uint32_t*)c->ws_line.p}, rec_layout, c->layout.p,
                  retry_count, 8u, c->layout_large, st);
    if (int r = check_launch(c, "layout")) return r;
    {
        Timed t(c, AMVHIP_K_UNSTUFF, st);
        launch_unstuff(d_blob, blob_bytes, d_offs, d_lens, n, (const uint32_t*)c->ws_line.p, (uint32_t*)c->ws.p, (uint32_t*)c->ws_bytes.p,
                       retry_list, retry_count, st);
    }
    if (int r = check_launch(c, "unstuff")) return r;
    unsigned long long* stats = c->want_stats ? (unsigned long long*)c->stats.p : nullptr;
    c->last_split = heavy_lanes && sinks.rec;
    if (heavy_lanes && sinks.rec) {
        if (int r = ensure(c, c->split, ((size_t)n * 2 + 8) * 4)) return r;   // [heavy count, light count, 6 spare | heavy list n | light list n]
        uint32_t* split_count = (uint32_t*)c->split.p;
        uint32_t *heavy = split_count + 8, *light = heavy + n;
        HIP_TRY(c, hipMemsetAsync(split_count, 0, 32, st));
        launch_split_by_weight(d_lens, n, (const uint32_t*)c->ws_line.p, heavy, light, split_count, st);
        if (int r = check_launch(c, "split")) return r;
        Timed t(c, AMVHIP_K_HUFFMAN, st);
        // the heavy frames first: their launch is a few waves deep and as long as its slowest frame's chain, the light
        // frames' launch behind it fills the chip
        launch_huffman_sync((const uint32_t*)c->ws.p, (const uint32_t*)c->ws_bytes.p, n, heavy, split_count, g, (const uint32_t*)c->ws_line.p,
                            heavy_lanes, c->d_dec, sinks, d_status, d_nmcu_ok, retry_count + 2, stats, c->cus, st);
        launch_huffman_sync((const uint32_t*)c->ws.p, (const uint32_t*)c->ws_bytes.p, n, light, split_count + 1, g, (const uint32_t*)c->ws_line.p,
                            lanes, c->d_dec, sinks, d_status, d_nmcu_ok, retry_count + 1, stats, c->cus, st);
    } else {
        Timed t(c, AMVHIP_K_HUFFMAN, st);
        launch_huffman_sync((const uint32_t*)c->ws.p, (const uint32_t*)c->ws_bytes.p, n, nullptr, nullptr, g, (const uint32_t*)c->ws_line.p,
                            lanes, c->d_dec, sinks, d_status, d_nmcu_ok, retry_count + 1, stats, c->cus, st);
    }
    fb = Fallback{retry_list, retry_count, n};
    return check_launch(c, "huffman_sync");
}

extern "C" int amvhip_huffman_decode_dev(amvhip_ctx* c, const uint8_t* d_blob, uint64_t blob_bytes,
                                         const uint64_t* d_offs, const uint32_t* d_lens, uint32_t n,
                                         uint32_t w, uint32_t h, int16_t* d_coef, int32_t* d_status,
                                         uint32_t* d_nmcu_ok, void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (!size_ok(w, h) || (n && (!d_blob || !d_offs || !d_lens || !d_coef || !d_status || !d_nmcu_ok)))
        return fail(c, AMVHIP_ERR_ARG, "huffman_decode: bad argument");
    if (((uintptr_t)d_blob & 3u) || ((uintptr_t)d_coef & 15u)) return fail(c, AMVHIP_ERR_ARG, "huffman_decode: blob must be 4-byte, coef 16-byte aligned");
    if (int r = use_device(c)) return r;
    if (n == 0) return AMVHIP_OK;
    std::lock_guard<std::mutex> lk(c->mu);
    const FrameGeom g = make_geom(w, h);
    hipStream_t st = (hipStream_t)stream;
    SyncSinks sinks{d_coef, nullptr, 0u, nullptr, nullptr, 0u, nullptr, nullptr, nullptr};
    Fallback fb;
    if (int r = entropy_front(c, d_blob, blob_bytes, d_offs, d_lens, n, g, sinks, d_status, d_nmcu_ok, c->retry, st, fb,
                              huffman_sync_lanes(n, c->cus, c->sync_lanes, (uint64_t)g.width * g.height, false)))
        return r;
    {   // the caller's array has a place for every frame: one launch, lines at the frames' own places
        Timed t(c, AMVHIP_K_HUFFMAN_SERIAL, st);
        launch_huffman(d_blob, blob_bytes, d_offs, d_lens, n, g, c->d_dec, d_coef, d_status, d_nmcu_ok, fb.list, fb.count, 0u,
                       fb.items, false, false, st);
    }
    return check_launch(c, "huffman");
}

static int reconstruct_launch(amvhip_ctx* c, const SyncSinks& sinks, const uint32_t* d_nmcu_ok, uint32_t n, const FrameSel& sel,
                              uint32_t items, const FrameGeom& g, uint32_t flags, uint8_t* d_out, hipStream_t st) {
    Timed t(c, AMVHIP_K_RECON, st);
    if (flags & AMVHIP_FLAG_FFMPEG)   // the patched FFmpeg's amv_decoder: YUVJ420P planes
        launch_reconstruct_yuv(sinks, d_nmcu_ok, n, sel, items, g, amvhip_yuv420_frame_bytes(g.width, g.height), d_out, st);
    else
        launch_reconstruct(sinks, d_nmcu_ok, n, sel, items, g, flags, d_out, st);
    return check_launch(c, "reconstruct");
}

// bytes of the output no kernel writes are cleared first: row padding (AMVDec.c:283), and in FFmpeg mode the plane
// rows mjpegdec.c:672-677 leaves untouched for some heights
static int clear_unwritten(amvhip_ctx* c, uint32_t n, const FrameGeom& g, uint32_t flags, uint8_t* d_out, hipStream_t st) {
    if (flags & AMVHIP_FLAG_FFMPEG_KEEP) return AMVHIP_OK;   // what no block covers stays as the caller had it
    if (flags & AMVHIP_FLAG_FFMPEG) {
        if (!yuv_store_covers_planes(g)) HIP_TRY(c, hipMemsetAsync(d_out, 0, amvhip_yuv420_frame_bytes(g.width, g.height) * n, st));
    } else if (g.stride != g.width * 3) {
        HIP_TRY(c, hipMemsetAsync(d_out, 0, g.frame_bytes * n, st));
    }
    return AMVHIP_OK;
}

extern "C" int amvhip_reconstruct_dev(amvhip_ctx* c, const int16_t* d_coef, const uint32_t* d_nmcu_ok,
                                      uint32_t n, uint32_t w, uint32_t h, uint32_t flags, uint8_t* d_out,
                                      void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (!size_ok(w, h) || (n && (!d_coef || !d_nmcu_ok || !d_out))) return fail(c, AMVHIP_ERR_ARG, "reconstruct: bad argument");
    if (((uintptr_t)d_out & 3u) || ((uintptr_t)d_coef & 15u)) return fail(c, AMVHIP_ERR_ARG, "reconstruct: out must be 4-byte, coef 16-byte aligned");
    if (int r = use_device(c)) return r;
    if (n == 0) return AMVHIP_OK;
    const FrameGeom g = make_geom(w, h);
    SyncSinks sinks{const_cast<int16_t*>(d_coef), nullptr, 0u, nullptr, nullptr, 0u, nullptr, nullptr, nullptr};
    if (int r = clear_unwritten(c, n, g, flags, d_out, (hipStream_t)stream)) return r;
    return reconstruct_launch(c, sinks, d_nmcu_ok, n, FrameSel{nullptr, nullptr, 0u, 0u}, n, g, flags, d_out, (hipStream_t)stream);
}

// dense coefficient lines the context keeps for frames that go through the serial kernel: a round's worth
// (one round up to 16 384 frames: every round is a pair of launches that usually find nothing to do, and a batch that
// small is latency-bound -- three rounds cost the 10 000-frame stream 3 % of its step)
static uint32_t dense_round(uint32_t n) { return n <= 16384u ? n : (n / 4u > 16384u ? (n + 3u) / 4u : 16384u); }

// What one decode call hands from the entropy stage to the reconstruction (the context has two such sets).
struct DecodeBufs {
    DevBuf &nmcu, &retry, &rec, &rec_line, &seg_start, &lane_tab, &rec_count;
};

static int decode_args_ok(amvhip_ctx* c, const uint8_t* d_blob, const uint64_t* d_offs, const uint32_t* d_lens, uint32_t n, uint32_t w,
                          uint32_t h, uint32_t flags, const uint8_t* d_out, const int32_t* d_status) {
    if (!size_ok(w, h)) return fail(c, AMVHIP_ERR_ARG, "decode: bad size %ux%u", w, h);
    if ((flags & AMVHIP_FLAG_FFMPEG_KEEP) && !(flags & AMVHIP_FLAG_FFMPEG))
        return fail(c, AMVHIP_ERR_ARG, "decode: AMVHIP_FLAG_FFMPEG_KEEP is a mode of AMVHIP_FLAG_FFMPEG");
    if (n == 0) return AMVHIP_OK;
    if (!d_blob || !d_offs || !d_lens || !d_out || !d_status) return fail(c, AMVHIP_ERR_ARG, "decode: null argument");
    if (((uintptr_t)d_blob & 3u) || ((uintptr_t)d_out & 3u)) return fail(c, AMVHIP_ERR_ARG, "decode: blob and out must be 4-byte aligned");
    return AMVHIP_OK;
}

// The entropy stage goes to stream `front`, everything that writes d_out to `back` (the same stream, or two of the
// context's own with `back` waiting for `front`).  Caller holds the lock.
static int decode_core(amvhip_ctx* c, const uint8_t* d_blob, uint64_t blob_bytes, const uint64_t* d_offs, const uint32_t* d_lens,
                       uint32_t n, uint32_t w, uint32_t h, uint32_t flags, uint8_t* d_out, int32_t* d_status, DecodeBufs b,
                       hipStream_t front, hipStream_t back) {
    const FrameGeom g = make_geom(w, h);
    // Between the two stages coefficients travel as records (one word per DC and per non-zero AC coefficient), every frame
    // in space of its own, sized from ITS chunk (round 4; one stride for all, from the batch's mean chunk, before: a heavy
    // frame in a light stream was silently decoded by the one-lane serial kernel).  A record costs at least 3 bits of scan,
    // in practice ~6, and every block has a DC symbol and (nearly always) an end-of-block symbol, which take a slot each in
    // the one-lane kernel's stride-aligned form: 2 words per byte of chunk + 2 per block + a stride's slack (never more than a
    // frame with every coefficient non-zero could fill; a frame denser than 2 records per byte goes to the serial kernel), in
    // whole 128-byte lines.  The space is laid out on
    // the device (the lengths are there); its total is bounded here by what the chunks occupy (chunks that overlap make the
    // sum larger than that: the layout saturates and the frames past the end are handed to the serial kernel).
    const uint32_t hi_rec = (g.blocks * 66u + 95u) & ~31u;   // every coefficient of every block non-zero, an end-of-block slot each
    const uint32_t add_rec = g.blocks * 2u + 64u;
    uint64_t cap_lines = (uint64_t)n * (hi_rec / 32u);
    {
        const uint64_t by_stream = (2u * blob_bytes + (uint64_t)n * (add_rec + 31u)) / 32u + 1u;
        if (by_stream < cap_lines) cap_lines = by_stream;
        if (cap_lines > 0xffffffffull) cap_lines = 0xffffffffull;
    }
    const uint32_t lanes = (uint32_t)huffman_sync_lanes(n, c->cus, c->sync_lanes, (uint64_t)g.width * g.height, true);
    // a batch that gets one lane per frame gives its heavy frames kHeavyLanes (entropy_front); AMVHIP_SPLIT=0: every frame one
    const uint32_t heavy_lanes = lanes == 1u && c->split_heavy ? c->heavy_lanes : 0u;
    const uint32_t tab_lanes = heavy_lanes ? heavy_lanes : lanes;       // a frame's row in lane_tab
    const uint32_t segs = ((g.mcu_cols + 9u) / 10u) * g.mcu_rows;
    // dense lines for a round of fall-back frames: by count (above), and never more than 2 GB of them -- at 640x480 a
    // block line is 128 bytes x 7 200 blocks, and 16 384 frames of that would be 15 GB kept for rounds that usually find nothing
    uint32_t round = dense_round(n);
    {
        const uint64_t by_bytes = (2ull << 30) / ((uint64_t)g.blocks * 128u);
        if (round > by_bytes) round = by_bytes > 64u ? (uint32_t)by_bytes : 64u;
        if (round > n) round = n;
    }
    if (int r = ensure(c, c->coef, (size_t)round * g.blocks * 128)) return r;
    if (int r = ensure(c, b.nmcu, (size_t)n * 4)) return r;
    if (int r = ensure(c, b.rec, (size_t)cap_lines * 128 + 16)) return r;   // + what a 16-byte read of a frame's last records may overshoot
    if (int r = ensure(c, b.rec_line, ((size_t)n + 1) * 4)) return r;
    const LayoutSpec rec_layout{4u, add_rec, hi_rec, 5u, (uint32_t)cap_lines, (uint32_t*)b.rec_line.p};   // laid out by entropy_front's launch
    if (int r = ensure(c, b.seg_start, (size_t)n * (segs + 1) * 8)) return r;
    if (int r = ensure(c, b.lane_tab, (size_t)n * tab_lanes * 16)) return r;
    if (int r = ensure(c, b.rec_count, (size_t)n * 4)) return r;
    SyncSinks sinks{(int16_t*)c->coef.p, (uint32_t*)b.rec.p, (const uint32_t*)b.rec_line.p, (uint32_t*)b.seg_start.p, (uint32_t*)b.lane_tab.p, tab_lanes,
                    (uint32_t*)b.rec_count.p, nullptr, nullptr};
    sinks.ok_in_blocks = (flags & AMVHIP_FLAG_FFMPEG_KEEP) ? 1u : 0u;   // b.nmcu is the context's own array: whole blocks, not whole MCUs
    uint32_t* d_nmcu = (uint32_t*)b.nmcu.p;
    c->last_decode_retry = &b.retry;
    Fallback fb;
    if (int r = entropy_front(c, d_blob, blob_bytes, d_offs, d_lens, n, g, sinks, d_status, d_nmcu, b.retry, front, fb, (int)lanes, (int)heavy_lanes,
                              rec_layout))
        return r;
    if (back != front) {
        HIP_TRY(c, hipEventRecord(c->ev_front, front));
        HIP_TRY(c, hipStreamWaitEvent(back, c->ev_front, 0));
    }
    hipStream_t st = back;
    if (int r = clear_unwritten(c, n, g, flags, d_out, st)) return r;
    if (fb.list)   // the frames in records form (a launch that skips the others)
        if (int r = reconstruct_launch(c, sinks, d_nmcu, n, FrameSel{nullptr, nullptr, 0u, 0u}, n, g, flags, d_out, st)) return r;
    // The others, a round of dense lines at a time.  With a list the count is on the device: the rounds past it find
    // nothing to do and leave at once (usually all of them: one pair of empty launches per round).
    for (uint32_t base = 0; base < fb.items; base += round) {
        const uint32_t items = fb.items - base < round ? fb.items - base : round;
        {
            Timed t(c, AMVHIP_K_HUFFMAN_SERIAL, st);
            launch_huffman(d_blob, blob_bytes, d_offs, d_lens, n, g, c->d_dec, sinks.coef, d_status, d_nmcu, fb.list, fb.count, base,
                           items, true, sinks.ok_in_blocks != 0u, st);
        }
        if (int r = check_launch(c, "huffman")) return r;
        if (int r = reconstruct_launch(c, sinks, d_nmcu, n, FrameSel{fb.list, fb.count, base, items}, items, g, flags, d_out, st)) return r;
    }
    c->ws_bytes_per_frame = (double)(c->ws.cap + c->coef.cap + c->ws_bytes.cap + c->ws_line.cap + c->rec.cap + c->rec_line.cap + c->second.rec_line.cap + c->seg_start.cap + c->lane_tab.cap +
                                     c->rec_count.cap + c->nmcu.cap + c->retry.cap + c->second.rec.cap + c->second.seg_start.cap +
                                     c->second.lane_tab.cap + c->second.rec_count.cap + c->second.nmcu.cap + c->second.retry.cap) / n;
    return AMVHIP_OK;
}

extern "C" int amvhip_decode_batch_dev(amvhip_ctx* c, const uint8_t* d_blob, uint64_t blob_bytes,
                                       const uint64_t* d_offs, const uint32_t* d_lens, uint32_t n,
                                       uint32_t w, uint32_t h, uint32_t flags, uint8_t* d_out,
                                       int32_t* d_status, void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (int r = decode_args_ok(c, d_blob, d_offs, d_lens, n, w, h, flags, d_out, d_status)) return r;
    if (n == 0) return AMVHIP_OK;
    if (int r = use_device(c)) return r;
    std::lock_guard<std::mutex> lk(c->mu);
    return decode_core(c, d_blob, blob_bytes, d_offs, d_lens, n, w, h, flags, d_out, d_status,
                       DecodeBufs{c->nmcu, c->retry, c->rec, c->rec_line, c->seg_start, c->lane_tab, c->rec_count}, (hipStream_t)stream,
                       (hipStream_t)stream);
}

// ---- the same in two halves, for a caller with more than one batch in hand ---------------------------------------
// submit: the batch's inputs are ready where `stream` stands now (an event is recorded there); the entropy stage is
// queued on the context's `front` stream, the reconstruction on its `back` stream, and `stream` is NOT made to wait.
// collect: `stream` waits for the oldest batch submitted and not yet collected.  With submit(k+1) called before
// collect(k), the entropy stage of batch k+1 runs beside the reconstruction of batch k -- the two are limited by
// different things (amv_huffman_fast_kernel by memory latency and scattered stores, amv_reconstruct_kernel by VALU
// issue).  At most two batches between submit and collect: the hand-over buffers exist twice.
extern "C" int amvhip_decode_submit_dev(amvhip_ctx* c, const uint8_t* d_blob, uint64_t blob_bytes,
                                        const uint64_t* d_offs, const uint32_t* d_lens, uint32_t n,
                                        uint32_t w, uint32_t h, uint32_t flags, uint8_t* d_out,
                                        int32_t* d_status, void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (int r = decode_args_ok(c, d_blob, d_offs, d_lens, n, w, h, flags, d_out, d_status)) return r;
    if (int r = select_device(c)) return r;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->submitted - c->collected >= 2) return fail(c, AMVHIP_ERR_ARG, "decode_submit: two batches are in flight, collect one first");
    if (!c->front) {
        // the entropy stage's few large workgroups (127 KB of LDS) ahead of the reconstruction's many small ones
        int least = 0, greatest = 0;
        HIP_TRY(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(c, hipStreamCreateWithPriority(&c->front, hipStreamNonBlocking, greatest));
        HIP_TRY(c, hipStreamCreateWithPriority(&c->back, hipStreamNonBlocking, least));
        for (hipEvent_t* e : {&c->ev_in, &c->ev_front, &c->ev_done[0], &c->ev_done[1]})
            HIP_TRY(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    const int which = (int)(c->submitted & 1u);
    HIP_TRY(c, hipEventRecord(c->ev_in, (hipStream_t)stream));
    HIP_TRY(c, hipStreamWaitEvent(c->front, c->ev_in, 0));
    // (the hand-over set this batch writes was last read by the reconstruction of the batch two before: on `back`,
    // ahead of the batch before this one -- whose entropy stage `front` has already gone through -- but not of `front`)
    if (c->submitted >= 2) HIP_TRY(c, hipStreamWaitEvent(c->front, c->ev_done[which], 0));
    if (n != 0) {
        DecodeBufs first{c->nmcu, c->retry, c->rec, c->rec_line, c->seg_start, c->lane_tab, c->rec_count};
        DecodeBufs second{c->second.nmcu, c->second.retry, c->second.rec, c->second.rec_line, c->second.seg_start, c->second.lane_tab, c->second.rec_count};
        if (int r = decode_core(c, d_blob, blob_bytes, d_offs, d_lens, n, w, h, flags, d_out, d_status, which ? second : first, c->front,
                                c->back))
            return r;
    } else {   // nothing to decode: `back` still has to pass the point where the inputs are ready
        HIP_TRY(c, hipEventRecord(c->ev_front, c->front));
        HIP_TRY(c, hipStreamWaitEvent(c->back, c->ev_front, 0));
    }
    HIP_TRY(c, hipEventRecord(c->ev_done[which], c->back));
    ++c->submitted;
    return AMVHIP_OK;
}

extern "C" int amvhip_decode_collect_dev(amvhip_ctx* c, void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (int r = select_device(c)) return r;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->collected == c->submitted) return fail(c, AMVHIP_ERR_ARG, "decode_collect: nothing submitted");
    HIP_TRY(c, hipStreamWaitEvent((hipStream_t)stream, c->ev_done[c->collected & 1u], 0));
    ++c->collected;
    return AMVHIP_OK;
}

// device workspace the last amvhip_decode_batch_dev call held, in bytes per frame of that call (a diagnostic)
extern "C" double amvhip_decode_workspace_per_frame(const amvhip_ctx* c) { return c ? c->ws_bytes_per_frame : 0.0; }

extern "C" int amvhip_decode_batch_async(amvhip_ctx* c, const uint8_t* blob, uint64_t blob_bytes,
                                         const uint64_t* offs, const uint32_t* lens, uint32_t n, uint32_t w,
                                         uint32_t h, uint32_t flags, uint8_t* out, int32_t* status) {
    if (!c) return AMVHIP_ERR_ARG;
    if (!size_ok(w, h) || (n && (!blob || !offs || !lens || !out))) return fail(c, AMVHIP_ERR_ARG, "decode: bad argument");
    if (n == 0) return AMVHIP_OK;
    if (int r = use_device(c)) return r;
    hipStream_t st;
    if (int r = host_stream(c, &st)) return r;
    const uint64_t fb = (flags & AMVHIP_FLAG_FFMPEG) ? amvhip_yuv420_frame_bytes(w, h) : amvhip_frame_bytes(w, h);
    // the staging buffers belong to the context: one host-buffer call at a time grows and fills them (hmu orders the
    // host-buffer entry points among themselves; mu, taken inside the _dev calls, orders the kernels' workspace)
    std::lock_guard<std::mutex> hlk(c->hmu);
    if (int r = ensure(c, c->h_in, blob_bytes + 16)) return r;
    if (int r = ensure(c, c->h_offs, (size_t)n * 8)) return r;
    if (int r = ensure(c, c->h_lens, (size_t)n * 4)) return r;
    if (!c->dstream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->dstream, hipStreamNonBlocking));
        for (hipEvent_t* e : {&c->ev_decoded, &c->ev_copied[0], &c->ev_copied[1]}) HIP_TRY(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    const uint32_t which = (uint32_t)(c->async_calls & 1u);
    DevBuf &d_frames = c->v_out[which], &d_st = c->v_status[which];
    if (int r = ensure(c, d_frames, fb * n)) return r;          // (growing one frees the old: hipFree waits for the device)
    if (int r = ensure(c, d_st, (size_t)n * 4)) return r;
    // the copy that last read this staging buffer (the call before the last one) must be done before the kernels write it
    if (c->async_calls >= 2) HIP_TRY(c, hipStreamWaitEvent(st, c->ev_copied[which], 0));
    HIP_TRY(c, hipMemcpyAsync(c->h_in.p, blob, blob_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->h_offs.p, offs, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->h_lens.p, lens, (size_t)n * 4, hipMemcpyHostToDevice, st));
    // AMVHIP_FLAG_FFMPEG_KEEP: what no block covers stays as the CALLER had it -- the caller's frames go up first
    if (flags & AMVHIP_FLAG_FFMPEG_KEEP) HIP_TRY(c, hipMemcpyAsync(d_frames.p, out, fb * n, hipMemcpyHostToDevice, st));
    if (int r = amvhip_decode_batch_dev(c, (const uint8_t*)c->h_in.p, blob_bytes, (const uint64_t*)c->h_offs.p,
                                        (const uint32_t*)c->h_lens.p, n, w, h, flags, (uint8_t*)d_frames.p, (int32_t*)d_st.p, st))
        return r;
    HIP_TRY(c, hipEventRecord(c->ev_decoded, st));
    HIP_TRY(c, hipStreamWaitEvent(c->dstream, c->ev_decoded, 0));
    // From the first copy queued on dstream on, a failure must not leave this staging buffer with a copy in flight that
    // no event stands for: the call after the next would pick the buffer again, wait for an event recorded two calls
    // earlier, and let its kernels write under the orphaned copy.  So whatever fails below, dstream is drained before
    // the call returns, and the call counts (the buffers keep taking turns).
    hipError_t e = hipMemcpyAsync(out, d_frames.p, fb * n, hipMemcpyDeviceToHost, c->dstream);
    if (e == hipSuccess && status) e = hipMemcpyAsync(status, d_st.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->dstream);
    if (e == hipSuccess) e = hipEventRecord(c->ev_copied[which], c->dstream);
    ++c->async_calls;
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(c->dstream);
        (void)hipEventRecord(c->ev_copied[which], c->dstream);   // (what the call after the next will wait for: nothing pending)
        return fail(c, AMVHIP_ERR_DEVICE, "decode_batch_async: copy back: %s", hipGetErrorString(e));
    }
    return AMVHIP_OK;
}

extern "C" int amvhip_sync(amvhip_ctx* c) {
    if (!c) return AMVHIP_ERR_ARG;
    if (int r = use_device(c)) return r;
    if (c->hstream) HIP_TRY(c, hipStreamSynchronize(c->hstream));
    if (c->dstream) HIP_TRY(c, hipStreamSynchronize(c->dstream));
    return AMVHIP_OK;
}

extern "C" int amvhip_decode_batch(amvhip_ctx* c, const uint8_t* blob, uint64_t blob_bytes,
                                   const uint64_t* offs, const uint32_t* lens, uint32_t n, uint32_t w,
                                   uint32_t h, uint32_t flags, uint8_t* out, int32_t* status) {
    if (int r = amvhip_decode_batch_async(c, blob, blob_bytes, offs, lens, n, w, h, flags, out, status)) return r;
    return amvhip_sync(c);
}

// page-locked host memory for the *_async entry points (pageable buffers work too, but their copies block)
extern "C" int amvhip_host_alloc(amvhip_ctx* c, void** p, size_t bytes) {
    if (!c || !p) return AMVHIP_ERR_ARG;
    *p = nullptr;
    if (int r = use_device(c)) return r;
    HIP_TRY(c, hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault));
    return AMVHIP_OK;
}

extern "C" void amvhip_host_free(amvhip_ctx* c, void* p) {
    if (!c || !p) return;
    (void)hipSetDevice(c->device);
    (void)hipHostFree(p);
}

// =============================================================================================
// video encode
// =============================================================================================

extern "C" int amvhip_encode_coefs_dev(amvhip_ctx* c, const uint8_t* d_pix, uint32_t pix_stride, int is_bgr,
                                       uint32_t n, uint32_t w, uint32_t h, uint32_t qbias, int16_t* d_coef,
                                       void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (!size_ok(w, h) || (w & 1) || (h & 1) || pix_stride < w * 3 || qbias > 255 || (n && (!d_pix || !d_coef)))
        return fail(c, AMVHIP_ERR_ARG, "encode: bad argument (width/height must be even)");
    if ((uintptr_t)d_coef & 15u) return fail(c, AMVHIP_ERR_ARG, "encode: coef must be 16-byte aligned");
    if (int r = use_device(c)) return r;
    const FrameGeom g = make_geom(w, h);
    {
        Timed t(c, AMVHIP_K_FDCT, (hipStream_t)stream);
        launch_forward(d_pix, pix_stride, is_bgr, n, FrameSel{nullptr, nullptr, 0u, 0u}, n, g, qbias, d_coef, (hipStream_t)stream);
    }
    return check_launch(c, "forward");
}

// dense coefficient lines the context keeps for frames that go through the two-stage route: a round's worth
static uint32_t encode_round(uint32_t n) { return n <= 1024u ? n : (n / 2u > 1024u ? (n + 1u) / 2u : 1024u); }

// Pixels -> chunks for n frames, RGB (yuv == nullptr) or planar YUVJ420P (the context is locked).  The one-kernel
// encoder takes the batch; what it hands back -- and the whole batch in AMVHIP_ENTROPY_SERIAL mode -- goes through
// amv_forward_kernel + amv_pack_kernel a round of dense lines at a time (with a list the count is on the device: the
// rounds past it find nothing to do and leave at once -- usually all of them).
static int encode_core(amvhip_ctx* c, const uint8_t* d_pix, uint32_t pix_stride, int is_bgr, const YuvSource* yuv, uint32_t n,
                       const FrameGeom& g, uint32_t qbias, uint8_t* d_blob, uint64_t blob_cap, uint64_t* d_offs, uint32_t* d_lens,
                       hipStream_t stream) {
    const uint32_t bound = amvhip_encode_bound(g.width, g.height);
    uint32_t round = encode_round(n);
    {   // (never more than 2 GB of dense coefficient lines, as in decode_core)
        const uint64_t by_bytes = (2ull << 30) / ((uint64_t)g.blocks * 128u);
        if (round > by_bytes) round = by_bytes > 64u ? (uint32_t)by_bytes : 64u;
        if (round > n) round = n;
    }
    if (int r = ensure(c, c->coef, (size_t)round * g.blocks * 128)) return r;
    if (int r = ensure(c, c->tmp, (size_t)n * bound)) return r;
    if (int r = ensure(c, c->flag, 16)) return r;
    if (int r = ensure(c, c->enc_retry, ((size_t)n + 4) * 8)) return r;   // (the encoder's own: the decode path's counter is read by amvhip_entropy_stats)
    uint32_t* retry_count = (uint32_t*)c->enc_retry.p;
    uint32_t* retry_list = retry_count + 8;
    HIP_TRY(c, hipMemsetAsync(retry_count, 0, 32, stream));
    const bool fused = c->entropy_mode != AMVHIP_ENTROPY_SERIAL;
    if (fused) {
        Timed t(c, AMVHIP_K_PACK, stream);
        launch_encode_frames(d_pix, pix_stride, is_bgr, yuv, n, g, qbias, c->d_enc, (uint8_t*)c->tmp.p, bound, d_lens, retry_list,
                             retry_count, stream);
    }
    if (int r = check_launch(c, "encode_frames")) return r;
    for (uint32_t base = 0; base < n; base += round) {
        const uint32_t items = n - base < round ? n - base : round;
        const FrameSel sel{fused ? retry_list : nullptr, fused ? retry_count : nullptr, base, items};
        {
            Timed t(c, AMVHIP_K_FDCT, stream);
            if (yuv) launch_forward_yuv(*yuv, n, sel, items, g, qbias, (int16_t*)c->coef.p, stream);
            else launch_forward(d_pix, pix_stride, is_bgr, n, sel, items, g, qbias, (int16_t*)c->coef.p, stream);
        }
        {
            Timed t(c, AMVHIP_K_PACK_SERIAL, stream);
            launch_pack((const int16_t*)c->coef.p, n, sel, items, g, c->d_enc, (uint8_t*)c->tmp.p, bound, d_lens, stream);
        }
        if (int r = check_launch(c, "forward + pack")) return r;
    }
    {
        Timed t(c, AMVHIP_K_COMPACT, stream);
        launch_compact((const uint8_t*)c->tmp.p, bound, d_lens, n, d_offs, d_blob, blob_cap, (int32_t*)c->flag.p, stream);
    }
    return check_launch(c, "compact");
}

extern "C" int amvhip_encode_batch_dev(amvhip_ctx* c, const uint8_t* d_pix, uint32_t pix_stride, int is_bgr,
                                       uint32_t n, uint32_t w, uint32_t h, uint32_t qbias, uint8_t* d_blob,
                                       uint64_t blob_cap, uint64_t* d_offs, uint32_t* d_lens, void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (!size_ok(w, h) || (w & 1) || (h & 1) || pix_stride < w * 3 || qbias > 255 || (n && !d_pix))
        return fail(c, AMVHIP_ERR_ARG, "encode: bad argument (width/height must be even)");
    if (n && (!d_blob || !d_offs || !d_lens)) return fail(c, AMVHIP_ERR_ARG, "encode: null output");
    if (n == 0) return AMVHIP_OK;
    if (int r = use_device(c)) return r;
    std::lock_guard<std::mutex> lk(c->mu);
    return encode_core(c, d_pix, pix_stride, is_bgr, nullptr, n, make_geom(w, h), qbias, d_blob, blob_cap, d_offs, d_lens,
                       (hipStream_t)stream);
}

static int encode_yuv_dev(amvhip_ctx* c, const uint8_t* d_y, const uint8_t* d_cb, const uint8_t* d_cr, uint32_t y_stride, uint32_t c_stride,
                          uint64_t y_frame_stride, uint64_t c_frame_stride, uint32_t rows422, uint32_t n, uint32_t w, uint32_t h, uint32_t qbias,
                          uint8_t* d_blob, uint64_t blob_cap, uint64_t* d_offs, uint32_t* d_lens, void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (!size_ok(w, h) || (w & 1) || (h & 1) || y_stride < w || c_stride < w / 2 || qbias > 255 ||
        (n && (!d_y || !d_cb || !d_cr || !d_blob || !d_offs || !d_lens)))
        return fail(c, AMVHIP_ERR_ARG, "encode_yuv: bad argument (width/height must be even)");
    if (n == 0) return AMVHIP_OK;
    if (int r = use_device(c)) return r;
    std::lock_guard<std::mutex> lk(c->mu);
    const YuvSource yuv{d_y, d_cb, d_cr, y_stride, c_stride, y_frame_stride, c_frame_stride, rows422};
    return encode_core(c, nullptr, 0u, 0, &yuv, n, make_geom(w, h), qbias, d_blob, blob_cap, d_offs, d_lens, (hipStream_t)stream);
}

extern "C" int amvhip_encode_yuv420_batch_dev(amvhip_ctx* c, const uint8_t* d_y, const uint8_t* d_cb, const uint8_t* d_cr,
                                              uint32_t y_stride, uint32_t c_stride, uint64_t y_frame_stride,
                                              uint64_t c_frame_stride, uint32_t n, uint32_t w, uint32_t h, uint32_t qbias,
                                              uint8_t* d_blob, uint64_t blob_cap, uint64_t* d_offs, uint32_t* d_lens,
                                              void* stream) {
    return encode_yuv_dev(c, d_y, d_cb, d_cr, y_stride, c_stride, y_frame_stride, c_frame_stride, 0u, n, w, h, qbias, d_blob, blob_cap, d_offs,
                          d_lens, stream);
}

extern "C" int amvhip_encode_yuv422_batch_dev(amvhip_ctx* c, const uint8_t* d_y, const uint8_t* d_cb, const uint8_t* d_cr,
                                              uint32_t y_stride, uint32_t c_stride, uint64_t y_frame_stride,
                                              uint64_t c_frame_stride, uint32_t n, uint32_t w, uint32_t h, uint32_t qbias,
                                              uint8_t* d_blob, uint64_t blob_cap, uint64_t* d_offs, uint32_t* d_lens,
                                              void* stream) {
    return encode_yuv_dev(c, d_y, d_cb, d_cr, y_stride, c_stride, y_frame_stride, c_frame_stride, 1u, n, w, h, qbias, d_blob, blob_cap, d_offs,
                          d_lens, stream);
}

// the device-to-host half of the host-buffer encoders: offs/lens, then the chunks
static int encode_fetch(amvhip_ctx* c, hipStream_t hs, uint32_t n, uint8_t* blob, uint64_t blob_cap, uint64_t* offs, uint32_t* lens) {
    int32_t overflow = 0;
    HIP_TRY(c, hipMemcpyAsync(offs, c->h_offs.p, (size_t)n * 8, hipMemcpyDeviceToHost, hs));
    HIP_TRY(c, hipMemcpyAsync(lens, c->h_lens.p, (size_t)n * 4, hipMemcpyDeviceToHost, hs));
    HIP_TRY(c, hipMemcpyAsync(&overflow, c->flag.p, 4, hipMemcpyDeviceToHost, hs));
    HIP_TRY(c, hipStreamSynchronize(hs));
    const uint64_t total = offs[n - 1] + lens[n - 1];
    if (overflow || total > blob_cap)
        return fail(c, AMVHIP_ERR_SPACE, "encode: the chunks need more than the %llu bytes of blob", (unsigned long long)blob_cap);
    HIP_TRY(c, hipMemcpyAsync(blob, c->h_out.p, total, hipMemcpyDeviceToHost, hs));
    HIP_TRY(c, hipStreamSynchronize(hs));
    return AMVHIP_OK;
}

extern "C" int amvhip_encode_batch(amvhip_ctx* c, const uint8_t* pix, uint32_t pix_stride, int is_bgr, uint32_t n,
                                   uint32_t w, uint32_t h, uint32_t qbias, uint8_t* blob, uint64_t blob_cap,
                                   uint64_t* offs, uint32_t* lens) {
    if (!c) return AMVHIP_ERR_ARG;
    if (n && (!pix || !blob || !offs || !lens)) return fail(c, AMVHIP_ERR_ARG, "encode: null argument");
    if (n == 0) return AMVHIP_OK;
    if (int r = use_device(c)) return r;
    hipStream_t hs;
    if (int r = host_stream(c, &hs)) return r;
    const size_t in_bytes = (size_t)pix_stride * h * n;
    std::lock_guard<std::mutex> hlk(c->hmu);   // the staging buffers: one host-buffer call at a time
    if (int r = ensure(c, c->h_in, in_bytes)) return r;
    if (int r = ensure(c, c->h_out, blob_cap + 16)) return r;
    if (int r = ensure(c, c->h_offs, (size_t)n * 8)) return r;
    if (int r = ensure(c, c->h_lens, (size_t)n * 4)) return r;
    HIP_TRY(c, hipMemcpyAsync(c->h_in.p, pix, in_bytes, hipMemcpyHostToDevice, hs));
    if (int r = amvhip_encode_batch_dev(c, (const uint8_t*)c->h_in.p, pix_stride, is_bgr, n, w, h, qbias,
                                        (uint8_t*)c->h_out.p, blob_cap, (uint64_t*)c->h_offs.p,
                                        (uint32_t*)c->h_lens.p, hs))
        return r;
    return encode_fetch(c, hs, n, blob, blob_cap, offs, lens);
}

static int encode_yuv_host(amvhip_ctx* c, const uint8_t* y, const uint8_t* cb, const uint8_t* cr, uint32_t y_stride, uint32_t c_stride,
                           uint64_t y_frame_stride, uint64_t c_frame_stride, uint32_t rows422, uint32_t n, uint32_t w, uint32_t h, uint32_t qbias,
                           uint8_t* blob, uint64_t blob_cap, uint64_t* offs, uint32_t* lens) {
    if (!c) return AMVHIP_ERR_ARG;
    if (!size_ok(w, h) || (w & 1) || (h & 1) || y_stride < w || c_stride < w / 2 || (n && (!y || !cb || !cr || !blob || !offs || !lens)))
        return fail(c, AMVHIP_ERR_ARG, "encode_yuv420: bad argument");
    if (n == 0) return AMVHIP_OK;
    if (int r = use_device(c)) return r;
    hipStream_t hs;
    if (int r = host_stream(c, &hs)) return r;
    // staged tight: Y w*h, Cb, Cr (w/2 x h/2, or w/2 x h when the source is 4:2:2) per frame
    const uint32_t cw = w / 2, chh = rows422 ? h : h / 2;
    const uint64_t fb = (uint64_t)w * h + 2ull * cw * chh;
    std::lock_guard<std::mutex> hlk(c->hmu);   // the staging buffers: one host-buffer call at a time
    if (int r = ensure(c, c->h_in, fb * n)) return r;
    if (int r = ensure(c, c->h_out, blob_cap + 16)) return r;
    if (int r = ensure(c, c->h_offs, (size_t)n * 8)) return r;
    if (int r = ensure(c, c->h_lens, (size_t)n * 4)) return r;
    uint8_t* d = (uint8_t*)c->h_in.p;
    for (uint32_t i = 0; i < n; ++i) {
        HIP_TRY(c, hipMemcpy2DAsync(d + i * fb, w, y + i * y_frame_stride, y_stride, w, h, hipMemcpyHostToDevice, hs));
        HIP_TRY(c, hipMemcpy2DAsync(d + i * fb + (uint64_t)w * h, cw, cb + i * c_frame_stride, c_stride, cw, chh, hipMemcpyHostToDevice, hs));
        HIP_TRY(c, hipMemcpy2DAsync(d + i * fb + (uint64_t)w * h + (uint64_t)cw * chh, cw, cr + i * c_frame_stride, c_stride, cw, chh, hipMemcpyHostToDevice, hs));
    }
    if (int r = encode_yuv_dev(c, d, d + (uint64_t)w * h, d + (uint64_t)w * h + (uint64_t)cw * chh, w, cw, fb, fb, rows422, n, w, h, qbias,
                               (uint8_t*)c->h_out.p, blob_cap, (uint64_t*)c->h_offs.p, (uint32_t*)c->h_lens.p, hs))
        return r;
    return encode_fetch(c, hs, n, blob, blob_cap, offs, lens);
}

extern "C" int amvhip_encode_yuv420_batch(amvhip_ctx* c, const uint8_t* y, const uint8_t* cb, const uint8_t* cr,
                                          uint32_t y_stride, uint32_t c_stride, uint64_t y_frame_stride,
                                          uint64_t c_frame_stride, uint32_t n, uint32_t w, uint32_t h, uint32_t qbias,
                                          uint8_t* blob, uint64_t blob_cap, uint64_t* offs, uint32_t* lens) {
    return encode_yuv_host(c, y, cb, cr, y_stride, c_stride, y_frame_stride, c_frame_stride, 0u, n, w, h, qbias, blob, blob_cap, offs, lens);
}

extern "C" int amvhip_encode_yuv422_batch(amvhip_ctx* c, const uint8_t* y, const uint8_t* cb, const uint8_t* cr,
                                          uint32_t y_stride, uint32_t c_stride, uint64_t y_frame_stride,
                                          uint64_t c_frame_stride, uint32_t n, uint32_t w, uint32_t h, uint32_t qbias,
                                          uint8_t* blob, uint64_t blob_cap, uint64_t* offs, uint32_t* lens) {
    return encode_yuv_host(c, y, cb, cr, y_stride, c_stride, y_frame_stride, c_frame_stride, 1u, n, w, h, qbias, blob, blob_cap, offs, lens);
}

// =============================================================================================
// picture rescale
// =============================================================================================

// av_build_filter(filter, factor, NB_TAPS = 4, NB_PHASES = 16, 1 << FILTER_BITS, type 0) -- libavcodec/resample2.c:93-140
// as img_resample_full_init calls it (imgresample.c:468-471): cubic, first-order derivative -0.5, every phase
// normalised to 256; host arithmetic in double / float exactly as there.
static void build_resample_filter(int16_t* filter, uint32_t out_size, uint32_t in_size) {
    double factor = (float)out_size / (float)in_size;
    if (factor > 1.0) factor = 1.0;                          // upsampling only interpolates
    for (int ph = 0; ph < 16; ++ph) {
        double tab[4], norm = 0;
        for (int i = 0; i < 4; ++i) {
            const float d = -0.5f;
            const double x = fabs(((double)(i - 1) - (double)ph / 16) * factor);
            const double y = x < 1.0 ? 1 - 3 * x * x + 2 * x * x * x + d * (-x * x + x * x * x)
                                     : d * (-4 + 8 * x - 5 * x * x + x * x * x);
            tab[i] = y;
            norm += y;
        }
        for (int i = 0; i < 4; ++i) {
            const long v = lrintf((float)(tab[i] * 256 / norm));
            filter[ph * 4 + i] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
        }
    }
}

extern "C" int amvhip_resample_yuv420_dev(amvhip_ctx* c, const uint8_t* d_src_y, const uint8_t* d_src_cb, const uint8_t* d_src_cr,
                                          uint32_t src_y_stride, uint32_t src_c_stride, uint64_t src_y_frame, uint64_t src_c_frame,
                                          uint32_t src_w, uint32_t src_h, uint8_t* d_dst_y, uint8_t* d_dst_cb, uint8_t* d_dst_cr,
                                          uint32_t dst_y_stride, uint32_t dst_c_stride, uint64_t dst_y_frame, uint64_t dst_c_frame,
                                          uint32_t dst_w, uint32_t dst_h, uint32_t n, void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (!size_ok(src_w, src_h) || !size_ok(dst_w, dst_h) || src_w < 2 || src_h < 2 || dst_w < 2 || dst_h < 2 ||
        src_y_stride < src_w || src_c_stride < src_w / 2 || dst_y_stride < dst_w || dst_c_stride < dst_w / 2 ||
        (n && (!d_src_y || !d_src_cb || !d_src_cr || !d_dst_y || !d_dst_cb || !d_dst_cr)))
        return fail(c, AMVHIP_ERR_ARG, "resample: bad argument");
    if (n == 0) return AMVHIP_OK;
    if (n > 65535u) return fail(c, AMVHIP_ERR_ARG, "resample: at most 65535 frames per call");
    if (int r = use_device(c)) return r;
    ResampleFilters f;
    f.h_incr = (int)(((uint64_t)src_w << 16) / dst_w);      // imgresample.c:465-466
    f.v_incr = (int)(((uint64_t)src_h << 16) / dst_h);
    build_resample_filter(f.h, dst_w, src_w);
    build_resample_filter(f.v, dst_h, src_h);
    ResamplePlanes src{const_cast<uint8_t*>(d_src_y), const_cast<uint8_t*>(d_src_cb), const_cast<uint8_t*>(d_src_cr), src_y_stride,
                       src_c_stride, src_y_frame, src_c_frame, src_w, src_h};
    ResamplePlanes dst{d_dst_y, d_dst_cb, d_dst_cr, dst_y_stride, dst_c_stride, dst_y_frame, dst_c_frame, dst_w, dst_h};
    launch_resample(src, dst, f, n, (hipStream_t)stream);
    return check_launch(c, "resample");
}

// rescale + encode in one call: what ffmpeg.c:757-814 does per picture (sws_scale, then avcodec_encode_video) for a
// source that is not the target size.  The rescaled planes live in the context's workspace.
extern "C" int amvhip_encode_yuv420_scaled_batch_dev(amvhip_ctx* c, const uint8_t* d_y, const uint8_t* d_cb, const uint8_t* d_cr,
                                                     uint32_t y_stride, uint32_t c_stride, uint64_t y_frame_stride,
                                                     uint64_t c_frame_stride, uint32_t src_w, uint32_t src_h, uint32_t n,
                                                     uint32_t w, uint32_t h, uint32_t qbias, uint8_t* d_blob, uint64_t blob_cap,
                                                     uint64_t* d_offs, uint32_t* d_lens, void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (!size_ok(w, h) || !size_ok(src_w, src_h) || (w & 1) || (h & 1) || qbias > 255 || (n && (!d_blob || !d_offs || !d_lens)))
        return fail(c, AMVHIP_ERR_ARG, "encode_scaled: bad argument (width/height must be even)");
    if (n == 0) return AMVHIP_OK;
    if (int r = use_device(c)) return r;
    const uint64_t fb = (uint64_t)w * h + 2ull * (w / 2) * (h / 2);
    // the rescaled planes are the context's: the lock is held from their allocation to the last launch that reads them
    std::lock_guard<std::mutex> lk(c->mu);
    if (int r = ensure(c, c->scaled, fb * n)) return r;
    uint8_t* p = (uint8_t*)c->scaled.p;
    if (int r = amvhip_resample_yuv420_dev(c, d_y, d_cb, d_cr, y_stride, c_stride, y_frame_stride, c_frame_stride, src_w, src_h, p,
                                           p + (uint64_t)w * h, p + (uint64_t)w * h + (uint64_t)(w / 2) * (h / 2), w, w / 2, fb, fb, w, h,
                                           n, stream))
        return r;
    const YuvSource yuv{p, p + (uint64_t)w * h, p + (uint64_t)w * h + (uint64_t)(w / 2) * (h / 2), w, w / 2, fb, fb, 0u};
    return encode_core(c, nullptr, 0u, 0, &yuv, n, make_geom(w, h), qbias, d_blob, blob_cap, d_offs, d_lens, (hipStream_t)stream);
}

// =============================================================================================
// ADPCM
// =============================================================================================

extern "C" int amvhip_adpcm_decode_batch_dev(amvhip_ctx* c, const uint8_t* d_blob, uint64_t blob_bytes,
                                             const uint64_t* d_offs, const uint32_t* d_lens, uint32_t n,
                                             int16_t* d_pcm, const uint64_t* d_pcm_offs,
                                             int32_t* d_final_state, void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (n && (!d_blob || !d_offs || !d_lens || !d_pcm || !d_pcm_offs)) return fail(c, AMVHIP_ERR_ARG, "adpcm_decode: null argument");
    if (int r = use_device(c)) return r;
    {
        Timed t(c, AMVHIP_K_ADPCM_DEC, (hipStream_t)stream);
        launch_adpcm_decode(d_blob, blob_bytes, d_offs, d_lens, n, d_pcm, d_pcm_offs, d_final_state, (hipStream_t)stream);
    }
    return check_launch(c, "adpcm_decode");
}

extern "C" int amvhip_adpcm_encode_batch_dev(amvhip_ctx* c, const int16_t* d_pcm, const uint64_t* d_pcm_offs,
                                             const uint32_t* d_nsamp, uint32_t n, const int32_t* d_step_in,
                                             uint8_t* d_blob, const uint64_t* d_offs, void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (n && (!d_pcm || !d_pcm_offs || !d_nsamp || !d_blob || !d_offs)) return fail(c, AMVHIP_ERR_ARG, "adpcm_encode: null argument");
    if (n == 0) return AMVHIP_OK;
    if (int r = use_device(c)) return r;
    std::lock_guard<std::mutex> lk(c->mu);
    Timed t(c, AMVHIP_K_ADPCM_ENC, (hipStream_t)stream);
    const uint32_t* need = nullptr;
    if (!d_step_in) {  // the reference's behaviour: step_index runs through the whole stream
        const size_t nb = adpcm_chain_blocks(n);
        if (int r = ensure(c, c->map, ((size_t)n + nb) * 96)) return r;
        if (int r = ensure(c, c->start, (nb + 1) * 4)) return r;    // the blocks' starts + a counter for the exhaustive route alone
        uint32_t* done = (uint32_t*)c->start.p + nb;
        if (c->adpcm_sweeps >= 0) {   // guessed starts + sweeps; the exhaustive route behind it runs only if they do not settle
            if (int r = ensure(c, c->chain, adpcm_chain_workspace(n))) return r;
            uint32_t sweeps = (uint32_t)c->adpcm_sweeps;
            if (!c->adpcm_sweeps_set) {
                // launched sweeps: until the list is expected to be a couple of hundred entries (it starts at ~0.41 n and
                // shrinks ~3.7x per sweep on ordinary audio; counted here as n shrinking 3.3x).  The front sweep behind them
                // (a workgroup per entry, four chunks looked ahead) and the one-workgroup settle kernel take the rest
                sweeps = 0;
                for (uint64_t left = n; left > 512u; left = left * 3u / 10u) ++sweeps;
                if (n <= 64u) sweeps = 0u;
            }
            c->chain_n = n;
            // (state, lists, counters and the flag live in the context's `chain` buffer: chained encodes of ONE context
            // must be ordered on the device -- one stream at a time, as for every _dev entry point; see amvhip.h)
            need = launch_adpcm_chain(d_pcm, d_pcm_offs, d_nsamp, n, d_blob, d_offs, c->chain.p, sweeps, c->adpcm_settle, (hipStream_t)stream);
            if (!need) return fail(c, AMVHIP_ERR_DEVICE, "adpcm_encode: clearing the chain counters failed");
            done = const_cast<uint32_t*>(need) - 1;   // zeroed with the flag
        } else {
            HIP_TRY(c, hipMemsetAsync(done, 0, 4, (hipStream_t)stream));
        }
        launch_adpcm_map(d_pcm, d_pcm_offs, d_nsamp, n, (uint8_t*)c->map.p, (int32_t*)c->start.p, done, need, (hipStream_t)stream);
        launch_adpcm_encode_mapped(d_pcm, d_pcm_offs, d_nsamp, n, (const uint8_t*)c->map.p, (const int32_t*)c->start.p, d_blob, d_offs, need,
                                   (hipStream_t)stream);
        return check_launch(c, "adpcm_encode");
    }
    launch_adpcm_encode(d_pcm, d_pcm_offs, d_nsamp, n, d_step_in, d_blob, d_offs, nullptr, (hipStream_t)stream);
    return check_launch(c, "adpcm_encode");
}

extern "C" void amvhip_adpcm_quotient_table(float out[89]) {
    if (out) adpcm_quotient_table(out);
}

extern "C" int amvhip_adpcm_chain_stats(amvhip_ctx* c, uint32_t out[64]) {
    if (!c || !out) return AMVHIP_ERR_ARG;
    if (int r = use_device(c)) return r;
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->chain_n) return fail(c, AMVHIP_ERR_ARG, "adpcm_chain_stats: no chained encode has run");
    uint32_t w[64];
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(w, (const uint8_t*)c->chain.p + (size_t)c->chain_n * 16, sizeof w, hipMemcpyDeviceToHost));
    out[0] = w[63];
    for (int k = 0; k < 62; ++k) out[k + 1] = w[k];
    out[63] = 0;
    return AMVHIP_OK;
}

extern "C" int amvhip_adpcm_decode_batch_async(amvhip_ctx* c, const uint8_t* blob, uint64_t blob_bytes,
                                               const uint64_t* offs, const uint32_t* lens, uint32_t n, int16_t* pcm,
                                               uint64_t pcm_samples, const uint64_t* pcm_offs, int32_t* final_state) {
    if (!c) return AMVHIP_ERR_ARG;
    if (n && (!blob || !offs || !lens || !pcm || !pcm_offs)) return fail(c, AMVHIP_ERR_ARG, "adpcm_decode: null argument");
    if (n == 0) return AMVHIP_OK;
    for (uint32_t i = 0; i < n; ++i)
        if (lens[i] > 8 && pcm_offs[i] + 2ull * (lens[i] - 8) > pcm_samples) return fail(c, AMVHIP_ERR_SPACE, "adpcm_decode: pcm too small for chunk %u", i);
    if (int r = use_device(c)) return r;
    hipStream_t st;
    if (int r = host_stream(c, &st)) return r;
    // audio staging sits behind the video staging of the same stream: separate buffers, so that a video batch and the
    // audio batch that travels with it can both be in flight
    std::lock_guard<std::mutex> hlk(c->hmu);   // the staging buffers: one host-buffer call at a time
    if (int r = ensure(c, c->a_in, blob_bytes + 16)) return r;
    if (int r = ensure(c, c->a_tab, (size_t)n * 28)) return r;
    if (int r = ensure(c, c->a_out, pcm_samples * 2)) return r;
    uint8_t* tab = (uint8_t*)c->a_tab.p;
    uint64_t* d_offs = (uint64_t*)tab;
    uint64_t* d_pcm_offs = (uint64_t*)(tab + (size_t)n * 8);
    int32_t* d_fin = (int32_t*)(tab + (size_t)n * 16);
    uint32_t* d_lens = (uint32_t*)(tab + (size_t)n * 24);
    HIP_TRY(c, hipMemcpyAsync(c->a_in.p, blob, blob_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d_offs, offs, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d_lens, lens, (size_t)n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(d_pcm_offs, pcm_offs, (size_t)n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(c, hipMemcpyAsync(c->a_out.p, pcm, pcm_samples * 2, hipMemcpyHostToDevice, st));  // keep untouched gaps
    if (int r = amvhip_adpcm_decode_batch_dev(c, (const uint8_t*)c->a_in.p, blob_bytes, d_offs, d_lens, n, (int16_t*)c->a_out.p,
                                              d_pcm_offs, d_fin, st))
        return r;
    HIP_TRY(c, hipMemcpyAsync(pcm, c->a_out.p, pcm_samples * 2, hipMemcpyDeviceToHost, st));
    if (final_state) HIP_TRY(c, hipMemcpyAsync(final_state, d_fin, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    return AMVHIP_OK;
}

extern "C" int amvhip_adpcm_decode_batch(amvhip_ctx* c, const uint8_t* blob, uint64_t blob_bytes,
                                         const uint64_t* offs, const uint32_t* lens, uint32_t n, int16_t* pcm,
                                         uint64_t pcm_samples, const uint64_t* pcm_offs, int32_t* final_state) {
    if (int r = amvhip_adpcm_decode_batch_async(c, blob, blob_bytes, offs, lens, n, pcm, pcm_samples, pcm_offs, final_state)) return r;
    return amvhip_sync(c);
}

extern "C" int amvhip_adpcm_encode_batch(amvhip_ctx* c, const int16_t* pcm, uint64_t pcm_samples,
                                         const uint64_t* pcm_offs, const uint32_t* nsamp, uint32_t n,
                                         const int32_t* step_in, uint8_t* blob, uint64_t blob_bytes,
                                         const uint64_t* offs) {
    if (!c) return AMVHIP_ERR_ARG;
    if (n && (!pcm || !pcm_offs || !nsamp || !blob || !offs)) return fail(c, AMVHIP_ERR_ARG, "adpcm_encode: null argument");
    if (n == 0) return AMVHIP_OK;
    for (uint32_t i = 0; i < n; ++i) {
        if (pcm_offs[i] + nsamp[i] > pcm_samples) return fail(c, AMVHIP_ERR_ARG, "adpcm_encode: chunk %u reads past pcm", i);
        if (offs[i] + 8ull + (nsamp[i] >> 1) > blob_bytes) return fail(c, AMVHIP_ERR_SPACE, "adpcm_encode: blob too small for chunk %u", i);
    }
    if (int r = use_device(c)) return r;
    hipStream_t hs;
    if (int r = host_stream(c, &hs)) return r;
    std::lock_guard<std::mutex> hlk(c->hmu);   // the staging buffers: one host-buffer call at a time
    if (int r = ensure(c, c->h_in, pcm_samples * 2 + 16)) return r;
    if (int r = ensure(c, c->h_offs, (size_t)n * 8)) return r;
    if (int r = ensure(c, c->h_lens, (size_t)n * 4)) return r;
    if (int r = ensure(c, c->h_out, blob_bytes)) return r;
    if (int r = ensure(c, c->h_aux, (size_t)n * 8)) return r;
    if (int r = ensure(c, c->h_status, (size_t)n * 4)) return r;
    HIP_TRY(c, hipMemcpyAsync(c->h_in.p, pcm, pcm_samples * 2, hipMemcpyHostToDevice, hs));
    HIP_TRY(c, hipMemcpyAsync(c->h_aux.p, pcm_offs, (size_t)n * 8, hipMemcpyHostToDevice, hs));
    HIP_TRY(c, hipMemcpyAsync(c->h_lens.p, nsamp, (size_t)n * 4, hipMemcpyHostToDevice, hs));
    HIP_TRY(c, hipMemcpyAsync(c->h_offs.p, offs, (size_t)n * 8, hipMemcpyHostToDevice, hs));
    HIP_TRY(c, hipMemcpyAsync(c->h_out.p, blob, blob_bytes, hipMemcpyHostToDevice, hs));
    if (step_in) HIP_TRY(c, hipMemcpyAsync(c->h_status.p, step_in, (size_t)n * 4, hipMemcpyHostToDevice, hs));
    if (int r = amvhip_adpcm_encode_batch_dev(c, (const int16_t*)c->h_in.p, (const uint64_t*)c->h_aux.p,
                                              (const uint32_t*)c->h_lens.p, n,
                                              step_in ? (const int32_t*)c->h_status.p : nullptr,
                                              (uint8_t*)c->h_out.p, (const uint64_t*)c->h_offs.p, hs))
        return r;
    HIP_TRY(c, hipMemcpyAsync(blob, c->h_out.p, blob_bytes, hipMemcpyDeviceToHost, hs));
    HIP_TRY(c, hipStreamSynchronize(hs));
    return AMVHIP_OK;
}

// The reference's `-trellis N` quality mode (adpcm_compress_trellis, adpcm.c:287-443) for independent chunks: every chunk
// starts from d_step_in[i] and reports the index it ends on in d_step_out[i] (optional).
extern "C" int amvhip_adpcm_encode_trellis_batch_dev(amvhip_ctx* c, const int16_t* d_pcm, const uint64_t* d_pcm_offs,
                                                     const uint32_t* d_nsamp, uint32_t n, const int32_t* d_step_in, uint32_t trellis,
                                                     uint8_t* d_blob, const uint64_t* d_offs, int32_t* d_step_out, void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (trellis < 1 || trellis > 5 || (n && (!d_pcm || !d_pcm_offs || !d_nsamp || !d_step_in || !d_blob || !d_offs)))
        return fail(c, AMVHIP_ERR_ARG, "adpcm_encode_trellis: bad argument (trellis 1..5, start indices required)");
    if (n == 0) return AMVHIP_OK;
    if (int r = use_device(c)) return r;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int r = ensure(c, c->trellis_ws, adpcm_trellis_workspace(n, trellis))) return r;
    Timed t(c, AMVHIP_K_ADPCM_ENC, (hipStream_t)stream);
    if (!launch_adpcm_trellis(d_pcm, d_pcm_offs, d_nsamp, n, d_step_in, trellis, d_blob, d_offs, d_step_out, (uint16_t*)c->trellis_ws.p,
                              (hipStream_t)stream))
        return fail(c, AMVHIP_ERR_DEVICE, "adpcm_encode_trellis: kernel attributes refused");
    return check_launch(c, "adpcm_trellis");
}

// One AMV audio chunk with the step index handed in and out: what adpcm_encode_frame (adpcm.c:461-498) does per
// call with the index it keeps in its context.  The end index is read off a decode of the fresh chunk (the decoder
// walks the same index chain), one synchronisation for both kernels.
static int adpcm_encode_frame_impl(amvhip_ctx* c, const int16_t* samples, uint32_t nsamp, int32_t* step_index, uint32_t trellis,
                                   uint8_t* chunk, uint32_t cap);

extern "C" int amvhip_adpcm_encode_frame(amvhip_ctx* c, const int16_t* samples, uint32_t nsamp, int32_t* step_index,
                                         uint8_t* chunk, uint32_t cap) {
    return adpcm_encode_frame_impl(c, samples, nsamp, step_index, 0u, chunk, cap);
}

extern "C" int amvhip_adpcm_encode_frame_trellis(amvhip_ctx* c, const int16_t* samples, uint32_t nsamp, int32_t* step_index,
                                                 uint32_t trellis, uint8_t* chunk, uint32_t cap) {
    if (trellis < 1 || trellis > 5) return c ? fail(c, AMVHIP_ERR_ARG, "adpcm_encode_frame_trellis: trellis 1..5") : AMVHIP_ERR_ARG;
    return adpcm_encode_frame_impl(c, samples, nsamp, step_index, trellis, chunk, cap);
}

static int adpcm_encode_frame_impl(amvhip_ctx* c, const int16_t* samples, uint32_t nsamp, int32_t* step_index, uint32_t trellis,
                                   uint8_t* chunk, uint32_t cap) {
    if (!c) return AMVHIP_ERR_ARG;
    const uint32_t len = 8u + (nsamp >> 1);
    if (!samples || !step_index || !chunk || (nsamp & 1u) || nsamp == 0 || *step_index < 0 || *step_index > 88)
        return fail(c, AMVHIP_ERR_ARG, "adpcm_encode_frame: bad argument (even, non-zero sample count; index 0..88)");
    if (cap < len) return fail(c, AMVHIP_ERR_SPACE, "adpcm_encode_frame: chunk needs %u bytes", len);
    if (int r = use_device(c)) return r;
    hipStream_t hs;
    if (int r = host_stream(c, &hs)) return r;
    // staging: [pcm | chunk | scratch pcm] + small tables {pcm_off, chunk_off, nsamp, len, step, final[2]}
    std::lock_guard<std::mutex> hlk(c->hmu);   // the staging buffers: one host-buffer call at a time
    if (int r = ensure(c, c->h_in, (size_t)nsamp * 2 + 16)) return r;
    if (int r = ensure(c, c->h_out, (size_t)len + 16 + (size_t)nsamp * 2 + 16)) return r;
    if (int r = ensure(c, c->h_aux, 64)) return r;
    struct { uint64_t pcm_off, chunk_off; uint32_t nsamp, len; int32_t step; int32_t final_state[2]; } tab = {0, 0, nsamp, len, *step_index, {0, 0}};
    uint8_t* aux = (uint8_t*)c->h_aux.p;
    uint8_t* d_chunk = (uint8_t*)c->h_out.p;
    int16_t* d_scratch = (int16_t*)(d_chunk + ((len + 15u) & ~15u));
    HIP_TRY(c, hipMemcpyAsync(c->h_in.p, samples, (size_t)nsamp * 2, hipMemcpyHostToDevice, hs));
    HIP_TRY(c, hipMemcpyAsync(aux, &tab, sizeof tab, hipMemcpyHostToDevice, hs));
    if (trellis) {
        if (int r = amvhip_adpcm_encode_trellis_batch_dev(c, (const int16_t*)c->h_in.p, (const uint64_t*)aux, (const uint32_t*)(aux + 16), 1,
                                                          (const int32_t*)(aux + 24), trellis, d_chunk, (const uint64_t*)(aux + 8), nullptr, hs))
            return r;
    } else if (int r = amvhip_adpcm_encode_batch_dev(c, (const int16_t*)c->h_in.p, (const uint64_t*)aux, (const uint32_t*)(aux + 16), 1,
                                                     (const int32_t*)(aux + 24), d_chunk, (const uint64_t*)(aux + 8), hs)) {
        return r;
    }
    if (int r = amvhip_adpcm_decode_batch_dev(c, d_chunk, len, (const uint64_t*)(aux + 8), (const uint32_t*)(aux + 20), 1, d_scratch,
                                              (const uint64_t*)aux, (int32_t*)(aux + 28), hs))
        return r;
    int32_t fin[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(chunk, d_chunk, len, hipMemcpyDeviceToHost, hs));
    HIP_TRY(c, hipMemcpyAsync(fin, aux + 28, 8, hipMemcpyDeviceToHost, hs));
    HIP_TRY(c, hipStreamSynchronize(hs));
    *step_index = fin[1];
    return (int)len;
}

// (amvhip_amv_audio_pairs / amvhip_amv_audio_frame_size, the AMV audio framing in host arithmetic, live in
// host/amv_container.c: plain C, no device -- they are part of every link of the host side, the FFmpeg one included)

extern "C" int amvhip_adpcm_wav_encode_frame(amvhip_ctx* c, const int16_t* samples, int frame_size,
                                             int32_t state[2], uint8_t* frame, int buf_size) {
    if (!c) return AMVHIP_ERR_ARG;
    const int groups = frame_size / 8;   // AdpcmIma.c:105
    if (!samples || !state || !frame || frame_size < 1 || buf_size < 4 + 4 * groups)
        return fail(c, AMVHIP_ERR_ARG, "adpcm_wav_encode: bad argument");
    if (int r = use_device(c)) return r;
    hipStream_t hs;
    if (int r = host_stream(c, &hs)) return r;
    const size_t ns = (size_t)1 + 8 * (size_t)groups;
    std::lock_guard<std::mutex> hlk(c->hmu);   // the staging buffers: one host-buffer call at a time
    if (int r = ensure(c, c->h_in, ns * 2)) return r;
    if (int r = ensure(c, c->h_out, 4 + 4 * (size_t)groups)) return r;
    if (int r = ensure(c, c->h_status, 8)) return r;
    HIP_TRY(c, hipMemcpyAsync(c->h_in.p, samples, ns * 2, hipMemcpyHostToDevice, hs));
    HIP_TRY(c, hipMemcpyAsync(c->h_status.p, state, 8, hipMemcpyHostToDevice, hs));
    launch_adpcm_wav_encode((const int16_t*)c->h_in.p, groups, (int32_t*)c->h_status.p, (uint8_t*)c->h_out.p, hs);
    if (int r = check_launch(c, "adpcm_wav_encode")) return r;
    HIP_TRY(c, hipMemcpyAsync(frame, c->h_out.p, 4 + 4 * (size_t)groups, hipMemcpyDeviceToHost, hs));
    HIP_TRY(c, hipMemcpyAsync(state, c->h_status.p, 8, hipMemcpyDeviceToHost, hs));
    HIP_TRY(c, hipStreamSynchronize(hs));
    return 4 + 4 * groups;
}

// =============================================================================================
// synthetic sources
// =============================================================================================

extern "C" int amvhip_synth_frames_dev(amvhip_ctx* c, uint32_t seed, uint32_t first, uint32_t n, uint32_t w,
                                       uint32_t h, uint8_t* d_rgb, void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (!size_ok(w, h) || (n && !d_rgb)) return fail(c, AMVHIP_ERR_ARG, "synth: bad argument");
    if (int r = use_device(c)) return r;
    {
        Timed t(c, AMVHIP_K_SYNTH, (hipStream_t)stream);
        launch_synth_frames(seed, first, n, w, h, d_rgb, (hipStream_t)stream);
    }
    return check_launch(c, "synth_frames");
}

extern "C" int amvhip_synth_audio_dev(amvhip_ctx* c, uint32_t seed, uint64_t first, uint64_t n, int16_t* d_pcm,
                                      void* stream) {
    if (!c) return AMVHIP_ERR_ARG;
    if (n && !d_pcm) return fail(c, AMVHIP_ERR_ARG, "synth: bad argument");
    if (int r = use_device(c)) return r;
    launch_synth_audio(seed, first, n, d_pcm, (hipStream_t)stream);
    return check_launch(c, "synth_audio");
}

// =============================================================================================
// timing
// =============================================================================================

extern "C" int amvhip_set_entropy_mode(amvhip_ctx* c, int mode) {
    if (!c || (mode != AMVHIP_ENTROPY_AUTO && mode != AMVHIP_ENTROPY_SERIAL)) return AMVHIP_ERR_ARG;
    c->entropy_mode = mode;
    return AMVHIP_OK;
}

extern "C" int amvhip_entropy_stats(amvhip_ctx* c, int enable, uint64_t out[10]) {
    if (!c) return AMVHIP_ERR_ARG;
    if (int r = use_device(c)) return r;
    std::lock_guard<std::mutex> lk(c->mu);
    if (int r = ensure(c, c->stats, amv::kStatsBytes)) return r;
    HIP_TRY(c, hipDeviceSynchronize());
    if (out) {
        if (c->want_stats) HIP_TRY(c, hipMemcpy(out, c->stats.p, 80, hipMemcpyDeviceToHost));
        else memset(out, 0, 80);
        // frames of the LAST decode call that the synchronising kernel handed to the one-lane-per-frame kernel (chunk
        // over the workspace window, long FF run, more records than the record space holds: see blob_bytes in amvhip.h)
        uint32_t handed = 0;
        if (c->last_decode_retry && c->last_decode_retry->p) HIP_TRY(c, hipMemcpy(&handed, c->last_decode_retry->p, 4, hipMemcpyDeviceToHost));
        out[3] = handed;
    }
    // the counters always; the per-task lines too when gathering goes on, so that amvhip_entropy_trace finds zeros behind the
    // tasks of the launches that follow (a fresh allocation holds whatever the pool last kept there)
    HIP_TRY(c, hipMemset(c->stats.p, 0, enable ? amv::kStatsBytes : 128));
    c->want_stats = enable != 0;
    return AMVHIP_OK;
}

extern "C" int amvhip_entropy_trace(amvhip_ctx* c, uint64_t* out, uint32_t tasks) {
    if (!c || (tasks && !out)) return AMVHIP_ERR_ARG;
    if (int r = use_device(c)) return r;
    std::lock_guard<std::mutex> lk(c->mu);
    if (tasks > amv::kTraceTasks) tasks = amv::kTraceTasks;
    if (!c->stats.p || c->stats.cap < amv::kStatsBytes) return fail(c, AMVHIP_ERR_ARG, "entropy_trace: gathering was never switched on");
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(out, (const uint64_t*)c->stats.p + amv::kTraceBase, (size_t)tasks * 64, hipMemcpyDeviceToHost));
    return (int)tasks;
}

extern "C" int amvhip_decode_split_stats(amvhip_ctx* c, uint32_t out[2]) {
    if (!c || !out) return AMVHIP_ERR_ARG;
    if (int r = use_device(c)) return r;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(c, hipDeviceSynchronize());
    out[0] = out[1] = 0;
    if (c->last_split && c->split.p) HIP_TRY(c, hipMemcpy(out, c->split.p, 8, hipMemcpyDeviceToHost));
    return AMVHIP_OK;
}

extern "C" void amvhip_prof_enable(amvhip_ctx* c, int on) { if (c) c->prof = on != 0; }

extern "C" void amvhip_prof_reset(amvhip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    drain(c);
    for (int k = 0; k < AMVHIP_K_COUNT; ++k) { c->launches[k] = 0; c->total_ms[k] = 0; }
}

extern "C" int amvhip_prof_read(amvhip_ctx* c, int kernel, uint64_t* launches, double* total_ms) {
    if (!c || kernel < 0 || kernel >= AMVHIP_K_COUNT) return AMVHIP_ERR_ARG;
    (void)hipSetDevice(c->device);
    drain(c);
    if (launches) *launches = c->launches[kernel];
    if (total_ms) *total_ms = c->total_ms[kernel];
    return AMVHIP_OK;
}

// The JPEG file header amvlib puts in front of a chunk's scan (AmvJpegPutHeader, AmvJpeg.c:315-414):
// SOI, JFIF APP0 (:282-313), two DQT segments with the fixed tables (:162-213), SOF0 4:2:0, the four
// K.3 Huffman tables (:245-279), SOS.  Host-only; the tables come from amv_tables.h.
extern "C" uint32_t amvhip_jpeg_header(uint16_t height, uint16_t width, uint8_t* out, uint32_t cap) {
    std::vector<uint8_t> b;
    auto put = [&](std::initializer_list<int> v) { for (int x : v) b.push_back((uint8_t)x); };
    put({0xff, 0xd8});                                                              // SOI
    put({0xff, 0xe0, 0x00, 0x10, 'J', 'F', 'I', 'F', 0x00, 0x01, 0x01, 0x01, 0x00, 0x60, 0x00, 0x60, 0x00, 0x00});
    for (int t = 0; t < 2; ++t) {                                                   // DQT, 8-bit, table t
        put({0xff, 0xdb, 0x00, 2 + 1 + 64, t});
        for (int i = 0; i < 64; ++i) b.push_back(t ? kQuantChroma[i] : kQuantLuma[i]);
    }
    put({0xff, 0xc0, 0, 17, 8, height >> 8, height & 0xff, width >> 8, width & 0xff, 3,
         1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1});                                      // SOF0
    const int order[4] = {0, 2, 1, 3};                                              // DC luma, AC luma, DC chroma, AC chroma (:247-278)
    for (int j = 0; j < 4; ++j) {
        const int t = order[j];
        int nsym = 0;
        for (int l = 0; l < 16; ++l) nsym += kHuffCount[t][l];
        const int len = 2 + 1 + 16 + nsym;                                          // 0x1F / 0xB5
        put({0xff, 0xc4, len >> 8, len & 0xff, ((t >= 2 ? 1 : 0) << 4) | (t & 1)});
        for (int l = 0; l < 16; ++l) b.push_back(kHuffCount[t][l]);
        const uint8_t* syms = symbols_of(t);
        for (int i = 0; i < nsym; ++i) b.push_back(syms[i]);
    }
    put({0xff, 0xda, 0, 12, 3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0});               // SOS
    if (out && cap >= b.size()) memcpy(out, b.data(), b.size());
    return (uint32_t)b.size();
}

extern "C" const char* amvhip_kernel_name(int kernel) {
    switch (kernel) {
        case AMVHIP_K_HUFFMAN: return "amv_huffman_fast_kernel|sync2|sync";   // whichever the batch got (huffman_sync_lanes; dense form: sync)
        case AMVHIP_K_UNSTUFF: return "amv_unstuff_kernel";
        case AMVHIP_K_HUFFMAN_SERIAL: return "amv_huffman_kernel";
        case AMVHIP_K_RECON: return "amv_reconstruct_kernel";
        case AMVHIP_K_FDCT: return "amv_forward_kernel";
        case AMVHIP_K_PACK: return "amv_encode_frame_kernel";
        case AMVHIP_K_PACK_SERIAL: return "amv_pack_kernel";
        case AMVHIP_K_COMPACT: return "amv_scan_kernel+amv_gather_kernel";
        case AMVHIP_K_ADPCM_DEC: return "amv_adpcm_decode_kernel";
        case AMVHIP_K_ADPCM_ENC: return "amv_adpcm_guess_kernel+amv_adpcm_sweep_kernel*+front+settle+check (+map, chain, encode_mapped when the chain does not settle)";
        case AMVHIP_K_SYNTH: return "amv_synth_frames_kernel";
        default: return "";
    }
}
