/*
 * host_stub.c -- TEST INFRASTRUCTURE: stands in for the device half of libamvhip so that the product's HOST C
 * (amv-codec-tools_amd/host/amvlib_compat.c, amv_container.c, and amvhip_lavc.c where the reference's avcodec.h is
 * at hand) can be built with -fsanitize=address,undefined and walked over hostile inputs on a machine without a GPU
 * (tests/test_abi_and_host.py::test_host_c_under_sanitizers).  Two modes, HOST_STUB_MODE in the environment:
 *   "fail" (default)  every device call fails the way a machine without a HIP device makes it fail
 *                     (amvhip_create -> AMVHIP_ERR_DEVICE): the paths a reader takes when nothing can be decoded;
 *   "zero"            device calls succeed and deliver zeros of the right SIZE (frames, samples, a minimal chunk):
 *                     the read-ahead windows, the lent pointers and the plugin's plane copies run in full, and a
 *                     wrong size anywhere is a sanitizer report.  Nothing here computes a codec result.
 * The pure-arithmetic entry points of the device half (sizes, strides) are restated: the host C's buffer sizes depend on
 * them.  (The AMV audio framing arithmetic is host C of the product, host/amv_container.c, and is linked, not restated.)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "amvhip.h"

struct amvhip_ctx { int device; };
static struct amvhip_ctx g_one;

static int zero_mode(void)
{
    const char *m = getenv("HOST_STUB_MODE");
    return m != NULL && strcmp(m, "zero") == 0;
}

int amvhip_create(amvhip_ctx **ctx, int device)
{
    if (ctx == NULL) return AMVHIP_ERR_ARG;
    *ctx = NULL;
    if (!zero_mode()) return AMVHIP_ERR_DEVICE;
    g_one.device = device;
    *ctx = &g_one;
    return AMVHIP_OK;
}
void amvhip_destroy(amvhip_ctx *ctx) { (void)ctx; }
const char *amvhip_last_error(const amvhip_ctx *ctx) { (void)ctx; return "stub"; }

uint32_t amvhip_stride(uint32_t width) { return ((width * 24u + 31u) / 32u) * 4u; }            /* AmvJpeg.c:1524 */
uint64_t amvhip_frame_bytes(uint32_t width, uint32_t height) { return (uint64_t)amvhip_stride(width) * height; }
uint64_t amvhip_yuv420_frame_bytes(uint32_t w, uint32_t h) { return (uint64_t)w * h + 2ull * ((w + 1) / 2) * ((h + 1) / 2); }
uint32_t amvhip_encode_bound(uint32_t w, uint32_t h) { return 4u + ((w + 15u) / 16u) * ((h + 15u) / 16u) * 6u * 64u * 4u + 64u; }
uint32_t amvhip_jpeg_header(uint16_t height, uint16_t width, uint8_t *out, uint32_t cap)
{
    if (out != NULL && cap >= 623u) {
        const uint32_t sof = 2 + 18 + 2 * 69;            /* where the real header has its SOF0 (amvlib_compat.c reads the size back from there) */
        memset(out, 0x5a, 623u);
        out[sof + 5] = (uint8_t)(height >> 8); out[sof + 6] = (uint8_t)height;
        out[sof + 7] = (uint8_t)(width >> 8); out[sof + 8] = (uint8_t)width;
    }
    return 623u;
}

int amvhip_sync(amvhip_ctx *ctx) { return ctx ? AMVHIP_OK : AMVHIP_ERR_ARG; }
int amvhip_host_alloc(amvhip_ctx *ctx, void **p, size_t bytes)
{
    if (ctx == NULL || p == NULL) return AMVHIP_ERR_ARG;
    *p = malloc(bytes ? bytes : 1);                       /* exact size: an overrun of a window buffer is a report */
    return *p ? AMVHIP_OK : AMVHIP_ERR_NOMEM;
}
void amvhip_host_free(amvhip_ctx *ctx, void *p) { (void)ctx; free(p); }

int amvhip_decode_batch_async(amvhip_ctx *ctx, const uint8_t *blob, uint64_t blob_bytes, const uint64_t *offs, const uint32_t *lens,
                              uint32_t n, uint32_t w, uint32_t h, uint32_t flags, uint8_t *out, int32_t *status)
{
    uint32_t i;
    volatile uint8_t sink = 0;
    if (ctx == NULL) return AMVHIP_ERR_DEVICE;
    for (i = 0; i < n; i++) {                             /* touch what the device would read: first and last byte of every chunk */
        if (offs[i] > blob_bytes || lens[i] > blob_bytes - offs[i]) { if (status) status[i] = 1; continue; }
        if (lens[i]) sink ^= (uint8_t)(blob[offs[i]] ^ blob[offs[i] + lens[i] - 1]);
        if (status) status[i] = 0;
    }
    memset(out, 0, (size_t)((flags & AMVHIP_FLAG_FFMPEG) ? amvhip_yuv420_frame_bytes(w, h) : amvhip_frame_bytes(w, h)) * n);
    (void)sink;
    return AMVHIP_OK;
}
int amvhip_decode_batch(amvhip_ctx *ctx, const uint8_t *blob, uint64_t blob_bytes, const uint64_t *offs, const uint32_t *lens,
                        uint32_t n, uint32_t w, uint32_t h, uint32_t flags, uint8_t *out, int32_t *status)
{
    return amvhip_decode_batch_async(ctx, blob, blob_bytes, offs, lens, n, w, h, flags, out, status);
}

int amvhip_adpcm_decode_batch_async(amvhip_ctx *ctx, const uint8_t *blob, uint64_t blob_bytes, const uint64_t *offs,
                                    const uint32_t *lens, uint32_t n, int16_t *pcm, uint64_t pcm_samples, const uint64_t *pcm_offs,
                                    int32_t *final_state)
{
    uint32_t i;
    volatile uint8_t sink = 0;
    if (ctx == NULL) return AMVHIP_ERR_DEVICE;
    for (i = 0; i < n; i++) {
        if (offs[i] > blob_bytes || lens[i] > blob_bytes - offs[i] || lens[i] <= 8) continue;
        sink ^= (uint8_t)(blob[offs[i]] ^ blob[offs[i] + lens[i] - 1]);
        if (pcm_offs[i] + 2ull * (lens[i] - 8) > pcm_samples) return AMVHIP_ERR_SPACE;
        memset(pcm + pcm_offs[i], 0, 4u * (size_t)(lens[i] - 8));
        if (final_state) { final_state[2 * i] = 0; final_state[2 * i + 1] = 0; }
    }
    (void)sink;
    return AMVHIP_OK;
}
int amvhip_adpcm_decode_batch(amvhip_ctx *ctx, const uint8_t *blob, uint64_t blob_bytes, const uint64_t *offs, const uint32_t *lens,
                              uint32_t n, int16_t *pcm, uint64_t pcm_samples, const uint64_t *pcm_offs, int32_t *final_state)
{
    return amvhip_adpcm_decode_batch_async(ctx, blob, blob_bytes, offs, lens, n, pcm, pcm_samples, pcm_offs, final_state);
}

static int tiny_chunks(uint32_t n, uint8_t *blob, uint64_t cap, uint64_t *offs, uint32_t *lens)
{
    uint32_t i;
    for (i = 0; i < n; i++) {
        if ((uint64_t)(i + 1) * 8u > cap) return AMVHIP_ERR_SPACE;
        memcpy(blob + 8u * i, "\xff\xd8\x00\x00\x00\x00\xff\xd9", 8);
        offs[i] = 8ull * i;
        lens[i] = 8;
    }
    return AMVHIP_OK;
}
int amvhip_encode_batch(amvhip_ctx *ctx, const uint8_t *pix, uint32_t pix_stride, int is_bgr, uint32_t n, uint32_t w, uint32_t h,
                        uint32_t qbias, uint8_t *blob, uint64_t cap, uint64_t *offs, uint32_t *lens)
{
    volatile uint8_t sink = 0;
    (void)is_bgr; (void)qbias;
    if (ctx == NULL) return AMVHIP_ERR_DEVICE;
    if (n && h) sink ^= (uint8_t)(pix[0] ^ pix[(size_t)(h - 1) * pix_stride + (size_t)w * 3u - 1u]);
    (void)sink;
    return tiny_chunks(n, blob, cap, offs, lens);
}
static int planes_touch(const uint8_t *y, const uint8_t *cb, const uint8_t *cr, uint32_t ys, uint32_t cs, uint32_t w, uint32_t h, uint32_t ch)
{
    volatile uint8_t sink = 0;
    uint32_t r;
    for (r = 0; r < h; r++) sink ^= (uint8_t)(y[(size_t)r * ys] ^ y[(size_t)r * ys + w - 1u]);
    for (r = 0; r < ch; r++) sink ^= (uint8_t)(cb[(size_t)r * cs + w / 2u - 1u] ^ cr[(size_t)r * cs + w / 2u - 1u]);
    return sink;
}
int amvhip_encode_yuv420_batch(amvhip_ctx *ctx, const uint8_t *y, const uint8_t *cb, const uint8_t *cr, uint32_t ys, uint32_t cs,
                               uint64_t yfs, uint64_t cfs, uint32_t n, uint32_t w, uint32_t h, uint32_t qbias, uint8_t *blob,
                               uint64_t cap, uint64_t *offs, uint32_t *lens)
{
    (void)yfs; (void)cfs; (void)qbias;
    if (ctx == NULL) return AMVHIP_ERR_DEVICE;
    if (n) (void)planes_touch(y, cb, cr, ys, cs, w, h, h / 2u);
    return tiny_chunks(n, blob, cap, offs, lens);
}
int amvhip_encode_yuv422_batch(amvhip_ctx *ctx, const uint8_t *y, const uint8_t *cb, const uint8_t *cr, uint32_t ys, uint32_t cs,
                               uint64_t yfs, uint64_t cfs, uint32_t n, uint32_t w, uint32_t h, uint32_t qbias, uint8_t *blob,
                               uint64_t cap, uint64_t *offs, uint32_t *lens)
{
    (void)yfs; (void)cfs; (void)qbias;
    if (ctx == NULL) return AMVHIP_ERR_DEVICE;
    if (n) (void)planes_touch(y, cb, cr, ys, cs, w, h, h);
    return tiny_chunks(n, blob, cap, offs, lens);
}

int amvhip_adpcm_encode_frame(amvhip_ctx *ctx, const int16_t *samples, uint32_t nsamp, int32_t *step_index, uint8_t *chunk, uint32_t cap)
{
    volatile int16_t sink;
    if (ctx == NULL) return AMVHIP_ERR_DEVICE;
    if (nsamp == 0 || (nsamp & 1u) || step_index == NULL) return AMVHIP_ERR_ARG;
    if (cap < 8u + nsamp / 2u) return AMVHIP_ERR_SPACE;
    sink = (int16_t)(samples[0] ^ samples[nsamp - 1u]);
    (void)sink;
    memset(chunk, 0, 8u + nsamp / 2u);
    /* the chunk's FRAMING (adpcm.c:464-467: le16 first sample, le16 step index, le32 sample count) -- what a walk over the
     * muxed file reads; the nibbles stay zero */
    chunk[0] = (uint8_t)samples[0]; chunk[1] = (uint8_t)((uint16_t)samples[0] >> 8);
    chunk[2] = (uint8_t)*step_index;
    chunk[4] = (uint8_t)nsamp; chunk[5] = (uint8_t)(nsamp >> 8); chunk[6] = (uint8_t)(nsamp >> 16); chunk[7] = (uint8_t)(nsamp >> 24);
    return (int)(8u + nsamp / 2u);
}
int amvhip_adpcm_encode_frame_trellis(amvhip_ctx *ctx, const int16_t *samples, uint32_t nsamp, int32_t *step_index, uint32_t trellis,
                                      uint8_t *chunk, uint32_t cap)
{
    (void)trellis;
    return amvhip_adpcm_encode_frame(ctx, samples, nsamp, step_index, chunk, cap);
}
int amvhip_adpcm_wav_encode_frame(amvhip_ctx *ctx, const int16_t *samples, int frame_size, int32_t state[2], uint8_t *frame, int buf_size)
{
    const int need = 4 + (frame_size - 1) / 2;
    (void)samples; (void)state;
    if (ctx == NULL) return AMVHIP_ERR_DEVICE;
    if (frame_size < 1 || buf_size < need) return AMVHIP_ERR_ARG;
    memset(frame, 0, (size_t)need);
    return need;
}
