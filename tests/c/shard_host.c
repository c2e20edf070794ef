/*
 * shard_host.c -- a plain C host that shards one AMV stream over several GPUs of a node, one amvhip context (and one
 * HIP stream) per device, no MPI and no Python: what BASELINE.json's north_star calls "the host stays in C ... frames
 * of a stream shard embarrassingly across the GPUs of one node".
 *
 *     shard_host <file.amv> [contexts]
 *
 * The video chunks of the file are read through the amvlib surface (AmvOpen / AmvReadNextFrame), then
 *   1. context r (on device r modulo the number of visible devices) gets the contiguous frame range
 *      [r * n / G, (r + 1) * n / G): its chunks go to that device, amvhip_decode_batch_dev runs on the context's own
 *      stream, and the decoded frames travel device to device into their place in ONE buffer on device 0
 *      (hipMemcpyPeerAsync on the same stream) -- every device works at once, nothing is ordered across devices
 *      until the final synchronisation;
 *   2. for comparison the whole stream is decoded by a single context through the host-buffer entry point.
 * Both results are hashed as the survey's harness hashed amvlib's output (chained FNV-1a-64 over the BGR frames).
 * With more contexts than devices (`contexts` argument; a one-GPU box) several contexts share a device: they are
 * independent -- workspace, tables and stream of their own -- and the same code path runs.
 *
 * Output: "key: value" lines.  tests/test_gpu_parity.py builds this with gcc against include/amvhip.h and the HIP
 * runtime's C API and checks the hashes against the figure amvlib itself produced for the reference's clip.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "amvhip.h"

#define MAX_CTX 16

#define HIP_OK(expr)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_));                          \
            return 2;                                                                           \
        }                                                                                       \
    } while (0)

static uint64_t fnv_frames(uint64_t h, const unsigned char *p, size_t n)
{
    size_t i;
    for (i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

struct rank {
    amvhip_ctx *ctx;
    int device;
    hipStream_t stream;
    uint32_t first, count;          /* frame range */
    uint8_t *d_blob, *d_out;
    uint64_t *d_offs;
    uint32_t *d_lens;
    int32_t *d_status;
    const char *path;               /* how its frames reach device 0 */
};

int main(int argc, char **argv)
{
    AMVDecoder *amv;
    unsigned char *blob = NULL;
    uint64_t *offs = NULL;
    uint32_t *lens = NULL;
    size_t blob_bytes = 0, blob_cap = 0;
    uint32_t n = 0, cap_n = 0, w, h, r, i;
    int ndev = 0, nctx, rc = 0;
    struct rank rk[MAX_CTX];
    uint64_t fb;
    uint8_t *d_all = NULL, *gathered = NULL, *single = NULL;
    int32_t *status = NULL, *st_r = NULL;
    uint64_t h_g, h_s;
    unsigned long bad = 0;

    if (argc < 2) {
        fprintf(stderr, "usage: shard_host <file.amv> [contexts]\n");
        return 2;
    }
    /* ---- the stream: every video chunk of the file, 4-byte aligned in one blob (AMVDec.c:150-238 through the library) */
    amv = AmvOpen(argv[1]);
    if (amv == NULL) {
        fprintf(stderr, "cannot open %s\n", argv[1]);
        return 2;
    }
    w = amv->amvinfo.dwWidth;
    h = amv->amvinfo.dwHeight;
    while (AmvReadNextFrame(amv) == 0 && amv->framebuf.framenum != -1) {
        const uint32_t len = amv->framebuf.videobufflen;
        if (n == cap_n) {
            cap_n = cap_n ? cap_n * 2 : 256;
            offs = (uint64_t *)realloc(offs, cap_n * sizeof *offs);
            lens = (uint32_t *)realloc(lens, cap_n * sizeof *lens);
        }
        if (blob_bytes + len + 16 > blob_cap) {
            blob_cap = (blob_bytes + len + 16) * 2;
            blob = (unsigned char *)realloc(blob, blob_cap);
        }
        if (!offs || !lens || !blob) return 2;
        memcpy(blob + blob_bytes, amv->framebuf.videobuff, len);
        offs[n] = blob_bytes;
        lens[n] = len;
        blob_bytes = (blob_bytes + len + 3u) & ~(size_t)3u;
        ++n;
    }
    AmvClose(amv);
    if (n == 0) {
        fprintf(stderr, "no frames\n");
        return 2;
    }
    fb = amvhip_frame_bytes(w, h);

    /* ---- one context per device; more contexts than devices wrap around */
    HIP_OK(hipGetDeviceCount(&ndev));
    nctx = argc > 2 ? atoi(argv[2]) : ndev;
    if (nctx < 1) nctx = 1;
    if (nctx > MAX_CTX) nctx = MAX_CTX;
    if ((uint32_t)nctx > n) nctx = (int)n;
    memset(rk, 0, sizeof rk);
    for (r = 0; r < (uint32_t)nctx; ++r) {
        rk[r].device = (int)(r % (uint32_t)ndev);
        if (amvhip_create(&rk[r].ctx, rk[r].device) != AMVHIP_OK) {
            fprintf(stderr, "amvhip_create on device %d failed\n", rk[r].device);
            return 2;
        }
        rk[r].first = (uint32_t)((uint64_t)r * n / (uint32_t)nctx);                 /* contiguous ranges */
        rk[r].count = (uint32_t)((uint64_t)(r + 1) * n / (uint32_t)nctx) - rk[r].first;
    }
    HIP_OK(hipSetDevice(0));
    HIP_OK(hipMalloc((void **)&d_all, (size_t)fb * n));
    /* how each context's frames will reach device 0: in place (same device), over the peer link (xGMI: hipMemcpyPeerAsync
     * moves device memory to device memory once peer access is on), or -- where the runtime refuses peer access -- by the
     * runtime's own staging through host memory, which the same call falls back to; said per context in the output */
    for (r = 0; r < (uint32_t)nctx; ++r) {
        rk[r].path = "same device";
        if (rk[r].device != 0) {
            int can = 0;
            hipError_t e;
            HIP_OK(hipDeviceCanAccessPeer(&can, rk[r].device, 0));
            rk[r].path = "staged by the runtime (no peer access)";
            if (can) {
                HIP_OK(hipSetDevice(rk[r].device));
                e = hipDeviceEnablePeerAccess(0, 0);
                (void)hipGetLastError();
                if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) rk[r].path = "peer link (access enabled)";
            }
        }
    }

    /* ---- 1. every range on its device, results gathered on device 0; nothing waits until the end */
    for (r = 0; r < (uint32_t)nctx; ++r) {
        struct rank *k = &rk[r];
        const uint64_t base = offs[k->first];
        const uint64_t end = k->first + k->count < n ? offs[k->first + k->count] : blob_bytes;
        uint64_t *rel = (uint64_t *)malloc(k->count * sizeof *rel);
        if (!rel) return 2;
        for (i = 0; i < k->count; ++i) rel[i] = offs[k->first + i] - base;
        HIP_OK(hipSetDevice(k->device));
        HIP_OK(hipStreamCreateWithFlags(&k->stream, hipStreamNonBlocking));
        HIP_OK(hipMalloc((void **)&k->d_blob, (size_t)(end - base) + 16));
        HIP_OK(hipMalloc((void **)&k->d_offs, k->count * sizeof(uint64_t)));
        HIP_OK(hipMalloc((void **)&k->d_lens, k->count * sizeof(uint32_t)));
        HIP_OK(hipMalloc((void **)&k->d_status, k->count * sizeof(int32_t)));
        HIP_OK(hipMalloc((void **)&k->d_out, (size_t)fb * k->count));
        HIP_OK(hipMemcpyAsync(k->d_blob, blob + base, (size_t)(end - base), hipMemcpyHostToDevice, k->stream));
        HIP_OK(hipMemcpyAsync(k->d_offs, rel, k->count * sizeof(uint64_t), hipMemcpyHostToDevice, k->stream));
        HIP_OK(hipMemcpyAsync(k->d_lens, lens + k->first, k->count * sizeof(uint32_t), hipMemcpyHostToDevice, k->stream));
        HIP_OK(hipStreamSynchronize(k->stream));                                    /* `rel` is pageable and about to be freed */
        free(rel);
        if (amvhip_decode_batch_dev(k->ctx, k->d_blob, end - base, k->d_offs, k->d_lens, k->count, w, h, 0, k->d_out, k->d_status,
                                    k->stream) != AMVHIP_OK) {
            fprintf(stderr, "decode on context %u: %s\n", r, amvhip_last_error(k->ctx));
            return 2;
        }
        HIP_OK(hipMemcpyPeerAsync(d_all + (size_t)fb * k->first, 0, k->d_out, k->device, (size_t)fb * k->count, k->stream));
    }
    gathered = (uint8_t *)malloc((size_t)fb * n);
    single = (uint8_t *)malloc((size_t)fb * n);
    status = (int32_t *)malloc(n * sizeof *status);
    st_r = (int32_t *)malloc(n * sizeof *st_r);
    if (!gathered || !single || !status || !st_r) return 2;
    for (r = 0; r < (uint32_t)nctx; ++r) {
        HIP_OK(hipSetDevice(rk[r].device));
        HIP_OK(hipStreamSynchronize(rk[r].stream));
        HIP_OK(hipMemcpy(st_r + rk[r].first, rk[r].d_status, rk[r].count * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    HIP_OK(hipSetDevice(0));
    HIP_OK(hipMemcpy(gathered, d_all, (size_t)fb * n, hipMemcpyDeviceToHost));
    for (i = 0; i < n; ++i) bad += st_r[i] != 0;

    /* ---- 2. the whole stream through one context, host buffers */
    if (amvhip_decode_batch(rk[0].ctx, blob, blob_bytes, offs, lens, n, w, h, 0, single, status) != AMVHIP_OK) {
        fprintf(stderr, "single-context decode: %s\n", amvhip_last_error(rk[0].ctx));
        return 2;
    }
    for (i = 0; i < n; ++i) bad += status[i] != 0;

    h_g = h_s = 1469598103934665603ull;                                             /* the survey's seed */
    for (i = 0; i < n; ++i) {
        h_g = fnv_frames(h_g, gathered + (size_t)fb * i, (size_t)fb);
        h_s = fnv_frames(h_s, single + (size_t)fb * i, (size_t)fb);
    }
    printf("size: %u x %u\n", w, h);
    printf("frames: %u\n", n);
    printf("devices: %d\n", ndev);
    printf("contexts: %d\n", nctx);
    for (r = 0; r < (uint32_t)nctx; ++r)
        printf("context %u: device %d frames %u..%u gather: %s\n", r, rk[r].device, rk[r].first, rk[r].first + rk[r].count, rk[r].path);
    printf("failed frames: %lu\n", bad);
    printf("gathered fnv1a64: %016llx\n", (unsigned long long)h_g);
    printf("single fnv1a64: %016llx\n", (unsigned long long)h_s);
    printf("match: %s\n", memcmp(gathered, single, (size_t)fb * n) == 0 ? "yes" : "no");
    rc = memcmp(gathered, single, (size_t)fb * n) == 0 && bad == 0 ? 0 : 1;

    for (r = 0; r < (uint32_t)nctx; ++r) {
        (void)hipSetDevice(rk[r].device);
        (void)hipFree(rk[r].d_blob); (void)hipFree(rk[r].d_offs); (void)hipFree(rk[r].d_lens);
        (void)hipFree(rk[r].d_status); (void)hipFree(rk[r].d_out);
        (void)hipStreamDestroy(rk[r].stream);
        amvhip_destroy(rk[r].ctx);
    }
    (void)hipSetDevice(0);
    (void)hipFree(d_all);
    free(gathered); free(single); free(status); free(st_r); free(blob); free(offs); free(lens);
    return rc;
}
