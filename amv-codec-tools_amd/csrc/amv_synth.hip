// amv_synth.hip -- the synthetic sources of BASELINE.md section 4 (seeded moving gradients + rotating checker + noise;
// three sines + noise), made on the device for bench.py and the tests: integer only, byte-identical to the CPU generator
// the parity tests hold it against (test_synthetic_sources_match_cpu_generator).  Not part of the codec path.
#include "amv_kernels.h"

namespace amv {

namespace {
__device__ __forceinline__ int clip16(int v) { return min(max(v, -32768), 32767); }
}  // namespace

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
