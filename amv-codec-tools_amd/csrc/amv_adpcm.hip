// amv_adpcm.hip -- IMA ADPCM (AMV chunk layout) and the synthetic-source generators, gfx950.
//
// Reference: decode  C-AMVDecoder/amvlib/AMVDec.c:312-320 (chunk header) and AdpcmIma.c:170-242
//            (AdpcmImaExpandNibble / AdpcmImaDecodeFrame, mono: high nibble first);
//            encode  AMVmuxer/ffmpeg/libavcodec/adpcm.c:219-227 (adpcm_ima_compress_sample)
//            and :461-498 (AMV framing: le16 first sample, le16 step index, le32 sample count).
//
// The predictor loop is a serial chain inside a chunk; chunks are independent on decode (each
// carries predictor + step index), so one lane owns one chunk.  On encode the reference carries
// step_index from chunk to chunk.  That chain is cut with the fact that step_index has only 89
// values: amv_adpcm_map_kernel runs every chunk from all 89 starts (state only, no output),
// amv_adpcm_chain_kernel walks the 89-entry maps, and the real encode then runs one lane per
// chunk from its now-known start.
#include "amv_kernels.h"

namespace amv {

namespace {

__device__ __forceinline__ int clip16(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ int clip_index(int v) { return min(max(v, 0), 88); }

// AdpcmImaExpandNibble, AdpcmIma.c:170-204 with shift 3
__device__ __forceinline__ int expand(int& predictor, int& index, uint32_t nibble) {
    const int step = kImaStep[index];
    index = clip_index(index + kImaIndexAdjust[nibble]);
    const int diff = ((2 * (int)(nibble & 7u) + 1) * step) >> 3;
    predictor = clip16((nibble & 8u) ? predictor - diff : predictor + diff);
    return predictor;
}

// adpcm_ima_compress_sample, adpcm.c:219-227
__device__ __forceinline__ uint32_t compress(int& prev, int& index, int sample) {
    const int delta = sample - prev;
    const int step = kImaStep[index];
    const int q = min(7, abs(delta) * 4 / step);
    const uint32_t nibble = (uint32_t)q + (delta < 0 ? 8u : 0u);
    const int mag = (step * (2 * q + 1)) / 8;   // step * yamaha_difflookup[nibble] / 8, C division
    prev = clip16(delta < 0 ? prev - mag : prev + mag);
    index = clip_index(index + kImaIndexAdjust[nibble]);
    return nibble;
}

}  // namespace

__global__ __launch_bounds__(64) void amv_adpcm_decode_kernel(
    const uint8_t* __restrict__ blob, uint64_t blob_bytes, const uint64_t* __restrict__ offs,
    const uint32_t* __restrict__ lens, uint32_t n, int16_t* __restrict__ pcm,
    const uint64_t* __restrict__ pcm_offs, int32_t* __restrict__ final_state) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    const uint64_t off = offs[i];
    const uint32_t len = lens[i];
    if (len <= 8 || off > blob_bytes || len > blob_bytes - off) return;
    const uint8_t* c = blob + off;
    int predictor = (int16_t)(c[0] | (c[1] << 8));   // AMVDec.c:312
    int index = clip_index(c[2]);                    // AMVDec.c:313 (the reference indexes its table unchecked)
    int16_t* o = pcm + pcm_offs[i];
    for (uint32_t k = 8; k < len; ++k) {             // AdpcmIma.c:225-237
        const uint32_t byte = c[k];
        *o++ = (int16_t)expand(predictor, index, byte >> 4);
        *o++ = (int16_t)expand(predictor, index, byte & 15u);
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

// state-only run of chunk i from start index s: where does step_index end up?
__global__ __launch_bounds__(128) void amv_adpcm_map_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs,
    const uint32_t* __restrict__ nsamp, uint32_t n, uint8_t* __restrict__ map /* [n][96] */) {
    const uint32_t i = blockIdx.x, s = threadIdx.x;
    if (i >= n || s >= 89) return;
    const int16_t* x = pcm + pcm_offs[i];
    const uint32_t m = nsamp[i] & ~1u;
    int prev = m ? x[0] : 0, index = (int)s;
    for (uint32_t k = 0; k < m; ++k) compress(prev, index, x[k]);
    map[i * 96u + s] = (uint8_t)index;
}

__global__ void amv_adpcm_chain_kernel(const uint8_t* __restrict__ map, uint32_t n, int32_t* __restrict__ start) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint32_t s = 0;   // the encoder context starts zeroed
    for (uint32_t i = 0; i < n; ++i) {
        start[i] = (int32_t)s;
        s = map[i * 96u + s];
    }
}

__global__ __launch_bounds__(64) void amv_adpcm_encode_kernel(
    const int16_t* __restrict__ pcm, const uint64_t* __restrict__ pcm_offs,
    const uint32_t* __restrict__ nsamp, uint32_t n, const int32_t* __restrict__ step_in,
    uint8_t* __restrict__ blob, const uint64_t* __restrict__ offs) {
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
    for (uint32_t k = 0; k < pairs; ++k) {          // :489-493 high nibble = earlier sample
        const uint32_t hi = compress(prev, index, x[2 * k]);
        const uint32_t lo = compress(prev, index, x[2 * k + 1]);
        d[8 + k] = (uint8_t)((hi << 4) | lo);
    }
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
    if (n == 0) return;
    hipLaunchKernelGGL(amv_adpcm_map_kernel, dim3(n), dim3(128), 0, s, pcm, pcm_offs, nsamp, n, map);
    hipLaunchKernelGGL(amv_adpcm_chain_kernel, dim3(1), dim3(64), 0, s, map, n, start);
}

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
