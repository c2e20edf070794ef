// What does rocprofv3's FETCH_SIZE count, per LOAD SHAPE?  The guide (MI355X_MICROARCH.md, section HBM) calibrates one
// shape -- 16 bytes per lane, coalesced, streaming: the counter reads HALF the bytes -- and says every other width is
// uncalibrated.  The kernels of this repository read in five shapes; each is reproduced here over a buffer far larger
// than the Infinity Cache (2 GiB, every byte requested exactly once per kernel, so bytes requested = bytes that must
// cross the fabric, up to line rounding), and the PMC pass gives FETCH_SIZE per kernel:
//
//   shape_a  4 bytes per lane, contiguous (256 B per instruction)
//   shape_b  16 bytes per lane, contiguous, 16-byte aligned (1 KB per instruction): the guide's shape, the control
//   shape_c  amv_adpcm_guess_kernel's staging fetch: four lanes per row, 16 bytes each (64 contiguous bytes of a row per
//            instruction and row), sixteen rows per instruction, rows 2 756 bytes apart (1378 samples), 2-byte aligned:
//            most pieces straddle a 64-byte boundary, half of them a 128-byte line.  One tile (64 B per row) per trip.
//   shape_c_aligned  the same with rows 2 816 bytes apart (44 x 64): no piece straddles anything
//   shape_d  amv_huffman_fast_kernel's stream requests: one lane per frame, 2 x 16 bytes (32 contiguous, 16-byte aligned)
//            per request, frames ~3.5 KB apart, a frame read front to back
//   shape_e  amv_reconstruct_kernel's record fetch: 16 bytes per lane as four 4-byte records, contiguous over the wave,
//            the range starting at any 4-byte boundary (1 KB per instruction, ~2 KB per wave)
//
// The _paced variants of c and d sleep between trips for about as long as the real kernels compute between theirs
// (a tile of 32 samples, four strides of eight symbols), with about as many waves resident: what a line's second half
// finds in the L2 when its turn comes.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/scratch/mb_fetch tools/microbench_fetch_shape.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r06_fetch_shape -o run --output-format csv -- tools/scratch/mb_fetch
//   python tools/summarize_fetch_shapes.py gpurun_out/r06_fetch_shape gpurun_out/r06_fetch_shape.txt profiles/r06_fetch_shapes.json
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint32_t U32x4 __attribute__((ext_vector_type(4)));
typedef U32x4 U32x4Align2 __attribute__((aligned(2)));
typedef U32x4 U32x4Align4 __attribute__((aligned(4)));

__device__ __forceinline__ uint32_t fold(U32x4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
__device__ __forceinline__ void nap(uint32_t units) {      // units x 8 128 cycles
    for (uint32_t i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(127);
}
__device__ __forceinline__ void keep(uint32_t* sink, uint32_t acc) {
    if (acc == 0x9e3779b9u) sink[threadIdx.x] = acc;     // (never: the buffer holds a pattern that cannot fold to this)
}

extern "C" __global__ __launch_bounds__(256) void shape_a(const uint32_t* __restrict__ p, uint64_t words, uint32_t* sink) {
    uint32_t acc = 0;
    const uint64_t per_wg = 256u * 16u;                   // sixteen instructions of 256 B per wave and workgroup trip
    for (uint64_t base = (uint64_t)blockIdx.x * per_wg; base < words; base += (uint64_t)gridDim.x * per_wg)
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) acc ^= p[base + k * 256u + threadIdx.x];
    keep(sink, acc);
}

extern "C" __global__ __launch_bounds__(256) void shape_b(const U32x4* __restrict__ p, uint64_t vecs, uint32_t* sink) {
    uint32_t acc = 0;
    const uint64_t per_wg = 256u * 4u;
    for (uint64_t base = (uint64_t)blockIdx.x * per_wg; base < vecs; base += (uint64_t)gridDim.x * per_wg)
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) acc ^= fold(p[base + k * 256u + threadIdx.x]);
    keep(sink, acc);
}

// one wave per 64 rows, as encode_rows: lane -> (row r0 + 16 j, piece), tiles of 64 bytes per row
template <bool kPaced>
__device__ __forceinline__ void rows_body(const uint8_t* p, uint64_t rows, uint32_t pitch, uint32_t tiles, uint32_t units, uint32_t* sink) {
    const uint32_t lane = threadIdx.x & 63u, piece = lane & 3u, r0 = lane >> 2;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t row0 = wave * 64u;
    if (row0 + 64u > rows) return;
    uint32_t acc = 0;
    const uint8_t* rp[4];
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) rp[j] = p + (row0 + r0 + 16u * j) * pitch + (piece << 4);
    for (uint32_t t = 0; t < tiles; ++t) {
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j)
            acc ^= fold(*(const __attribute__((address_space(1))) U32x4Align2*)(uint64_t)(rp[j] + (uint64_t)t * 64u));
        if (kPaced) nap(units);
    }
    keep(sink, acc);
}
extern "C" __global__ __launch_bounds__(256) void shape_c(const uint8_t* p, uint64_t rows, uint32_t pitch, uint32_t tiles, uint32_t units, uint32_t* sink) {
    rows_body<false>(p, rows, pitch, tiles, units, sink);
}
extern "C" __global__ __launch_bounds__(256) void shape_c_aligned(const uint8_t* p, uint64_t rows, uint32_t pitch, uint32_t tiles, uint32_t units, uint32_t* sink) {
    rows_body<false>(p, rows, pitch, tiles, units, sink);
}
extern "C" __global__ __launch_bounds__(256) void shape_c_paced(const uint8_t* p, uint64_t rows, uint32_t pitch, uint32_t tiles, uint32_t units, uint32_t* sink) {
    rows_body<true>(p, rows, pitch, tiles, units, sink);
}

// one lane per frame, 32 bytes per request, the frame front to back
template <bool kPaced>
__device__ __forceinline__ void frames_body(const uint8_t* p, uint64_t frames, uint32_t pitch, uint32_t requests, uint32_t units, uint32_t* sink) {
    const uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    const U32x4* s = reinterpret_cast<const U32x4*>(p + f * pitch);
    uint32_t acc = 0;
    for (uint32_t r = 0; r < requests; ++r) {
        acc ^= fold(s[2u * r]) ^ fold(s[2u * r + 1u]);
        if (kPaced) nap(units);
    }
    keep(sink, acc);
}
extern "C" __global__ __launch_bounds__(1024) void shape_d(const uint8_t* p, uint64_t frames, uint32_t pitch, uint32_t requests, uint32_t units, uint32_t* sink) {
    frames_body<false>(p, frames, pitch, requests, units, sink);
}
extern "C" __global__ __launch_bounds__(1024) void shape_d_paced(const uint8_t* p, uint64_t frames, uint32_t pitch, uint32_t requests, uint32_t units, uint32_t* sink) {
    frames_body<true>(p, frames, pitch, requests, units, sink);
}

// one wave per ~2 KB range of 4-byte records starting at a 4-byte boundary (never a 16-byte one), 16 bytes per lane
extern "C" __global__ __launch_bounds__(256) void shape_e(const uint32_t* __restrict__ p, uint64_t ranges, uint32_t words_per_range, uint32_t* sink) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave >= ranges) return;
    const uint32_t* rec = p + wave * words_per_range + 1u;          // + 4 bytes: what a record range's start usually is
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t j = 0; j < 2u; ++j) {
        const uint32_t i = j * 256u + lane * 4u;
        if (i + 4u <= words_per_range - 4u) acc ^= fold(*(const __attribute__((address_space(1))) U32x4Align4*)(uint64_t)(rec + i));
    }
    keep(sink, acc);
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const uint64_t bytes = 2ull << 30;
    uint8_t* d;
    uint32_t* sink;
    CHECK(hipMalloc(&d, bytes + 4096));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(d, 0x11, bytes + 4096));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    CHECK(hipFuncSetAttribute((const void*)shape_d_paced, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
    float ms;
    // every kernel is launched three times; the summary takes the median FETCH_SIZE per kernel name
    auto report = [&](const char* name, double requested, const char* shape) {
        printf("%-18s requested_bytes %.0f  ms %.3f  GB/s %.1f  | %s\n", name, requested, ms, requested / ms / 1e6, shape);
    };
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(shape_a, dim3(8192), dim3(256), 0, 0, (const uint32_t*)d, bytes / 4, sink);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep == 2) report("shape_a", (double)bytes, "4 B per lane, contiguous");
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(shape_b, dim3(8192), dim3(256), 0, 0, (const U32x4*)d, bytes / 16, sink);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
        if (rep == 2) report("shape_b", (double)bytes, "16 B per lane, contiguous, aligned");
        {   // rows of 2 756 bytes, 43 tiles of 64 bytes read of each
            const uint32_t pitch = 2756, tiles = 43;
            const uint64_t rows = (bytes / pitch) & ~63ull;
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(shape_c, dim3((uint32_t)(rows / 256)), dim3(256), 0, 0, d, rows, pitch, tiles, 0u, sink);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) report("shape_c", (double)(rows / 256 * 256) * tiles * 64.0, "4 lanes x 16 B per row, rows 2756 B apart, 2-byte aligned");
            CHECK(hipEventRecord(a));
            // 40 KB of LDS per workgroup: four workgroups = sixteen waves per CU, about what the guess pass keeps resident
            hipLaunchKernelGGL(shape_c_paced, dim3((uint32_t)(rows / 256)), dim3(256), 40960, 0, d, rows, pitch, tiles, 2u, sink);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) report("shape_c_paced", (double)(rows / 256 * 256) * tiles * 64.0, "the same, ~16 000 cycles between a row's tiles, 16 waves per CU");
            const uint32_t pitch2 = 2816;
            const uint64_t rows2 = (bytes / pitch2) & ~63ull;
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(shape_c_aligned, dim3((uint32_t)(rows2 / 256)), dim3(256), 0, 0, d, rows2, pitch2, tiles, 0u, sink);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) report("shape_c_aligned", (double)(rows2 / 256 * 256) * tiles * 64.0, "rows 2816 B apart (64-byte aligned pieces); 2752 of every 2816 bytes read");
        }
        {   // frames of 3 520 bytes (110 requests of 32)
            const uint32_t pitch = 3520, requests = 110;
            const uint64_t frames = (bytes / pitch) & ~1023ull;
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(shape_d, dim3((uint32_t)(frames / 1024)), dim3(1024), 0, 0, d, frames, pitch, requests, 0u, sink);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) report("shape_d", (double)frames * requests * 32.0, "one lane per frame, 2 x 16 B per request, frames 3520 B apart");
            CHECK(hipEventRecord(a));
            // 96 KB of LDS per workgroup: one workgroup = sixteen waves per CU, as amv_huffman_fast_kernel runs
            hipLaunchKernelGGL(shape_d_paced, dim3((uint32_t)(frames / 1024)), dim3(1024), 98304, 0, d, frames, pitch, requests, 4u, sink);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) report("shape_d_paced", (double)frames * requests * 32.0, "the same, ~32 000 cycles between a frame's requests, 16 waves per CU");
        }
        {   // ranges of 500 records (2 000 bytes), each starting 4 bytes past a 16-byte boundary... of its own: ranges are 2000 B apart
            const uint32_t words = 500;
            const uint64_t ranges = (bytes / (words * 4ull)) & ~3ull;
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(shape_e, dim3((uint32_t)(ranges / 4)), dim3(256), 0, 0, (const uint32_t*)d, ranges, words, sink);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); CHECK(hipEventElapsedTime(&ms, a, b));
            // lanes 0..60 of trip 0 read 16 bytes, trip 1 lanes with i + 4 <= 496: i = 256 + 4 lane <= 492 -> lanes 0..59
            if (rep == 2) report("shape_e", (double)ranges * (64.0 + 60.0) * 16.0, "4 x 4-byte records per lane, contiguous, ranges at 4-byte boundaries");
        }
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
