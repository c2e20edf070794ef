// microbench_lds_lookup.hip -- what a table look-up with a DIFFERENT row per lane costs the LDS of one CU on MI355X
// (gfx950) when every SIMD has waves asking: the ADPCM encoder's guess pass runs three waves per SIMD and each sample of
// each lane is one such look-up (cell (step index, quotient) -> the next state).  Twelve waves of one workgroup (three per
// SIMD) issue reads whose rows come from a per-lane generator, eight reads in flight per wave; the figure is shader cycles
// of the CU per wave-instruction.
//   b128 / 16-byte cells      the round-4 cell table
//   b64  /  8-byte cells
//   b64  / 16-byte stride, even lanes the low half of a cell, odd lanes the high half (two copies of an 8-byte cell)
//   b32  /  4-byte cells
//   one row for all lanes     the broadcast floor
// Build: hipcc -O2 --offload-arch=gfx950 tools/microbench_lds_lookup.hip -o /tmp/lds; output: profiles/r04_lds_lookup.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

struct Out { long long cycles; uint32_t sink; };

typedef uint32_t U4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kRows = 712;     // 89 x 8 cells

template <int kBytes, int kStride, int kMode>   // kMode 0: a row per lane, 1: one row for all, 2: + 8 * (lane & 1)
__global__ __launch_bounds__(768) void lookup(Out* out, int n) {
    __shared__ uint4 s_tab[kRows + 312];
    for (uint32_t i = threadIdx.x; i < kRows + 312; i += blockDim.x) s_tab[i] = make_uint4(i, i * 3u, i * 5u, i * 7u);
    __syncthreads();
    uint32_t x = kMode == 1 ? 12345u : threadIdx.x * 2654435761u + 99991u;
    uint32_t acc = 0;
    const uint32_t half = kMode == 2 ? (threadIdx.x & 1u) * 8u : 0u;
    const uint8_t* base = reinterpret_cast<const uint8_t*>(s_tab);
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
        uint32_t a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x = x * 5u + 0x3c6ef35fu;                       // (v_lshl_add + v_add: cheap, and every lane its own sequence)
            x ^= x >> 13;
            uint32_t row = (x >> 7) & 1023u;
            row = row >= kRows ? row - 312u : row;
            a[k] = row * (uint32_t)kStride + half;
        }
        if (kBytes == 16 && kMode == 3) {            // the 16-byte cell as two 8-byte reads of one instruction
            U4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1" : "=v"(v[k]) : "v"((uint32_t)(uintptr_t)base + a[k]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t lo = v[k].x, hi = v[k].w;
                asm volatile("" : "+v"(lo), "+v"(hi));      // (not before the wait)
                acc += lo ^ hi;
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (kBytes == 16 && kMode == 4) {        // ... and as two instructions
                const uint2 lo = *reinterpret_cast<const uint2*>(base + a[k]);
                const uint2 hi = *reinterpret_cast<const uint2*>(base + a[k] + 8u);
                acc += lo.x ^ hi.y;
            } else if (kBytes == 16) { const uint4 v = *reinterpret_cast<const uint4*>(base + a[k]); acc += v.x ^ v.w; }
            else if (kBytes == 8) { const uint2 v = *reinterpret_cast<const uint2*>(base + a[k]); acc += v.x ^ v.y; }
            else { acc += *reinterpret_cast<const uint32_t*>(base + a[k]); }
        }
    }
    const long long c1 = clock64();
    if (acc == 0x12345u || threadIdx.x == 0) { out->cycles = c1 - c0; out->sink = acc; }
}

// the same generator without the reads: what the addresses alone cost
__global__ __launch_bounds__(768) void no_lookup(Out* out, int n) {
    uint32_t x = threadIdx.x * 2654435761u + 99991u, acc = 0;
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x = x * 5u + 0x3c6ef35fu;
            x ^= x >> 13;
            uint32_t row = (x >> 7) & 1023u;
            row = row >= kRows ? row - 312u : row;
            acc += row * 16u;
        }
    }
    const long long c1 = clock64();
    if (acc == 0x12345u || threadIdx.x == 0) { out->cycles = c1 - c0; out->sink = acc; }
}

template <typename K>
static void run(const char* name, K kernel, int threads, Out* d_out) {
    const int n = 4096;
    Out h;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kernel, dim3(1), dim3(threads), 0, 0, d_out, n);
        hipDeviceSynchronize();
    }
    hipMemcpy(&h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    // clock64 ticks at 100 MHz on this part (s_memtime); the shader clock is read from the ratio the dep_add chain gives
    printf("%-62s %3d waves %9.2f ticks per wave-instruction of the CU\n", name, waves, (double)h.cycles / ((double)n * 8.0 * waves));
}

__global__ void dep_add(Out* out, int n) {
    int x = threadIdx.x;
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(n));
    }
    const long long c1 = clock64();
    if (threadIdx.x == 0) { out->cycles = c1 - c0; out->sink = (uint32_t)x; }
}

int main() {
    Out* d_out;
    hipMalloc(&d_out, sizeof(Out));
    {
        Out h;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(dep_add, dim3(1), dim3(64), 0, 0, d_out, 4096); hipDeviceSynchronize(); }
        hipMemcpy(&h, d_out, sizeof(h), hipMemcpyDeviceToHost);
        printf("dependent v_add_u32 (9 shader cycles each): %.4f ticks -> one tick = %.2f shader cycles\n", (double)h.cycles / (4096.0 * 64.0),
               9.0 / ((double)h.cycles / (4096.0 * 64.0)));
    }
    for (int threads : {768, 256, 64}) {
        printf("-- %d waves of one workgroup\n", threads / 64);
        run("address generator alone", no_lookup, threads, d_out);
        run("ds_read_b128, 16-byte cells, a row per lane", lookup<16, 16, 0>, threads, d_out);
        run("ds_read2_b64 offset1:1, 16-byte cells, a row per lane", lookup<16, 16, 3>, threads, d_out);
        run("two ds_read_b64, 16-byte cells, a row per lane", lookup<16, 16, 4>, threads, d_out);
        run("ds_read_b64, 8-byte cells, a row per lane", lookup<8, 8, 0>, threads, d_out);
        run("ds_read_b64, 16-byte stride, a row per lane", lookup<8, 16, 0>, threads, d_out);
        run("ds_read_b64, 16-byte stride, odd lanes the high half", lookup<8, 16, 2>, threads, d_out);
        run("ds_read_b32, 4-byte cells, a row per lane", lookup<4, 4, 0>, threads, d_out);
        run("ds_read_b32, 16-byte stride, a row per lane", lookup<4, 16, 0>, threads, d_out);
        run("ds_read_b128, one row for all lanes", lookup<16, 16, 1>, threads, d_out);
        run("ds_read_b64, one row for all lanes", lookup<8, 8, 1>, threads, d_out);
    }
    return 0;
}
