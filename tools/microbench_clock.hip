// microbench_clock.hip -- the shader clock MI355X (gfx950) sustains under integer VALU load: clock64() (s_memtime, shader
// cycles) against wall_clock64() (s_memrealtime, 100 MHz) around a loop of eight independent v_mad_i32_i24 chains, with one
// workgroup on the chip and with every SIMD holding `waves` waves; with and without a 12-byte store per lane and trip.
// Build: hipcc -O2 --offload-arch=gfx950 tools/microbench_clock.hip -o /tmp/clock; output: profiles/r03_shader_clock.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

struct Stamp { long long c0, c1, w0, w1; };

template <bool kStore>
__global__ void k(int* sink, Stamp* stamps, int n, int a, uint8_t* out) {
    int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const long long w0 = wall_clock64(), c0 = clock64();
    uint8_t* p = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 12u;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(x0) : "v"(a));
            asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(x1) : "v"(a));
            asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(x2) : "v"(a));
            asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(x3) : "v"(a));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(x4) : "v"(a));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(x5) : "v"(a));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(x6) : "v"(a));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(x7) : "v"(a));
        }
        if (kStore) {
            struct __attribute__((aligned(4))) P { int w[3]; };
            *reinterpret_cast<P*>(p + (size_t)(i & 63) * gridDim.x * blockDim.x * 12u) = P{{x0, x1, x2}};
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Stamp{c0, c1, w0, w1};
    if (x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 == 0x7fffffff) *sink = 1;
}

template <bool kStore>
static void run(const char* what, int blocks, int threads, int n, int* sink, Stamp* d_st, uint8_t* out) {
    const int waves = blocks * threads / 64;
    std::vector<Stamp> st(waves);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k<kStore>, dim3(blocks), dim3(threads), 0, 0, sink, d_st, n, 3, out);
        hipDeviceSynchronize();
    }
    hipMemcpy(st.data(), d_st, sizeof(Stamp) * waves, hipMemcpyDeviceToHost);
    std::vector<double> mhz;
    for (auto& s : st) mhz.push_back(double(s.c1 - s.c0) / double(s.w1 - s.w0) * 100.0);
    std::sort(mhz.begin(), mhz.end());
    const double ms = double(st[0].w1 - st[0].w0) / 1e5;
    // 16 * 8 instructions per trip per wave: 4 full-rate-class (v_add_u32) + 4 VOP3 (v_mad_i32_i24) per group
    printf("%-46s waves %6d  shader clock MHz: min %.0f median %.0f max %.0f   (wave 0 ran %.3f ms)\n", what, waves, mhz.front(),
           mhz[mhz.size() / 2], mhz.back(), ms);
}

int main() {
    int dev_cus = 0;
    hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0);
    int* sink; Stamp* d_st; uint8_t* out;
    hipMalloc(&sink, 4);
    hipMalloc(&d_st, sizeof(Stamp) * 65536);
    const size_t out_bytes = (size_t)dev_cus * 8 * 256 * 12 * 64;
    hipMalloc(&out, out_bytes);
    printf("compute units: %d\n", dev_cus);
    run<false>("one wave, VALU only", 1, 64, 20000, sink, d_st, out);
    run<false>("one workgroup per CU (1 wave per SIMD), VALU only", dev_cus, 256, 20000, sink, d_st, out);
    run<false>("5 waves per SIMD on every CU, VALU only", dev_cus * 5, 256, 8000, sink, d_st, out);
    run<false>("8 waves per SIMD on every CU, VALU only", dev_cus * 8, 256, 8000, sink, d_st, out);
    run<true>("5 waves per SIMD on every CU, VALU + 12-byte stores", dev_cus * 5, 256, 8000, sink, d_st, out);
    run<true>("8 waves per SIMD on every CU, VALU + 12-byte stores", dev_cus * 8, 256, 8000, sink, d_st, out);
    return 0;
}
