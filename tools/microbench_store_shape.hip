// How fast does MI355X take a frame-sized write stream, by store shape?  (amv_reconstruct_kernel writes 12 bytes per
// lane, 480-byte row pieces, rows bottom-up.)   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_store tools/microbench_store_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct __attribute__((aligned(4))) Px12 { uint32_t w[3]; };
// one wave per 7 680-byte piece (16 rows of 480 bytes), as the reconstruction's store stage
template <int kShape>
__global__ __launch_bounds__(64) void store_kernel(uint8_t* out, uint32_t v) {
    const uint32_t lane = threadIdx.x;
    uint8_t* base = out + (uint64_t)blockIdx.x * 7680u;
    if (kShape == 0) {          // 16 bytes per lane, line after line: 7 full 1-KB instructions + a half
        for (uint32_t i = lane; i < 480u; i += 64u) reinterpret_cast<uint4*>(base)[i] = make_uint4(v, v + i, v, v);
    } else if (kShape == 1) {   // 12 bytes per lane, contiguous: 10 instructions of 768 bytes
        for (uint32_t i = lane; i < 640u; i += 64u) { Px12 p{{v, v + i, v}}; reinterpret_cast<Px12*>(base)[i] = p; }
    } else {                    // the kernel's shape: patch t -> rows 2*(t/40), +1 (descending), 12 bytes at column group t%40
        for (uint32_t t = lane; t < 320u; t += 64u) {
            const uint32_t i2 = t / 40u, gi = t % 40u;
            Px12 p{{v, v + t, v}};
            uint8_t* r = base + (15u - 2u * i2) * 480u + gi * 12u;
            *reinterpret_cast<Px12*>(r) = p;
            *reinterpret_cast<Px12*>(r - 480u) = p;
        }
    }
}
int main() {
    const uint32_t pieces = 1280000;   // 160 000 frames x 8 MCU rows
    const size_t bytes = (size_t)pieces * 7680;
    uint8_t* d;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const char* names[3] = {"16 B/lane, 1 KB per instruction", "12 B/lane contiguous, 768 B per instruction", "12 B/lane, the kernel's row pieces"};
    for (int shape = 0; shape < 3; ++shape) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(a);
            if (shape == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(pieces), dim3(64), 0, 0, d, (uint32_t)rep);
            else if (shape == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(pieces), dim3(64), 0, 0, d, (uint32_t)rep);
            else hipLaunchKernelGGL(store_kernel<2>, dim3(pieces), dim3(64), 0, 0, d, (uint32_t)rep);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep && ms < best) best = ms;
        }
        printf("%-48s %.3f ms  %.2f TB/s\n", names[shape], best, bytes / best / 1e9);
    }
    return 0;
}
