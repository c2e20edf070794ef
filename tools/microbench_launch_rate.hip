// microbench_launch_rate.hip -- how many single-wave workgroups per microsecond the chip starts (the reconstruction
// launches 1.28 M of them per 160 000 frames).  Build: hipcc -O2 --offload-arch=gfx950; output: profiles/r02_launch_rate.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int LDS, int REGS>
__global__ __launch_bounds__(64) void k(int* out, int n) {
    __shared__ int s[LDS / 4 > 0 ? LDS / 4 : 1];
    int v[REGS];
    for (int i = 0; i < REGS; ++i) v[i] = threadIdx.x * i + n;
    if (LDS) s[threadIdx.x] = v[0];
    __syncthreads();
    int acc = 0;
    for (int i = 0; i < REGS; ++i) acc += v[i] * (i + 1);
    if (acc == 0x12345678) out[0] = LDS ? s[(threadIdx.x + 1) & 63] : 1;
}
template <int LDS, int REGS> void run(const char* name, dim3 grid) {
    int* d; (void)hipMalloc(&d, 4);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<LDS, REGS>), grid, dim3(64), 0, 0, d, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<LDS, REGS>), grid, dim3(64), 0, 0, d, 1);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double wgs = (double)grid.x * grid.y * grid.z;
    printf("%-28s grid %u x %u x %u: %.3f ms, %.1f workgroups/us\n", name, grid.x, grid.y, grid.z, ms, wgs / ms / 1e3);
}
int main() {
    run<0, 4>("no LDS, few regs", dim3(160000, 8, 1));
    run<7680, 4>("7680 B LDS, few regs", dim3(160000, 8, 1));
    run<7680, 80>("7680 B LDS, 80 regs", dim3(160000, 8, 1));
    run<7680, 80>("same, 1-D grid", dim3(1280000, 1, 1));
    run<0, 4>("no LDS, 1-D", dim3(1280000, 1, 1));
    run<7680, 4>("7680 B LDS, 1-D", dim3(1280000, 1, 1));
    return 0;
}
