// microbench_lone_wave.hip -- what ONE wave pays per instruction on MI355X (gfx950): the ADPCM index-chain sweeps and the
// small-batch entropy launches are one wave per SIMD deep, so their time is a dependent chain's latency, not the chip's
// issue rate (tools/microbench_valu_rate.hip has that).  Shader cycles (clock64 = s_memtime) per instruction of
//   * a chain of DEPENDENT instructions of one kind,
//   * eight INDEPENDENT chains of the same kind (the issue interval a lone wave gets),
//   * LDS pointer chases (ds_read_b32 / b64 / b128, ds_bpermute_b32), an L2-resident global pointer chase,
//   * VALU -> SALU -> VALU hops (v_readfirstlane_b32, v_readlane_b32 with an SGPR lane select).
// Build: hipcc -O2 --offload-arch=gfx950 tools/microbench_lone_wave.hip -o /tmp/lone; output: profiles/r04_lone_wave.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

struct Out { long long cycles; int sink; };

#define DEP_KERNEL(name, init, text)                                                          \
    __global__ void name(Out* out, int n, int a, float fa) {                                  \
        int x = threadIdx.x + a;                                                              \
        float f = (float)threadIdx.x + fa;                                                    \
        (void)f; (void)x;                                                                     \
        init;                                                                                 \
        const long long c0 = clock64();                                                       \
        for (int i = 0; i < n; ++i) { REP16(text) }                                           \
        const long long c1 = clock64();                                                       \
        if (threadIdx.x == 0) { out->cycles = c1 - c0; out->sink = x + (int)f; }              \
    }

// dependent chains
DEP_KERNEL(dep_add, , asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));)
DEP_KERNEL(dep_lshl_add, , asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x) : "v"(a));)
DEP_KERNEL(dep_add3, , asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(a));)
DEP_KERNEL(dep_mad24, , asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(x) : "v"(a));)
DEP_KERNEL(dep_mul_lo, , asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(a));)
DEP_KERNEL(dep_med3_i, , asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(x) : "v"(a));)
DEP_KERNEL(dep_min_u, , asm volatile("v_min_u32 %0, %0, %1" : "+v"(x) : "v"(a));)
DEP_KERNEL(dep_bfe, , asm volatile("v_bfe_u32 %0, %0, %1, 30" : "+v"(x) : "v"(a));)
DEP_KERNEL(dep_bfi, , asm volatile("v_bfi_b32 %0, %1, %0, %0" : "+v"(x) : "v"(a));)
DEP_KERNEL(dep_perm, , asm volatile("v_perm_b32 %0, %0, %0, %1" : "+v"(x) : "v"(a));)
DEP_KERNEL(dep_sad, , asm volatile("v_sad_u32 %0, %0, %1, 0" : "+v"(x) : "v"(a));)
DEP_KERNEL(dep_and_or, , asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x) : "v"(a));)
DEP_KERNEL(dep_mul_f32, , asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f) : "v"(fa));)
DEP_KERNEL(dep_add_f32, , asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(fa));)
DEP_KERNEL(dep_fma_f32, , asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f) : "v"(fa));)
DEP_KERNEL(dep_med3_f, , asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(f) : "v"(fa));)
DEP_KERNEL(dep_min_f, , asm volatile("v_min_f32 %0, %0, %1" : "+v"(f) : "v"(fa));)
DEP_KERNEL(dep_trunc_f, , asm volatile("v_trunc_f32 %0, %0" : "+v"(f));)
DEP_KERNEL(dep_rcp_f, , asm volatile("v_rcp_f32 %0, %0" : "+v"(f));)
DEP_KERNEL(dep_cvt_u2f, , asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x));)
DEP_KERNEL(dep_cvt_f2u, , asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(x));)
DEP_KERNEL(dep_cmp_cnd, , asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");)
DEP_KERNEL(dep_dpp_mov, , asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));)
DEP_KERNEL(dep_dpp_add, , asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x));)
DEP_KERNEL(dep_sdwa_add, , asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "+v"(x) : "v"(a));)
DEP_KERNEL(dep_pk_add, , asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(x) : "v"(a));)
DEP_KERNEL(dep_readfirst, int s = 0;, asm volatile("v_readfirstlane_b32 %1, %0\n v_add_u32 %0, %1, %2" : "+v"(x), "+s"(s) : "v"(a));)
DEP_KERNEL(dep_readlane, int s = 0;, asm volatile("v_readfirstlane_b32 %1, %0\n s_and_b32 %1, %1, 63\n s_nop 3\n v_readlane_b32 %1, %0, %1\n v_add_u32 %0, %1, %2" : "+v"(x), "+s"(s) : "v"(a));)
DEP_KERNEL(dep_salu, int s = a;, asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(a) : "scc");)
DEP_KERNEL(dep_smul, int s = a;, asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s) : "s"(a));)

// eight independent chains of one kind
#define IND_KERNEL(name, op)                                                                                   \
    __global__ void name(Out* out, int n, int a, float fa) {                                                   \
        int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
        (void)fa;                                                                                              \
        const long long c0 = clock64();                                                                        \
        for (int i = 0; i < n; ++i) {                                                                          \
            REP4(asm volatile(op : "+v"(x0) : "v"(a)); asm volatile(op : "+v"(x1) : "v"(a));                   \
                 asm volatile(op : "+v"(x2) : "v"(a)); asm volatile(op : "+v"(x3) : "v"(a));)                   \
            REP4(asm volatile(op : "+v"(x4) : "v"(a)); asm volatile(op : "+v"(x5) : "v"(a));                   \
                 asm volatile(op : "+v"(x6) : "v"(a)); asm volatile(op : "+v"(x7) : "v"(a));)                   \
        }                                                                                                      \
        const long long c1 = clock64();                                                                        \
        if (threadIdx.x == 0) { out->cycles = (c1 - c0) / 2; out->sink = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; } \
    }
IND_KERNEL(ind_add, "v_add_u32 %0, %0, %1")
IND_KERNEL(ind_mad24, "v_mad_i32_i24 %0, %0, %1, %0")
IND_KERNEL(ind_fma, "v_fma_f32 %0, %0, %1, %1")
IND_KERNEL(ind_med3, "v_med3_i32 %0, %0, %1, %1")
IND_KERNEL(ind_perm, "v_perm_b32 %0, %0, %0, %1")
IND_KERNEL(ind_cvt, "v_cvt_f32_u32 %0, %0")

// LDS pointer chases: the loaded word is the next address
template <int kBytes>
__global__ void lds_chase(Out* out, int n, int a, float) {
    __shared__ __attribute__((aligned(16))) uint32_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (uint32_t)(((i / 4 * 4 + 16 * 17) & 4095) * 4);   // byte address of another 16-byte slot
    __syncthreads();
    uint32_t p = (threadIdx.x * 16u + (uint32_t)a) & 16383u;
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (kBytes == 4) { REP16(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(p));) }
        if (kBytes == 8) { REP16({ uint64_t q; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(p)); p = (uint32_t)q; }) }
        if (kBytes == 16) { REP16({ uint4 q; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(p)); p = q.x; }) }
    }
    const long long c1 = clock64();
    if (threadIdx.x == 0) { out->cycles = c1 - c0; out->sink = (int)p; }
}

__global__ void bpermute_chase(Out* out, int n, int a, float) {
    uint32_t p = (threadIdx.x * 4u + 4u * (uint32_t)a) & 255u, v = ((threadIdx.x + 17u) & 63u) * 4u;
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) { REP16(asm volatile("ds_bpermute_b32 %0, %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(p) : "v"(v));) }
    const long long c1 = clock64();
    if (threadIdx.x == 0) { out->cycles = c1 - c0; out->sink = (int)p; }
}

// LDS reads that are NOT on the chain: sixteen independent ds_read_b128 per trip behind one wait (what look-ahead costs)
__global__ void lds_stream(Out* out, int n, int a, float) {
    __shared__ __attribute__((aligned(16))) uint32_t s[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (uint32_t)i;
    __syncthreads();
    const uint32_t p = (threadIdx.x * 16u + (uint32_t)a) & 16383u;
    uint32_t acc = 0;
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
        REP16({ uint4 q; asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(p)); asm volatile("s_waitcnt lgkmcnt(8)"); acc += q.x; })
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long c1 = clock64();
    if (threadIdx.x == 0) { out->cycles = c1 - c0; out->sink = (int)acc; }
}

__global__ void global_chase(Out* out, int n, int a, float, const uint32_t* __restrict__ ring) {
    uint32_t p = (threadIdx.x * 64u + (uint32_t)a) & 65535u;
    const long long c0 = clock64();
    for (int i = 0; i < n; ++i) { REP16(p = __builtin_nontemporal_load(ring + p) & 65535u; asm volatile("" : "+v"(p));) }
    const long long c1 = clock64();
    if (threadIdx.x == 0) { out->cycles = c1 - c0; out->sink = (int)p; }
}

template <typename K, typename... A>
static void run(const char* what, K kern, int threads, Out* d_out, A... extra) {
    const int n = 20000;
    Out o{};
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, d_out, n, 3, 1.0009765625f, extra...);
        hipDeviceSynchronize();
    }
    hipMemcpy(&o, d_out, sizeof(o), hipMemcpyDeviceToHost);
    printf("%-58s %7.2f cycles per instruction\n", what, (double)o.cycles / (16.0 * n));
}

int main() {
    Out* d_out;
    hipMalloc(&d_out, sizeof(Out));
    uint32_t* ring;
    hipMalloc(&ring, 65536 * 4);
    {
        uint32_t* h = new uint32_t[65536];
        for (uint32_t i = 0; i < 65536; ++i) h[i] = (i * 64u + 4099u * 64u) & 65535u;
        hipMemcpy(ring, h, 65536 * 4, hipMemcpyHostToDevice);
        delete[] h;
    }
    printf("one wave on the chip, dependent chain of one instruction kind:\n");
#define D(k) run(#k, k, 64, d_out)
    D(dep_add); D(dep_lshl_add); D(dep_add3); D(dep_mad24); D(dep_mul_lo); D(dep_med3_i); D(dep_min_u); D(dep_bfe); D(dep_bfi); D(dep_perm);
    D(dep_sad); D(dep_and_or); D(dep_mul_f32); D(dep_add_f32); D(dep_fma_f32); D(dep_med3_f); D(dep_min_f); D(dep_trunc_f); D(dep_rcp_f);
    D(dep_cvt_u2f); D(dep_cvt_f2u); D(dep_dpp_mov); D(dep_dpp_add); D(dep_sdwa_add); D(dep_pk_add);
    run("dep v_cmp + v_cndmask (per pair)", dep_cmp_cnd, 64, d_out);
    run("v_readfirstlane -> v_add (per pair)", dep_readfirst, 64, d_out);
    run("v_readfirstlane, s_and, v_readlane (SGPR select), v_add (per four + s_nop 3)", dep_readlane, 64, d_out);
    D(dep_salu); D(dep_smul);
    printf("one wave, eight independent chains (issue interval):\n");
    D(ind_add); D(ind_mad24); D(ind_fma); D(ind_med3); D(ind_perm); D(ind_cvt);
    printf("one wave, memory on the chain (per access, wait included):\n");
    run("ds_read_b32 chase", lds_chase<4>, 64, d_out);
    run("ds_read_b64 chase", lds_chase<8>, 64, d_out);
    run("ds_read_b128 chase", lds_chase<16>, 64, d_out);
    run("ds_bpermute_b32 chase", bpermute_chase, 64, d_out);
    run("ds_read_b128, sixteen in flight (per read)", lds_stream, 64, d_out);
    run("global load chase, 256 KB ring (L2 / MALL resident)", global_chase, 64, d_out, (const uint32_t*)ring);
    printf("four waves of one workgroup (one per SIMD), same chains:\n");
    run("dep_add x4 waves", dep_add, 256, d_out);
    run("ds_read_b128 chase x4 waves", lds_chase<16>, 256, d_out);
    printf("two, three and four waves per SIMD (workgroups of 512, 768, 1024), the same dependent chains -- per instruction of ONE wave:\n");
    for (int t = 512; t <= 1024; t += 256) {
        char name[96];
        snprintf(name, sizeof name, "dep_add x%d waves", t / 64); run(name, dep_add, t, d_out);
        snprintf(name, sizeof name, "dep_fma_f32 x%d waves", t / 64); run(name, dep_fma_f32, t, d_out);
        snprintf(name, sizeof name, "dep_sdwa_add x%d waves", t / 64); run(name, dep_sdwa_add, t, d_out);
        snprintf(name, sizeof name, "dep_med3_f x%d waves", t / 64); run(name, dep_med3_f, t, d_out);
        snprintf(name, sizeof name, "ds_read_b64 chase x%d waves", t / 64); run(name, lds_chase<8>, t, d_out);
        snprintf(name, sizeof name, "ds_read_b128 chase x%d waves", t / 64); run(name, lds_chase<16>, t, d_out);
    }
    return 0;
}
