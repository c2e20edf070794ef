// amv_resample.hip -- the picture rescaler in front of the encoder: img_resample of the reference
// (AMVmuxer/ffmpeg/libavcodec/imgresample.c:474-495), the routine behind the sws_scale shim (:599) that
// ffmpeg.c:757 calls when the source is not the 160x120 an AMV player wants (AMVmuxer/Makefile:15-17).
//
// component_resample (:341-405): every source line is filtered horizontally by a four-tap, 16-phase polyphase
// filter into BYTES (h_resample :312-339, the sum >> 8 clipped to 0..255), and four such lines are filtered
// vertically the same way (v_resample :119-153).  Positions are 16.16 fixed point, one pixel left of / above the
// filter's centre at the start; taps that fall outside the picture repeat the edge sample (h_resample_slow :288-310,
// the line clamp of :361-366).  The same increments and filters serve luma and chroma (img_resample :474-495).
//
// One thread per output sample: four horizontal sums over 4 x 4 source bytes, one vertical sum -- the work the
// reference saves by keeping filtered lines in a ring is recomputed (16 multiply-adds per sample), which costs less
// than passing the intermediate lines through memory.  The filters (2 x 64 int16, built on the host exactly as
// av_build_filter does, libavcodec/resample2.c:93-140) sit in LDS.
#include "amv_kernels.h"

namespace amv {

namespace {
constexpr int kPosBits = 16, kPhaseBits = 4, kFilterBits = 8;

__device__ __forceinline__ int phase_of(int pos) { return (pos >> (kPosBits - kPhaseBits)) & 15; }
__device__ __forceinline__ int clip8(int sum) { return min(max(sum >> kFilterBits, 0), 255); }
}  // namespace

__global__ __launch_bounds__(256) void amv_resample_kernel(ResamplePlanes src, ResamplePlanes dst, ResampleFilters f, uint32_t n) {
    __shared__ int16_t s_f[128];
    if (threadIdx.x < 128) s_f[threadIdx.x] = threadIdx.x < 64 ? f.h[threadIdx.x] : f.v[threadIdx.x - 64];
    __syncthreads();
    const uint32_t plane = blockIdx.y, frame = blockIdx.z;
    const int iw = plane ? (int)(src.width >> 1) : (int)src.width, ih = plane ? (int)(src.height >> 1) : (int)src.height;
    const int ow = plane ? (int)(dst.width >> 1) : (int)dst.width, oh = plane ? (int)(dst.height >> 1) : (int)dst.height;
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= (uint32_t)(ow * oh)) return;
    const int y = (int)(t / (uint32_t)ow), x = (int)(t - (uint32_t)y * (uint32_t)ow);
    const uint8_t* in = (plane == 0 ? src.y : (plane == 1 ? src.cb : src.cr)) + (uint64_t)frame * (plane ? src.c_frame : src.y_frame);
    uint8_t* out = (plane == 0 ? dst.y : (plane == 1 ? dst.cb : dst.cr)) + (uint64_t)frame * (plane ? dst.c_frame : dst.y_frame);
    const uint32_t istride = plane ? src.c_stride : src.y_stride, ostride = plane ? dst.c_stride : dst.y_stride;

    const int hpos = -(1 << kPosBits) + x * f.h_incr;             // src_start = -FCENTER * POS_FRAC (:367)
    const int s0 = hpos >> kPosBits;
    const int16_t* hf = s_f + phase_of(hpos) * 4;
    int sx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) sx[j] = min(max(s0 + j, 0), iw - 1);
    const int vpos = 2 * (1 << kPosBits) + y * f.v_incr;          // (last_src_y + NB_TAPS) * POS_FRAC (:350)
    const int y1 = vpos >> kPosBits;
    const int16_t* vf = s_f + 64 + phase_of(vpos) * 4;
    int sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int line = min(max(y1 - 3 + j, 0), ih - 1);        // the ring's last four lines (:400), clamped (:361-366)
        const uint8_t* row = in + (uint64_t)line * istride;
        const int h = row[sx[0]] * hf[0] + row[sx[1]] * hf[1] + row[sx[2]] * hf[2] + row[sx[3]] * hf[3];
        sum += clip8(h) * vf[j];
    }
    out[(uint64_t)y * ostride + x] = (uint8_t)clip8(sum);
}

void launch_resample(const ResamplePlanes& src, const ResamplePlanes& dst, const ResampleFilters& f, uint32_t n, hipStream_t s) {
    if (n == 0) return;
    const uint32_t px = dst.width * dst.height;
    hipLaunchKernelGGL(amv_resample_kernel, dim3((px + 255u) / 256u, 3, n), dim3(256), 0, s, src, dst, f, n);
}

}  // namespace amv
