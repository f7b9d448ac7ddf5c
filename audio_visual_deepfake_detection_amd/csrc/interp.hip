// K1: feature-length linear interpolation + channel concat, on device.
// Replaces the F.interpolate(mode='linear', align_corners=False) x3 + torch.cat block of
// DeepFakeVideoAudioDataset.__getitem__ (libs/datasets/deepfake_video_audio.py:1722-1756), which the
// reference runs on the CPU in the DataLoader workers (65 ms/clip).  Source streams are the
// on-disk [T_s, C_s] time-major arrays, so the resample reads two contiguous rows per output row
// and writes straight into the time-major [Tout, Ctot] model input at channel offset `coff`.
// Pure HBM streaming: 16-byte loads/stores, one output float4 per thread.
//
// Index/weight arithmetic restates ATen's area_pixel_compute_source_index + guard_index_and_lambda
// (fp32): src = scale*(t+0.5)-0.5 clamped at 0, i0 = min(floor(src), Ts-1), lam = clamp(src-i0,0,1),
// i1 = i0 + (i0 < Ts-1), out = (1-lam)*x[i0] + lam*x[i1].
// Rounding: ATen's CPU kernels are built with FMA contraction, i.e. src = fma(scale, t+0.5, -0.5)
// and out = fma(1-lam, x[i0], round(lam*x[i1])); the same two fused operations are used here, which
// reproduces the reference's output bit-for-bit on the golden vectors (tests/golden/interp.npz).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void interp_linear_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int Tsrc, int Csrc, int Tout, int Ctot, int coff,
                                                            long src_bstride, long dst_bstride) {
  src += (size_t)blockIdx.y * src_bstride;        // blockIdx.y = clip of a batched call
  dst += (size_t)blockIdx.y * dst_bstride;
  const int C4 = Csrc >> 2;
  const long total = (long)Tout * C4;
  const float scale = (float)Tsrc / (float)Tout;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const int t = (int)(i / C4);
    float4 o;
    if (Tsrc == Tout) {
      o = *reinterpret_cast<const float4*>(src + (size_t)t * Csrc + c);
    } else {
      float sp = fmaf(scale, (float)t + 0.5f, -0.5f);
      if (sp < 0.f) sp = 0.f;
      int i0 = (int)floorf(sp);
      if (i0 > Tsrc - 1) i0 = Tsrc - 1;
      float lam = sp - (float)i0;
      lam = fminf(fmaxf(lam, 0.f), 1.f);
      const int i1 = i0 + (i0 < Tsrc - 1 ? 1 : 0);
      const float w0 = 1.f - lam, w1 = lam;
      const float4 a = *reinterpret_cast<const float4*>(src + (size_t)i0 * Csrc + c);
      const float4 b = *reinterpret_cast<const float4*>(src + (size_t)i1 * Csrc + c);
      o.x = fmaf(w0, a.x, __fmul_rn(w1, b.x));
      o.y = fmaf(w0, a.y, __fmul_rn(w1, b.y));
      o.z = fmaf(w0, a.z, __fmul_rn(w1, b.z));
      o.w = fmaf(w0, a.w, __fmul_rn(w1, b.w));
    }
    *reinterpret_cast<float4*>(dst + (size_t)t * Ctot + coff + c) = o;
  }
}

}  // namespace

extern "C" int avdf_interp_linear(const float* src, float* dst, int Tsrc, int Csrc, int Tout, int Ctot,
                                  int coff, void* stream) {
  AVDF_REQUIRE(src && dst && Tsrc > 0 && Csrc > 0 && Tout > 0 && Ctot >= coff + Csrc && coff >= 0, "interp_linear: bad args");
  AVDF_REQUIRE(Csrc % 4 == 0 && coff % 4 == 0 && Ctot % 4 == 0, "interp_linear: channel counts/offset must be multiples of 4");
  AVDF_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "interp_linear: 16-byte alignment");
  long g = ((long)Tout * Csrc / 4 + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(interp_linear_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, src, dst, Tsrc, Csrc, Tout,
                     Ctot, coff, 0L, 0L);
  AVDF_CHECK_LAUNCH("interp_linear");
  return AVDF_OK;
}

// nb clips in one launch: clip b reads src + b*src_bstride ([Tsrc, Csrc]) and writes dst + b*dst_bstride ([Tout, Ctot])
extern "C" int avdf_interp_linear_batched(const float* src, float* dst, int nb, int64_t src_bstride, int64_t dst_bstride,
                                          int Tsrc, int Csrc, int Tout, int Ctot, int coff, void* stream) {
  AVDF_REQUIRE(src && dst && nb > 0 && Tsrc > 0 && Csrc > 0 && Tout > 0 && Ctot >= coff + Csrc && coff >= 0, "interp_linear_batched: bad args");
  AVDF_REQUIRE(Csrc % 4 == 0 && coff % 4 == 0 && Ctot % 4 == 0 && src_bstride % 4 == 0 && dst_bstride % 4 == 0,
               "interp_linear_batched: channel counts / offsets / strides must be multiples of 4");
  AVDF_REQUIRE(src_bstride >= (int64_t)Tsrc * Csrc && dst_bstride >= (int64_t)Tout * Ctot, "interp_linear_batched: strides too small");
  AVDF_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "interp_linear_batched: 16-byte alignment");
  long g = ((long)Tout * Csrc / 4 + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(interp_linear_kernel, dim3((unsigned)g, nb), dim3(256), 0, (hipStream_t)stream, src, dst, Tsrc, Csrc, Tout,
                     Ctot, coff, (long)src_bstride, (long)dst_bstride);
  AVDF_CHECK_LAUNCH("interp_linear_batched");
  return AVDF_OK;
}
