// Audio front-end + BYOL-A encoder glue (frozen feature extractor, forward only).
//   * log-mel front-end of the reference's BYOL-A extraction (audio_feature/content_audio/
//     extract_audio_feature_batch.py:42-60, config.yaml:4-10): torchaudio MelSpectrogram(16 kHz,
//     n_fft 1024, win 1024, hop 160, 64 mel, 60-7800 Hz, power 2, centre + reflect pad, periodic Hann,
//     HTK, no norm) -> log(x + eps) -> (x - mean) / std.  The STFT is a windowed-frame x DFT-matrix GEMM and
//     the mel projection a second GEMM, both on the fp32 MFMA conv kernel (gemm_conv.hip); the kernels here
//     are the HBM-bound pieces around them.
//   * AudioNTT2020Task6 CNN (audio_feature/content_audio/byol_a/models.py:53-85): conv 3x3 + BN(eval,
//     folded) + ReLU + MaxPool 2x2, three times; conv2/conv3 run as ONE k=3 conv1d GEMM over time on a
//     mel-im2col'd tensor (channels = 3 mel taps x 64).
// Layout of CNN activations: [M (mel), T (time), C] fp32, C contiguous.
#include "common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// frames[f, j] = window[j] * x[reflect(f*hop + j - nfft/2)]
__global__ __launch_bounds__(256) void frame_window_kernel(const float* __restrict__ x, const float* __restrict__ win,
                                                           float* __restrict__ frames, long N, int F, int nfft, int hop) {
  const long total = (long)F * nfft;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int j = (int)(i % nfft);
    const long f = i / nfft;
    long s = f * hop + j - nfft / 2;
    if (s < 0) s = -s;                       // reflect (no edge repeat), as torch.stft(pad_mode='reflect')
    if (s >= N) s = 2 * (N - 1) - s;
    frames[i] = win[j] * x[s];
  }
}

// spec [F, 2*KP] = (re[0..KP), im[0..KP)) -> power [F, KP]
__global__ __launch_bounds__(256) void power_kernel(const float* __restrict__ spec, float* __restrict__ pw, long F, int KP) {
  const long total = F * KP;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long f = i / KP;
    const int k = (int)(i % KP);
    const float re = spec[f * 2 * KP + k], im = spec[f * 2 * KP + KP + k];
    pw[i] = re * re + im * im;
  }
}

// lms[m, f] (mel-major image, time contiguous) = (log(mel[f, m] + eps) - mean) / std
__global__ __launch_bounds__(256) void lognorm_kernel(const float* __restrict__ mel, float* __restrict__ lms,
                                                      long F, int M, float eps, float mean, float stdv) {
  const long total = F * M;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long f = i / M;
    const int m = (int)(i % M);
    lms[(long)m * F + f] = (logf(mel[i] + eps) - mean) / stdv;
  }
}

// conv1 (1 -> C channels, 3x3, pad 1) + folded BN + ReLU + MaxPool 2x2, fused.
// x: [M, T] (mel-major image), w: [C, 9] (kh = mel tap, kw = time tap), out: [M/2, T/2, C]
__global__ __launch_bounds__(256) void conv2d_c1_pool_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             float* __restrict__ out, int M, int T, int C) {
  const int M2 = M / 2, T2 = T / 2;
  const long total = (long)M2 * T2 * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long mt = i / C;
    const int t2 = (int)(mt % T2), m2 = (int)(mt / T2);
    float wv[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) wv[q] = w[c * 9 + q];
    float patch[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int m = 2 * m2 - 1 + a, t = 2 * t2 - 1 + b;
        patch[a][b] = (m >= 0 && m < M && t >= 0 && t < T) ? x[(long)m * T + t] : 0.f;
      }
    float best = 0.f;                       // ReLU output is >= 0
    const float sc = scale[c], sh = shift[c];
#pragma unroll
    for (int pm = 0; pm < 2; ++pm)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        float s = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) s += patch[pm + kh][pt + kw] * wv[kh * 3 + kw];
        best = fmaxf(best, s * sc + sh);
      }
    out[i] = best;
  }
}

// xcol[m, t, dm*C + c] = x[m + dm - 1, t, c] (zero outside), x: [M, T, C]
__global__ __launch_bounds__(256) void mel_im2col_kernel(const float* __restrict__ x, float* __restrict__ xcol, int M, int T, int C) {
  const int C4 = C >> 2;
  const long total = (long)M * T * 3 * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    long r = i / C4;
    const int dm = (int)(r % 3);
    r /= 3;
    const int t = (int)(r % T), m = (int)(r / T);
    const int ms = m + dm - 1;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ms >= 0 && ms < M) v = ld4(x + ((long)ms * T + t) * C + c);
    st4(xcol + ((long)m * T + t) * 3 * C + dm * C + c, v);
  }
}

// MaxPool 2x2 over (m, t) of x [M, T, C]; t_major = 0: out [M/2, T/2, C]; 1: out [T/2, M/2, C]
__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                         int M, int T, int C, int t_major) {
  const int C4 = C >> 2, M2 = M / 2, T2 = T / 2;
  const long total = (long)M2 * T2 * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    const long mt = i / C4;
    const int t2 = (int)(mt % T2), m2 = (int)(mt / T2);
    float4 a = ld4(x + ((long)(2 * m2) * T + 2 * t2) * C + c);
    const float4 b = ld4(x + ((long)(2 * m2) * T + 2 * t2 + 1) * C + c);
    const float4 d = ld4(x + ((long)(2 * m2 + 1) * T + 2 * t2) * C + c);
    const float4 e = ld4(x + ((long)(2 * m2 + 1) * T + 2 * t2 + 1) * C + c);
    a.x = fmaxf(fmaxf(a.x, b.x), fmaxf(d.x, e.x)); a.y = fmaxf(fmaxf(a.y, b.y), fmaxf(d.y, e.y));
    a.z = fmaxf(fmaxf(a.z, b.z), fmaxf(d.z, e.z)); a.w = fmaxf(fmaxf(a.w, b.w), fmaxf(d.w, e.w));
    const long o = t_major ? ((long)t2 * M2 + m2) : ((long)m2 * T2 + t2);
    st4(out + o * C + c, a);
  }
}

inline int grid_for(long total) {
  long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int avdf_frame_window(const float* pcm, const float* window, float* frames, int64_t n_samples,
                                 int n_frames, int n_fft, int hop, void* stream) {
  AVDF_REQUIRE(pcm && window && frames && n_samples > n_fft / 2 && n_frames > 0 && n_fft > 0 && hop > 0, "frame_window: bad args");
  AVDF_REQUIRE((int64_t)(n_frames - 1) * hop <= n_samples, "frame_window: n_frames exceeds 1 + n_samples/hop");
  hipLaunchKernelGGL(frame_window_kernel, dim3(grid_for((long)n_frames * n_fft)), dim3(256), 0, (hipStream_t)stream,
                     pcm, window, frames, (long)n_samples, n_frames, n_fft, hop);
  AVDF_CHECK_LAUNCH("frame_window");
  return AVDF_OK;
}

extern "C" int avdf_power_spectrum(const float* spec, float* power, int64_t n_frames, int kp, void* stream) {
  AVDF_REQUIRE(spec && power && n_frames > 0 && kp > 0, "power_spectrum: bad args");
  hipLaunchKernelGGL(power_kernel, dim3(grid_for(n_frames * kp)), dim3(256), 0, (hipStream_t)stream, spec, power, (long)n_frames, kp);
  AVDF_CHECK_LAUNCH("power_spectrum");
  return AVDF_OK;
}

extern "C" int avdf_logmel_norm(const float* mel, float* lms, int64_t n_frames, int n_mels, float eps, float mean,
                                float stdv, void* stream) {
  AVDF_REQUIRE(mel && lms && n_frames > 0 && n_mels > 0 && stdv != 0.f, "logmel_norm: bad args");
  hipLaunchKernelGGL(lognorm_kernel, dim3(grid_for(n_frames * n_mels)), dim3(256), 0, (hipStream_t)stream, mel, lms,
                     (long)n_frames, n_mels, eps, mean, stdv);
  AVDF_CHECK_LAUNCH("logmel_norm");
  return AVDF_OK;
}

extern "C" int avdf_conv2d_c1_bn_relu_pool(const float* x, const float* w, const float* scale, const float* shift,
                                           float* out, int M, int T, int C, void* stream) {
  AVDF_REQUIRE(x && w && scale && shift && out && M >= 2 && T >= 2 && C > 0, "conv2d_c1_bn_relu_pool: bad args");
  hipLaunchKernelGGL(conv2d_c1_pool_kernel, dim3(grid_for((long)(M / 2) * (T / 2) * C)), dim3(256), 0, (hipStream_t)stream,
                     x, w, scale, shift, out, M, T, C);
  AVDF_CHECK_LAUNCH("conv2d_c1_bn_relu_pool");
  return AVDF_OK;
}

extern "C" int avdf_mel_im2col(const float* x, float* xcol, int M, int T, int C, void* stream) {
  AVDF_REQUIRE(x && xcol && M > 0 && T > 0 && C > 0 && C % 4 == 0 && aligned16(x) && aligned16(xcol), "mel_im2col: bad args");
  hipLaunchKernelGGL(mel_im2col_kernel, dim3(grid_for((long)M * T * 3 * C / 4)), dim3(256), 0, (hipStream_t)stream, x, xcol, M, T, C);
  AVDF_CHECK_LAUNCH("mel_im2col");
  return AVDF_OK;
}

extern "C" int avdf_maxpool2x2(const float* x, float* out, int M, int T, int C, int t_major, void* stream) {
  AVDF_REQUIRE(x && out && M >= 2 && T >= 2 && C > 0 && C % 4 == 0 && aligned16(x) && aligned16(out), "maxpool2x2: bad args");
  hipLaunchKernelGGL(maxpool2x2_kernel, dim3(grid_for((long)(M / 2) * (T / 2) * C / 4)), dim3(256), 0, (hipStream_t)stream,
                     x, out, M, T, C, t_major);
  AVDF_CHECK_LAUNCH("maxpool2x2");
  return AVDF_OK;
}
