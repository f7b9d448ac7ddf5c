// Small data-movement kernels that replace the torch-side glue of a training step (cat / index_select / relu / mul /
// strided mask copies and their autograd-generated backward): the FPN levels of a batch are laid out as ONE gapped
// sequence for the shared heads (libs/modeling/av_fd_no_recon.py:75-89,144-159), the heads' outputs are compacted back
// to the point order, the per-level Scale + ReLU of the regression head (blocks.py:1374-1392, av_fd_no_recon.py:155-156)
// rides along, and the mask pyramid of MaskedConv1D's nearest down-sampling (blocks.py:51-55) is one launch.
// All HBM-bound, a few hundred KB to a few MB per launch.
#include "common.h"

namespace {

constexpr int kMaxLevels = 8;
struct LevelPtrs { const float* p[kMaxLevels]; };
struct LevelOut { float* p[kMaxLevels]; };
struct LevelLens { int len[kMaxLevels]; int off[kMaxLevels]; int n; };       // off: first row of the level in the gapped sequence

__device__ __forceinline__ int level_of_gapped_row(const LevelLens& L, int g, int& t) {
  // rows [off_l, off_l + len_l) belong to level l; row off_l + len_l is its gap row
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) {
    if (l >= L.n) break;
    if (g < L.off[l] + L.len[l] + 1) { t = g - L.off[l]; return t < L.len[l] ? l : -1; }
  }
  t = 0;
  return -1;
}

// out[b, off_l + t, :] = x_l[b, t, :]; gap rows = 0.  mask_out[b, g] likewise from the level masks (optional).
__global__ __launch_bounds__(256) void gap_concat_fwd_kernel(LevelPtrs x, LevelPtrs m, LevelLens L, float* __restrict__ out,
                                                             float* __restrict__ mask_out, int B, int C, int Tg) {
  const int C4 = C >> 2;
  const long total = (long)B * Tg * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const long row = i / C4;
    const int g = (int)(row % Tg), b = (int)(row / Tg);
    int t;
    const int l = level_of_gapped_row(L, g, t);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (l >= 0) v = *reinterpret_cast<const float4*>(x.p[l] + ((size_t)b * L.len[l] + t) * C + c4 * 4);
    *reinterpret_cast<float4*>(out + (size_t)row * C + c4 * 4) = v;
    if (mask_out != nullptr && c4 == 0) mask_out[row] = l >= 0 ? m.p[l][(size_t)b * L.len[l] + t] : 0.f;
  }
}

// dx_l[b, t, :] = dout[b, off_l + t, :]
__global__ __launch_bounds__(256) void gap_concat_bwd_kernel(const float* __restrict__ dout, LevelOut dx, LevelLens L, int B, int C,
                                                             int Tg) {
  const int C4 = C >> 2;
  const long total = (long)B * Tg * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % C4);
    const long row = i / C4;
    const int g = (int)(row % Tg), b = (int)(row / Tg);
    int t;
    const int l = level_of_gapped_row(L, g, t);
    if (l < 0 || dx.p[l] == nullptr) continue;
    *reinterpret_cast<float4*>(dx.p[l] + ((size_t)b * L.len[l] + t) * C + c4 * 4) =
        *reinterpret_cast<const float4*>(dout + (size_t)row * C + c4 * 4);
  }
}

// out[b, p, j] = f(y[b, g(p), j] * scale_l)   (f = relu when `relu`; scale == null -> 1); p runs over the real rows in level order
__global__ __launch_bounds__(256) void head_out_fwd_kernel(const float* __restrict__ y, LevelPtrs scale, LevelLens L,
                                                           float* __restrict__ out, int B, int NC, int Tg, int P, int relu) {
  const long total = (long)B * Tg * NC;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int j = (int)(i % NC);
    const long row = i / NC;
    const int g = (int)(row % Tg), b = (int)(row / Tg);
    int t;
    const int l = level_of_gapped_row(L, g, t);
    if (l < 0) continue;
    float v = y[i];
    if (scale.p[l] != nullptr) v *= scale.p[l][0];
    if (relu) v = fmaxf(v, 0.f);
    out[((size_t)b * P + (L.off[l] - l) + t) * NC + j] = v;          // off_l - l = number of real rows before level l
  }
}

// dy[b, g, j] = dout[b, p, j] * scale_l * [out > 0] (gap rows 0);  dscale_l += sum dout * [out > 0] * y   (block partials + atomics)
__global__ __launch_bounds__(256) void head_out_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                           const float* __restrict__ out, LevelPtrs scale, LevelOut dscale,
                                                           LevelLens L, float* __restrict__ dy, int B, int NC, int Tg, int P,
                                                           int relu) {
  __shared__ float part[kMaxLevels];
  if (threadIdx.x < kMaxLevels) part[threadIdx.x] = 0.f;
  __syncthreads();
  const long total = (long)B * Tg * NC;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int j = (int)(i % NC);
    const long row = i / NC;
    const int g = (int)(row % Tg), b = (int)(row / Tg);
    int t;
    const int l = level_of_gapped_row(L, g, t);
    float d = 0.f;
    if (l >= 0) {
      const size_t o = ((size_t)b * P + (L.off[l] - l) + t) * NC + j;
      float go = dout[o];
      if (relu && !(out[o] > 0.f)) go = 0.f;
      if (scale.p[l] != nullptr) {
        if (dscale.p[l] != nullptr && go != 0.f) atomicAdd(&part[l], go * y[i]);
        go *= scale.p[l][0];
      }
      d = go;
    }
    dy[i] = d;
  }
  __syncthreads();
  if (threadIdx.x < L.n && dscale.p[threadIdx.x] != nullptr && part[threadIdx.x] != 0.f)
    atomicAdd(dscale.p[threadIdx.x], part[threadIdx.x]);
}

// out_l[b, t] = mask[b, t * stride^l]  for l = 1 .. n  (MaskedConv1D applies mask[:, ::stride] level after level)
__global__ __launch_bounds__(256) void mask_pyramid_kernel(const float* __restrict__ mask, LevelOut out, LevelLens L, int B, int T) {
  const int b = blockIdx.y;
  for (int l = 0; l < L.n; ++l) {
    const int step = L.off[l];                         // stride^(l+1)
    for (int t = blockIdx.x * 256 + threadIdx.x; t < L.len[l]; t += gridDim.x * 256)
      out.p[l][(size_t)b * L.len[l] + t] = mask[(size_t)b * T + (size_t)t * step];
  }
}

// stochastic-depth / dropout keep factors from uniform draws u in [0, 1):
//   mode 0 (AffineDropPath, blocks.py:1403-1409): out[r, b] = floor(keep_prob[r] + u[r, b]) / keep_prob[r]
//   mode 1 (nn.Dropout(p)):                          out[i]    = u[i] >= p ? 1 / (1 - p) : 0
__global__ __launch_bounds__(256) void keep_from_uniform_kernel(const float* __restrict__ u, const float* __restrict__ kp,
                                                                float* __restrict__ out, long n, int cols, int mode, float p) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (mode == 0) {
      const float k = kp[i / cols];
      out[i] = floorf(k + u[i]) / k;
    } else {
      out[i] = u[i] >= p ? 1.f / (1.f - p) : 0.f;
    }
  }
}

inline int grid_for(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

int fill_lens(LevelLens& L, const int* lens, int n) {
  L.n = n;
  int off = 0;
  for (int l = 0; l < n; ++l) { L.len[l] = lens[l]; L.off[l] = off; off += lens[l] + 1; }
  for (int l = n; l < kMaxLevels; ++l) { L.len[l] = 0; L.off[l] = off; }
  return off;
}

}  // namespace

extern "C" int avdf_gap_concat_fwd(const float* const* x, const float* const* masks, const int* lens, int n_levels, float* out,
                                   float* mask_out, int B, int C, void* stream) {
  AVDF_REQUIRE(x && lens && out && n_levels >= 1 && n_levels <= kMaxLevels && B > 0 && C > 0 && C % 4 == 0, "gap_concat_fwd: bad args");
  AVDF_REQUIRE((mask_out == nullptr) == (masks == nullptr), "gap_concat_fwd: masks and mask_out go together");
  LevelPtrs px{}, pm{};
  LevelLens L;
  const int Tg = fill_lens(L, lens, n_levels);
  for (int l = 0; l < n_levels; ++l) {
    AVDF_REQUIRE(x[l] && lens[l] > 0 && ((uintptr_t)x[l] & 15) == 0, "gap_concat_fwd: level %d", l);
    px.p[l] = x[l];
    pm.p[l] = masks ? masks[l] : nullptr;
  }
  AVDF_REQUIRE(((uintptr_t)out & 15) == 0, "gap_concat_fwd: 16-byte alignment");
  hipLaunchKernelGGL(gap_concat_fwd_kernel, dim3(grid_for((long)B * Tg * (C / 4))), dim3(256), 0, (hipStream_t)stream, px, pm, L, out,
                     mask_out, B, C, Tg);
  AVDF_CHECK_LAUNCH("gap_concat_fwd");
  return AVDF_OK;
}

extern "C" int avdf_gap_concat_bwd(const float* dout, float* const* dx, const int* lens, int n_levels, int B, int C, void* stream) {
  AVDF_REQUIRE(dout && dx && lens && n_levels >= 1 && n_levels <= kMaxLevels && B > 0 && C > 0 && C % 4 == 0, "gap_concat_bwd: bad args");
  LevelOut pd{};
  LevelLens L;
  const int Tg = fill_lens(L, lens, n_levels);
  for (int l = 0; l < n_levels; ++l) {
    AVDF_REQUIRE(((uintptr_t)dx[l] & 15) == 0, "gap_concat_bwd: level %d alignment", l);
    pd.p[l] = dx[l];
  }
  hipLaunchKernelGGL(gap_concat_bwd_kernel, dim3(grid_for((long)B * Tg * (C / 4))), dim3(256), 0, (hipStream_t)stream, dout, pd, L, B, C, Tg);
  AVDF_CHECK_LAUNCH("gap_concat_bwd");
  return AVDF_OK;
}

extern "C" int avdf_head_out_fwd(const float* y, const float* const* scale, const int* lens, int n_levels, float* out, int B, int NC,
                                 int relu, void* stream) {
  AVDF_REQUIRE(y && lens && out && n_levels >= 1 && n_levels <= kMaxLevels && B > 0 && NC > 0, "head_out_fwd: bad args");
  LevelPtrs ps{};
  LevelLens L;
  const int Tg = fill_lens(L, lens, n_levels);
  for (int l = 0; l < n_levels; ++l) ps.p[l] = scale ? scale[l] : nullptr;
  const int P = Tg - n_levels;
  hipLaunchKernelGGL(head_out_fwd_kernel, dim3(grid_for((long)B * Tg * NC)), dim3(256), 0, (hipStream_t)stream, y, ps, L, out, B, NC, Tg,
                     P, relu);
  AVDF_CHECK_LAUNCH("head_out_fwd");
  return AVDF_OK;
}

extern "C" int avdf_head_out_bwd(const float* dout, const float* y, const float* out, const float* const* scale, float* const* dscale,
                                 const int* lens, int n_levels, float* dy, int B, int NC, int relu, void* stream) {
  AVDF_REQUIRE(dout && y && out && lens && dy && n_levels >= 1 && n_levels <= kMaxLevels && B > 0 && NC > 0, "head_out_bwd: bad args");
  LevelPtrs ps{};
  LevelOut pd{};
  LevelLens L;
  const int Tg = fill_lens(L, lens, n_levels);
  for (int l = 0; l < n_levels; ++l) {
    ps.p[l] = scale ? scale[l] : nullptr;
    pd.p[l] = dscale ? dscale[l] : nullptr;
  }
  const int P = Tg - n_levels;
  long blocks = grid_for((long)B * Tg * NC);
  if (blocks > 64) blocks = 64;                        // <= 64 x n_levels global atomics for the scale gradients
  hipLaunchKernelGGL(head_out_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, y, out, ps, pd, L, dy, B, NC,
                     Tg, P, relu);
  AVDF_CHECK_LAUNCH("head_out_bwd");
  return AVDF_OK;
}

extern "C" int avdf_mask_pyramid(const float* mask, float* const* out, int n_levels, int B, int T, int stride, void* stream) {
  AVDF_REQUIRE(mask && out && n_levels >= 1 && n_levels <= kMaxLevels && B > 0 && T > 0 && stride >= 2, "mask_pyramid: bad args");
  LevelOut po{};
  LevelLens L;
  L.n = n_levels;
  int step = 1, len = T;
  for (int l = 0; l < n_levels; ++l) {
    AVDF_REQUIRE(len % stride == 0 && out[l], "mask_pyramid: level %d: length %d not divisible by %d", l, len, stride);
    step *= stride;
    len /= stride;
    L.len[l] = len;
    L.off[l] = step;
    po.p[l] = out[l];
  }
  hipLaunchKernelGGL(mask_pyramid_kernel, dim3(cdiv(T / stride, 256), B), dim3(256), 0, (hipStream_t)stream, mask, po, L, B, T);
  AVDF_CHECK_LAUNCH("mask_pyramid");
  return AVDF_OK;
}

extern "C" int avdf_keep_from_uniform(const float* u, const float* keep_prob, float* out, int64_t n, int cols, int mode, float p,
                                      void* stream) {
  AVDF_REQUIRE(u && out && n > 0 && cols > 0 && (mode == 0 || mode == 1), "keep_from_uniform: bad args");
  AVDF_REQUIRE(mode == 1 || keep_prob, "keep_from_uniform: drop-path mode needs the per-row keep probabilities");
  AVDF_REQUIRE(mode == 0 || (p >= 0.f && p < 1.f), "keep_from_uniform: dropout p in [0, 1)");
  hipLaunchKernelGGL(keep_from_uniform_kernel, dim3(grid_for((long)n)), dim3(256), 0, (hipStream_t)stream, u, keep_prob, out, (long)n,
                     cols, mode, p);
  AVDF_CHECK_LAUNCH("keep_from_uniform");
  return AVDF_OK;
}
