// LAV-DF visual encoder (C3D) kernels for gfx950: frame preparation, 3x3x3 Conv3d as implicit GEMM on
// exact-fp32 MFMA, (1,p,p) max pooling.  Frozen extractor: forward only.
//
// Replaces video_feature/aligned_video/model/video_encoder.py:27-62 (C3DVideoEncoder blocks) and the
// per-frame part of extract_video_features.py:19-37 (read_video /255, tail zero pad, nearest resize).
//
// Layout: activations are voxel-major / channels-last [B, T, H, W, C] fp32 (the reference is [B, C, T, H, W]),
// so the C channels of one voxel are contiguous (float4 loads) and a (h, w) patch of one frame is the M tile.
// Weights are consumed tap-major: wp[27][Cout][Cin] (packed once at load from the reference [Cout,Cin,3,3,3]).
//
// GEMM view: M = B*T*H*W voxels, N = Cout, K = 27*Cin.  One workgroup owns a TT x TH x TW voxel patch and BN output
// channels; the K loop runs dt (3 frame offsets) x Cin chunks of KC; per iteration one haloed patch
// [TT][(TH+2)][(TW+2)][KC] and the 9 in-plane taps' weights [9][BN][KC] are staged through LDS
// (register prefetch of the next stage overlaps the MFMAs).  MFMA operand maps as in gemm_conv.hip.
#include "common.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <int N>
struct F4Buf { float4 v[N]; };

// split-bf16 operands (see gemm_conv.hip): x = hi + lo, a*b ~ hi*hi + hi*lo + lo*hi on the bf16 matrix cores
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned short bf16_bits(__bf16 v) { return __builtin_bit_cast(unsigned short, v); }
__device__ __forceinline__ void split_bf16x4(float4 v, uint2& hi, uint2& lo) {
  const __bf16 h0 = (__bf16)v.x, h1 = (__bf16)v.y, h2 = (__bf16)v.z, h3 = (__bf16)v.w;
  const __bf16 l0 = (__bf16)(v.x - (float)h0), l1 = (__bf16)(v.y - (float)h1), l2 = (__bf16)(v.z - (float)h2),
               l3 = (__bf16)(v.w - (float)h3);
  hi.x = bf16_bits(h0) | ((unsigned)bf16_bits(h1) << 16); hi.y = bf16_bits(h2) | ((unsigned)bf16_bits(h3) << 16);
  lo.x = bf16_bits(l0) | ((unsigned)bf16_bits(l1) << 16); lo.y = bf16_bits(l2) | ((unsigned)bf16_bits(l3) << 16);
}

template <int TT, int TH, int TW, int WM, int WN, int MI, int NI, int KC, bool BF3 = false>
__global__ __launch_bounds__(256) void conv3d_k3_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y,
    int B, int T, int H, int W, int Cin, int Cout, int act, int tilesT, int tilesH, int tilesW) {
  static_assert(WM * WN == 4, "4 waves");
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  static_assert(TT * TH * TW == BM, "patch must cover the M tile");
  constexpr int HH = TH + 2, HW = TW + 2;
  constexpr int HV = TT * HH * HW;               // haloed voxels of one stage
  // VEC (exact fp32, KC a multiple of 8): rows padded by 4 floats, 16-byte staging stores, and a lane fetches four
  // k-steps of an operand row with one ds_read_b128 (MFMA j of an 8-deep step contracts channels {cc + j, cc + 4 + j})
  constexpr bool VEC = !BF3 && (KC % 8 == 0);
  constexpr int LDA = VEC ? KC + 4 : KC + 1, LDB = LDA;
  constexpr int A4 = KC / 4;
  constexpr int NA = (HV * A4 + 255) / 256, NB = (9 * BN * A4 + 255) / 256;
  constexpr int AS = (HV * LDA + 3) / 4 * 4;
  // BF3: hi / lo bf16 planes, rows of exactly KC elements; the 16-byte chunk index of a row is XOR-swizzled with bit 3
  // of the row number (swz), so the 16 consecutive rows one quarter-wave reads land on 64 distinct banks without row
  // padding: 58.6 KB instead of 87.9 KB per workgroup for the 64-channel layers = two workgroups per CU
  constexpr int LDH = KC;
  auto swz = [](int row, int chunk) { return chunk ^ ((row >> 3) & 1); };
  constexpr int ASH = HV * LDH, BSH = 9 * BN * LDH;
  static_assert(!BF3 || KC % 16 == 0, "bf16x3 needs 16-deep channel chunks");
  constexpr int SMF = BF3 ? (ASH + BSH) : (AS + 9 * BN * LDB);
  __shared__ __attribute__((aligned(16))) float smem[SMF];
  float* As = smem;
  float* Bs = smem + AS;
  __bf16* Ah = reinterpret_cast<__bf16*>(smem);
  __bf16* Al = Ah + ASH;
  __bf16* Bh = Al + ASH;
  __bf16* Bl = Bh + BSH;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int li = lane & 31, lh = lane >> 5;
  int bx = blockIdx.x;
  const int tw = bx % tilesW; bx /= tilesW;
  const int th = bx % tilesH; bx /= tilesH;
  const int tt = bx % tilesT;
  const int b = bx / tilesT;
  const int t0 = tt * TT, h0 = th * TH, w0 = tw * TW;
  const int n0 = blockIdx.y * BN;

  floatx16 acc[MI * NI];
#pragma unroll
  for (int a = 0; a < MI * NI; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  // per-lane LDS offset of its A rows (voxel m -> patch coordinates)
  int abase[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = (wm * MI + mi) * 32 + li;
    const int pt = m / (TH * TW), ph = (m / TW) % TH, pw = m % TW;
    abase[mi] = BF3 ? ((pt * HH + ph) * HW + pw) : ((pt * HH + ph) * HW + pw) * LDA + (VEC ? lh * 4 : lh);     // BF3: the patch ROW (swizzled per read)
  }

  F4Buf<NA> ra;
  F4Buf<NB> rb;
  const int nc = (Cin + KC - 1) / KC;
  const int nstage = 3 * nc;
  auto load_stage = [&](int s) {
    const int dt = s / nc, c0 = (s % nc) * KC;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int q = tid + i * 256;
      float4 v = zero4();
      if (q < HV * A4) {
        const int vox = q / A4, c = c0 + (q % A4) * 4;
        const int pt = vox / (HH * HW), ph = (vox / HW) % HH, pw = vox % HW;
        const int t = t0 + pt + dt - 1, h = h0 + ph - 1, w = w0 + pw - 1;
        if (t >= 0 && t < T && h >= 0 && h < H && w >= 0 && w < W && c < Cin)
          v = ld4(x + ((((size_t)b * T + t) * H + h) * W + w) * Cin + c);
      }
      ra.v[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int q = tid + i * 256;
      float4 v = zero4();
      if (q < 9 * BN * A4) {
        const int row = q / A4, c = c0 + (q % A4) * 4;     // row = tap9 * BN + n
        const int tap = dt * 9 + row / BN, n = n0 + row % BN;
        if (n < Cout && c < Cin) v = ld4(wp + ((size_t)tap * Cout + n) * Cin + c);
      }
      rb.v[i] = v;
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int q = tid + i * 256;
      if (q < HV * A4) {
        if (BF3) {
          uint2 hi, lo;
          split_bf16x4(ra.v[i], hi, lo);
          const int row = q / A4, c4 = q % A4;
          const int off = row * LDH + swz(row, c4 >> 1) * 8 + (c4 & 1) * 4;
          *reinterpret_cast<uint2*>(Ah + off) = hi;
          *reinterpret_cast<uint2*>(Al + off) = lo;
        } else if (VEC) {
          *reinterpret_cast<float4*>(As + (q / A4) * LDA + (q % A4) * 4) = ra.v[i];
        } else {
          float* d = As + (q / A4) * LDA + (q % A4) * 4;
          d[0] = ra.v[i].x; d[1] = ra.v[i].y; d[2] = ra.v[i].z; d[3] = ra.v[i].w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int q = tid + i * 256;
      if (q < 9 * BN * A4) {
        if (BF3) {
          uint2 hi, lo;
          split_bf16x4(rb.v[i], hi, lo);
          const int row = q / A4, c4 = q % A4;
          const int off = row * LDH + swz(row, c4 >> 1) * 8 + (c4 & 1) * 4;
          *reinterpret_cast<uint2*>(Bh + off) = hi;
          *reinterpret_cast<uint2*>(Bl + off) = lo;
        } else if (VEC) {
          *reinterpret_cast<float4*>(Bs + (q / A4) * LDB + (q % A4) * 4) = rb.v[i];
        } else {
          float* d = Bs + (q / A4) * LDB + (q % A4) * 4;
          d[0] = rb.v[i].x; d[1] = rb.v[i].y; d[2] = rb.v[i].z; d[3] = rb.v[i].w;
        }
      }
    }
  };

  load_stage(0);
  store_stage();
  __syncthreads();
  for (int s = 0; s < nstage; ++s) {
    if (s + 1 < nstage) load_stage(s + 1);
    if constexpr (BF3) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int arow_off = (tap / 3) * HW + (tap % 3);
#pragma unroll
        for (int cc = 0; cc < KC; cc += 16) {
          bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const int row = abase[mi] + arow_off;
            const int off = row * LDH + swz(row, cc / 8 + lh) * 8;
            ah[mi] = *reinterpret_cast<const bf16x8*>(Ah + off);
            al[mi] = *reinterpret_cast<const bf16x8*>(Al + off);
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const int row = tap * BN + (wn * NI + ni) * 32 + li;
            const int off = row * LDH + swz(row, cc / 8 + lh) * 8;
            bh[ni] = *reinterpret_cast<const bf16x8*>(Bh + off);
            bl[ni] = *reinterpret_cast<const bf16x8*>(Bl + off);
          }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              floatx16 c_ = acc[mi * NI + ni];
              c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], bh[ni], c_, 0, 0, 0);
              c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bl[ni], c_, 0, 0, 0);
              c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mi], bh[ni], c_, 0, 0, 0);
              acc[mi * NI + ni] = c_;
            }
        }
      }
    } else if constexpr (VEC) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int aoff = ((tap / 3) * HW + (tap % 3)) * LDA;
#pragma unroll
        for (int cc = 0; cc < KC; cc += 8) {
          float4 a4[MI], b4[NI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) a4[mi] = *reinterpret_cast<const float4*>(As + abase[mi] + aoff + cc);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            b4[ni] = *reinterpret_cast<const float4*>(Bs + (tap * BN + (wn * NI + ni) * 32 + li) * LDB + cc + lh * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni) {
                const float av = j == 0 ? a4[mi].x : (j == 1 ? a4[mi].y : (j == 2 ? a4[mi].z : a4[mi].w));
                const float bv = j == 0 ? b4[ni].x : (j == 1 ? b4[ni].y : (j == 2 ? b4[ni].z : b4[ni].w));
                acc[mi * NI + ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi * NI + ni], 0, 0, 0);
              }
        }
      }
    } else {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int aoff = ((tap / 3) * HW + (tap % 3)) * LDA;
#pragma unroll
      for (int cc = 0; cc < KC; cc += 2) {
        float a[MI], bv[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[mi] = As[abase[mi] + aoff + cc];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bv[ni] = Bs[(tap * BN + (wn * NI + ni) * 32 + li) * LDB + cc + lh];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi * NI + ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], bv[ni], acc[mi * NI + ni], 0, 0, 0);
      }
    }
    }
    __syncthreads();
    if (s + 1 < nstage) {
      store_stage();
      __syncthreads();
    }
  }

#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + (wn * NI + ni) * 32 + li;
      if (n >= Cout) continue;
      const float bn = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int t = t0 + m / (TH * TW), h = h0 + (m / TW) % TH, w = w0 + m % TW;
        if (t >= T || h >= H || w >= W) continue;
        y[((((size_t)b * T + t) * H + h) * W + w) * Cout + n] = act_f(acc[mi * NI + ni][r] + bn, act);
      }
    }
  }
}

// uint8 frames [T, H0, W0, 3] -> fp32 [Tp, S, S, 4]: /255, nearest resize (ATen `nearest`: src = min(floor(dst *
// (in/out)), in-1) with a float scale), channel 3 and frames >= T are zero (tail padding of the chunk).
__global__ __launch_bounds__(256) void video_prep_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                         int T, int H0, int W0, int Tp, int S) {
  const float sh = (float)H0 / (float)S, sw = (float)W0 / (float)S;
  const long total = (long)Tp * S * S;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int w = (int)(i % S), h = (int)((i / S) % S), t = (int)(i / ((long)S * S));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < T) {
      const int hs = min((int)floorf(h * sh), H0 - 1), ws = min((int)floorf(w * sw), W0 - 1);
      const uint8_t* p = src + (((size_t)t * H0 + hs) * W0 + ws) * 3;
      v.x = (float)p[0] / 255.f; v.y = (float)p[1] / 255.f; v.z = (float)p[2] / 255.f;
    }
    *reinterpret_cast<float4*>(dst + i * 4) = v;
  }
}

// [N, H, W, C] -> [N, H/p, W/p, C] (MaxPool3d((1,p,p)), floor mode)
__global__ __launch_bounds__(256) void maxpool_hw_kernel(const float* __restrict__ x, float* __restrict__ y, long N,
                                                         int H, int W, int C, int p) {
  const int Ho = H / p, Wo = W / p, C4 = C >> 2;
  const long total = N * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4) * 4;
    long r = i / C4;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const long n = r / Ho;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int dh = 0; dh < p; ++dh)
      for (int dw = 0; dw < p; ++dw) {
        const float4 v = ld4(x + ((n * H + ho * p + dh) * W + wo * p + dw) * C + c);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    *reinterpret_cast<float4*>(y + i * 4) = m;
  }
}

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 256L * 32 ? 256L * 32 : g));
}
inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

static thread_local int g_c3d_mode = 0;      // 0 exact fp32 MFMA, 1 split-bf16 operands (layers with Cin >= 16); per calling thread
static int g_c3d_n96_small = [] { const char* e = getenv("AVDF_C3D_N96_SMALL"); return e ? atoi(e) : 1; }();
extern "C" int avdf_conv3d_mode(int mode) {
  AVDF_REQUIRE(mode == 0 || mode == 1, "conv3d_mode: 0 (fp32) or 1 (bf16x3)");
  g_c3d_mode = mode;
  return AVDF_OK;
}

extern "C" int avdf_conv3d_k3(const float* x, const float* wp, const float* bias, float* y, int B, int T, int H, int W,
                              int Cin, int Cout, int act, void* stream) {
  AVDF_REQUIRE(x && wp && y, "conv3d_k3: null pointer");
  AVDF_REQUIRE(B > 0 && T > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "conv3d_k3: bad dims");
  AVDF_REQUIRE(Cin % 4 == 0, "conv3d_k3: Cin must be a multiple of 4 (%d)", Cin);
  AVDF_REQUIRE(aligned16(x) && aligned16(wp), "conv3d_k3: 16-byte alignment");
  AVDF_REQUIRE((int64_t)B * T * H * W * (int64_t)(Cin > Cout ? Cin : Cout) < ((int64_t)1 << 40), "conv3d_k3: too large");
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH_C3(TT_, TH_, TW_, WM_, WN_, MI_, NI_, KC_)                                                          \
  do {                                                                                                             \
    constexpr int BN_ = WN_ * NI_ * 32;                                                                            \
    const int tT = cdiv(T, TT_), tH = cdiv(H, TH_), tW = cdiv(W, TW_);                                             \
    const int64_t nblk = (int64_t)B * tT * tH * tW;                                                                \
    AVDF_REQUIRE(nblk < ((int64_t)1 << 31), "conv3d_k3: grid too large");                                          \
    if constexpr (KC_ % 16 == 0) {                                                                                 \
      if (g_c3d_mode == 1) {                                                                                       \
        hipLaunchKernelGGL((conv3d_k3_kernel<TT_, TH_, TW_, WM_, WN_, MI_, NI_, KC_, true>),                       \
                           dim3((unsigned)nblk, cdiv(Cout, BN_)), dim3(256), 0, st, x, wp, bias, y, B, T, H, W,    \
                           Cin, Cout, act, tT, tH, tW);                                                            \
        break;                                                                                                     \
      }                                                                                                            \
    }                                                                                                              \
    hipLaunchKernelGGL((conv3d_k3_kernel<TT_, TH_, TW_, WM_, WN_, MI_, NI_, KC_>), dim3((unsigned)nblk, cdiv(Cout, BN_)), \
                       dim3(256), 0, st, x, wp, bias, y, B, T, H, W, Cin, Cout, act, tT, tH, tW);                  \
  } while (0)
  const bool n96 = (Cout % 96 == 0) && (Cout % 64 != 0);
  if (Cin <= 4) {
    if (W >= 32) LAUNCH_C3(1, 8, 32, 4, 1, 2, 2, 4);
    else LAUNCH_C3(4, 4, 4, 2, 2, 1, 1, 4);
  } else if (W >= 32) {
    // 96 output channels: 6 accumulator tiles per wave need 320 registers (one workgroup per CU); in split-bf16 mode
    // a 128-voxel tile (3 accumulator tiles) fits two workgroups per CU
    if (n96 && g_c3d_mode == 1 && g_c3d_n96_small) LAUNCH_C3(1, 8, 16, 4, 1, 1, 3, 16);
    else if (n96) LAUNCH_C3(1, 8, 32, 4, 1, 2, 3, 16);
    else LAUNCH_C3(1, 8, 32, 4, 1, 2, 2, 16);
  } else if (W >= 16) {
    if (n96 && g_c3d_mode == 1 && g_c3d_n96_small) LAUNCH_C3(1, 8, 16, 4, 1, 1, 3, 16);
    else if (n96) LAUNCH_C3(1, 16, 16, 4, 1, 2, 3, 16);
    else LAUNCH_C3(1, 16, 16, 4, 1, 2, 2, 16);
  } else if (W >= 8) {
    LAUNCH_C3(2, 8, 8, 2, 2, 2, 1, 16);
  } else {
    LAUNCH_C3(4, 4, 4, 2, 2, 1, 1, 16);
  }
#undef LAUNCH_C3
  AVDF_CHECK_LAUNCH("conv3d_k3");
  return AVDF_OK;
}

extern "C" int avdf_video_prep(const uint8_t* frames, float* out, int T, int H0, int W0, int Tp, int S, void* stream) {
  AVDF_REQUIRE(frames && out && T >= 0 && H0 > 0 && W0 > 0 && Tp > 0 && S > 0 && T <= Tp, "video_prep: bad args");
  AVDF_REQUIRE(aligned16(out), "video_prep: out must be 16-byte aligned");
  hipLaunchKernelGGL(video_prep_kernel, dim3(grid_for((long)Tp * S * S)), dim3(256), 0, (hipStream_t)stream, frames, out,
                     T, H0, W0, Tp, S);
  AVDF_CHECK_LAUNCH("video_prep");
  return AVDF_OK;
}

extern "C" int avdf_maxpool_hw(const float* x, float* y, int64_t N, int H, int W, int C, int p, void* stream) {
  AVDF_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && p > 0 && H >= p && W >= p, "maxpool_hw: bad args");
  AVDF_REQUIRE(C % 4 == 0 && aligned16(x) && aligned16(y), "maxpool_hw: C %% 4 and 16-byte alignment required");
  hipLaunchKernelGGL(maxpool_hw_kernel, dim3(grid_for(N * (H / p) * (W / p) * C / 4)), dim3(256), 0, (hipStream_t)stream,
                     x, y, (long)N, H, W, C, p);
  AVDF_CHECK_LAUNCH("maxpool_hw");
  return AVDF_OK;
}
