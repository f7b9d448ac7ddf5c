// Wave-specialised, persistent exact-fp32 GEMM core for the FORWARD of the pointwise layers (Conv1d k=1 / Linear: q, k, v,
// proj, fc1, fc2 of every transformer block, libs/modeling/blocks.py:957-966, 1291-1297) on v_mfma_f32_32x32x2_f32 -- an
// alternative to the register-staged kernels of gemm_conv.hip, selected per shape by plan codes >= 8 (avdf_tuned_extra /
// avdf_gemm_override; no row of the shipped plan table selects it: in situ it ties with the table plans).
//
// A workgroup is 4 + NL + NE waves and loops over BM x BN output tiles:
//   * waves 0-3 ("MFMA waves") touch only LDS and the matrix pipe.  Per 32-deep K-step: one s_barrier, then the MFMAs of
//     stage g from one fragment register set while the ds_read_b128s of stage g+1 fill the other, one read in the shadow
//     of each MFMA (a dependent fp32 MFMA holds the pipe for 64 cycles).  At a tile end they park the accumulators in an
//     LDS tile and go on with the next tile.
//   * NL "loader waves" own every global -> LDS transfer, all by LDS-DMA (global_load_lds): the operand stages, ST steps
//     ahead and across tile boundaries; the tile parameters (bias / scale columns, mask / keep rows); the residual operand
//     of the tile being drained, two steps ahead of its use.  Nothing but DMAs enters their vmcnt queue, which retires in
//     issue order, so "at most the previous step's instructions outstanding" before a step's barrier certifies all of it.
//   * NE "epilogue waves" own every LDS -> global transfer: they drain the parked tile of the PREVIOUS tile in per-step
//     chunks -- bias, pre-activation copy, GELU, mask, residual, 16-byte row-contiguous stores -- with plain LDS reads and
//     stores nobody waits for.
// Why the split (measured on MI355X: in-kernel s_memtime stamps and the ablation switches below, profiles/r03_h_*):
//   - a wave issues one 1 KiB LDS-DMA per ~85-100 cycles however many are in flight (tools/micro/dma_rate.hip); an
//     in-order wave cannot hide that behind its own MFMAs (16 KB per K-step cost the MFMA waves 350-400 of every 1024
//     MFMA cycles), and the rate scales with the number of issuing waves up to ~16 cycles per KiB per CU;
//   - vmcnt retires in issue order: a wave that stores its epilogue and then waits for a younger DMA waits until the L2
//     has acknowledged its stores (with loads and stores in one wave the stage wait cost ~1200 cycles per K-step although
//     the stage had been issued two steps earlier);
//   - a wave that has LDS-DMAs in flight must read LDS by inline asm (a compiler-visible ds_read behind a pending DMA gets
//     a vmcnt(0)), and an asm read's result register may be copied or spilled by the compiler BEFORE the wait that makes it
//     valid -- so the only waves that read LDS by asm are the MFMA waves, whose loop is small enough to inspect (and
//     tests/test_pw_gemm_build.py keeps every kernel here at zero scratch); the epilogue waves issue no DMA and use plain
//     reads.
//
// LDS image of a stage: lane-linear per DMA instruction (1 KiB = 8 rows x 128 B), so the bank swizzle sits on the SOURCE
// address: the 16-byte piece p of row r is fetched into slot p ^ ((r >> 1) & 7) and the fragment reads apply the same XOR
// -- a 16-lane group of a ds_read_b128 covers all 64 banks once.  A lane's ds_read_b128 holds FOUR k-steps of its operand
// row (MFMA j of an 8-deep step contracts k = {8q + j, 8q + 4 + j}, the same pairing for A and B).  The MFMAs take the
// WEIGHT fragment as the A operand: the accumulator tile is y^T, registers 4j..4j+3 of a lane are four consecutive output
// columns of one token row, so parking is four ds_write_b128 per 32x32 tile.
#include "pw_gemm.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

namespace {

typedef const __attribute__((address_space(1))) void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 64 lanes x 16 bytes: lane l's 16 bytes land at lds_base + 16*l (the LDS base is wave-uniform)
__device__ __forceinline__ void glds16(const float* g, float* lds_base) {
  __builtin_amdgcn_global_load_lds((gas_ptr)g, (las_ptr)lds_base, 16, 0, 0);
}
// 64 lanes x 4 bytes: lane l's dword lands at lds_base + 4*l
__device__ __forceinline__ void glds4(const float* g, float* lds_base) {
  __builtin_amdgcn_global_load_lds((gas_ptr)g, (las_ptr)lds_base, 4, 0, 0);
}
// s_waitcnt vmcnt(N): simm16 = vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]; expcnt / lgkmcnt left at their maxima
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits on gfx9");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
// run-time n, rounded DOWN to the next supported immediate (waiting for more is always safe)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  if (n >= 48) wait_vmcnt<48>();
  else if (n >= 40) wait_vmcnt<40>();
  else if (n >= 32) wait_vmcnt<32>();
  else if (n >= 28) wait_vmcnt<28>();
  else if (n >= 24) wait_vmcnt<24>();
  else if (n >= 20) wait_vmcnt<20>();
  else if (n >= 16) wait_vmcnt<16>();
  else if (n >= 14) wait_vmcnt<14>();
  else if (n >= 12) wait_vmcnt<12>();
  else if (n >= 10) wait_vmcnt<10>();
  else if (n >= 8) wait_vmcnt<8>();
  else if (n >= 7) wait_vmcnt<7>();
  else if (n >= 6) wait_vmcnt<6>();
  else if (n >= 5) wait_vmcnt<5>();
  else if (n >= 4) wait_vmcnt<4>();
  else if (n >= 3) wait_vmcnt<3>();
  else if (n >= 2) wait_vmcnt<2>();
  else if (n >= 1) wait_vmcnt<1>();
  else wait_vmcnt<0>();
}
__device__ __forceinline__ f32x4 lds_read16(unsigned byte_addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
  return v;
}
template <int OFF>
__device__ __forceinline__ f32x4 lds_read16_off(unsigned byte_addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ float lds_read4(unsigned byte_addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(byte_addr));
  return v;
}
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

constexpr int BK = 32;                      // floats of K per stage: one 128-byte line per tile row
constexpr int ST = 3;                       // LDS stages; the DMA issued in step g fills the buffer of stage g (free since barrier g) with stage g + ST

// blocks b and b + 8 share an XCD (round-robin dispatch): give the blocks of one XCD a contiguous run of tile ids, so
// the column tiles of one row panel (consecutive ids) hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int b, int G) {
  const int q = G >> 3, r = G & 7, xcd = b & 7, j = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

template <class T>
__device__ __forceinline__ T pick3(T const (&a)[3], int p) { return p == 0 ? a[0] : (p == 1 ? a[1] : a[2]); }

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int WM, int WN, int KG, int MI, int NI, int NL, int NE, bool RES>
struct PwGeom {
  static constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  static constexpr int STAGE = (BM + BN) * BK;                  // floats per stage: A rows, then B rows
  static constexpr int LDE = BN + 4;                            // row stride of the parked tile (a 16-byte bank rotation per row)
  static constexpr int TILE = BM * BN;
  static constexpr int WAVES = 4 + NL + NE;                     // MFMA waves, loader waves, epilogue waves
  static constexpr int IT = TILE / (NE * 256);                  // float4 items per epilogue-wave lane per tile
  static constexpr int CPK = (IT + 3) / 4;                      // static bound of the items per K-step chunk
  static constexpr int PB = (cmax(BM, BN) + 63) / 64 * 64;      // floats per tile-parameter vector (whole 64-lane DMA instructions)
  // float offsets of the regions
  static constexpr int EPI = ST * STAGE;                        // parked output tile [BM][LDE]
  static constexpr int RED = EPI + BM * LDE;                    // K-group partials [(KG-1)][wave-major accumulators]
  static constexpr int PAR = RED + (KG - 1) * TILE;             // [2 tiles][bias, rscale, mask, keep][PB]
  static constexpr int RING = PAR + 2 * 4 * PB;                 // residual operand [3 chunks][CPK][epilogue wave][64 lanes x 4]
  static constexpr int LDS_FLOATS = RING + (RES ? 3 * CPK * NE * 256 : 0);
  static constexpr int LDS_BYTES = LDS_FLOATS * 4;
  static constexpr int WPC = LDS_BYTES * 2 <= 160 * 1024 ? 2 : 1;   // co-resident workgroups per CU the kernel is built for
};

// diagnostic stamps (tools/pwgemm_bench.py --clock): per workgroup {s_memtime, s_memrealtime} at entry and exit of wave 0
// plus its cycles in barrier waits / MFMA+read blocks / parking; the in-kernel shader clock is d(memtime) / d(memrealtime)
// x 100 MHz.  NULL in production (one uniform branch per workgroup).
__device__ unsigned long long* g_pw_stamps = nullptr;
// diagnostic ablation switches (tools/pwgemm_bench.py --ablate): 2 no epilogue items, 4 no operand DMAs, 8 no s_setprio

__device__ int g_pw_flags = 0;

template <int WM, int WN, int KG, int MI, int NI, int NL, int NE, bool RES>
__global__ __launch_bounds__((64 * (4 + NL + NE)), ((4 + NL + NE) * PwGeom<WM, WN, KG, MI, NI, NL, NE, RES>::WPC / 4))
void pw_fwd_kernel(PwFwd a, int tilesM, int tilesN, int total) {
  static_assert(WM * WN * KG == 4, "four MFMA waves per workgroup");
  static_assert((NL == 2 || NL == 4) && (NE == 2 || NE == 4), "loader / epilogue wave counts");
  using Gm = PwGeom<WM, WN, KG, MI, NI, NL, NE, RES>;
  constexpr int BM = Gm::BM, BN = Gm::BN, STAGE = Gm::STAGE, LDE = Gm::LDE, IT = Gm::IT, CPK = Gm::CPK, PB = Gm::PB;
  static_assert(Gm::LDS_BYTES <= 160 * 1024, "tile configuration does not fit the LDS");
  __shared__ __attribute__((aligned(1024))) float smem[Gm::LDS_FLOATS];
  float* const epi = smem + Gm::EPI;
  float* const red = smem + Gm::RED;
  float* const par = smem + Gm::PAR;
  unsigned long long* const stamps = g_pw_stamps;
  const int dbg = g_pw_flags;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x;
  const int M = a.M, N = a.N, K = a.K, nk = K / BK;
  const int per = tilesM * tilesN;
  auto coords = [&](int L, int& prob, int& m0, int& n0) {       // tile L -> (problem, m0, n0)
    int t = xcd_remap(L, total);
    prob = t / per;
    t -= prob * per;
    m0 = (t / tilesN) * BM;
    n0 = (t % tilesN) * BN;
  };
  int my_tiles = 0;
  for (int L = blockIdx.x; L < total; L += G) ++my_tiles;
  const int total_steps = my_tiles * nk;
  // Barrier sequence, identical for all eight waves: one before the first fragment read, one per K-step, two per tile end
  // (E0: the previous parked tile is drained and the K-group partials are written; E1: the new tile is parked).

  // Epilogue geometry, shared by the loader waves (which fetch the residual operand in the same order) and the epilogue
  // waves: item `it` of lane l of epilogue wave e = float4 number (it*NE + e)*64 + l of the parked [BM][BN] tile;
  // 64*NE % (BN/4) == 0, so a lane's column group is the same for every item and its rows step by 256*NE/BN.
  constexpr int CG = BN / 4, RPI = NE * 64 / CG;
  const int ipc = (IT + nk - 3) / (nk - 2);                     // items per K-step chunk (host-checked <= CPK): chunk c is drained in step c + 2
  auto n_items = [&](int c) -> int {
    const int left = IT - c * ipc;
    return left < 0 ? 0 : (left < ipc ? left : ipc);
  };
  float* const ring = smem + Gm::RING;                          // [3 chunks][CPK items][NE waves][64 lanes x 4 floats]

  if (wid >= 4 && wid < 4 + NL) {
    // ---------------------------------------------------- loader waves ----------------------------------------------------
    // Every global -> LDS transfer of the workgroup, all by LDS-DMA: with nothing but DMAs in this wave's vmcnt queue
    // (which retires in issue order) "at most the previous step's instructions outstanding" certifies everything older.
    const int lw = wid - 4;
    constexpr int NPL = (BM + BN) / (8 * NL);                   // operand DMA instructions of this wave per stage
    constexpr int NPA = BM / (8 * NL);                          // ... of which the first NPA fetch x rows
    const float* src[NPL];
    int Li = blockIdx.x, ikt = 0, ibuf = 0;
    bool ivalid = Li < total;
    auto setup = [&](int L) {                                   // instruction q = lw + NL*i covers rows 8q .. 8q+7 of the stage
      int prob, m0, n0;
      coords(L, prob, m0, n0);
      const float* __restrict__ x = pick3(a.x, prob);
      const float* __restrict__ w = pick3(a.w, prob);
#pragma unroll
      for (int i = 0; i < NPL; ++i) {
        const int row = (lw + NL * i) * 8 + (lane >> 3) - (i < NPA ? 0 : BM);      // row inside its own operand tile
        const int piece = (lane & 7) ^ ((row >> 1) & 7);
        if (i < NPA) {
          int gm = m0 + row;
          gm = gm < M ? gm : M - 1;
          src[i] = x + (size_t)gm * K + piece * 4;
        } else {
          int gn = n0 + row;
          gn = gn < N ? gn : N - 1;
          src[i] = w + (size_t)gn * K + piece * 4;
        }
      }
    };
    if (ivalid) setup(Li);
    auto issue_stage = [&]() -> int {
      if (!ivalid) return 0;
      float* dst = smem + ibuf * STAGE + lw * 256;
      if (!(dbg & 4))
#pragma unroll
      for (int i = 0; i < NPL; ++i) glds16(src[i] + ikt * BK, dst + i * (NL * 256));
      ibuf = ibuf + 1 == ST ? 0 : ibuf + 1;
      if (++ikt == nk) {
        ikt = 0;
        Li += G;
        ivalid = Li < total;
        if (ivalid) setup(Li);
      }
      return (dbg & 4) ? 0 : NPL;
    };
    // tile parameters of tile L into parameter set `ps`: vector k of {bias, rscale, mask, keep} by loader k % NL
    // (selects, not an if-chain: hipcc 7.2 lowered a four-way branch on the wave index to a flow that handed one wave the
    // pointer of another)
    const float* const rscale = RES ? a.rscale : nullptr;
    const float* const rkeep = RES ? a.rkeep : nullptr;
    const int resT = a.resT;
    auto issue_vec = [&](const float* g, float* dst, int n, int first, int lim, bool per_clip) -> int {
      if (!g) return 0;
      int cnt = 0;
      for (int i = 0; i < n; i += 64, ++cnt) {
        int e = first + i + lane;
        e = e < lim ? e : lim - 1;
        if (per_clip) e /= resT;
        glds4(g + e, dst + i);
      }
      return cnt;
    };
    auto issue_params = [&](int L, int ps) -> int {
      int prob, m0, n0;
      coords(L, prob, m0, n0);
      const float* gcol = (lw & 1) ? rscale : pick3(a.bias, prob);        // per output column
      const float* grow = (lw & 1) ? rkeep : pick3(a.mask, prob);         // per token row
      int cnt = 0;
      if (NL == 2 || lw < 2) cnt += issue_vec(gcol, par + (ps * 4 + (lw & 1)) * PB, BN, n0, N, false);
      if (NL == 2 || lw >= 2) cnt += issue_vec(grow, par + (ps * 4 + 2 + (lw & 1)) * PB, BM, m0, M, (lw & 1) != 0);
      return cnt;
    };
    // residual operand of the tile being drained: item (it, e) is one 1 KiB instruction, dealt round-robin to the loaders
    const float* const res = RES ? a.res : nullptr;
    int pm0 = 0, pn0 = 0;
    auto res_one = [&](int it, int e, float* dst) {
      const int f = e * 64 + lane;
      int gr = pm0 + f / CG + it * RPI, gc = pn0 + (f % CG) * 4;
      gr = gr < M ? gr : M - 1;
      gc = gc < N ? gc : N - 4;
      glds16(res + (size_t)gr * N + gc, dst);
    };
    auto issue_res = [&](int c) -> int {                        // chunk c into ring slot c % 3
      if (!RES) return 0;
      const int n = n_items(c);
      int cnt = 0;
      for (int d = lw; d < n * NE; d += NL, ++cnt) res_one(c * ipc + d / NE, d % NE, ring + (((c % 3) * CPK + d / NE) * NE + d % NE) * 256);
      return cnt;
    };

    const bool mst = stamps != nullptr && lw == 0;              // diagnostic: phases of loader wave 0
    unsigned long long mph[3] = {0, 0, 0}, mt = 0;
    auto lap = [&](int i) {
      if (mst) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        mph[i] += t - mt;
        mt = t;
      }
    };
    static_assert(ST == 3, "the waits below assume that a stage is consumed two steps after the step that issued it");
    int prev = 0;                                               // instructions issued in the previous step
    {
      const int c0 = issue_stage(), c1 = issue_stage(), c2 = issue_stage();
      (void)c0;
      wait_vmcnt_dyn(c1 + c2);                                  // stage 0 has landed
      prev = c2;
    }
    __builtin_amdgcn_s_barrier();
    if (mst) mt = __builtin_amdgcn_s_memtime();
    int tpar = 0;
    bool have_prev = false;
    for (int L = blockIdx.x; L < total; L += G, tpar ^= 1) {
      for (int kt = 0; kt < nk; ++kt) {
        // everything issued before the previous step has landed: stage gs+1 (read by the MFMA waves during this step),
        // the residual chunk the epilogue waves drain in this step, the tile parameters from step 2 on
        wait_vmcnt_dyn(prev);
        lap(0);
        __builtin_amdgcn_s_barrier();
        lap(1);
        prev = issue_stage();                                   // stage gs + ST into the buffer of stage gs
        if (kt == 0) prev += issue_params(L, tpar);
        if (have_prev) prev += issue_res(kt);                   // chunk kt, drained in step kt + 2
        lap(2);
      }
      __builtin_amdgcn_s_barrier();                             // E0
      __builtin_amdgcn_s_barrier();                             // E1
      int prob;
      coords(L, prob, pm0, pn0);
      have_prev = true;
      lap(1);
    }
    // the last tile: its whole residual into the (now free) stage buffers, [item][epilogue wave][64 lanes x 4]
    if (RES && have_prev)
      for (int d = lw; d < IT * NE; d += NL) res_one(d / NE, d % NE, smem + d * 256);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                               // F
    if (mst && lane == 0)
      for (int i = 0; i < 3; ++i) stamps[(G + blockIdx.x) * 8 + i] = mph[i];
    return;
  }

  if (wid >= 4 + NL) {
    // --------------------------------------------------- epilogue waves ---------------------------------------------------
    // LDS -> global only: plain (compiler-visible) LDS reads -- these waves issue no LDS-DMA, so nothing makes the compiler
    // drain vmcnt in front of them -- and 16-byte stores, which nobody waits for.
    const int lw = wid - 4 - NL;
    const int c4 = ((lw * 64 + lane) % CG) * 4, r0 = (lw * 64 + lane) / CG;
    float* __restrict__ yres = RES ? a.yres : nullptr;
    const bool has_scale = RES && a.rscale != nullptr, has_keep = RES && a.rkeep != nullptr;
    const int act = a.act & 255;
    const float* const my_epi = epi + r0 * LDE + c4;
    const float* const my_ring = ring + lw * 256 + lane * 4;

    // the tile being drained ("previous") and its operands
    int pm0 = 0, pn0 = 0;
    const float* ppar = par;
    bool pmask = false;
    float* __restrict__ y = nullptr;
    float* __restrict__ ypre = nullptr;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), sv = make_float4(1.f, 1.f, 1.f, 1.f);

    // one item: parked accumulators -> global.  rres: this lane's residual float4 in LDS
    auto item = [&](int it, const float* rres) {
      const int row = r0 + it * RPI, gr = pm0 + row, gc = pn0 + c4;
      if (gr >= M || gc >= N) return;
      const float4 v = *reinterpret_cast<const float4*>(my_epi + it * (RPI * LDE));
      const float m_ = pmask ? ppar[2 * PB + row] : 1.f;
      const size_t o = (size_t)gr * N + gc;
      float4 u = make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w);
      if (ypre) *reinterpret_cast<float4*>(ypre + o) = u;
      if (act == AVDF_ACT_GELU) { u.x = gelu_f(u.x); u.y = gelu_f(u.y); u.z = gelu_f(u.z); u.w = gelu_f(u.w); }
      else if (act != AVDF_ACT_NONE) { u.x = act_f(u.x, act); u.y = act_f(u.y, act); u.z = act_f(u.z, act); u.w = act_f(u.w, act); }
      u.x *= m_; u.y *= m_; u.z *= m_; u.w *= m_;
      if (y) *reinterpret_cast<float4*>(y + o) = u;
      if (RES) {
        const float k_ = has_keep ? ppar[3 * PB + row] : 1.f;
        const float4 r_ = *reinterpret_cast<const float4*>(rres);
        *reinterpret_cast<float4*>(yres + o) = make_float4(fmaf(sv.x * k_, u.x, r_.x), fmaf(sv.y * k_, u.y, r_.y),
                                                           fmaf(sv.z * k_, u.z, r_.z), fmaf(sv.w * k_, u.w, r_.w));
      }
    };
    auto begin_tile = [&](int L, int ps) {                      // tile L has just been parked; its parameter set has landed
      int prob;
      coords(L, prob, pm0, pn0);
      ppar = par + ps * 4 * PB;
      pmask = pick3(a.mask, prob) != nullptr;
      y = pick3(a.y, prob);
      ypre = pick3(a.ypre, prob);
      bv = pick3(a.bias, prob) ? *reinterpret_cast<const float4*>(ppar + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      sv = has_scale ? *reinterpret_cast<const float4*>(ppar + PB + c4) : make_float4(1.f, 1.f, 1.f, 1.f);
    };
    // s_barrier plus a compiler fence: LDS accesses must not move across it (no instruction is emitted for the fence)
    auto sync = [&]() {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };

    const bool mst = stamps != nullptr && lw == 0;              // diagnostic: phases of epilogue wave 0
    unsigned long long mph[2] = {0, 0}, mt = 0;
    auto lap = [&](int i) {
      if (mst) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        mph[i] += t - mt;
        mt = t;
      }
    };
    sync();
    if (mst) mt = __builtin_amdgcn_s_memtime();
    int tpar = 0;
    bool have_prev = false;
    for (int L = blockIdx.x; L < total; L += G, tpar ^= 1) {
      for (int kt = 0; kt < nk; ++kt) {
        sync();
        lap(0);
        if (have_prev && kt >= 2 && !(dbg & 2)) {
          const int c = kt - 2, n = n_items(c);
          for (int j = 0; j < n; ++j) item(c * ipc + j, my_ring + (((c % 3) * CPK + j) * NE) * 256);
        }
        lap(1);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the drain's LDS reads are done before the tile is overwritten
      sync();                                                   // E0
      sync();                                                   // E1: tile L is parked
      begin_tile(L, tpar);
      have_prev = true;
      lap(0);
    }
    sync();                                                     // F: the last tile's residual has landed in the stage buffers
    if (have_prev && !(dbg & 2))
      for (int it = 0; it < IT; ++it) item(it, smem + (it * NE + lw) * 256 + lane * 4);
    if (mst && lane == 0)
      for (int i = 0; i < 2; ++i) stamps[(2 * G + blockIdx.x) * 8 + i] = mph[i];
    return;
  }

  // ------------------------------------------------------ MFMA waves ------------------------------------------------------
  unsigned long long t0 = 0, rt0 = 0, ph[3] = {0, 0, 0};
  if (stamps) { t0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
  const int li = lane & 31, lh = lane >> 5;
  const int kg = wid / (WM * WN), wp = wid % (WM * WN), wm = wp / WN, wn = wp % WN;
  // fragment byte address inside a stage: row * 128 + ((2q + lh) ^ swz(row)) * 16 for the q-th 8-deep k-step; with
  // row = 32 * tile + li the swizzle (row >> 1) & 7 depends on li only, so one register per q serves every 32-row tile of
  // an operand (the tile index is an instruction offset)
  constexpr int NQ = 4 / KG;                                    // 8-deep k-steps of a stage handled by this K-group
  unsigned xa[NQ], xb[NQ];
#pragma unroll
  for (int qq = 0; qq < NQ; ++qq) {
    const unsigned pc = 2 * (kg + qq * KG) + lh, sw = (li >> 1) & 7;
    const unsigned in_tile = lds_addr(smem) + li * (BK * 4) + ((pc ^ sw) << 4);
    xa[qq] = in_tile + (wm * MI * 32) * (BK * 4);
    xb[qq] = in_tile + (BM + wn * NI * 32) * (BK * 4);
  }
  floatx16 acc[MI * NI];
  f32x4 fa[2][NQ][MI], fb[2][NQ][NI];                           // two fragment register sets
  auto read_frag = [&](auto SC, auto EC, int buf) {             // E-th fragment read of a stage into register set S
    constexpr int S = decltype(SC)::value, E = decltype(EC)::value;
    constexpr int qq = E / (MI + NI), r = E % (MI + NI);
    const unsigned sb = buf * (STAGE * 4);
    constexpr int ra = r < MI ? r : 0, rb = r < MI ? 0 : r - MI;
    if (r < MI) fa[S][qq][ra] = lds_read16_off<ra * 32 * BK * 4>(xa[qq] + sb);
    else fb[S][qq][rb] = lds_read16_off<rb * 32 * BK * 4>(xb[qq] + sb);
  };
  int gs = 0, rbuf = 1;                                         // rbuf: buffer of stage gs + 1
  auto step = [&](auto SC) {
    constexpr int S = decltype(SC)::value;
    constexpr int NM = NQ * 4 * MI * NI, NR = NQ * (MI + NI);
    unsigned long long c0 = 0, c1 = 0;
    if (stamps) c0 = __builtin_amdgcn_s_memtime();
    lds_wait();                                                 // the fragments of THIS step (read during the previous one): their buffer is free from here
    __builtin_amdgcn_s_barrier();                               // the memory waves have seen stage gs+1 land
    if (stamps) c1 = __builtin_amdgcn_s_memtime();
    // (the last step reads the fragments of a stage that does not exist: in-bounds LDS garbage that no MFMA consumes --
    // a second loop body without the reads made the compiler shuffle the accumulators between the two)
    static_for<(NM > NR ? NM : NR)>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      if constexpr (m < NM) {
        constexpr int qq = m / (4 * MI * NI), j = (m / (MI * NI)) % 4, mi = (m / NI) % MI, ni = m % NI;
        acc[mi * NI + ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[S][qq][ni][j], fa[S][qq][mi][j], acc[mi * NI + ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (m < NR) read_frag(std::integral_constant<int, S ^ 1>{}, mc, rbuf);
      __builtin_amdgcn_sched_barrier(0);
    });
    ++gs;
    rbuf = rbuf + 1 == ST ? 0 : rbuf + 1;
    if (stamps) { ph[0] += c1 - c0; ph[1] += __builtin_amdgcn_s_memtime() - c1; }
  };

  if (!(dbg & 8)) __builtin_amdgcn_s_setprio(2);
  __builtin_amdgcn_s_barrier();                                 // stage 0 has landed
  static_for<NQ * (MI + NI)>([&](auto ec) { read_frag(std::integral_constant<int, 0>{}, ec, 0); });
  for (int L = blockIdx.x; L < total; L += G) {
#pragma unroll
    for (int i = 0; i < MI * NI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int kt = 0; kt < nk; kt += 2) {                        // nk is even (host-checked): the register set is static
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
    }
    const unsigned long long p0 = stamps ? __builtin_amdgcn_s_memtime() : 0;
    if (KG > 1 && kg > 0) {                                     // K-groups > 0 hand their accumulators to group 0
#pragma unroll
      for (int i = 0; i < MI * NI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((kg - 1) * (WM * WN) + wp) * MI * NI + i) * 1024 + r * 64 + lane] = acc[i][r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // E0
    if (kg == 0) {
      if (KG > 1) {
#pragma unroll
        for (int g = 1; g < KG; ++g)
#pragma unroll
          for (int i = 0; i < MI * NI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] += red[(((g - 1) * (WM * WN) + wp) * MI * NI + i) * 1024 + r * 64 + lane];
      }
      // park y^T: lane li = token row, registers 4j..4j+3 = columns 8j + 4*lh + {0..3}
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          float* dst = epi + ((wm * MI + mi) * 32 + li) * LDE + (wn * NI + ni) * 32 + 4 * lh;
          const floatx16& c = acc[mi * NI + ni];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(dst + 8 * j) = make_float4(c[4 * j], c[4 * j + 1], c[4 * j + 2], c[4 * j + 3]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                               // E1
    if (stamps) ph[2] += __builtin_amdgcn_s_memtime() - p0;
  }
  __builtin_amdgcn_s_barrier();                                 // F (see the loader waves)
  if (stamps && tid == 0) {
    stamps[blockIdx.x * 8 + 0] = t0;
    stamps[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime();
    stamps[blockIdx.x * 8 + 2] = rt0;
    stamps[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    stamps[blockIdx.x * 8 + 4] = ph[0];
    stamps[blockIdx.x * 8 + 5] = 0;
    stamps[blockIdx.x * 8 + 6] = ph[1];
    stamps[blockIdx.x * 8 + 7] = ph[2];
  }
}

struct PwCfg { int wm, wn, kg, mi, ni, nl, ne; };
// index = plan code - 8; BM = wm*mi*32, BN = wn*ni*32; nl loader and ne epilogue waves beside the four MFMA waves
constexpr PwCfg kPwCfgs[] = {
    {2, 2, 1, 1, 1, 2, 2},   // 0:  64 x 64, 8 waves
    {2, 1, 2, 1, 1, 2, 2},   // 1:  64 x 32, two K-groups
    {1, 2, 2, 1, 1, 2, 2},   // 2:  32 x 64, two K-groups
    {1, 1, 4, 1, 1, 2, 2},   // 3:  32 x 32, four K-groups
    {2, 2, 1, 2, 1, 4, 4},   // 4: 128 x 64, 12 waves
    {2, 2, 1, 1, 2, 4, 4},   // 5:  64 x 128, 12 waves
    {2, 2, 1, 1, 1, 4, 2},   // 6:  64 x 64, 10 waves
    {2, 2, 1, 2, 1, 2, 2},   // 7: 128 x 64, 8 waves
    {2, 2, 1, 2, 1, 4, 2},   // 8: 128 x 64, 10 waves
    {1, 2, 2, 3, 1, 2, 2},   // 9:  96 x 64, two K-groups (waves 96 x 32), 8 waves: [6144, 256] outputs are 256 tiles, one per CU
    {1, 2, 2, 3, 1, 4, 4},   // 10: 96 x 64, two K-groups, 12 waves
};
constexpr int kNumPwCfgs = sizeof(kPwCfgs) / sizeof(kPwCfgs[0]);

static std::atomic<int> g_pw_launches{0};   // launches so far (tests assert that a forced plan really ran here)
static int g_pw_wpc = [] { const char* e = getenv("AVDF_PW_WPC"); return e ? atoi(e) : 0; }();

template <int WM, int WN, int KG, int MI, int NI, int NL, int NE>
int pw_fwd_go(const PwFwd& a, int wpc_plan, hipStream_t st) {
  using Gm = PwGeom<WM, WN, KG, MI, NI, NL, NE, true>;
  using Gn = PwGeom<WM, WN, KG, MI, NI, NL, NE, false>;
  const int tilesM = cdiv(a.M, Gm::BM), tilesN = cdiv(a.N, Gm::BN);
  const int total = tilesM * tilesN * a.nprob;
  // persistent grid: the workgroups that fit the chip at once, each loops over tiles
  int wpc = a.res ? Gm::WPC : Gn::WPC;
  if (wpc_plan >= 1 && wpc_plan < wpc) wpc = wpc_plan;
  if (g_pw_wpc > 0 && wpc > g_pw_wpc) wpc = g_pw_wpc;
  int grid = 256 * wpc;
  if (grid > total) grid = total;
  if (a.res)
    hipLaunchKernelGGL((pw_fwd_kernel<WM, WN, KG, MI, NI, NL, NE, true>), dim3(grid), dim3(64 * Gm::WAVES), 0, st, a, tilesM, tilesN, total);
  else
    hipLaunchKernelGGL((pw_fwd_kernel<WM, WN, KG, MI, NI, NL, NE, false>), dim3(grid), dim3(64 * Gm::WAVES), 0, st, a, tilesM, tilesN, total);
  AVDF_CHECK_LAUNCH("pw_fwd");
  ++g_pw_launches;
  return AVDF_OK;
}

}  // namespace

extern "C" int avdf_pw_launch_count() { return g_pw_launches.load(); }

int pw_fwd_cfg_count() { return kNumPwCfgs; }

// the kernel's preconditions on the contraction length: whole, even number of 32-deep stages, at least ST + 1 of them (the
// tile parameters ride on the stage certification), and an epilogue chunk per step within the static bound
bool pw_fwd_supported(int cfg, int K) {
  if (cfg < 0 || cfg >= kNumPwCfgs || K % (2 * BK) != 0) return false;
  const PwCfg& c = kPwCfgs[cfg];
  const int nk = K / BK, it = 4 * c.wm * c.mi * c.wn * c.ni / c.ne;          // TILE / (NE * 256)
  return nk >= ST + 1 && (it + nk - 3) / (nk - 2) <= (it + 3) / 4;
}

extern "C" int avdf_pw_flags(int flags) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_pw_flags), &flags, sizeof(flags));
  AVDF_REQUIRE(e == hipSuccess, "pw_flags: %s", hipGetErrorString(e));
  return AVDF_OK;
}

// diagnostic: device buffer of 8 x u64 per workgroup, or NULL to switch the stamps off
extern "C" int avdf_pw_stamps(void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_pw_stamps), &p, sizeof(p));
  AVDF_REQUIRE(e == hipSuccess, "pw_stamps: %s", hipGetErrorString(e));
  return AVDF_OK;
}

int pw_fwd_launch(const PwFwd& a, int cfg, int wpc, hipStream_t st) {
  AVDF_REQUIRE(pw_fwd_supported(cfg, a.K), "pw_fwd: configuration %d does not support K=%d", cfg, a.K);
  AVDF_REQUIRE(a.nprob >= 1 && a.nprob <= 3 && a.M > 0 && a.N >= 4 && a.N % 4 == 0, "pw_fwd: bad problem (N %% 4 != 0?)");
  switch (cfg) {
    case 0: return pw_fwd_go<2, 2, 1, 1, 1, 2, 2>(a, wpc, st);
    case 1: return pw_fwd_go<2, 1, 2, 1, 1, 2, 2>(a, wpc, st);
    case 2: return pw_fwd_go<1, 2, 2, 1, 1, 2, 2>(a, wpc, st);
    case 3: return pw_fwd_go<1, 1, 4, 1, 1, 2, 2>(a, wpc, st);
    case 4: return pw_fwd_go<2, 2, 1, 2, 1, 4, 4>(a, wpc, st);
    case 5: return pw_fwd_go<2, 2, 1, 1, 2, 4, 4>(a, wpc, st);
    case 6: return pw_fwd_go<2, 2, 1, 1, 1, 4, 2>(a, wpc, st);
    case 7: return pw_fwd_go<2, 2, 1, 2, 1, 2, 2>(a, wpc, st);
    case 8: return pw_fwd_go<2, 2, 1, 2, 1, 4, 2>(a, wpc, st);
    case 9: return pw_fwd_go<1, 2, 2, 3, 1, 2, 2>(a, wpc, st);
    case 10: return pw_fwd_go<1, 2, 2, 3, 1, 4, 4>(a, wpc, st);
    default: break;
  }
  AVDF_REQUIRE(false, "pw_fwd: unknown tile configuration %d", cfg);
}
