// Pipelined exact-fp32 GEMM core for the FORWARD of the pointwise layers (Conv1d k=1 / Linear: q, k, v, proj, fc1, fc2 of
// every transformer block, libs/modeling/blocks.py:957-966, 1291-1297) on v_mfma_f32_32x32x2_f32 -- an alternative to the
// register-staged kernels of gemm_conv.hip, selected per shape by plan codes >= 8 (table rows / avdf_gemm_override).
//
//   * persistent workgroups (4 waves) loop over output tiles; operand tiles go global -> LDS by LDS-DMA
//     (global_load_lds_dwordx4): no staging registers, no ds_write pass; the DMA stream runs ST-1 K-steps ahead across
//     ONE raw s_barrier per step (counted vmcnt) and across tile boundaries;
//   * the LDS image is lane-linear (a DMA wave-instruction writes 1 KiB contiguously), so the bank swizzle sits on the
//     SOURCE address: 16-byte piece p of row r is fetched into slot p ^ ((r >> 1) & 7) of the row's 128 bytes and the
//     fragment reads apply the same XOR -- a 16-lane group of a ds_read_b128 then covers all 64 banks exactly once;
//   * a lane's ds_read_b128 holds FOUR k-steps of its operand row (MFMA j of an 8-deep step contracts
//     k = {8q + j, 8q + 4 + j}, the same pairing for A and B);
//   * the MFMAs take the WEIGHT fragment as the A operand, so the accumulator tile is y^T and a lane's registers 4j..4j+3
//     are four consecutive output columns of one token row: the epilogue is 16-byte row-piece stores straight from the
//     accumulators (no LDS round trip, no barrier), 16-byte bias / residual loads, one row factor per lane.
//
// Status (round 3, profiles/r03_*pw*): correct on every shape (tests/test_gpu_ops.py, tools/pwgemm_bench.py), but NOT
// faster than the table plans of gemm_conv.hip -- 0.83-1.03x in isolation at 8 and 32 clips per step -- so no table row
// selects it.  In-kernel stamps (one workgroup per CU, fc1 at 8 clips) put a K-step at ~2300 cycles for 1024 cycles of
// MFMA: ~350 cycles to ISSUE four 1 KiB DMAs per wave (the CU's address path serialises them: 16 KB per step is ~40 % of
// what the L2 -> LDS path of a CU sustains), ~700 in the counted wait + barrier, ~250 in fragment reads before the first
// MFMA can issue; the row-piece epilogue is 17-22 % of a workgroup's life.  At K = 256 a tile has only eight such steps, so
// neither core gets past ~50 % of the fp32 matrix peak on these shapes (in-kernel clock 2.1-2.3 GHz under this load, not
// 2.4).  Kept as the measured starting point of a software-pipelined variant (DMA issue and next-step fragment reads
// interleaved with the MFMAs of the current step).
#include "pw_gemm.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef const __attribute__((address_space(1))) void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

// 64 lanes x 16 bytes: lane l's 16 bytes land at lds_base + 16*l (the LDS base is wave-uniform)
__device__ __forceinline__ void glds16(const float* g, float* lds_base) {
  __builtin_amdgcn_global_load_lds((gas_ptr)g, (las_ptr)lds_base, 16, 0, 0);
}
// s_waitcnt vmcnt(N) as the BUILTIN (simm16: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]; expcnt / lgkmcnt
// left at their maxima): the compiler's own wait-count pass sees a real S_WAITCNT and credits it, so it does not add a
// vmcnt(0) of its own in front of the fragment reads of the stage this wait retires (an inline-asm wait is opaque to it)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits on gfx9");
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}
// Fragment reads as inline asm: hipcc orders every compiler-visible ds_read behind ALL pending LDS-DMAs with a vmcnt(0)
// of its own (it cannot tell the stage being read from the stages being filled), which would drain the prefetch at every
// K-step.  An asm read is invisible to that pass; the waits are placed by hand: the counted vmcnt + barrier that retire a
// stage before its reads, lds_wait() + a scheduling fence between the reads and the MFMAs that consume them.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 lds_read16(unsigned byte_addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
  return v;
}
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ float f4e(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

constexpr int BK = 32;                      // floats of K per stage: one 128-byte line per tile row

// blocks b and b + 8 share an XCD (round-robin dispatch): give the blocks of one XCD a contiguous run of tile ids, so
// the column tiles of one row panel (consecutive ids) hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int b, int G) {
  const int q = G >> 3, r = G & 7, xcd = b & 7, j = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

template <class T>
__device__ __forceinline__ T pick3(T const (&a)[3], int p) { return p == 0 ? a[0] : (p == 1 ? a[1] : a[2]); }

// ---------------------------------------------------------------------------------------------------------------
// forward:  y[M, N] = act(x[M, K] w[N, K]^T + bias) * mask
// ---------------------------------------------------------------------------------------------------------------
// Persistent workgroups of 4 waves: each loops over BM x BN output tiles; the LDS-DMA stream runs ST-1 K-steps ahead of
// the MFMAs and crosses tile boundaries (the first stages of tile i+1 are in flight during the epilogue of tile i), so
// the load latency is paid once per workgroup, not once per tile.  The epilogue stores straight from the accumulators
// (a register = 2 rows x 32 consecutive columns = two whole 128-byte lines), all of its global LOADS issued before its
// first store: vmcnt retires in issue order, a load behind a store would wait for the store.  The waits of the next
// tile's first ST-1 steps allow for the epilogue's stores (they are younger than the prefetched DMAs), so the stores drain
// under the next tile's MFMAs.  Co-resident workgroups start staggered (`stagger`) so that they are not all in their
// epilogues at once.
template <int WM, int WN, int KG, int MI, int NI, int ST>
struct PwGeom {
  static constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  static constexpr int STAGE = (BM + BN) * BK;                  // floats per stage
  static constexpr int NPS = (BM + BN) / 32;                    // DMA wave-instructions per wave per stage
  static constexpr int TILE = BM * BN;                          // floats of an output tile
  static constexpr int LDS_FLOATS = ST * STAGE + (KG - 1) * TILE;   // stages | (KG-1) reduction tiles
  static constexpr int LDS_BYTES = LDS_FLOATS * 4;
};

// s_waitcnt vmcnt(n) for a run-time n: rounded DOWN to the next supported immediate (waiting for more is always safe)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  if (n >= 56) wait_vmcnt<56>();
  else if (n >= 48) wait_vmcnt<48>();
  else if (n >= 40) wait_vmcnt<40>();
  else if (n >= 36) wait_vmcnt<36>();
  else if (n >= 32) wait_vmcnt<32>();
  else if (n >= 24) wait_vmcnt<24>();
  else if (n >= 20) wait_vmcnt<20>();
  else if (n >= 16) wait_vmcnt<16>();
  else if (n >= 12) wait_vmcnt<12>();
  else if (n >= 8) wait_vmcnt<8>();
  else if (n >= 6) wait_vmcnt<6>();
  else if (n >= 4) wait_vmcnt<4>();
  else if (n >= 3) wait_vmcnt<3>();
  else if (n >= 2) wait_vmcnt<2>();
  else if (n >= 1) wait_vmcnt<1>();
  else wait_vmcnt<0>();
}

// diagnostic stamps (tools/pwgemm_bench.py --clock): per workgroup {s_memtime, s_memrealtime} at entry and exit; the
// in-kernel shader clock is d(memtime) / d(memrealtime) x 100 MHz.  NULL in production (one uniform branch per workgroup).
__device__ unsigned long long* g_pw_stamps = nullptr;

template <int WM, int WN, int KG, int MI, int NI, int ST, bool PIPE, bool RES>
__global__ __launch_bounds__(PIPE ? 384 : 256, PIPE ? 3 : 1) void pw_fwd_kernel(PwFwd a, int tilesM, int tilesN, int total, int stagger) {
  unsigned long long* const stamps = g_pw_stamps;
  unsigned long long t0 = 0, r0 = 0;
  if (stamps) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  static_assert(WM * WN * KG == 4, "four waves per workgroup");
  static_assert(ST == 2 || ST == 3 || ST == 4, "2-4 stages");
  using Gm = PwGeom<WM, WN, KG, MI, NI, ST>;
  constexpr int BM = Gm::BM, BN = Gm::BN, STAGE = Gm::STAGE, NPS = Gm::NPS;
  __shared__ __attribute__((aligned(1024))) float smem[Gm::LDS_FLOATS];
  float* const red = smem + ST * STAGE;                         // [(KG-1)][wave-major accumulators]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int kg = wid / (WM * WN), wp = wid % (WM * WN), wm = wp / WN, wn = wp % WN;
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
  if (stagger > 0) {
    const int rank = blockIdx.x >> 8;                           // blocks b, b + 256, ... tend to share a CU
    for (int i = 0; i < rank * stagger; ++i) __builtin_amdgcn_s_sleep(8);
  }

  if (PIPE && wid >= 4) {
    // ------------------------------------------------ loader waves (PIPE) ------------------------------------------------
    // Waves 4 and 5 issue every LDS-DMA of the workgroup (half of a stage's 1 KiB instructions each) and do the counted
    // vmcnt waits; the four MFMA waves never touch the vector-memory queue inside the K loop: a DMA takes ~100 cycles to
    // ISSUE, which an in-order wave cannot hide behind its own MFMAs, and their epilogue stores are not ordered against DMAs.
    // Barrier sequence (all six waves): one before the first fragment read, one per K-step, two per tile end if KG > 1.
    const int lw = wid - 4;
    constexpr int LA = BM / 16, LB = BN / 16;                   // instructions per loader wave per stage
    const float* sa[LA];
    const float* sb[LB];
    int Li = blockIdx.x, ikt = 0, ibuf = 0;
    bool ivalid = Li < total;
    auto setup = [&](int L) {
      int prob, m0, n0;
      coords(L, prob, m0, n0);
      const float* __restrict__ x = pick3(a.x, prob);
      const float* __restrict__ w = pick3(a.w, prob);
#pragma unroll
      for (int i = 0; i < LA; ++i) {
        const int row = (lw + 2 * i) * 8 + (lane >> 3), piece = (lane & 7) ^ ((row >> 1) & 7);
        int gm = m0 + row;
        gm = gm < M ? gm : M - 1;
        sa[i] = x + (size_t)gm * K + piece * 4;
      }
#pragma unroll
      for (int i = 0; i < LB; ++i) {
        const int row = (lw + 2 * i) * 8 + (lane >> 3), piece = (lane & 7) ^ ((row >> 1) & 7);
        int gn = n0 + row;
        gn = gn < N ? gn : N - 1;
        sb[i] = w + (size_t)gn * K + piece * 4;
      }
    };
    if (ivalid) setup(Li);
    int issued = 0;
    auto issue_stage = [&]() {
      if (!ivalid) return;
      float* sA = smem + ibuf * STAGE;
      float* sB = sA + BM * BK;
#pragma unroll
      for (int i = 0; i < LA; ++i) glds16(sa[i] + ikt * BK, sA + (lw + 2 * i) * 256);
#pragma unroll
      for (int i = 0; i < LB; ++i) glds16(sb[i] + ikt * BK, sB + (lw + 2 * i) * 256);
      ++issued;
      ibuf = ibuf + 1 == ST ? 0 : ibuf + 1;
      if (++ikt == nk) {
        ikt = 0;
        Li += G;
        ivalid = Li < total;
        if (ivalid) setup(Li);
      }
    };
    constexpr int NPL = LA + LB;
    int my_tiles = 0;
    for (int L = blockIdx.x; L < total; L += G) ++my_tiles;
    const int total_steps = my_tiles * nk;
#pragma unroll
    for (int s = 0; s < ST - 1; ++s) issue_stage();
    wait_vmcnt_dyn((issued - 1) * NPL);                         // stage 0
    __builtin_amdgcn_s_barrier();
    int gs = 0;
    for (int t = 0; t < my_tiles; ++t) {
      for (int kt = 0; kt < nk; ++kt, ++gs) {
        if (gs + 1 < total_steps) {                             // stage gs+1 must have landed before this step's barrier
          const int n1 = issued - gs - 2;
          if (n1 >= 2) wait_vmcnt<2 * NPL>(); else if (n1 == 1) wait_vmcnt<NPL>(); else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        issue_stage();                                          // into the buffer of stage gs-1 (read two steps ago)
      }
      if (KG > 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
    }
    return;
  }

  // DMA sources of the tile being ISSUED (runs up to ST-1 stages ahead of the tile being computed)
  const float* srcA[BM / 32];
  const float* srcB[BN / 32];
  auto setup = [&](int L) {
    int prob, m0, n0;
    coords(L, prob, m0, n0);
    const float* __restrict__ x = pick3(a.x, prob);
    const float* __restrict__ w = pick3(a.w, prob);
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) {                         // instruction q covers rows 8q .. 8q+7 (8 lanes per 128-B row)
      const int row = (wid + 4 * i) * 8 + (lane >> 3), piece = (lane & 7) ^ ((row >> 1) & 7);
      int gm = m0 + row;
      gm = gm < M ? gm : M - 1;
      srcA[i] = x + (size_t)gm * K + piece * 4;
    }
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) {
      const int row = (wid + 4 * i) * 8 + (lane >> 3), piece = (lane & 7) ^ ((row >> 1) & 7);
      int gn = n0 + row;
      gn = gn < N ? gn : N - 1;
      srcB[i] = w + (size_t)gn * K + piece * 4;
    }
  };
  int Li = blockIdx.x, ikt = 0, ibuf = 0;
  bool ivalid = Li < total;
  if (ivalid && !PIPE) setup(Li);
  constexpr int NA = BM / 32, NB = BN / 32;                     // DMA instructions of this wave per stage (A rows, B rows)
  auto issue_one = [&](int i) {                                 // i-th DMA instruction of the stage under the issue cursor
    float* sA = smem + ibuf * STAGE;
    if (i < NA) glds16(srcA[i < NA ? i : 0] + ikt * BK, sA + (wid + 4 * i) * 256);
    else glds16(srcB[i < NA ? 0 : i - NA] + ikt * BK, sA + BM * BK + (wid + 4 * (i - NA)) * 256);
  };
  auto issue_advance = [&]() {                                  // cursor -> next stage (next tile's pointers at a tile end)
    ibuf = ibuf + 1 == ST ? 0 : ibuf + 1;
    if (++ikt == nk) {
      ikt = 0;
      Li += G;
      ivalid = Li < total;
      if (ivalid) setup(Li);
    }
  };
  auto issue_next = [&]() -> int {
    if (!ivalid) return 0;
#pragma unroll
    for (int i = 0; i < NA + NB; ++i) issue_one(i);
    issue_advance();
    return 1;
  };

  // fragment byte offsets inside a stage: row * 128 + ((2q + lh) ^ swz(row)) * 16, q = 0..3
  unsigned offA[MI], swA[MI], offB[NI], swB[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int row = (wm * MI + mi) * 32 + li;
    offA[mi] = row * (BK * 4);
    swA[mi] = (row >> 1) & 7;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int row = (wn * NI + ni) * 32 + li;
    offB[ni] = (BM + row) * (BK * 4);
    swB[ni] = (row >> 1) & 7;
  }
  const unsigned sbase = lds_addr(smem);
  floatx16 acc[MI * NI];
  auto compute = [&](int buf) {
    constexpr int NQ = 4 / KG;
    const unsigned sb = sbase + buf * (STAGE * 4);
    f32x4 a4[NQ][MI], b4[NQ][NI];
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq) {
      const unsigned pc = 2 * (kg + qq * KG) + lh;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a4[qq][mi] = lds_read16(sb + offA[mi] + ((pc ^ swA[mi]) << 4));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b4[qq][ni] = lds_read16(sb + offB[ni] + ((pc ^ swB[ni]) << 4));
    }
    lds_wait();
#pragma unroll
    for (int qq = 0; qq < NQ; ++qq)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi * NI + ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b4[qq][ni][j], a4[qq][mi][j], acc[mi * NI + ni], 0, 0, 0);
  };

  // RES: the fused residual epilogue (fc2) as a compile-time switch -- its operands cost 32 VGPRs, which would take a
  // co-resident workgroup away from every other layer
  const float* __restrict__ res = RES ? a.res : nullptr;
  const float* __restrict__ rscale = a.rscale;
  const float* __restrict__ rkeep = a.rkeep;
  float* __restrict__ yres = a.yres;
  const int act = a.act & 255, resT = a.resT;

  int pend = 0;                                                 // stores of the previous tile's epilogue (this wave)
  unsigned long long ph[4] = {0, 0, 0, 0};                      // diagnostic: cycles in wait+barrier / DMA issue / reads+MFMA / epilogue
  // ---- software-pipelined K loop (PIPE): two fragment register sets; during the MFMAs of stage g (set g & 1) the wave
  // reads the fragments of stage g+1 into the other set and issues the DMAs of stage g+ST-1, one such instruction in the
  // shadow of every MFMA (a dependent fp32 MFMA occupies the pipe for 64 cycles; whatever issues meanwhile is free).  The
  // wait + barrier at the top of a step therefore certifies stage g+1, not g.  Needs ST >= 3 (the DMA target of step g is
  // the buffer of stage g-1, the reads go to the buffer of stage g+1) and an even number of K-steps per tile (the register
  // set of a step must be a compile-time constant).
  constexpr int NQ = 4 / KG;
  f32x4 fa[2][NQ][MI], fb[2][NQ][NI];
  auto read_frag = [&](auto SC, int e, int buf) {               // e-th fragment read of a stage into register set S
    constexpr int S = decltype(SC)::value;
    const unsigned sb = sbase + buf * (STAGE * 4);
    const int qq = e / (MI + NI), r = e % (MI + NI);
    const unsigned pc = 2 * (kg + qq * KG) + lh;
    if (r < MI) fa[S][qq][r < MI ? r : 0] = lds_read16(sb + offA[r < MI ? r : 0] + ((pc ^ swA[r < MI ? r : 0]) << 4));
    else fb[S][qq][r < MI ? 0 : r - MI] = lds_read16(sb + offB[r < MI ? 0 : r - MI] + ((pc ^ swB[r < MI ? 0 : r - MI]) << 4));
  };
  int gs = 0;                                                   // global step index of this workgroup
  int my_tiles = 0;
  for (int L = blockIdx.x; L < total; L += G) ++my_tiles;
  const int total_steps = my_tiles * nk;
  auto pipe_step = [&](auto SC, int kt) {
    constexpr int S = decltype(SC)::value;
    constexpr int NM = NQ * 4 * MI * NI, NR = NQ * (MI + NI);
    unsigned long long c0 = 0, c1 = 0;
    if (stamps) c0 = __builtin_amdgcn_s_memtime();
    const bool more = gs + 1 < total_steps;                     // stage gs+1 exists: its fragments are read during this step
    __builtin_amdgcn_s_barrier();                               // the loader waves have seen stage gs+1 land
    lds_wait();                                                 // the fragments of THIS step (read during the previous one)
    if (stamps) c1 = __builtin_amdgcn_s_memtime();
    const int nbuf = (gs + 1) % ST;
    auto body = [&](auto MC) {
      constexpr bool MORE = decltype(MC)::value;
#pragma unroll
      for (int m = 0; m < (NM > NR ? NM : NR); ++m) {
        if (m < NM) {
          const int qq = m / (4 * MI * NI), j = (m / (MI * NI)) % 4, mi = (m / NI) % MI, ni = m % NI;
          acc[mi * NI + ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[S][qq][ni][j], fa[S][qq][mi][j], acc[mi * NI + ni], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (m < NR && MORE) read_frag(std::integral_constant<int, S ^ 1>{}, m, nbuf);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (more) body(std::true_type{}); else body(std::false_type{});
    ++gs;
    if (stamps) { ph[0] += c1 - c0; ph[2] += __builtin_amdgcn_s_memtime() - c1; }
  };

  int ahead = 0;                                                // stages issued and not yet consumed
  if (!PIPE) {
#pragma unroll
    for (int s = 0; s < ST - 1; ++s) ahead += issue_next();
  }
  int rbuf = 0;
  if (PIPE) {                                                   // fragments of the very first stage
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int e = 0; e < NQ * (MI + NI); ++e) read_frag(std::integral_constant<int, 0>{}, e, 0);
  }
  for (int L = blockIdx.x; L < total; L += G) {
#pragma unroll
    for (int i = 0; i < MI * NI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if (PIPE) {
      for (int kt = 0; kt < nk; kt += 2) {
        pipe_step(std::integral_constant<int, 0>{}, kt);
        pipe_step(std::integral_constant<int, 1>{}, kt + 1);
      }
    } else {
    for (int kt = 0; kt < nk; ++kt) {
      // the stage about to be read has landed once only the (ahead - 1) younger stages' DMAs -- and, for the stages that
      // were prefetched before the previous tile's epilogue, that epilogue's stores -- are outstanding
      unsigned long long c0 = 0, c1 = 0, c2 = 0;
      if (stamps) c0 = __builtin_amdgcn_s_memtime();
      wait_vmcnt_dyn((ahead - 1) * NPS + (kt < ST - 1 ? pend : 0));
      __builtin_amdgcn_s_barrier();                             // every wave's part has landed; the stage read last step is free
      if (stamps) c1 = __builtin_amdgcn_s_memtime();
      ahead += issue_next();
      if (stamps) c2 = __builtin_amdgcn_s_memtime();
      compute(rbuf);
      if (stamps) { ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += __builtin_amdgcn_s_memtime() - c2; }
      rbuf = rbuf + 1 == ST ? 0 : rbuf + 1;
      --ahead;
    }
    }
    pend = 0;
    unsigned long long e0 = stamps ? __builtin_amdgcn_s_memtime() : 0;
    if (KG > 1) {                                               // fold the K-groups into group 0 through LDS
      __builtin_amdgcn_s_barrier();                             // (the previous tile's reduction reads are long done)
      if (kg > 0) {
#pragma unroll
        for (int i = 0; i < MI * NI; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[(((kg - 1) * (WM * WN) + wp) * MI * NI + i) * 1024 + r * 64 + lane] = acc[i][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (kg == 0) {
        const unsigned rb = lds_addr(red) + ((wp * MI * NI) * 1024 + lane) * 4;
#pragma unroll
        for (int g = 1; g < KG; ++g)
#pragma unroll
          for (int i = 0; i < MI * NI; ++i) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
              asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[r]) : "v"(rb + ((g - 1) * (WM * WN) * MI * NI + i) * 4096), "n"(r * 256));
            lds_wait();
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] += v[r];
          }
      }
    }
    if (kg == 0) {
      // The MFMAs ran with the WEIGHT fragment as the A operand, so the accumulator tile is y^T: lane li owns token row
      // m = li of the wave tile and registers 4j .. 4j+3 hold output columns n = 8j + 4*lh + {0..3} -- four consecutive
      // columns: every store is a 16-byte row piece (4 per tile and output instead of 16 dword stores), bias / residual
      // are 16-byte loads, and the row factors (mask, drop-path keep) are ONE scalar per lane.
      int prob, m0, n0;
      coords(L, prob, m0, n0);
      const float* __restrict__ bias = pick3(a.bias, prob);
      const float* __restrict__ mask = pick3(a.mask, prob);
      float* __restrict__ y = pick3(a.y, prob);
      float* __restrict__ ypre = pick3(a.ypre, prob);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = m0 + (wm * MI + mi) * 32 + li;
        const bool rok = row < M;
        // ---- loads first: row factors, column factors, residual
        const float mk = (mask && rok) ? mask[row] : 1.f;
        const float kp = (res && rkeep && rok) ? rkeep[row / resT] : 1.f;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int cb = n0 + (wn * NI + ni) * 32 + 4 * lh;
          float4 bv[4], rs[RES ? 4 : 1], rv[RES ? 4 : 1];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int col = cb + 8 * j;
            const bool ok = rok && col < N;
            bv[j] = (bias && ok) ? *reinterpret_cast<const float4*>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (RES) {
              rs[RES ? j : 0] = (res && ok) ? *reinterpret_cast<const float4*>(rscale + col) : make_float4(0.f, 0.f, 0.f, 0.f);
              rv[RES ? j : 0] = (res && ok) ? *reinterpret_cast<const float4*>(res + (size_t)row * N + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
          // ---- then only stores
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int col = cb + 8 * j;
            if (!rok || col >= N) continue;
            const size_t o = (size_t)row * N + col;
            const floatx16& c = acc[mi * NI + ni];
            float4 v = make_float4(c[4 * j] + bv[j].x, c[4 * j + 1] + bv[j].y, c[4 * j + 2] + bv[j].z, c[4 * j + 3] + bv[j].w);
            if (ypre) *reinterpret_cast<float4*>(ypre + o) = v;
            if (act == AVDF_ACT_GELU) { v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); }
            else if (act != AVDF_ACT_NONE) { v.x = act_f(v.x, act); v.y = act_f(v.y, act); v.z = act_f(v.z, act); v.w = act_f(v.w, act); }
            v.x *= mk; v.y *= mk; v.z *= mk; v.w *= mk;
            if (y) *reinterpret_cast<float4*>(y + o) = v;
            if (RES && res) {
              const float4 r_ = rv[RES ? j : 0], s_ = rs[RES ? j : 0];
              const float4 o4 = make_float4(fmaf(s_.x * kp, v.x, r_.x), fmaf(s_.y * kp, v.y, r_.y), fmaf(s_.z * kp, v.z, r_.z),
                                            fmaf(s_.w * kp, v.w, r_.w));
              *reinterpret_cast<float4*>(yres + o) = o4;
            }
          }
          pend += 4 * ((ypre ? 1 : 0) + (y ? 1 : 0) + (res ? 1 : 0));
        }
      }
    }
    if (stamps) ph[3] += __builtin_amdgcn_s_memtime() - e0;
  }
  if (stamps && threadIdx.x == 0) {
    stamps[blockIdx.x * 8 + 0] = t0;
    stamps[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_memtime();
    stamps[blockIdx.x * 8 + 2] = r0;
    stamps[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    stamps[blockIdx.x * 8 + 4] = ph[0];
    stamps[blockIdx.x * 8 + 5] = ph[1];
    stamps[blockIdx.x * 8 + 6] = ph[2];
    stamps[blockIdx.x * 8 + 7] = ph[3];
  }
}

struct PwCfg { int wm, wn, kg, mi, ni; };
// index = plan code; BM = wm*mi*32, BN = wn*ni*32
constexpr PwCfg kPwCfgs[] = {
    {2, 2, 1, 2, 2},   // 0: 128 x 128
    {2, 2, 1, 2, 1},   // 1: 128 x 64
    {2, 2, 1, 1, 2},   // 2:  64 x 128
    {2, 2, 1, 1, 1},   // 3:  64 x 64
    {2, 1, 2, 1, 1},   // 4:  64 x 32, two K-groups
    {1, 2, 2, 1, 1},   // 5:  32 x 64, two K-groups
    {2, 1, 2, 1, 2},   // 6:  64 x 64, two K-groups (waves 32 x 64)
    {1, 1, 4, 1, 1},   // 7:  32 x 32, four K-groups
    {1, 1, 4, 1, 2},   // 8:  32 x 64, four K-groups
    {2, 1, 2, 2, 2},   // 9: 128 x 64, two K-groups (waves 64 x 64)
};
constexpr int kNumPwCfgs = sizeof(kPwCfgs) / sizeof(kPwCfgs[0]);

static int g_pw_stagger = [] { const char* e = getenv("AVDF_PW_STAGGER"); return e ? atoi(e) : 0; }();
static int g_pw_wpc = [] { const char* e = getenv("AVDF_PW_WPC"); return e ? atoi(e) : 0; }();

static int g_pw_pipe = [] { const char* e = getenv("AVDF_PW_PIPE"); return e ? atoi(e) : 1; }();

template <int WM, int WN, int KG, int MI, int NI, int ST>
int pw_fwd_go2(const PwFwd& a, hipStream_t st) {
  using Gm = PwGeom<WM, WN, KG, MI, NI, ST>;
  const int tilesM = cdiv(a.M, Gm::BM), tilesN = cdiv(a.N, Gm::BN);
  const int total = tilesM * tilesN * a.nprob;
  // persistent grid: as many workgroups as fit the chip at once (LDS-limited), every one loops over tiles
  int wpc = (160 * 1024) / Gm::LDS_BYTES;
  if (wpc > 5) wpc = 5;
  if (g_pw_wpc > 0 && wpc > g_pw_wpc) wpc = g_pw_wpc;
  if (wpc < 1) wpc = 1;
  int grid = 256 * wpc;
  if (grid > total) grid = total;
  if constexpr (ST >= 3) {
    if (g_pw_pipe && (a.K / BK) % 2 == 0) {
      if (a.res)
        hipLaunchKernelGGL((pw_fwd_kernel<WM, WN, KG, MI, NI, ST, true, true>), dim3(grid), dim3(384), 0, st, a, tilesM, tilesN, total, g_pw_stagger);
      else
        hipLaunchKernelGGL((pw_fwd_kernel<WM, WN, KG, MI, NI, ST, true, false>), dim3(grid), dim3(384), 0, st, a, tilesM, tilesN, total, g_pw_stagger);
      AVDF_CHECK_LAUNCH("pw_fwd");
      return AVDF_OK;
    }
  }
  hipLaunchKernelGGL((pw_fwd_kernel<WM, WN, KG, MI, NI, ST, false, true>), dim3(grid), dim3(256), 0, st, a, tilesM, tilesN, total, g_pw_stagger);
  AVDF_CHECK_LAUNCH("pw_fwd");
  return AVDF_OK;
}
template <int WM, int WN, int KG, int MI, int NI>
int pw_fwd_go(const PwFwd& a, int stages, hipStream_t st) {
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  if constexpr (4 * (BM + BN) * BK * 4 + (KG - 1) * BM * BN * 4 <= 80 * 1024) {
    if (stages >= 4) return pw_fwd_go2<WM, WN, KG, MI, NI, 4>(a, st);
  }
  if constexpr (3 * (BM + BN) * BK * 4 + (KG - 1) * BM * BN * 4 <= 80 * 1024) {
    if (stages >= 3) return pw_fwd_go2<WM, WN, KG, MI, NI, 3>(a, st);
  }
  return pw_fwd_go2<WM, WN, KG, MI, NI, 2>(a, st);
}

}  // namespace

int pw_fwd_cfg_count() { return kNumPwCfgs; }

// diagnostic: device buffer of 8 x u64 per workgroup, or NULL to switch the stamps off
extern "C" int avdf_pw_stamps(void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_pw_stamps), &p, sizeof(p));
  AVDF_REQUIRE(e == hipSuccess, "pw_stamps: %s", hipGetErrorString(e));
  return AVDF_OK;
}

int pw_fwd_launch(const PwFwd& a, int cfg, int stages, hipStream_t st) {
  AVDF_REQUIRE(a.K % BK == 0 && a.K >= BK, "pw_fwd: K must be a multiple of %d (K=%d)", BK, a.K);
  AVDF_REQUIRE(a.nprob >= 1 && a.nprob <= 3 && a.M > 0 && a.N > 0 && a.N % 4 == 0, "pw_fwd: bad problem (N %% 4 != 0?)");
  switch (cfg) {
    case 0: return pw_fwd_go<2, 2, 1, 2, 2>(a, stages, st);
    case 1: return pw_fwd_go<2, 2, 1, 2, 1>(a, stages, st);
    case 2: return pw_fwd_go<2, 2, 1, 1, 2>(a, stages, st);
    case 3: return pw_fwd_go<2, 2, 1, 1, 1>(a, stages, st);
    case 4: return pw_fwd_go<2, 1, 2, 1, 1>(a, stages, st);
    case 5: return pw_fwd_go<1, 2, 2, 1, 1>(a, stages, st);
    case 6: return pw_fwd_go<2, 1, 2, 1, 2>(a, stages, st);
    case 7: return pw_fwd_go<1, 1, 4, 1, 1>(a, stages, st);
    case 8: return pw_fwd_go<1, 1, 4, 1, 2>(a, stages, st);
    case 9: return pw_fwd_go<2, 1, 2, 2, 2>(a, stages, st);
    default: break;
  }
  AVDF_REQUIRE(false, "pw_fwd: unknown tile configuration %d", cfg);
}
