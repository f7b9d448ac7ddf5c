// LDS-DMA fill rate per CU by access pattern (MI355X): how fast can loader waves pull GEMM operand tiles out of L2?
//   build: hipcc -O3 --offload-arch=gfx950 tools/micro/dma_rate.hip -o tools/micro/dma_rate
//   run:   tools/micro/dma_rate
// Every wave issues global_load_lds_dwordx4 (1 KiB per instruction) with `depth` instructions in flight.  Patterns: an
// instruction covers R rows x (1024 / R) bytes, rows `stride` bytes apart; successive instructions of a wave walk along the
// rows (the K direction of a GEMM operand) and then to the next row block.  Source footprint 8 MiB (L2 / Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef const __attribute__((address_space(1))) void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14)); }

template <int DEPTH>
__global__ __launch_bounds__(512) void dma_kernel(const float* src, size_t src_floats, int R, int stride_floats, int row_floats,
                                                   int iters, unsigned long long* cyc) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  float* dst = lds + wid * DEPTH * 256;
  const int per_row = 64 / R;                       // lanes per row
  const int r = lane / per_row, c = (lane % per_row) * 4;
  const int seg = 256 / R;                          // floats of a row one instruction covers
  const int steps_per_row = row_floats / seg;       // instructions to walk a row block
  const size_t gw = (size_t)blockIdx.x * nw + wid;  // global wave id
  const unsigned nblocks = (unsigned)(src_floats / ((size_t)R * stride_floats));   // a power of two
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int slot = 0, k = 0;
  unsigned blk = ((unsigned)gw * 7u) & (nblocks - 1);
  const float* g = src + (size_t)blk * R * stride_floats + (size_t)r * stride_floats + c;
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_global_load_lds((gas_ptr)(g + k * seg), (las_ptr)(dst + slot * 256), 16, 0, 0);
    slot = slot + 1 == DEPTH ? 0 : slot + 1;
    if (++k == steps_per_row) {                   // next row block
      k = 0;
      blk = (blk + 1) & (nblocks - 1);
      g = src + (size_t)blk * R * stride_floats + (size_t)r * stride_floats + c;
    }
    wait_vmcnt<DEPTH - 1>();
  }
  wait_vmcnt<0>();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[gw] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int DEPTH>
void run(const float* src, size_t floats, unsigned long long* cyc, int wgs_per_cu, int waves, int R, int stride_b, int row_b, const char* tag) {
  const int iters = 2048, grid = 256 * wgs_per_cu;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const size_t shm = (size_t)waves * DEPTH * 1024;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(dma_kernel<DEPTH>, dim3(grid), dim3(waves * 64), shm, 0, src, floats, R, stride_b / 4, row_b / 4, iters, cyc);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
  }
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = (double)grid * waves * iters * 1024.0;
  std::vector<unsigned long long> h((size_t)grid * waves);
  CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
  double avg = 0;
  for (auto v : h) avg += (double)v;
  avg /= h.size();
  printf("%-28s wg/CU %d waves %d depth %2d: %7.1f us  %6.2f TB/s chip  %5.1f GB/s per CU  %6.0f cycles per DMA per wave  (%5.1f per CU)\n", tag,
         wgs_per_cu, waves, DEPTH, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256, avg / iters, avg / iters / (waves * wgs_per_cu));
  fflush(stdout);
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? atoi(argv[1]) : 8;
  printf("source footprint %zu MiB\n", mib);
  const size_t floats = (size_t)(mib << 20) / 4;
  float* src; unsigned long long* cyc;
  CK(hipMalloc(&src, floats * 4)); CK(hipMemset(src, 0, floats * 4));
  CK(hipMalloc(&cyc, 8 * 4096 * 8));
  struct P { int R, stride, row; const char* tag; } pats[] = {
      {1, 1024, 1024, "1 x 1024 B contiguous"},
      {8, 1024, 1024, "8 x 128 B, stride 1 KiB"},
      {8, 4096, 4096, "8 x 128 B, stride 4 KiB"},
      {4, 1024, 1024, "4 x 256 B, stride 1 KiB"},
      {2, 1024, 1024, "2 x 512 B, stride 1 KiB"},
      {4, 4096, 4096, "4 x 256 B, stride 4 KiB"},
  };
  for (auto& p : pats) {
    run<4>(src, floats, cyc, 1, 2, p.R, p.stride, p.row, p.tag);
    run<8>(src, floats, cyc, 1, 2, p.R, p.stride, p.row, p.tag);
    run<8>(src, floats, cyc, 1, 4, p.R, p.stride, p.row, p.tag);
    run<16>(src, floats, cyc, 1, 4, p.R, p.stride, p.row, p.tag);
    run<8>(src, floats, cyc, 2, 4, p.R, p.stride, p.row, p.tag);
    run<8>(src, floats, cyc, 1, 8, p.R, p.stride, p.row, p.tag);
  }
  return 0;
}
