// Phase-timestamp probe of conv_fwd_kernel (developer tool, not part of the library).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on tools/probe/gemm_probe.hip -o tools/probe/gemm_probe
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
__device__ unsigned long long* g_probe = nullptr;
#define AVDF_PROBE(slot)                                                                              \
  do {                                                                                                \
    if (threadIdx.x == 0 && g_probe)                                                                  \
      g_probe[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (slot)] = wall_clock64();          \
  } while (0)
#include "../../audio_visual_deepfake_detection_amd/csrc/gemm_conv.hip"
void avdf_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }

int main(int argc, char** argv) {
  const int B = 8, T = argc > 1 ? atoi(argv[1]) : 768, Cin = argc > 2 ? atoi(argv[2]) : 256, Cout = argc > 3 ? atoi(argv[3]) : 256;
  const int K = 1;
  float *x, *w, *y;
  hipMalloc(&x, (size_t)B * T * Cin * 4); hipMalloc(&w, (size_t)Cout * Cin * K * 4); hipMalloc(&y, (size_t)B * T * Cout * 4);
  hipMemset(x, 0, (size_t)B * T * Cin * 4); hipMemset(w, 0, (size_t)Cout * Cin * K * 4);
  unsigned long long* d; const size_t nslots = 16 * 65536;
  hipMalloc(&d, nslots * 8); hipMemset(d, 0, nslots * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &d, sizeof(d));
  hipStream_t st; hipStreamCreate(&st);
  for (int i = 0; i < 5; ++i) avdf_conv1d_fwd(x, w, nullptr, nullptr, y, nullptr, B, T, Cin, T, Cout, K, 1, 0, st);
  hipStreamSynchronize(st);
  hipMemset(d, 0, nslots * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  avdf_conv1d_fwd(x, w, nullptr, nullptr, y, nullptr, B, T, Cin, T, Cout, K, 1, 0, st);
  hipEventRecord(e1, st);
  hipStreamSynchronize(st);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nslots);
  hipMemcpy(h.data(), d, nslots * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, t1 = 0; int nb = 0;
  for (size_t b = 0; b < 65536; ++b) if (h[b * 16]) { nb++; t0 = std::min(t0, h[b * 16]); t1 = std::max(t1, h[b * 16 + 15]); }
  printf("M=%d Cin=%d Cout=%d: event %.2f us, blocks %d, first-start..last-end %.2f us\n", B * T, Cin, Cout, ms * 1e3, nb, (t1 - t0) * 0.01);
  // average phase durations (10 ns ticks), and start-time distribution
  double ph[16] = {0}; int cnt[16] = {0};
  std::vector<double> starts, ends;
  for (size_t b = 0; b < 65536; ++b) {
    if (!h[b * 16]) continue;
    starts.push_back((h[b * 16] - t0) * 0.01); ends.push_back((h[b * 16 + 15] - t0) * 0.01);
    unsigned long long prev = h[b * 16];
    for (int s = 1; s < 16; ++s) if (h[b * 16 + s]) { ph[s] += (h[b * 16 + s] - prev) * 0.01; cnt[s]++; prev = h[b * 16 + s]; }
  }
  for (int s = 1; s < 16; ++s) if (cnt[s]) printf("  phase ->%2d: avg %.2f us (n=%d)\n", s, ph[s] / cnt[s], cnt[s]);
  std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
  printf("  block start: min %.2f med %.2f max %.2f | block end: min %.2f med %.2f max %.2f\n", starts.front(),
         starts[starts.size() / 2], starts.back(), ends.front(), ends[ends.size() / 2], ends.back());
  return 0;
}
