// Isolated timing of the row-block fused kernels (csrc/block_fused.hip) at the shapes of the exp10 pyramid.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on tools/experimental/block_fused/bf_bench.hip -o tools/experimental/block_fused/bf_bench
//   run:   tools/experimental/block_fused/bf_bench [B=8] [T=768] [train=1]
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "block_fused.hip"

void avdf_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

static float* dalloc(size_t n, float v) {
  float* p; hipMalloc(&p, n * sizeof(float));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = v == 0.f ? 0.f : v * (float)((i * 2654435761u >> 8) & 1023) / 1024.f - v * 0.5f;
  hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return p;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8, T = argc > 2 ? atoi(argv[2]) : 768, train = argc > 3 ? atoi(argv[3]) : 1;
  const int C = 256, H = 4, W = 7;
  const size_t M = (size_t)B * T;
  float* x = dalloc(M * C, 1.f);
  float *ga = dalloc(C, 1.f), *ba = dalloc(C, 1.f), *dw = dalloc(C * 3, 1.f), *gb = dalloc(C, 1.f), *bb = dalloc(C, 1.f);
  float *pw[4], *pb[4], *u[3], *y[3], *sa[3], *sb[3], *o[3];
  for (int i = 0; i < 4; ++i) { pw[i] = dalloc((size_t)C * C, 0.1f); pb[i] = dalloc(C, 0.1f); }
  for (int i = 0; i < 3; ++i) { u[i] = dalloc(M * C, 0.f); y[i] = dalloc(M * C, 0.f); sa[i] = dalloc(2 * M, 0.f); sb[i] = dalloc(2 * M, 0.f); o[i] = dalloc(M * C, 0.f); }
  float* mask = dalloc(M, 0.f);
  { std::vector<float> h(M, 1.f); hipMemcpy(mask, h.data(), M * sizeof(float), hipMemcpyHostToDevice); }
  float *att = dalloc(M * C, 0.f), *probs = dalloc(M * H * W, 0.f), *r = dalloc(M * C, 0.f), *stats = dalloc(2 * M, 0.f), *out = dalloc(M * C, 0.f), *h = dalloc(M * C, 0.f);
  float* scale = dalloc(C, 1.f);
  const float* xs[3] = {x, x, x}; const float* gas[3] = {ga, ga, ga}; const float* bas[3] = {ba, ba, ba};
  const float* dws[3] = {dw, dw, dw}; const float* gbs[3] = {gb, gb, gb}; const float* bbs[3] = {bb, bb, bb};
  const float* pws[3] = {pw[0], pw[1], pw[2]}; const float* pbs[3] = {pb[0], pb[1], pb[2]};
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto f1 = [&]() { return avdf_qkv_fused_fwd(xs, gas, bas, dws, gbs, bbs, pws, pbs, mask, train ? u : nullptr, train ? y : nullptr, train ? sa : nullptr, train ? sb : nullptr, o, nullptr, B, T, T, C, 1, 1e-5f, st); };
  auto f2 = [&]() { return avdf_attn_out_fused_fwd(o[0], o[1], o[2], mask, nullptr, nullptr, pw[3], pb[3], x, scale, nullptr, ga, ba, train ? att : nullptr, train ? probs : nullptr, train ? r : nullptr, train ? stats : nullptr, out, h, B, T, C, H, W, 1e-5f, st); };
  for (int which = 0; which < 2; ++which) {
    for (int i = 0; i < 5; ++i) if ((which ? f2() : f1()) != 0) return 1;
    hipStreamSynchronize(st);
    const int N = 50;
    hipEventRecord(e0, st);
    for (int i = 0; i < N; ++i) which ? f2() : f1();
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (which ? 1.0 : 3.0) * 2.0 * M * C * C;
    printf("%s B=%d T=%d train=%d: %.2f us per launch, %.1f TFLOP/s\n", which ? "attn_out_fused_fwd" : "qkv_fused_fwd     ", B, T, train, ms * 1e3 / N, fl / (ms * 1e-3 / N) / 1e12);
  }
  return 0;
}
