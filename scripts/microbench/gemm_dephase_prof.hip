// Stand-alone profiler for the de-phased 256 x 256 GEMM kernel (gemm_dephase.hip, next to this file), with per-segment cycle stamps.
// Never part of the library: it includes the kernel's translation unit with SA_DP_STAMPS defined.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-inline-asm -DSA_DP_STAMPS gemm_dephase_prof.hip -o bin/gemm_dephase_prof
//   run:   ./gemm_dephase_prof M N K epi(1|3|5|6) D(4|8) reps
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "gemm_dephase.hip"

extern "C" void sa_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\n");
  va_end(ap);
}
int sagemm::g_cu_budget = 0;

static uint16_t f2bf_host(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 63744, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  const int epi = argc > 4 ? atoi(argv[4]) : 6, D = argc > 5 ? atoi(argv[5]) : 8, reps = argc > 6 ? atoi(argv[6]) : 10;
  const size_t na = (size_t)M * K, nb = (size_t)N * K;
  std::vector<uint16_t> ha(na), hb(nb);
  uint32_t st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 32768.0f - 1.0f; };   // uniform [-1, 1): random data
  for (auto& v : ha) v = f2bf_host(rnd());
  for (auto& v : hb) v = f2bf_host(rnd() * 0.05f);
  void *dA, *dB, *dO, *dAux;
  float *dBias, *dRes, *dO32, *dCs;
  hipMalloc(&dA, na * 2); hipMalloc(&dB, nb * 2); hipMalloc(&dO, (size_t)M * N * 2); hipMalloc(&dAux, (size_t)M * N * 2);
  hipMalloc(&dBias, N * 4); hipMalloc(&dRes, (size_t)M * N * 4); hipMalloc(&dO32, (size_t)M * N * 4); hipMalloc(&dCs, (size_t)((M + 63) / 64) * N * 4);
  hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hb.data(), nb * 2, hipMemcpyHostToDevice);
  hipMemset(dBias, 0, N * 4); hipMemset(dRes, 0, (size_t)M * N * 4); hipMemset(dAux, 0, (size_t)M * N * 2);
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = (const char*)dA; p.B = (const char*)dB;
  p.lda = K; p.ldb = K;
  p.a_bytes = (uint32_t)(na * 2); p.b_bytes = (uint32_t)(nb * 2);
  p.M = M; p.N = N; p.K = K; p.alpha = 1.f; p.bias = dBias;
  p.epi_kind = epi; p.gm256 = 4; p.nt_store = getenv("NT") ? atoi(getenv("NT")) : 1; p.split_k = 1;
  if (epi == 3) { p.out_f32 = dO32; p.ldo_f32 = N; p.residual = dRes; p.ldr = N; }
  else { p.out_bf16 = (bf16_t*)dO; p.ldo_bf16 = N; }
  if (epi == 5) { p.aux_in = (const bf16_t*)dAux; p.ldaux = N; p.act = 4; p.bias = nullptr; p.colsum_ws = dCs; }
  if (epi == 6) { p.aux_out = (bf16_t*)dAux; p.ldaux = N; p.act = 3; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i)
    if (sagemm::launch_dephase(p, D, 0) != 0) { fprintf(stderr, "launch failed / not covered\n"); return 1; }
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) sagemm::launch_dephase(p, D, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("NT M=%d N=%d K=%d epi=%d D=%d: %.1f us  %.1f TFLOP/s\n", M, N, K, epi, D, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
  unsigned long long h[4][12];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(sa_dp_prof), sizeof(h));
  const int KS = (K + 63) / 64;
  const int ntiles = ((M + 255) / 256) * ((N + 255) / 256);
  for (int w = 0; w < 4; ++w) {
    const int wg = w < 2 ? 0 : 133;
    const int mine = (ntiles - wg + 255) / 256;            // (approximate: the XCD-aware remap changes which tiles, not how many +-1)
    const double solo = (double)mine * D, both = (double)mine * (KS - D), chunks = (double)mine * D;
    printf("  wg %3d row %d (%d tiles): solo step issue %.0f mma %.0f vmcnt %.0f barrier %.0f = %.0f | both step issue %.0f mma %.0f vmcnt %.0f barrier %.0f = %.0f"
           " | epi chunk %.0f + barrier %.0f | idle %llu total %llu\n",
           wg, w & 1, mine, h[w][0] / solo, h[w][1] / solo, h[w][2] / solo, h[w][3] / solo, (h[w][0] + h[w][1] + h[w][2] + h[w][3]) / solo,
           h[w][4] / both, h[w][5] / both, h[w][6] / both, h[w][7] / both, (h[w][4] + h[w][5] + h[w][6] + h[w][7]) / both,
           h[w][8] / chunks, h[w][9] / chunks, h[w][10], h[w][11]);
  }
  return 0;
}
