// Stand-alone profiler for the phased 256 x 256 GEMM kernel (ssl_audio_amd/csrc/gemm_phase.hip), with per-segment cycle stamps.
// Never part of the library: it includes the kernel's translation unit with SA_PHASE_STAMPS defined.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DSA_PHASE_STAMPS gemm_phase_prof.hip -o gemm_phase_prof
//   run:   ./gemm_phase_prof M N K layout(NT|NN|TN) epi(1|3|5|6) reps [split_k]
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../ssl_audio_amd/csrc/gemm_phase.hip"

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
  const int M = argc > 1 ? atoi(argv[1]) : 63744, N = argc > 2 ? atoi(argv[2]) : 768, K = argc > 3 ? atoi(argv[3]) : 3072;
  const char* layout = argc > 4 ? argv[4] : "NT";
  const int epi = argc > 5 ? atoi(argv[5]) : 1, reps = argc > 6 ? atoi(argv[6]) : 10, split = argc > 7 ? atoi(argv[7]) : 1;
  const bool a_km = layout[0] == 'N', b_km = layout[1] == 'T';
  const size_t na = (size_t)M * K, nb = (size_t)N * K;
  std::vector<uint16_t> ha(na), hb(nb);
  uint32_t st = 12345u;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xFFFF) / 32768.0f - 1.0f; };   // uniform [-1, 1): random data (rule 25)
  for (auto& v : ha) v = f2bf_host(rnd());
  for (auto& v : hb) v = f2bf_host(rnd());
  void *dA, *dB, *dO, *dAux;
  float *dBias, *dRes, *dO32, *dCs;
  hipMalloc(&dA, na * 2); hipMalloc(&dB, nb * 2); hipMalloc(&dO, (size_t)M * N * 2); hipMalloc(&dAux, (size_t)M * N * 2);
  hipMalloc(&dBias, N * 4); hipMalloc(&dRes, (size_t)M * N * 4); hipMalloc(&dO32, (size_t)M * N * 4); hipMalloc(&dCs, (size_t)((M + 63) / 64) * N * 4);
  hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hb.data(), nb * 2, hipMemcpyHostToDevice);
  hipMemset(dBias, 0, N * 4); hipMemset(dRes, 0, (size_t)M * N * 4); hipMemset(dAux, 0, (size_t)M * N * 2);
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = (const char*)dA; p.B = (const char*)dB;
  p.lda = a_km ? K : M; p.ldb = b_km ? K : N;
  p.a_bytes = (uint32_t)(na * 2); p.b_bytes = (uint32_t)(nb * 2);
  p.M = M; p.N = N; p.K = K; p.alpha = 1.f; p.bias = dBias;
  p.epi_kind = epi; p.gm256 = 4; p.nt_store = getenv("NT") ? atoi(getenv("NT")) : 1; p.split_k = split;
  if (split > 1) { p.out_f32 = dO32; p.ldo_f32 = N; p.bias = nullptr; }
  else if (epi == 3) { p.out_f32 = dO32; p.ldo_f32 = N; p.residual = dRes; p.ldr = N; }
  else { p.out_bf16 = (bf16_t*)dO; p.ldo_bf16 = N; }
  if (epi == 5) { p.aux_in = (const bf16_t*)dAux; p.ldaux = N; p.act = 4; p.bias = nullptr; p.colsum_ws = dCs; }
  if (epi == 6) { p.aux_out = (bf16_t*)dAux; p.ldaux = N; p.act = 3; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i)
    if (sagemm::launch_phase(p, a_km, b_km, split > 1, 0) != 0) { fprintf(stderr, "launch failed\n"); return 1; }
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) sagemm::launch_phase(p, a_km, b_km, split > 1, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("%s M=%d N=%d K=%d epi=%d split=%d: %.1f us  %.1f TFLOP/s\n", layout, M, N, K, epi, split, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
#ifdef SA_PHASE_STAMPS
  unsigned long long h[4][8];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(sa_phase_prof), sizeof(h));
  const char* names[8] = {"L-issue", "vmcnt", "barrier1", "lgkmcnt", "MFMA-issue", "barrier2", "epilogue", "total"};
  const int ksteps = (K + 63) / 64;
  for (int w = 0; w < 4; ++w) {
    printf("  wg %d row %d:", w < 2 ? 0 : 133, w & 1);
    for (int k = 0; k < 8; ++k) printf(" %s=%llu", names[k], h[w][k]);
    const double phases = (double)(h[w][0] + h[w][1] + h[w][2] + h[w][3] + h[w][4] + h[w][5]);
    printf("  | K-loop share %.2f\n", phases / (double)h[w][7]);
  }
  (void)ksteps;
#endif
  return 0;
}
