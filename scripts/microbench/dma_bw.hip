// Micro-benchmark: how fast can one persistent workgroup per CU stream GEMM operand tiles L2/HBM -> LDS with
// `buffer_load ... lds` on gfx950, as a function of bytes in flight?  Mirrors the access pattern of the 256 x 256 GEMM
// (A tile: 256 rows x RB bytes, B tile: 256 rows x RB bytes per K-step), no MFMA, no fragment reads.
//   build: hipcc -O3 --offload-arch=gfx950 dma_bw.hip -o dma_bw      run: ./dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// RB = bytes per row per stage (128 or 64); DEPTH = stages in flight; NW waves
template <int RB, int DEPTH, int NW, bool CONTIG = false>
__global__ __launch_bounds__(64 * NW) void dma_kernel(const char* A, uint32_t a_bytes, int lda_b, const char* B, uint32_t b_bytes, int ldb_b,
                                                      int tiles_m, int tiles_n, int ksteps, float* sink, int GM, int GN, int reps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 512 * RB;
  constexpr int INSTR = STAGE / 1024;          // wave-instructions per stage
  constexpr int PER_WAVE = INSTR / NW;
  constexpr int LPR = RB / 16;                 // lanes per row
  constexpr int RPI = 64 / LPR;                // rows per instruction
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, a_bytes), rb = make_rsrc(B, b_bytes);
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int ntiles = tiles_m * tiles_n;
  const int per_pass = ((ntiles - lid + nwg - 1) / nwg) * ksteps;
  const int total = per_pass * reps;   // K-tiles this workgroup streams
  auto issue = [&](int g, char* buf) {
    g = g % per_pass;
    const int ti = g / ksteps, kt = g - ti * ksteps;
    const int t = lid + ti * nwg;
    // super-groups of GM row-panels; inside, column blocks of GN tiles; inside, m fastest
    const int sg = t / (GM * tiles_n), w0 = t - sg * GM * tiles_n;
    const int gm = min(GM, tiles_m - sg * GM);
    const int cb = w0 / (gm * GN), w1 = w0 - cb * gm * GN;
    const int gn = min(GN, tiles_n - cb * GN);
    (void)gn;
    const int m0 = (sg * GM + w1 % gm) * 256, n0 = (cb * GN + w1 / gm) * 256;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int q = wave * PER_WAVE + i;       // 0 .. INSTR-1; first half A, second half B
      const int row = (q % (INSTR / 2)) * RPI + lane / LPR;
      const uint32_t col = (uint32_t)kt * RB + (lane % LPR) * 16;
      if constexpr (CONTIG) {
        // pre-tiled operand: the K-tile of a 256-row panel is one contiguous 32 KiB block, instruction q reads 1 KiB of it
        const uint32_t hq = q % (INSTR / 2);
        if (q < INSTR / 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(buf + q * 1024), 16, ((uint32_t)(m0 / 256) * ksteps + kt) * (256u * RB) + hq * 1024 + lane * 16, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(buf + q * 1024), 16, ((uint32_t)(n0 / 256) * ksteps + kt) * (256u * RB) + hq * 1024 + lane * 16, 0, 0, 0);
      } else {
      if (q < INSTR / 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(buf + q * 1024), 16, (uint32_t)(m0 + row) * lda_b + col, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(buf + q * 1024), 16, (uint32_t)(n0 + row) * ldb_b + col, 0, 0, 0);
      }
    }
  };
  int g = 0;
  for (; g < DEPTH && g < total; ++g) issue(g, smem + (g % DEPTH) * STAGE);
  float acc = 0.f;
  for (int c = 0; c < total; ++c) {
    if (c + DEPTH - 1 < total) wait_vm<(DEPTH - 1) * PER_WAVE>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    acc += *reinterpret_cast<const float*>(smem + (c % DEPTH) * STAGE + threadIdx.x * 4);   // touch the stage
    __builtin_amdgcn_s_barrier();
    if (g < total) { issue(g, smem + (c % DEPTH) * STAGE); ++g; }
  }
  if (acc == 123.456f) sink[0] = acc;
}


// "2.5-stage" ring: 32 KiB half-stages (A half = 256 rows x 128 B, B half likewise), INFL halves kept in flight after each
// issue, the two oldest retired per step (what a GEMM K-step consumes).
template <int SLOTS, int INFL>
__global__ __launch_bounds__(512) void ring_kernel(const char* A, uint32_t a_bytes, int lda_b, const char* B, uint32_t b_bytes, int ldb_b,
                                                   int tiles_m, int tiles_n, int ksteps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, a_bytes), rb = make_rsrc(B, b_bytes);
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int ntiles = tiles_m * tiles_n;
  const int total = ((ntiles - lid + nwg - 1) / nwg) * ksteps * 2;   // half-tiles this workgroup streams
  auto issue = [&](int h) {
    const int g = h >> 1, isb = h & 1;
    const int ti = g / ksteps, kt = g - ti * ksteps;
    const int t = lid + ti * nwg;
    const int grp = t / (4 * tiles_n), within = t - grp * 4 * tiles_n;
    const int gm = min(4, tiles_m - grp * 4);
    const int m0 = (grp * 4 + within % gm) * 256, n0 = (within / gm) * 256;
    char* buf = smem + (h % SLOTS) * 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = wave * 4 + i;              // 32 instructions of 8 rows
      const int row = q * 8 + (lane >> 3);
      const uint32_t col = (uint32_t)kt * 128 + (lane & 7) * 16;
      if (!isb) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(buf + q * 1024), 16, (uint32_t)(m0 + row) * lda_b + col, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(buf + q * 1024), 16, (uint32_t)(n0 + row) * ldb_b + col, 0, 0, 0);
    }
  };
  int h = 0;
  for (; h < INFL && h < total; ++h) issue(h);
  float acc = 0.f;
  for (int c = 0; c < total; c += 2) {
    if (c + INFL <= total) wait_vm<(INFL - 2) * 4>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    acc += *reinterpret_cast<const float*>(smem + (c % SLOTS) * 32768 + threadIdx.x * 4);
    __builtin_amdgcn_s_barrier();
    for (int k = 0; k < 2; ++k)
      if (h < total) { issue(h); ++h; }
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int SLOTS, int INFL>
void run_ring(const char* A, size_t a_bytes, int lda_b, const char* B, size_t b_bytes, int ldb_b, int M, int N, int Kbytes, float* sink) {
  const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256, ksteps = Kbytes / 128;
  const int lds = SLOTS * 32768;
  hipFuncSetAttribute(reinterpret_cast<const void*>(ring_kernel<SLOTS, INFL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w)
    hipLaunchKernelGGL((ring_kernel<SLOTS, INFL>), dim3(256), dim3(512), lds, 0, A, (uint32_t)a_bytes, lda_b, B, (uint32_t)b_bytes, ldb_b, tiles_m, tiles_n, ksteps, sink);
  hipEventRecord(e0);
  const int nrep = 5;
  for (int w = 0; w < nrep; ++w)
    hipLaunchKernelGGL((ring_kernel<SLOTS, INFL>), dim3(256), dim3(512), lds, 0, A, (uint32_t)a_bytes, lda_b, B, (uint32_t)b_bytes, ldb_b, tiles_m, tiles_n, ksteps, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= nrep;
  hipError_t err = hipGetLastError();
  const double bytes = (double)tiles_m * tiles_n * ksteps * 65536.0;
  printf("ring slots=%d in-flight-halves=%d (%3d KB after issue, %3d KB at the wait)  %8.1f us  %5.1f B/clk/CU (%s)\n", SLOTS, INFL, INFL * 32, (INFL - 2) * 32,
         ms * 1e3, bytes / (ms * 1e-3) / 256 / 2.4e9, err == hipSuccess ? "ok" : hipGetErrorString(err));
}


// Same stream through VGPRs: buffer_load_dwordx4 into registers, then (optionally) ds_write_b128 into LDS.  DEPTH stages of
// 8 x 16 B per lane are kept in flight in registers.
template <int DEPTH, bool TO_LDS>
__global__ __launch_bounds__(512) void reg_kernel(const char* A, uint32_t a_bytes, int lda_b, const char* B, uint32_t b_bytes, int ldb_b,
                                                  int tiles_m, int tiles_n, int ksteps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, a_bytes), rb = make_rsrc(B, b_bytes);
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int ntiles = tiles_m * tiles_n;
  const int total = ((ntiles - lid + nwg - 1) / nwg) * ksteps;
  u32x4 r[DEPTH][8];
  auto issue = [&](int g, u32x4 (&dst)[8]) {
    const int ti = g / ksteps, kt = g - ti * ksteps;
    const int t = lid + ti * nwg;
    const int grp = t / (4 * tiles_n), within = t - grp * 4 * tiles_n;
    const int gm = min(4, tiles_m - grp * 4);
    const int m0 = (grp * 4 + within % gm) * 256, n0 = (within / gm) * 256;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = wave * 8 + i;              // 64 instructions: 32 A, 32 B
      const int row = (q & 31) * 8 + (lane >> 3);
      const uint32_t col = (uint32_t)kt * 128 + (lane & 7) * 16;
      if (q < 32) dst[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (uint32_t)(m0 + row) * lda_b + col, 0, 0));
      else dst[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (uint32_t)(n0 + row) * ldb_b + col, 0, 0));
    }
  };
  u32x4 accv = {0, 0, 0, 0};
  int g = 0;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) if (d < total) { issue(d, r[d]); ++g; }
  for (int c = 0; c < total; c += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (c + d >= total) break;
      // consume stage d (the compiler inserts the counted vmcnt), then refill it
      if (TO_LDS) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(smem + ((c + d) & 1) * 65536 + (wave * 8 + i) * 1024 + lane * 16) = r[d][i];
        __builtin_amdgcn_s_barrier();
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) accv ^= r[d][i];
      }
      if (g < total) { issue(g, r[d]); ++g; }
    }
  }
  if (TO_LDS) accv[0] ^= *reinterpret_cast<unsigned int*>(smem + threadIdx.x * 4);
  if ((accv[0] ^ accv[1] ^ accv[2] ^ accv[3]) == 0x12345678u) sink[0] = 1.f;
}

template <int DEPTH, bool TO_LDS>
void run_reg(const char* A, size_t a_bytes, int lda_b, const char* B, size_t b_bytes, int ldb_b, int M, int N, int Kbytes, float* sink) {
  const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256, ksteps = Kbytes / 128;
  const int lds = TO_LDS ? 131072 : 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(reg_kernel<DEPTH, TO_LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w)
    hipLaunchKernelGGL((reg_kernel<DEPTH, TO_LDS>), dim3(256), dim3(512), lds, 0, A, (uint32_t)a_bytes, lda_b, B, (uint32_t)b_bytes, ldb_b, tiles_m, tiles_n, ksteps, sink);
  hipEventRecord(e0);
  const int nrep = 5;
  for (int w = 0; w < nrep; ++w)
    hipLaunchKernelGGL((reg_kernel<DEPTH, TO_LDS>), dim3(256), dim3(512), lds, 0, A, (uint32_t)a_bytes, lda_b, B, (uint32_t)b_bytes, ldb_b, tiles_m, tiles_n, ksteps, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= nrep;
  hipError_t err = hipGetLastError();
  const double bytes = (double)tiles_m * tiles_n * ksteps * 65536.0;
  printf("via VGPRs depth=%d (%3d KB in flight) %s  %8.1f us  %5.1f B/clk/CU (%s)\n", DEPTH, DEPTH * 64, TO_LDS ? "+ds_write" : "         ",
         ms * 1e3, bytes / (ms * 1e-3) / 256 / 2.4e9, err == hipSuccess ? "ok" : hipGetErrorString(err));
}


// Mixed: per K-step the A half goes HBM/L2 -> LDS by LDS-DMA, the B half through VGPRs + ds_write_b128 (same LDS image).
// One stage in flight on each path (what a double-buffered GEMM can afford: 64 KiB of LDS per stage + 16 VGPRs).
template <bool B_ALSO_DMA>
__global__ __launch_bounds__(512) void mix_kernel(const char* A, uint32_t a_bytes, int lda_b, const char* B, uint32_t b_bytes, int ldb_b,
                                                  int tiles_m, int tiles_n, int ksteps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, a_bytes), rb = make_rsrc(B, b_bytes);
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int ntiles = tiles_m * tiles_n;
  const int total = ((ntiles - lid + nwg - 1) / nwg) * ksteps;
  u32x4 r[4];
  float acc = 0.f;
  for (int g = 0; g < total; ++g) {
    const int ti = g / ksteps, kt = g - ti * ksteps;
    const int t = lid + ti * nwg;
    const int grp = t / (4 * tiles_n), within = t - grp * 4 * tiles_n;
    const int gm = min(4, tiles_m - grp * 4);
    const int m0 = (grp * 4 + within % gm) * 256, n0 = (within / gm) * 256;
    char* buf = smem + (g & 1) * 65536;
#pragma unroll
    for (int i = 0; i < 4; ++i) {              // B half: 32 instructions over 8 waves
      const int q = wave * 4 + i;
      const int row = q * 8 + (lane >> 3);
      const uint32_t off = (uint32_t)(n0 + row) * ldb_b + (uint32_t)kt * 128 + (lane & 7) * 16;
      if (B_ALSO_DMA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(buf + 32768 + q * 1024), 16, off, 0, 0, 0);
      else r[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {              // A half by LDS-DMA
      const int q = wave * 4 + i;
      const int row = q * 8 + (lane >> 3);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(buf + q * 1024), 16, (uint32_t)(m0 + row) * lda_b + (uint32_t)kt * 128 + (lane & 7) * 16, 0, 0, 0);
    }
    if (!B_ALSO_DMA) {
      wait_vm<4>();                            // the four register loads are older than the four LDS-DMA requests
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(buf + 32768 + (wave * 4 + i) * 1024 + lane * 16) = r[i];
    }
    wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    acc += *reinterpret_cast<const float*>(buf + threadIdx.x * 4) + *reinterpret_cast<const float*>(buf + 32768 + threadIdx.x * 4);
    __builtin_amdgcn_s_barrier();
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <bool B_ALSO_DMA>
void run_mix(const char* A, size_t a_bytes, int lda_b, const char* B, size_t b_bytes, int ldb_b, int M, int N, int Kbytes, float* sink) {
  const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256, ksteps = Kbytes / 128;
  const int lds = 131072;
  hipFuncSetAttribute(reinterpret_cast<const void*>(mix_kernel<B_ALSO_DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w)
    hipLaunchKernelGGL((mix_kernel<B_ALSO_DMA>), dim3(256), dim3(512), lds, 0, A, (uint32_t)a_bytes, lda_b, B, (uint32_t)b_bytes, ldb_b, tiles_m, tiles_n, ksteps, sink);
  hipEventRecord(e0);
  const int nrep = 5;
  for (int w = 0; w < nrep; ++w)
    hipLaunchKernelGGL((mix_kernel<B_ALSO_DMA>), dim3(256), dim3(512), lds, 0, A, (uint32_t)a_bytes, lda_b, B, (uint32_t)b_bytes, ldb_b, tiles_m, tiles_n, ksteps, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= nrep;
  hipError_t err = hipGetLastError();
  const double bytes = (double)tiles_m * tiles_n * ksteps * 65536.0;
  printf("serial d1, A by LDS-DMA, B by %s  %8.1f us  %5.1f B/clk/CU (%s)\n", B_ALSO_DMA ? "LDS-DMA        " : "VGPR + ds_write", ms * 1e3,
         bytes / (ms * 1e-3) / 256 / 2.4e9, err == hipSuccess ? "ok" : hipGetErrorString(err));
}

template <int RB, int DEPTH, int NW, bool CONTIG = false>
void run(const char* name, const char* A, size_t a_bytes, int lda_b, const char* B, size_t b_bytes, int ldb_b, int M, int N, int Kbytes, float* sink, int grid, int GM, int GN, int reps = 1) {
  const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256, ksteps = Kbytes / RB;
  const int lds = DEPTH * 512 * RB;
  hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<RB, DEPTH, NW, CONTIG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w)
    hipLaunchKernelGGL((dma_kernel<RB, DEPTH, NW, CONTIG>), dim3(grid), dim3(64 * NW), lds, 0, A, (uint32_t)a_bytes, lda_b, B, (uint32_t)b_bytes, ldb_b, tiles_m, tiles_n, ksteps, sink, GM, GN, reps);
  hipEventRecord(e0);
  const int nrep = 5;
  for (int w = 0; w < nrep; ++w)
    hipLaunchKernelGGL((dma_kernel<RB, DEPTH, NW, CONTIG>), dim3(grid), dim3(64 * NW), lds, 0, A, (uint32_t)a_bytes, lda_b, B, (uint32_t)b_bytes, ldb_b, tiles_m, tiles_n, ksteps, sink, GM, GN, reps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= nrep;
  hipError_t err = hipGetLastError();
  const double bytes = (double)tiles_m * tiles_n * ksteps * 512.0 * RB * reps;
  printf("%-30s GM=%2d GN=%2d lds=%3d KB grid=%3d  %8.1f us  %6.2f TB/s  %5.1f B/clk/CU  (%s)\n", name, GM, GN, lds / 1024, grid, ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9,
         err == hipSuccess ? "ok" : hipGetErrorString(err));
}

int main() {
  float* sink; hipMalloc(&sink, 4);
  const int Ms[2] = {64256, 64256}, Ns[2] = {3072, 768}, Ks[2] = {768, 3072};
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int M = Ms[cfg], K = Ks[cfg], N = Ns[cfg];
    const int ld_b = K * 2;
    const size_t a_bytes = (size_t)(M + 256) * ld_b, b_bytes = (size_t)N * ld_b;
    char *A, *B; hipMalloc(&A, a_bytes + 65536); hipMalloc(&B, b_bytes + 65536);
    hipMemset(A, 1, a_bytes); hipMemset(B, 1, b_bytes);
    printf("---- M=%d N=%d K=%d\n", M, N, K);
    run<128, 1, 8>("row-major d1", A, a_bytes, ld_b, B, b_bytes, ld_b, M, N, K * 2, sink, 256, 4, N / 256, 1);
    run<128, 2, 8>("row-major d2", A, a_bytes, ld_b, B, b_bytes, ld_b, M, N, K * 2, sink, 256, 4, N / 256, 1);
    run<128, 1, 8, true>("pre-tiled d1", A, a_bytes, ld_b, B, b_bytes, ld_b, M, N, K * 2, sink, 256, 4, N / 256, 1);
    run<128, 2, 8, true>("pre-tiled d2", A, a_bytes, ld_b, B, b_bytes, ld_b, M, N, K * 2, sink, 256, 4, N / 256, 1);
    hipFree(A); hipFree(B);
  }
  return 0;
}
