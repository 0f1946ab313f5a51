// EXPERIMENT (round 3, not part of the library; it was wired in as SA_GEMM_TILE=W through sagemm::launch_w4 and passed
// tests/gemm_mode_check.py).  Result: correct, but the main loop takes ~4 200 cycles per 256 x 256 x 64 K-tile against the 3 200 of the
// phased 8-wave kernel: with ONE wave per SIMD nothing issues MFMAs while that wave is stalled issuing an LDS-DMA instruction (~72 cycles
// each, 16 per wave and K-tile = 1 150 cycles; the same loop without its requests runs 270 us on the fc1-dgrad shape, 374 us with them),
// so the epilogue/main-loop overlap this geometry allows (the finished tile parked as bf16 in 128 of the 512 registers) cannot pay.
//
// Four-wave persistent 256 x 256 x 64 bf16 MFMA GEMM for gfx950 (forward layout, both operands k-contiguous): ONE wave per SIMD with the
// whole 512-entry register file, so that a finished tile can stay in registers -- converted to its bf16 output form -- while the next
// tile's MFMAs already run, and its epilogue (activation math, LDS staging, global stores) is issued between them instead of in front of
// them.  See the comment above the kernel.
#include "gemm_common.h"

namespace {

// =====================================================================================================
// Geometry.  256 threads = 4 waves as 2 (rows) x 2 (columns); a wave owns a 128 x 128 block of the tile: 8 x 8 MFMA 16x16x32 tiles,
// 256 accumulator registers.  LDS: two 64 KiB stages [A rows 0..127 | A rows 128..255 | B cols 0..127 | B cols 128..255], each a
// 128 x 64 half-tile image of gemm_common.h (128-byte rows, 16-byte chunks XOR-ed with row & 7), + 8 KiB of epilogue scratch per wave.
//
// Time line of a tile: K-tile s + 1 is requested (16 LDS-DMA instructions per wave) while K-tile s is consumed; one barrier per K-tile.
// After the last K-tile the accumulators are turned into the PARKED form (alpha, bias, bf16: 128 registers) in one pass of ~250 VALU
// instructions and are free again; during the first DRAIN_STEPS K-tiles of the NEXT tile each wave drains one 32 x 64 chunk of the parked
// tile per K-tile: [GELU pair on the unpacked values] -> wave-private LDS staging -> 16-byte global stores.  A K-tile's stores are issued
// AFTER its LDS-DMA requests, so `s_waitcnt vmcnt(<stores of this K-tile>)` at its end still proves that every request has landed while
// the stores stay in flight (vmcnt retires in issue order).
constexpr int W4_STAGE = 4 * TILE_BYTES;
constexpr int W4_LDS = 2 * W4_STAGE + 4 * 8192;

template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm256_w4_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int ntiles = p.tiles_m * p.tiles_n;
  const int GM = p.gm256;
  const int group_sz = GM * p.tiles_n;
  auto coords = [&](int t, int& m0, int& n0) {
    const int grp = t / group_sz, within = t - grp * group_sz;
    const int gm = min(GM, p.tiles_m - grp * GM);
    m0 = (grp * GM + within % gm) * 256;
    n0 = (within / gm) * 256;
  };
  const int ksteps = (p.K + BK - 1) / BK;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);
  auto stage_all = [&](char* buf, int m0, int n0, int k0) {
    stage_tile<true, 4>(ra, buf, p.lda, m0, k0, wave, lane);
    stage_tile<true, 4>(ra, buf + TILE_BYTES, p.lda, m0 + 128, k0, wave, lane);
    stage_tile<true, 4>(rb, buf + 2 * TILE_BYTES, p.ldb, n0, k0, wave, lane);
    stage_tile<true, 4>(rb, buf + 3 * TILE_BYTES, p.ldb, n0 + 128, k0, wave, lane);
  };

  int t = lid;
  if (t >= ntiles) return;
  int m0, n0;
  coords(t, m0, n0);
  stage_all(smem, m0, n0, 0);
  int cur = 0;
  char* const wl = smem + 2 * W4_STAGE + wave * 8192;
  while (true) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int tnext = t + nwg;
    const bool has_next = tnext < ntiles;
    int m0n = 0, n0n = 0;
    if (has_next) coords(tnext, m0n, n0n);

    f32x4 acc[2][2][4][4];                             // [row block][column block][i][j]: four 64 x 64 blocks of 4 x 4 MFMA tiles
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < ksteps; ++kt) {
      const bool more = kt + 1 < ksteps;
      // next K-tile (or the next tile's first one): this wave's 16 LDS-DMA instructions, two per MFMA group of the first k-slice
      char* const nbuf = smem + (cur ^ 1) * W4_STAGE;
      const int nm0 = more ? m0 : m0n, nn0 = more ? n0 : n0n, nk0 = more ? (kt + 1) * BK : 0;
      const bool nvalid = more || has_next;
      auto request = [&](int n) {                       // n = 0..15: half-tile n >> 2 (A lo, A hi, B lo, B hi), instruction 4 * wave + (n & 3)
        const int h = n >> 2, q = wave * 4 + (n & 3);
        if (h < 2) stage_one_v<true>(ra, nbuf + h * TILE_BYTES, p.lda, nm0 + h * 128, nk0, q, lane, nvalid);
        else stage_one_v<true>(rb, nbuf + h * TILE_BYTES, p.ldb, nn0 + (h - 2) * 128, nk0, q, lane, nvalid);
      };
      const char* ta = smem + cur * W4_STAGE + wr * TILE_BYTES;
      const char* tb = smem + cur * W4_STAGE + (2 + wc) * TILE_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fb[8], fa[2];
#pragma unroll
        for (int j = 0; j < 8; ++j) fb[j] = load_frag<true>(tb, j * 16, ks, lane);
        fa[0] = load_frag<true>(ta, 0, ks, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i < 7) fa[(i + 1) & 1] = load_frag<true>(ta, (i + 1) * 16, ks, lane);      // one A fragment ahead of the MFMAs that use it
          if (ks == 0 && !(p.ring_phase & 2)) { request(2 * i); request(2 * i + 1); }      // (ring_phase bits: timing experiments, SA_W4_DBG)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            acc[i >> 2][j >> 2][i & 3][j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i & 1], acc[i >> 2][j >> 2][i & 3][j & 3], 0, 0, 0);
          // pin the order: [1 fragment read + 2 requests] then 8 MFMAs, group after group (the scheduler otherwise sinks every read to
          // just above its first use and waits for it there: ~100 exposed cycles per group with no partner wave to cover them)
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // VMEM read (LDS-DMA)
          __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // MFMA
        }
      }
      if (more) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(p.ring_phase & 4)) __builtin_amdgcn_s_barrier();
        cur ^= 1;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // raw: the next tile's LDS-DMA stays in flight
    // ---- epilogue (serial form: four 64 x 64 blocks through the compact epilogue of the 8-wave kernels)
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
      for (int bj = 0; bj < 2; ++bj)
        wave_epilogue_compact<true, EPI>(p, acc[bi][bj], m0 + wr * 128 + bi * 64, n0 + wc * 128 + bj * 64, wl, lane);
    if (!has_next) break;
    t = tnext; m0 = m0n; n0 = n0n;
    cur ^= 1;
  }
}

template <int EPI>
int launch_w4_one(const GemmParams& p, hipStream_t stream, int slots) {
  static bool cfg = false;
  if (!cfg) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_w4_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS) != hipSuccess) {
      sa_set_error("sa_gemm_bf16: 160 KiB of LDS per workgroup refused");
      return 2;
    }
    cfg = true;
  }
  const int ntiles = p.tiles_m * p.tiles_n;
  const dim3 grid(ntiles < budget_slots(slots) ? ntiles : budget_slots(slots));
  hipLaunchKernelGGL((gemm256_w4_kernel<EPI>), grid, dim3(256), W4_LDS, stream, p);
  SA_LAUNCH_CHECK("sa_gemm_bf16(256 four-wave)");
  return 0;
}

}  // namespace

// returns -1 when the four-wave kernel does not cover the problem
int sagemm::launch_w4(GemmParams p, hipStream_t stream) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  static int slots = 0;
  if (slots == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      sa_set_error("sa_gemm_bf16: cannot query the device");
      return 2;
    }
    slots = prop.multiProcessorCount;
  }
  static const char* dbg = getenv("SA_W4_DBG");
  p.ring_phase = dbg ? atoi(dbg) : 0;
  switch (p.epi_kind) {
    case 1: return launch_w4_one<1>(p, stream, slots);
    default: return -1;
  }
}
