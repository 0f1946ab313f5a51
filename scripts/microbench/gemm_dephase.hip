// De-phased persistent 256 x 256 x 64 bf16 MFMA GEMM for gfx950 (forward operand layout, compact epilogues 1 and 6).
// EXPERIMENT, NOT PART OF THE LIBRARY (round 4, VERDICT r3 #1: "two independent tiles per CU so one's epilogue runs under the other's
// MFMAs ... the de-phased wave-row form").  Correct (bf16 outputs 1.7e-3 from fp32 torch products at ViT-B and ragged shapes, K = 768 .. 3072; gpurun_out/r04/dp_check1.txt),
// measured NEGATIVE: qkv forward 249 -> 264 us, fc1 forward (GELU pair) 397 -> 423 us (D = 4), 467 (D = 8).  The per-segment cycle
// stamps (profiles/r04_gemm_dephase_stamps.txt, gemm_dephase_prof.hip) say why:
//   * a K-step in which ONE wave row computes alone (48 KiB requested instead of 64) is no shorter than a shared one (issue + MFMA
//     ~2 600 cycles either way): a step is bound by issuing its requests and waiting for them to land, not by bytes or MFMA slots;
//   * an epilogue chunk run by one wave per SIMD next to a main-loop wave takes as long as the lock-step kernel needs for BOTH rows'
//     share (GELU pair: 16 rows in ~5 000 stamped cycles; bias only: 32 rows in ~3 000): the lock-step epilogue is VALU-bound with two
//     waves per SIMD covering each other's LDS / store latencies, a lone wave is latency-bound -- so de-phasing doubles the epilogue's
//     wall time and the main-loop row waits at every barrier for the chunk;
//   * the younger wave row (waves 4..7) loses the issue arbitration: its MFMA segment takes 2 300 - 2 800 cycles against 1 300.
#include "../../ssl_audio_amd/csrc/gemm_common.h"

namespace {

// =====================================================================================================
// Why: the persistent 256^2 kernel (gemm_bf16.hip) runs a tile as [main loop: 12 K-steps at K = 768][epilogue], both wave rows in
// lock step.  Its main loop is bound by the operand stream L2 -> LDS (64 KiB per K-step, ~3 200 cycles against 2 048 of MFMA), and
// during the epilogue (9 - 23 k cycles: VALU / store bound) both the stream and the matrix pipe idle: nobody consumes K-tiles, and
// LDS holds no more than the one prefetched stage.
//
// Here wave row 1 (waves 4..7: tile rows 128..255) runs D K-steps BEHIND wave row 0 (waves 0..3: rows 0..127) and walks its
// reduction ROTATED by D K-tiles, so that whenever both rows are in their main loops they are on the same K-tile of the same tile
// and share its B stage:
//
//   step j of tile i   0 .. D-1               D .. KS-1          KS .. KS+D-1
//   row 0              main k = j             main k = j         epilogue chunk j - KS
//   row 1              epilogue of tile i-1   main k = j         main k = j - KS            (KS = K / 64, period KS + D steps)
//
// One wave of every SIMD is therefore always in a main loop: the stream never stops, the epilogue's VALU / store work runs under
// the other row's MFMAs, and a step in which one row computes alone moves 48 KiB (its A half + B) instead of 64.  The price is B
// fetched for KS + D steps instead of KS.  The sum over k is the same set of products in a rotated order for row 1 (fp32
// accumulation: rows 128..255 of a tile differ from the lock-step kernels by rounding only; bit-reproducible run to run).
//
// Steps are separated by ONE workgroup barrier, as in the lock-step kernel.  All LDS-DMA requests of a step are issued by the waves
// that are in a main loop during it (1/8 of the stage each when both rows are, 1/4 each when one is); a wave in its epilogue only
// executes its chunk (one D-th of its 128 x 64 accumulators through its private 4 KiB of scratch) and joins the barrier.
// LDS: two 64 KiB stages [A rows 0..127 | A rows 128..255 | B cols 0..127 | B cols 128..255] + 8 x 4 KiB scratch = 160 KiB.
constexpr int DP_STAGE = 4 * TILE_BYTES;
constexpr int DP_SCRATCH = 4096;
constexpr int DP_LDS = 2 * DP_STAGE + 8 * DP_SCRATCH;

// SA_DP_STAMPS (scripts/microbench/gemm_dephase_prof.hip only; never in the library): cycle stamps of waves 0 and 4 of two workgroups,
// summed per segment kind -- [0..3] issue / MFMA / vmcnt wait / barrier of a main step run ALONE, [4..7] the same with both rows in
// their main loops, [8] epilogue chunk, [9] its barrier wait, [10] idle steps, [11] whole kernel
#ifdef SA_DP_STAMPS
__device__ unsigned long long sa_dp_prof[4][12];
#define DP_T(k) { __builtin_amdgcn_sched_barrier(0); tnow = __builtin_readcyclecounter(); pt[k] += tnow - tprev; tprev = tnow; __builtin_amdgcn_sched_barrier(0); }
#else
#define DP_T(k)
#endif

template <int EPI, int D>
__global__ __launch_bounds__(512, 2) void gemm256_dephase_kernel(const GemmParams p) {
  static_assert(D == 4 || D == 8, "epilogue chunks of 32 or 16 rows");
  constexpr int NI = 8 / D;                            // 16-row groups per epilogue chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;

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
  const int KS = (p.K + BK - 1) / BK;                  // (the host guarantees KS >= D)
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);
  char* const scratch = smem + 2 * DP_STAGE + wave * DP_SCRATCH;

  // this wave's share of the requests that fill `buf` for K-tile k of tile (m0, n0): the A half of row 0 / row 1 and B, as asked
  auto issue = [&](char* buf, int m0, int n0, int k, bool a0, bool a1, bool b, bool solo) {
    const int k0 = k * BK;
    if (solo) {
      const int w = wave & 3;
      if (a0) stage_tile<true, 4>(ra, buf, p.lda, m0, k0, w, lane);
      if (a1) stage_tile<true, 4>(ra, buf + TILE_BYTES, p.lda, m0 + 128, k0, w, lane);
      if (b) {
        stage_tile<true, 4>(rb, buf + 2 * TILE_BYTES, p.ldb, n0, k0, w, lane);
        stage_tile<true, 4>(rb, buf + 3 * TILE_BYTES, p.ldb, n0 + 128, k0, w, lane);
      }
    } else {
      if (a0) stage_tile<true, 8>(ra, buf, p.lda, m0, k0, wave, lane);
      if (a1) stage_tile<true, 8>(ra, buf + TILE_BYTES, p.lda, m0 + 128, k0, wave, lane);
      if (b) {
        stage_tile<true, 8>(rb, buf + 2 * TILE_BYTES, p.ldb, n0, k0, wave, lane);
        stage_tile<true, 8>(rb, buf + 3 * TILE_BYTES, p.ldb, n0 + 128, k0, wave, lane);
      }
    }
  };

  int t = lid;
  if (t >= ntiles) return;
  int m0, n0;
  coords(t, m0, n0);
  issue(smem, m0, n0, 0, true, false, true, false);    // step 0 of the first tile: row 0 alone on K-tile 0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int cur = 0;
#ifdef SA_DP_STAMPS
  unsigned long long pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tnow, tprev = __builtin_readcyclecounter();
  const unsigned long long tbegin = tprev;
#endif
  // end of a step: requests of this step landed (only waves that issued any wait), everybody past the stage that is overwritten next
#define DP_STEP_END(issued)                                          \
  do {                                                               \
    if (issued) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_s_barrier();                                    \
    cur ^= 1;                                                        \
  } while (0)

  f32x4 acc[8][4];
  bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
  // one K-step of this wave's 128 x 64 output from stage `st` (the lock-step kernel's order: two 64 x 32 quadrants per A half)
  auto mma_step = [&](const char* st) {
    const char* ta = st + wr * TILE_BYTES;
    const char* tb = st + (2 + (wc >> 1)) * TILE_BYTES;
    const int bcol = (wc & 1) * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[i][ks] = load_frag<true>(ta, i * 16, ks, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb0[j][ks] = load_frag<true>(tb, bcol + j * 16, ks, lane);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j][ks], fa[i][ks], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb1[j][ks] = load_frag<true>(tb, bcol + 32 + j * 16, ks, lane);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j][ks], fa[i][ks], acc[i][2 + j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[i][ks] = load_frag<true>(ta, 64 + i * 16, ks, lane);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j][ks], fa[i][ks], acc[4 + i][2 + j], 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j][ks], fa[i][ks], acc[4 + i][j], 0, 0, 0);
  };
  // epilogue chunk c of the tile at (em0, en0): 16 * NI rows of this wave's accumulators
  auto epi_chunk = [&](int c, int em0, int en0) {
    const int nb = en0 + wc * 64;
#pragma unroll
    for (int cc = 0; cc < D; ++cc)
      if (c == cc) {
        f32x4(&blk)[NI][4] = reinterpret_cast<f32x4(&)[NI][4]>(acc[cc * NI]);
        wave_epilogue_compact<true, EPI, NI>(p, blk, em0 + wr * 128 + cc * NI * 16, nb, scratch, lane);
      }
  };
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  if (wr == 0) {
    // ------------------------------------------------------------------ wave row 0
    while (true) {
      const int tnext = t + nwg;
      zero_acc();
      for (int j = 0; j < KS; ++j) {
        // next step j + 1 <= KS < KS + D: row 0 needs its A half while it is still in its main loop, row 1 from step D on;
        // K-tile j + 1, or 0 again when row 1 wraps around (row 0 then leaves for its epilogue)
        const int kn = (j + 1 < KS) ? j + 1 : 0;
        const int so = (j < D) ? 0 : 4;
        issue(smem + (cur ^ 1) * DP_STAGE, m0, n0, kn, j + 1 < KS, j + 1 >= D, true, j < D);
        DP_T(so + 0)
        mma_step(smem + cur * DP_STAGE);
        DP_T(so + 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DP_T(so + 2)
        DP_STEP_END(false);
        DP_T(so + 3)
      }
      for (int c = 0; c < D; ++c) {
        epi_chunk(c, m0, n0);
        DP_T(8)
        DP_STEP_END(false);
        DP_T(9)
      }
      if (tnext >= ntiles) break;
      t = tnext;
      coords(t, m0, n0);
    }
    for (int c = 0; c < D; ++c) DP_STEP_END(false);          // row 1's last epilogue
    DP_T(10)
  } else {
    // ------------------------------------------------------------------ wave row 1: D steps behind, reduction rotated by D K-tiles
    for (int c = 0; c < D; ++c) DP_STEP_END(false);
    DP_T(10)
    while (true) {
      const int tnext = t + nwg;
      const bool has_next = tnext < ntiles;
      int m0n = 0, n0n = 0;
      if (has_next) coords(tnext, m0n, n0n);
      zero_acc();
      for (int j = D; j < KS + D; ++j) {
        const bool solo = j >= KS;                              // row 0 is in its epilogue
        if (j + 1 < KS + D) {
          const int kn = (j + 1 < KS) ? j + 1 : j + 1 - KS;
          issue(smem + (cur ^ 1) * DP_STAGE, m0, n0, kn, j + 1 < KS, true, true, solo);
        } else {
          // last step of this tile: the next step is step 0 of the next tile, row 0 alone on its K-tile 0
          issue(smem + (cur ^ 1) * DP_STAGE, m0n, n0n, 0, has_next, false, has_next, solo);
        }
        const int so = solo ? 0 : 4;
        DP_T(so + 0)
        mma_step(smem + cur * DP_STAGE);
        DP_T(so + 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DP_T(so + 2)
        DP_STEP_END(false);
        DP_T(so + 3)
      }
      for (int c = 0; c < D; ++c) {
        epi_chunk(c, m0, n0);
        DP_T(8)
        DP_STEP_END(false);
        DP_T(9)
      }
      if (!has_next) break;
      t = tnext; m0 = m0n; n0 = n0n;
    }
  }
#undef DP_STEP_END
#ifdef SA_DP_STAMPS
  if ((blockIdx.x == 0 || blockIdx.x == 133) && (threadIdx.x & 255) == 0) {     // waves 0 and 4 of two workgroups
    unsigned long long* o = sa_dp_prof[(blockIdx.x == 0 ? 0 : 2) + wr];
    for (int k = 0; k < 11; ++k) o[k] = pt[k];
    o[11] = __builtin_readcyclecounter() - tbegin;
  }
#endif
}

template <int EPI, int D>
int launch_dephase_one(const GemmParams& p, hipStream_t stream, int slots) {
  static bool cfg = false;
  if (!cfg) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_dephase_kernel<EPI, D>), hipFuncAttributeMaxDynamicSharedMemorySize, DP_LDS) !=
        hipSuccess) {
      sa_set_error("sa_gemm_bf16: 160 KiB of LDS per workgroup refused");
      return 2;
    }
    cfg = true;
  }
  const int ntiles = p.tiles_m * p.tiles_n;
  const dim3 grid(ntiles < budget_slots(slots) ? ntiles : budget_slots(slots));
  hipLaunchKernelGGL((gemm256_dephase_kernel<EPI, D>), grid, dim3(512), DP_LDS, stream, p);
  SA_LAUNCH_CHECK("sa_gemm_bf16(256 de-phased)");
  return 0;
}

}  // namespace

// returns -1 when the problem is not one this kernel covers (the caller then takes the lock-step kernels)
namespace sagemm { int launch_dephase(GemmParams p, int d, hipStream_t stream); }
int sagemm::launch_dephase(GemmParams p, int d, hipStream_t stream) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  const int KS = (p.K + BK - 1) / BK;
  if (KS < 8 || p.tile_counter) return -1;
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
  switch (p.epi_kind) {
    case 1: return d == 8 ? launch_dephase_one<1, 8>(p, stream, slots) : launch_dephase_one<1, 4>(p, stream, slots);
    case 6: return d == 4 ? launch_dephase_one<6, 4>(p, stream, slots) : launch_dephase_one<6, 8>(p, stream, slots);
    default: return -1;
  }
}
