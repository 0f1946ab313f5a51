// bf16 MFMA GEMM for gfx950 with fused epilogues (bias / GELU / GELU' / residual / row scatter / split-K).
//
// Tile 128 x 128 x 64, 256 threads = 4 waves (2 x 2), each wave 64 x 64 = 4 x 4 MFMA 16x16x32 tiles
// (64 accumulator VGPRs).  Both operand tiles are brought HBM -> LDS by `buffer_load_dwordx4 ... lds`
// (no VGPR round trip, hardware bounds check gives zero fill for ragged M / N / K-rows), two LDS
// buffers, one barrier per K step.  Each operand can be stored with the reduction index contiguous
// ("k-major": 128-byte LDS rows, XOR-swizzled 16-byte chunks, ds_read_b128 fragments) or strided
// ("k-strided": 256-byte LDS rows, ds_read_b64_tr_b16 transposing reads), so forward (NT), dgrad (NN)
// and wgrad (TN) are the same kernel and no transposed copy of a weight or activation is ever made.
// LDS destination of an LDS-DMA is lane-linear, so swizzles are applied to the per-lane SOURCE address
// and again on the fragment read (cdna_hip_programming.md rule 21).
#include <string.h>
#include "gemm_common.h"

int sagemm::g_cu_budget = 0;

// ---- dynamic tile hand-out ---------------------------------------------------------------------------------------------------------
// Under data parallelism RCCL's kernels hold some CUs for the length of an all-reduce.  A persistent grid with STATIC striding then
// finishes late by a whole share of tiles on every workgroup that had to wait for a CU; with tickets drawn from one counter a late
// workgroup just draws fewer tiles.  One word per launch out of a ring (launches on one stream are ordered; 256 launches later a slot
// is long dead), zeroed by a memset node ahead of the kernel so the entry point stays capturable and stateless.
namespace {
__device__ int g_tile_tickets[256];
int g_dynamic_tiles = -1;                 // -1: not decided yet (SA_GEMM_DYNAMIC; default off: measured 1.5-2 % on an undisturbed chip)
}  // namespace

int* sagemm::next_tile_counter(hipStream_t stream) {
  if (g_dynamic_tiles < 0) {
    const char* e = getenv("SA_GEMM_DYNAMIC");
    g_dynamic_tiles = (e && e[0] == '1') ? 1 : 0;
  }
  if (!g_dynamic_tiles) return nullptr;
  static int* base = nullptr;
  static unsigned turn = 0;
  if (!base && hipGetSymbolAddress(reinterpret_cast<void**>(&base), HIP_SYMBOL(g_tile_tickets)) != hipSuccess) return nullptr;
  int* slot = base + (turn++ & 255u);
  if (hipMemsetAsync(slot, 0, sizeof(int), stream) != hipSuccess) return nullptr;
  return slot;
}

extern "C" int sa_set_dynamic_tiles(int32_t on) {
  g_dynamic_tiles = on ? 1 : 0;
  return 0;
}

namespace {

template <bool A_KM, bool B_KM, bool SWAP>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][A tile | B tile]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware, bijective remap: the 8 XCDs are dealt blocks round-robin; give each a contiguous run of
  // logical ids so neighbouring tiles (same A row-panel, adjacent B panels) share one L2.
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  // split-K slice is the SLOWEST index (blocks that run together reduce over the same rows and share them in L2);
  // tiles are walked in groups of GM row-panels with m fastest, so the ~64 blocks resident on one XCD cover about
  // 8 x 8 tiles: 8 A panels + 8 B panels stay in its 4 MiB L2 instead of streaming every B panel per tile.
  const int ntiles = p.tiles_m * p.tiles_n;
  const int ks_id = lid / ntiles;
  const int tile = lid - ks_id * ntiles;
  const int GM = p.gm;
  const int group_sz = GM * p.tiles_n;
  const int grp = tile / group_sz, within = tile - grp * group_sz;
  const int gm = min(GM, p.tiles_m - grp * GM);
  const int tm = grp * GM + within % gm, tn = within / gm;
  const int m0 = tm * BM, n0 = tn * BN;

  const int ksteps = (p.K + BK - 1) / BK;
  const int chunk = (ksteps + p.split_k - 1) / p.split_k;
  const int kt_begin = ks_id * chunk;
  const int kt_end = min(ksteps, kt_begin + chunk);
  if (kt_begin >= kt_end) return;  // (only possible for split_k > 1: nothing to add; the deterministic reduce skips empty slices)

  constexpr bool HID = !(A_KM && B_KM);    // kernels with a transposing-read operand hide their LDS-DMA from the compiler (lds_dma16)
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LDS layout: [buffer 0: A tile | B tile][buffer 1: A tile | B tile]
  stage_tile<A_KM, 4, HID>(ra, smem, p.lda, m0, kt_begin * BK, wave, lane);
  stage_tile<B_KM, 4, HID>(rb, smem + TILE_BYTES, p.ldb, n0, kt_begin * BK, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    if (kt + 1 < kt_end) {
      char* nxt = smem + (cur ^ 1) * 2 * TILE_BYTES;
      stage_tile<A_KM, 4, HID>(ra, nxt, p.lda, m0, (kt + 1) * BK, wave, lane);
      stage_tile<B_KM, 4, HID>(rb, nxt + TILE_BYTES, p.ldb, n0, (kt + 1) * BK, wave, lane);
    }
    const char* ta = smem + cur * 2 * TILE_BYTES;
    const char* tb = ta + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = load_frag<A_KM>(ta, wr * 64 + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = load_frag<B_KM>(tb, wc * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (SWAP)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // ---------------------------------------------------------------- epilogue
  const int g = lane >> 4, c = lane & 15;
  if constexpr (!SWAP) {
    // split-K: lane owns rows m = 4g + r of one column n -> 16 consecutive dwords per row per instruction
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wr * 64 + i * 16 + 4 * g + r;
          if (m < p.M && n < p.N) {
            if (p.split_ws) p.split_ws[((int64_t)ks_id * p.M + m) * p.N + n] = p.alpha * acc[i][j][r];
            else atomicAdd(p.out_f32 + (int64_t)m * p.ldo_f32 + n, p.alpha * acc[i][j][r]);
          }
        }
      }
  } else {
    wave_epilogue_64x64(p, acc, m0 + wr * 64, n0 + wc * 64, smem + wave * EPI_BYTES, lane);
  }
}

// =====================================================================================================
// 256 x 256 x 64 tile, 512 threads = 8 waves (2 x 4), each wave 128 x 64 (8 x 4 MFMA tiles, 128 accumulator VGPRs),
// one workgroup per CU (2 x 64 KiB LDS buffers).  Why: a CU pulls L2-resident operands at only ~30 B/clk, so the
// 128^2 kernel (32 KiB of operands per 2.1 MFLOP) is bound by that path long before the MFMA pipe; 256^2 halves
// the bytes per FLOP.  Each operand tile is two of the 128-wide half-tiles above, so staging and fragment reads are
// the same code.  Per K-tile the wave walks four C-quadrants (A-half0 x B-half0, A0 x B1, A1 x B1, A1 x B0) so only
// 12 fragment registers sets are live; the next K-tile's 4 half-tiles are requested at the top of the iteration and
// awaited at its end (one barrier per K-tile).
template <bool A_KM, bool B_KM, bool SPLIT>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][A half0 | A half1 | B half0 | B half1]
  constexpr int BUF = 4 * TILE_BYTES;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;      // wave rows wr*128.., cols wc*64..

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  constexpr int GM = 4;                          // 32 blocks per XCD ~ 4 x 8 tiles in flight
  const int ntiles = p.tiles_m * p.tiles_n;
  const int ks_id = SPLIT ? lid / ntiles : 0;    // split-K slice slowest (co-resident workgroups share their K rows in L2)
  const int tile = lid - ks_id * ntiles;
  const int group_sz = GM * p.tiles_n;
  const int grp = tile / group_sz, within = tile - grp * group_sz;
  const int gm = min(GM, p.tiles_m - grp * GM);
  const int tm = grp * GM + within % gm, tn = within / gm;
  const int m0 = tm * 256, n0 = tn * 256;

  const int ksteps_all = (p.K + BK - 1) / BK;
  const int chunk = (ksteps_all + p.split_k - 1) / p.split_k;
  const int kt0 = SPLIT ? ks_id * chunk : 0;
  const int ksteps = SPLIT ? min(ksteps_all - kt0, chunk) : ksteps_all;
  if (ksteps <= 0) return;
  constexpr bool HID = !(A_KM && B_KM);    // kernels with a transposing-read operand hide their LDS-DMA from the compiler (lds_dma16)
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage_all = [&](char* buf, int k0) {
    stage_tile<A_KM, 8, HID>(ra, buf, p.lda, m0, k0, wave, lane);
    stage_tile<A_KM, 8, HID>(ra, buf + TILE_BYTES, p.lda, m0 + 128, k0, wave, lane);
    stage_tile<B_KM, 8, HID>(rb, buf + 2 * TILE_BYTES, p.ldb, n0, k0, wave, lane);
    stage_tile<B_KM, 8, HID>(rb, buf + 3 * TILE_BYTES, p.ldb, n0 + 128, k0, wave, lane);
  };

  if (p.stagger & 4) { if (wr == 1) __builtin_amdgcn_s_setprio(1); }     // (see gemm256_persist_kernel; not the default here: no gain measured)
  stage_all(smem, kt0 * BK);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int cur = 0;
  for (int kt = 0; kt < ksteps; ++kt) {
    // the second wave row requests its share of the next K-tile mid-step (see gemm256_persist_kernel).  Only with compiler-visible
    // requests: there the mid-step position staggered the drain hipcc puts behind them (lds_dma16) between a SIMD's two waves, 12-14 %
    // on the wgrad shapes; with hidden requests there is no drain and requesting at the top is 3-5 % faster still
    const bool late = !HID && p.stagger && wr == 1;
    if (kt + 1 < ksteps && !late) stage_all(smem + (cur ^ 1) * BUF, (kt0 + kt + 1) * BK);
    const char* ta = smem + cur * BUF + wr * TILE_BYTES;                 // this wave's A half-tile (128 rows)
    const char* tb = smem + cur * BUF + (2 + (wc >> 1)) * TILE_BYTES;    // this wave's B half-tile
    const int bcol = (wc & 1) * 64;
    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    // quadrant (0,0)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[i][ks] = load_frag<A_KM>(ta, i * 16, ks, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb0[j][ks] = load_frag<B_KM>(tb, bcol + j * 16, ks, lane);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (SPLIT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], fb0[j][ks], acc[i][j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j][ks], fa[i][ks], acc[i][j], 0, 0, 0));
    // quadrant (0,1)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fb1[j][ks] = load_frag<B_KM>(tb, bcol + 32 + j * 16, ks, lane);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][2 + j] = (SPLIT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], fb1[j][ks], acc[i][2 + j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j][ks], fa[i][ks], acc[i][2 + j], 0, 0, 0));
    if (kt + 1 < ksteps && late) stage_all(smem + (cur ^ 1) * BUF, (kt0 + kt + 1) * BK);
    // quadrant (1,1)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) fa[i][ks] = load_frag<A_KM>(ta, 64 + i * 16, ks, lane);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = (SPLIT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], fb1[j][ks], acc[4 + i][2 + j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j][ks], fa[i][ks], acc[4 + i][2 + j], 0, 0, 0));
    // quadrant (1,0)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[4 + i][j] = (SPLIT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], fb0[j][ks], acc[4 + i][j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j][ks], fa[i][ks], acc[4 + i][j], 0, 0, 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  const int g = lane >> 4, c = lane & 15;
  if constexpr (SPLIT) {
    // split-K: lane owns rows 4g + r of one column -> 64-byte row segments per atomic instruction
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wc * 64 + j * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wr * 128 + i * 16 + 4 * g + r;
          if (m < p.M && n < p.N) {
            if (p.split_ws) p.split_ws[((int64_t)ks_id * p.M + m) * p.N + n] = p.alpha * acc[i][j][r];
            else atomicAdd(p.out_f32 + (int64_t)m * p.ldo_f32 + n, p.alpha * acc[i][j][r]);
          }
        }
      }
    return;
  }
  // ---------------------------------------------------------------- epilogue (lane: one row m, 4 consecutive n)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wr * 128 + i * 16 + c;
    if (m >= p.M) continue;
    const int64_t orow = p.row_group > 0 ? (int64_t)m + m / p.row_group + 1 : (int64_t)m;
    const int64_t rrow = p.res_mod > 0 ? (int64_t)(m % p.res_mod) : (int64_t)m;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + 4 * g;
      if (n + 3 >= p.N) {
        for (int r = 0; r < 4 && n + r < p.N; ++r) {
          float x = acc[i][j][r] * p.alpha;
          if (p.bias) x += p.bias[n + r];
          if (p.act == 1 || p.act == 3) {
            if (p.aux_out) p.aux_out[(int64_t)m * p.ldaux + n + r] = f2bf(p.act == 3 ? dgelu_f(x) : x);
            x = gelu_f(x);
          } else if (p.act == 2 || p.act == 4) {
            const float h = bf2f(p.aux_in[(int64_t)m * p.ldaux + n + r]);
            x *= p.act == 4 ? h : dgelu_f(h);
          }
          if (p.residual) x += p.residual[rrow * p.ldr + n + r];
          if (p.out_f32) {
            float* o = p.out_f32 + orow * p.ldo_f32 + n + r;
            *o = p.accumulate ? *o + x : x;
          }
          if (p.out_bf16) p.out_bf16[orow * p.ldo_bf16 + n + r] = f2bf(x);
        }
        continue;
      }
      float v[4] = {acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha};
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (p.act == 1 || p.act == 3) {
        if (p.aux_out) {
          bf16x4 h = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
          if (p.act == 3) h = bf16x4{f2bf(dgelu_f(v[0])), f2bf(dgelu_f(v[1])), f2bf(dgelu_f(v[2])), f2bf(dgelu_f(v[3]))};
          *reinterpret_cast<bf16x4*>(p.aux_out + (int64_t)m * p.ldaux + n) = h;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
      } else if (p.act == 2 || p.act == 4) {
        const bf16x4 h = *reinterpret_cast<const bf16x4*>(p.aux_in + (int64_t)m * p.ldaux + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= p.act == 4 ? bf2f(h[r]) : dgelu_f(bf2f(h[r]));
      }
      if (p.residual) {
        const float4 rv = *reinterpret_cast<const float4*>(p.residual + rrow * p.ldr + n);
        v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
      }
      if (p.out_f32) {
        float* o = p.out_f32 + orow * p.ldo_f32 + n;
        if (p.accumulate) {
          const float4 old = *reinterpret_cast<const float4*>(o);
          v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w;
        }
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      }
      if (p.out_bf16) {
        bf16x4 h = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *reinterpret_cast<bf16x4*>(p.out_bf16 + orow * p.ldo_bf16 + n) = h;
      }
    }
  }
}

template <bool A_KM, bool B_KM, bool SPLIT = false>
int launch256(GemmParams p, hipStream_t stream) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_kernel<A_KM, B_KM, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              8 * TILE_BYTES);
    configured = true;
  }
  hipLaunchKernelGGL((gemm256_kernel<A_KM, B_KM, SPLIT>), dim3(p.tiles_m * p.tiles_n * (SPLIT ? p.split_k : 1)), dim3(512), 8 * TILE_BYTES,
                     stream, p);
  SA_LAUNCH_CHECK("sa_gemm_bf16(256)");
  return 0;
}

// =====================================================================================================
// Persistent 256 x 256 x 64 kernel (forward / dgrad): one workgroup per CU walks tiles lid, lid + grid, ...  With a single
// resident workgroup nothing else hides the pipeline fill and the store tail of a 12-iteration (K = 768) tile, so the
// kernel hides them itself: the next tile's first K-tile is requested during the last K-iteration and lands while the
// epilogue runs, and the epilogue stages through the 64 KiB pipeline stage that was just consumed (8 KiB per wave, two
// 64-row passes).
template <bool A_KM, bool B_KM, int EPI = 0>
__global__ __launch_bounds__(512, 2) void gemm256_persist_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][A half0 | A half1 | B half0 | B half1]
  constexpr int BUF = 4 * TILE_BYTES;
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
  const int ksteps = (p.K + BK - 1) / BK;
  constexpr bool HID = !(A_KM && B_KM);    // kernels with a transposing-read operand hide their LDS-DMA from the compiler (lds_dma16)
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);
  auto stage_all = [&](char* buf, int m0, int n0, int k0) {
    stage_tile<A_KM, 8, HID>(ra, buf, p.lda, m0, k0, wave, lane);
    stage_tile<A_KM, 8, HID>(ra, buf + TILE_BYTES, p.lda, m0 + 128, k0, wave, lane);
    stage_tile<B_KM, 8, HID>(rb, buf + 2 * TILE_BYTES, p.ldb, n0, k0, wave, lane);
    stage_tile<B_KM, 8, HID>(rb, buf + 3 * TILE_BYTES, p.ldb, n0 + 128, k0, wave, lane);
  };

  int t = lid;
  if (t >= ntiles) return;
  // static priority for the second-dispatched wave row (waves 4..7): both at priority 0, the older waves win every issue arbitration and
  // the younger ones' MFMA segments take twice as long (cycle stamps of scripts/microbench/gemm_dephase_prof.hip: 2 300 - 2 800 against
  // 1 300 cycles); MI355X_MICROARCH.md, "Two waves per SIMD", item 4
  // -- measured (scripts/bench_gemm.py, A/B/A/B on one box): qkv forward 251-259 -> 240-241 us, fc1 forward 401-412 -> 384-386,
  // fc2 dgrad 378-380 -> 357-360, proj dgrad 76-79 -> 71.5; default on (SA_GEMM_STAGGER bit 1)
  if (p.stagger & 2) { if (wr == 1) __builtin_amdgcn_s_setprio(1); }
  int m0, n0;
  coords(t, m0, n0);
  stage_all(smem, m0, n0, 0);
  int cur = 0;
  // dynamic hand-out (p.tile_counter; the host sets it only when ksteps >= 2): lane 0 of wave 0 draws the NEXT tile's ticket at the
  // top of a tile, the value crosses to the other waves through one LDS word behind the first K-step's barrier, and is first needed
  // in the last K-step (prefetch of the next tile's first K-tile)
  const bool dyn = p.tile_counter != nullptr;
  int* const ticket_word = reinterpret_cast<int*>(smem + 2 * BUF);
  while (true) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int tnext = t + nwg;
    int ticket = 0;
    if (dyn && threadIdx.x == 0) ticket = atomicAdd(p.tile_counter, 1);
    bool has_next = tnext < ntiles;
    int m0n = 0, n0n = 0;
    if (!dyn && has_next) coords(tnext, m0n, n0n);

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    for (int kt = 0; kt < ksteps; ++kt) {
      const bool more = kt + 1 < ksteps;
      // the second wave row requests its share of the next K-tile mid-step instead of at the top: halves the burst that blocks
      // every wave on the memory pipe's issue queue right after the barrier (measured 2-4 % on the forward shapes)
      const bool late = (p.stagger & 1) && wr == 1;
      if (!late) {
        if (more) stage_all(smem + (cur ^ 1) * BUF, m0, n0, (kt + 1) * BK);
        else if (has_next) stage_all(smem + (cur ^ 1) * BUF, m0n, n0n, 0);
      }
      const char* ta = smem + cur * BUF + wr * TILE_BYTES;
      const char* tb = smem + cur * BUF + (2 + (wc >> 1)) * TILE_BYTES;
      const int bcol = (wc & 1) * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[i][ks] = load_frag<A_KM>(ta, i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fb0[j][ks] = load_frag<B_KM>(tb, bcol + j * 16, ks, lane);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb0[j][ks], fa[i][ks], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fb1[j][ks] = load_frag<B_KM>(tb, bcol + 32 + j * 16, ks, lane);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb1[j][ks], fa[i][ks], acc[i][2 + j], 0, 0, 0);
      if (late) {
        if (more) stage_all(smem + (cur ^ 1) * BUF, m0, n0, (kt + 1) * BK);
        else if (has_next) stage_all(smem + (cur ^ 1) * BUF, m0n, n0n, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[i][ks] = load_frag<A_KM>(ta, 64 + i * 16, ks, lane);
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
      if (more) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (dyn && kt == 0) {
          if (threadIdx.x == 0) *ticket_word = nwg + ticket;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (dyn && kt == 0) {
          tnext = __builtin_amdgcn_readfirstlane(*ticket_word);
          has_next = tnext < ntiles;
          if (has_next) coords(tnext, m0n, n0n);
        }
        cur ^= 1;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // raw: the next tile's LDS-DMA stays in flight
    char* wl = smem + cur * BUF + wave * 8192;
    if constexpr (EPI == 4 || EPI == 5) {
      // (as in the ring kernel) the wave's whole aux_in block is requested before the first store
      const int nb = n0 + wc * 64;
      uint4 auxv[16];
      {   // buffer loads: one per-lane offset, the row block in an SGPR, rows past M read as zeros (see staged_store_bf16)
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.aux_in, nb < p.N ? (uint32_t)((((int64_t)p.M - 1) * p.ldaux + p.N) * 2) : 0u);
        const uint32_t ld2 = (uint32_t)p.ldaux * 2u;
        const uint32_t voff = (uint32_t)(lane >> 3) * ld2 + (uint32_t)(lane & 7) * 16u;
        const uint32_t soff = (uint32_t)(m0 + wr * 128) * ld2 + (uint32_t)nb * 2u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, voff, soff + (uint32_t)q * 8u * ld2, 0);
          auxv[q] = __builtin_bit_cast(uint4, v);
        }
      }
      wave_epilogue_compact<true, EPI>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[0]), m0 + wr * 128, nb, wl, lane, nullptr, nullptr, &auxv[0]);
      wave_epilogue_compact<true, EPI>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[4]), m0 + wr * 128 + 64, nb, wl, lane, nullptr, nullptr, &auxv[8]);
    } else if constexpr (EPI != 0) {
      float4 bias4[4];
      const bool has_bias = p.bias != nullptr && n0 + wc * 64 < p.N;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bias4[j] = has_bias ? *reinterpret_cast<const float4*>(p.bias + n0 + wc * 64 + j * 16 + 4 * (lane >> 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
      wave_epilogue_compact<true, EPI>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[0]), m0 + wr * 128, n0 + wc * 64, wl, lane, nullptr, bias4);
      wave_epilogue_compact<true, EPI>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[4]), m0 + wr * 128 + 64, n0 + wc * 64, wl, lane, nullptr, bias4);
    } else {
      wave_epilogue_64x64<true>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[0]), m0 + wr * 128, n0 + wc * 64, wl, lane);
      wave_epilogue_64x64<true>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[4]), m0 + wr * 128 + 64, n0 + wc * 64, wl, lane);
    }
    if (!has_next) break;
    t = tnext; m0 = m0n; n0 = n0n;
    cur ^= 1;
  }
}

constexpr int PERSIST_LDS = 8 * TILE_BYTES + 16;     // two 64 KiB stages + the ticket word of the dynamic tile hand-out

template <bool A_KM, bool B_KM>
int launch256_persist(GemmParams p, hipStream_t stream) {
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
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_persist_kernel<A_KM, B_KM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              PERSIST_LDS);
  }
  const int ntiles = p.tiles_m * p.tiles_n;
  const dim3 grid(ntiles < budget_slots(slots) ? ntiles : budget_slots(slots));
  p.tile_counter = ((p.K + BK - 1) / BK >= 2 && ntiles > (int)grid.x) ? sagemm::next_tile_counter(stream) : nullptr;
  if constexpr (A_KM && B_KM) {            // the forward layout gets kernels specialised on the compact epilogues 1..3
#define SA_EPI_CASE(E)                                                                                                     \
  case E: {                                                                                                                \
    static bool cfg = false;                                                                                               \
    if (!cfg) {                                                                                                            \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_persist_kernel<true, true, E>),                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, PERSIST_LDS);                                  \
      cfg = true;                                                                                                          \
    }                                                                                                                      \
    hipLaunchKernelGGL((gemm256_persist_kernel<true, true, E>), grid, dim3(512), PERSIST_LDS, stream, p);                \
    SA_LAUNCH_CHECK("sa_gemm_bf16(256 persistent, compact epilogue)");                                                     \
    return 0;                                                                                                              \
  }
    switch (p.epi_kind) { SA_EPI_CASE(1) SA_EPI_CASE(3) SA_EPI_CASE(5) SA_EPI_CASE(6) SA_EPI_CASE(8) default: break; }   // kind 2 (separate erf-GELU pass) spills: general path
#undef SA_EPI_CASE
  }
  hipLaunchKernelGGL((gemm256_persist_kernel<A_KM, B_KM>), grid, dim3(512), PERSIST_LDS, stream, p);
  SA_LAUNCH_CHECK("sa_gemm_bf16(256 persistent)");
  return 0;
}


// =====================================================================================================
// Persistent 256 x 256 x 64 kernel over a FIVE-slot ring of 32 KiB half-stages ("mode 8"; forward / dgrad and split-K wgrad).
// Measured with scripts/microbench/dma_bw.hip: one workgroup per CU streams operand tiles L2 -> LDS at ~23 B/clk/CU with one
// 64 KiB stage in flight and ~34 B/clk/CU with two -- the 256^2 tile needs 32 B/clk/CU at MFMA peak, so bytes in flight are
// the limiter, and 160 KiB of LDS cannot hold three 64 KiB stages.  Splitting a stage into its A half and B half lets the
// whole LDS work as a ring: two halves are being read, THREE are in flight (the K-tile after next's A half included).
//   slot(h) = h mod 5 for the h-th half-tile of this workgroup's K-tile stream (A_0 B_0 A_1 B_1 ...), across tile borders;
//   end of K-step g: s_waitcnt vmcnt(4) leaves only A_{g+2} outstanding, barrier, then B_{g+2} and A_{g+3} are requested
//   into the two slots step g has just finished reading.
// Requests past the end of the stream go out of range (zero fill, no traffic) so the outstanding count stays exact; the
// K-step right after an epilogue waits vmcnt(0) because stores / atomics sit between the loads in the counter.
template <bool A_KM, bool B_KM, bool SPLIT, int EPI = 0>
__global__ __launch_bounds__(512, 2) void gemm256_ring_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HALF = 2 * TILE_BYTES;               // 32 KiB: [rows 0..127 | rows 128..255] of one operand
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int ntiles = p.tiles_m * p.tiles_n;
  const int nunits = ntiles * (SPLIT ? p.split_k : 1);
  const int GM = p.gm256;
  const int group_sz = GM * p.tiles_n;
  const int ksteps_all = (p.K + BK - 1) / BK;
  const int chunk = SPLIT ? (ksteps_all + p.split_k - 1) / p.split_k : ksteps_all;
  // unit -> (m0, n0, first K-step, number of K-steps); split-K slice slowest so co-resident workgroups share K rows in L2
  auto unit = [&](int u, int& m0, int& n0, int& kt0, int& nk) {
    const int ks_id = SPLIT ? u / ntiles : 0;
    const int t = u - ks_id * ntiles;
    const int grp = t / group_sz, within = t - grp * group_sz;
    const int gm = min(GM, p.tiles_m - grp * GM);
    m0 = (grp * GM + within % gm) * 256;
    n0 = (within / gm) * 256;
    kt0 = ks_id * chunk;
    nk = min(ksteps_all - kt0, chunk);
  };
  constexpr bool HID = !(A_KM && B_KM);    // kernels with a transposing-read operand hide their LDS-DMA from the compiler (lds_dma16)
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);

  int u = lid;
  if (u >= nunits) return;
  int m0, n0, kt0, nk;
  unit(u, m0, n0, kt0, nk);

  // ---- request cursor (runs 2.5 K-tiles ahead of the compute)
  int ru = u, rm0 = m0, rn0 = n0, rk = kt0, rkend = kt0 + nk, rside = 0, rslot = 0;
  struct HalfReq { int row0, k0, slot, valid; };      // one pending half-tile request (A if issued on an even turn, else B)
  auto next_half = [&]() {                             // advance the cursor, return what to fetch
    HalfReq h;
    h.valid = ru < nunits;
    h.slot = rslot;
    h.k0 = rk * BK;
    h.row0 = rside == 0 ? rm0 : rn0;
    if (rside == 1) {
      if (h.valid && ++rk == rkend) {
        ru += nwg;
        if (ru < nunits) {
          int nk2;
          unit(ru, rm0, rn0, rk, nk2);
          rkend = rk + nk2;
        }
      }
    }
    rside ^= 1;
    rslot = rslot == 4 ? 0 : rslot + 1;
    return h;
  };
  // piece i (0..3) of a half: this wave's instruction (i & 1) of 16-KiB tile (i >> 1)
  auto issue_a = [&](const HalfReq& h, int i) {
    stage_one_v<A_KM, HID>(ra, smem + h.slot * HALF + (i >> 1) * TILE_BYTES, p.lda, h.row0 + (i >> 1) * 128, h.k0, wave * 2 + (i & 1), lane, h.valid);
  };
  auto issue_b = [&](const HalfReq& h, int i) {
    stage_one_v<B_KM, HID>(rb, smem + h.slot * HALF + (i >> 1) * TILE_BYTES, p.ldb, h.row0 + (i >> 1) * 128, h.k0, wave * 2 + (i & 1), lane, h.valid);
  };
  auto request_half = [&]() {                          // burst form (prologue, after an epilogue)
    const bool is_a = rside == 0;
    const HalfReq h = next_half();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (is_a) issue_a(h, i); else issue_b(h, i);
    }
  };
#pragma unroll 1
  for (int i = 0; i < 5; ++i) request_half();
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");     // A_0, B_0 landed; A_1, B_1, A_2 in flight
  __builtin_amdgcn_s_barrier();

  int ca = 0, cb = 1;                                    // slots of the K-tile being consumed
  HalfReq pb = {0, 0, 0, 0}, pa = {0, 0, 0, 0};         // requests to trickle out during the current K-step (B half first)
  bool pending = false;
  bool fresh = false;
  const int bcol = (wc & 1) * 64;
#define SA_MM(FA, FB, ACC) (SPLIT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA, FB, ACC, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB, FA, ACC, 0, 0, 0))
  while (true) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int ea = 0, eb = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const char* ta = smem + ca * HALF + wr * TILE_BYTES;
      const char* tb = smem + cb * HALF + (wc >> 1) * TILE_BYTES;
      bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
      // request phases: wave row 0 sends its eight requests during Q0/Q1 (two per 8 MFMAs), row 1 during Q2/Q3 -- each SIMD's two
      // waves are then never both stalled on the memory pipe's issue queue (3-8 % over the uniform one-per-8-MFMAs trickle)
      const bool phased = p.ring_phase != 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[i][ks] = load_frag<A_KM>(ta, i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fb0[j][ks] = load_frag<B_KM>(tb, bcol + j * 16, ks, lane);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = SA_MM(fa[i][ks], fb0[j][ks], acc[i][j]);
            if (i == 3 && j == 1 && pending) {                          // requests trickle out every 8 MFMAs: a burst of 8 stalls
              if (!phased) issue_b(pb, ks);
              else if (wr == 0) { issue_b(pb, 2 * ks); issue_b(pb, 2 * ks + 1); }
            }
          }                                                              // every wave on the memory pipe's issue queue
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fb1[j][ks] = load_frag<B_KM>(tb, bcol + 32 + j * 16, ks, lane);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][2 + j] = SA_MM(fa[i][ks], fb1[j][ks], acc[i][2 + j]);
            if (i == 3 && j == 1 && pending) {
              if (!phased) issue_b(pb, 2 + ks);
              else if (wr == 0) { issue_a(pa, 2 * ks); issue_a(pa, 2 * ks + 1); }
            }
          }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[i][ks] = load_frag<A_KM>(ta, 64 + i * 16, ks, lane);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[4 + i][2 + j] = SA_MM(fa[i][ks], fb1[j][ks], acc[4 + i][2 + j]);
            if (i == 3 && j == 1 && pending) {
              if (!phased) issue_a(pa, ks);
              else if (wr == 1) { issue_b(pb, 2 * ks); issue_b(pb, 2 * ks + 1); }
            }
          }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[4 + i][j] = SA_MM(fa[i][ks], fb0[j][ks], acc[4 + i][j]);
            if (i == 3 && j == 1 && pending) {
              if (!phased) issue_a(pa, 2 + ks);
              else if (wr == 1) { issue_a(pa, 2 * ks); issue_a(pa, 2 * ks + 1); }
            }
          }
      // ---- K-step barrier: the next K-tile has landed, this one is no longer read
      if (fresh) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      pending = false;
      if (kt + 1 < nk) {
        pb = next_half();                               // B_{g+2}: due at the end of the next step -> first half of it
        pa = next_half();                               // A_{g+3}: a step of slack -> second half
        pending = true;
        fresh = false;
      } else {
        ea = ca; eb = cb;                               // the unit's epilogue borrows these two slots first
      }
      ca = ca >= 3 ? ca - 3 : ca + 2;
      cb = cb >= 3 ? cb - 3 : cb + 2;
    }
    if constexpr (SPLIT) {
      const int g = lane >> 4, c = lane & 15;           // lane owns rows 4g + r of one column -> 64-byte row segments per atomic
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + wc * 64 + j * 16 + c;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = m0 + wr * 128 + i * 16 + 4 * g + r;
            if (m < p.M && n < p.N) atomicAdd(p.out_f32 + (int64_t)m * p.ldo_f32 + n, p.alpha * acc[i][j][r]);
          }
        }
    } else {
      char* wl = smem + (wave < 4 ? ea * HALF + wave * 8192 : eb * HALF + (wave - 4) * 8192);
      if constexpr (EPI == 4 || EPI == 5) {
        // the wave's whole aux_in block (128 rows x 64 columns) is requested before the first store: a load issued after a store
        // sits behind it in the in-order vmcnt queue, and waiting for it would drain the first half's stores (an HBM round trip)
        const int nb = n0 + wc * 64;
        uint4 auxv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = m0 + wr * 128 + q * 8 + (lane >> 3);
          auxv[q] = (m < p.M && nb < p.N) ? *reinterpret_cast<const uint4*>(p.aux_in + (int64_t)m * p.ldaux + nb + (lane & 7) * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
        wave_epilogue_compact<true, EPI>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[0]), m0 + wr * 128, nb, wl, lane, nullptr, nullptr, &auxv[0]);
        wave_epilogue_compact<true, EPI>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[4]), m0 + wr * 128 + 64, nb, wl, lane, nullptr, nullptr, &auxv[8]);
      } else if constexpr (EPI != 0) {
        wave_epilogue_compact<true, EPI>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[0]), m0 + wr * 128, n0 + wc * 64, wl, lane);
        wave_epilogue_compact<true, EPI>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[4]), m0 + wr * 128 + 64, n0 + wc * 64, wl, lane);
      } else {
        wave_epilogue_64x64<true>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[0]), m0 + wr * 128, n0 + wc * 64, wl, lane);
        wave_epilogue_64x64<true>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[4]), m0 + wr * 128 + 64, n0 + wc * 64, wl, lane);
      }
    }
    u += nwg;
    if (u >= nunits) break;
    unit(u, m0, n0, kt0, nk);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave is done with its epilogue scratch
    request_half();
    request_half();
    fresh = true;
  }
#undef SA_MM
}

template <bool A_KM, bool B_KM, bool SPLIT>
int launch256_ring(GemmParams p, hipStream_t stream) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  if (SPLIT) {                                          // no empty K slices: the kernel's two cursors must agree on the unit list
    const int ksteps_all = (p.K + BK - 1) / BK;
    const int chunk = (ksteps_all + p.split_k - 1) / p.split_k;
    p.split_k = (ksteps_all + chunk - 1) / chunk;
  } else {
    p.split_k = 1;
  }
  static int slots = 0;
  if (slots == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      sa_set_error("sa_gemm_bf16: cannot query the device");
      return 2;
    }
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_ring_kernel<A_KM, B_KM, SPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            10 * TILE_BYTES) != hipSuccess) {
      sa_set_error("sa_gemm_bf16: 160 KiB of LDS per workgroup refused");
      return 2;
    }
    slots = prop.multiProcessorCount;
  }
  const int nunits = p.tiles_m * p.tiles_n * p.split_k;
  const dim3 grid(nunits < budget_slots(slots) ? nunits : budget_slots(slots));
  if constexpr (A_KM && !B_KM && !SPLIT) {  // the dgrad layout gets kernels specialised on the compact epilogues 1 and 4
#define SA_EPI_CASE(E)                                                                                                     \
  case E: {                                                                                                                \
    static bool cfg = false;                                                                                               \
    if (!cfg) {                                                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_ring_kernel<true, false, false, E>),                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 10 * TILE_BYTES) != hipSuccess) {                \
        sa_set_error("sa_gemm_bf16: 160 KiB of LDS per workgroup refused");                                                \
        return 2;                                                                                                          \
      }                                                                                                                    \
      cfg = true;                                                                                                          \
    }                                                                                                                      \
    hipLaunchKernelGGL((gemm256_ring_kernel<true, false, false, E>), grid, dim3(512), 10 * TILE_BYTES, stream, p);          \
    SA_LAUNCH_CHECK("sa_gemm_bf16(256 ring, compact epilogue)");                                                           \
    return 0;                                                                                                              \
  }
    switch (p.epi_kind) { SA_EPI_CASE(1) SA_EPI_CASE(4) SA_EPI_CASE(5) default: break; }
#undef SA_EPI_CASE
  }
  hipLaunchKernelGGL((gemm256_ring_kernel<A_KM, B_KM, SPLIT>), grid, dim3(512), 10 * TILE_BYTES, stream, p);
  SA_LAUNCH_CHECK("sa_gemm_bf16(256 ring)");
  return 0;
}


template <bool A_KM, bool B_KM, bool SWAP>
int launch(const GemmParams& p, hipStream_t stream) {
  const int nwg = p.tiles_m * p.tiles_n * p.split_k;
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<A_KM, B_KM, SWAP>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    configured = true;
  }
  hipLaunchKernelGGL((gemm_kernel<A_KM, B_KM, SWAP>), dim3(nwg), dim3(NTHREADS), 4 * TILE_BYTES, stream, p);
  SA_LAUNCH_CHECK("sa_gemm_bf16");
  return 0;
}

// ---------------------------------------------------------------- small helpers
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      const float4 a = *reinterpret_cast<const float4*>(src + i);
      const float4 b = *reinterpret_cast<const float4*>(src + i + 4);
      bf16x8 o = {f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w), f2bf(b.x), f2bf(b.y), f2bf(b.z), f2bf(b.w)};
      *reinterpret_cast<bf16x8*>(dst + i) = o;
    } else {
      for (int64_t k = i; k < n; ++k) dst[k] = f2bf(src[k]);
    }
  }
}

// column sums: block handles 64 columns x a slab of rows; 256 threads = 4 row-phases x 64 columns
__global__ void colsum_bf16_kernel(const bf16_t* __restrict__ x, int64_t ld, int M, int N, float* __restrict__ out, int rows_per_block,
                                   float* __restrict__ ws, int64_t range_stride) {
  __shared__ float red[4][64];
  x += blockIdx.z * range_stride; out += blockIdx.z * range_stride;          // column range z of the same matrix (q / v bias gradients)
  if (ws) ws += (int64_t)blockIdx.z * gridDim.y * N;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ph = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (col < N)
    for (int r = r0 + ph; r < r1; r += 4) s += bf2f(x[(int64_t)r * ld + col]);
  red[ph][threadIdx.x & 63] = s;
  __syncthreads();
  if (ph == 0 && col < N) {
    const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (ws) ws[(int64_t)blockIdx.y * N + col] = t;          // ordered form: row-slab partials, summed in slab order by colsum_ws_reduce_kernel
    else atomicAdd(out + col, t);
  }
}

// vectorised column sums: a block covers 256 columns (32 lanes x 8 bf16 = 512 contiguous bytes per row) x 8 row lanes,
// so every load is a full-line 16-byte access; partial sums meet in LDS and leave as one atomic per column per block -- or, with a
// workspace, as one plain store per column per block (ws[row slab][N]; a second launch adds the slabs in order: bit-reproducible).
__global__ __launch_bounds__(256) void colsum_bf16_vec_kernel(const bf16_t* __restrict__ x, int64_t ld, int M, int N, float* __restrict__ out,
                                                              int rows_per_block, float* __restrict__ ws, int64_t range_stride) {
  __shared__ float red[8][256 + 8];
  x += blockIdx.z * range_stride; out += blockIdx.z * range_stride;          // column range z of the same matrix (q / v bias gradients)
  if (ws) ws += (int64_t)blockIdx.z * gridDim.y * N;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int col = blockIdx.x * 256 + tx * 8;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < N) {
    int r = r0 + ty;
    for (; r + 24 < r1; r += 32) {                     // four independent 16-byte loads in flight per thread
      bf16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(x + (int64_t)(r + 8 * u) * ld + col);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += bf2f(v[u][k]);
    }
    for (; r < r1; r += 8) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (int64_t)r * ld + col);
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] += bf2f(v[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[ty][tx * 8 + k] = a[k];
  __syncthreads();
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < N) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][threadIdx.x];
    if (ws) ws[(int64_t)blockIdx.y * N + c] = s;
    else atomicAdd(out + c, s);
  }
}

}  // namespace

namespace {
// out[n] += sum over the rows of ws[rows][N]; 16 waves split the rows, lanes are columns
__global__ __launch_bounds__(1024) void colsum_ws_reduce_kernel(const float* __restrict__ ws, int rows, int N, float* __restrict__ out,
                                                                int assign = 0, int64_t range_stride = 0) {
  __shared__ float red[16][64];
  ws += (int64_t)blockIdx.y * rows * N; out += blockIdx.y * range_stride;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (col < N) {
    const float* base = ws + col;
    int r = wave;
    for (; r + 7 * 16 < rows; r += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(int64_t)(r + u * 16) * N];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; r < rows; r += 16) acc += base[(int64_t)r * N];
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && col < N) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][lane];
    out[col] = assign ? t : out[col] + t;
  }
}

int gemm_dispatch(const SaGemmArgs* a, hipStream_t stream);
}  // namespace

extern "C" int sa_set_cu_budget(int32_t cus) {
  SA_CHECK_ARG(cus >= 0, "sa_set_cu_budget: negative CU count");
  sagemm::g_cu_budget = cus;
  return 0;
}

extern "C" int64_t sa_gemm_colsum_workspace_bytes(int32_t M, int32_t N) { return (int64_t)((M + 63) / 64) * N * (int64_t)sizeof(float); }

namespace {
// deterministic split-K, second launch: out[m][n] += ws[0][m][n] + ws[1][m][n] + ... in slice order (one thread per 4 columns)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int nslice, int M, int N, float* __restrict__ out, int64_t ldo) {
  const int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n4 = N >> 2;
  if (i4 >= (int64_t)M * n4) return;
  const int m = (int)(i4 / n4), n = (int)(i4 - (int64_t)m * n4) * 4;
  float4 acc = *reinterpret_cast<const float4*>(ws + (int64_t)m * N + n);
  for (int s = 1; s < nslice; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(ws + ((int64_t)s * M + m) * N + n);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  float4* o = reinterpret_cast<float4*>(out + (int64_t)m * ldo + n);
  float4 t = *o;
  t.x += acc.x; t.y += acc.y; t.z += acc.z; t.w += acc.w;
  *o = t;
}

// the same sum for many thin slices (a narrow weight gradient split 40 - 128 ways): the NW waves of a block take every NW-th slice of
// 64 float4 columns, their partials meet in LDS and are added in wave order -- a fixed order, and NW x the loads in flight
template <int NW>
__global__ __launch_bounds__(64 * NW) void splitk_reduce_wide_kernel(const float* __restrict__ ws, int nslice, int M, int N, float* __restrict__ out, int64_t ldo) {
  __shared__ float4 part[NW][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i4 = (int64_t)blockIdx.x * 64 + lane;
  const int n4 = N >> 2;
  const bool ok = i4 < (int64_t)M * n4;
  const int m = ok ? (int)(i4 / n4) : 0, n = ok ? (int)(i4 - (int64_t)m * n4) * 4 : 0;
  const int64_t slice = (int64_t)M * N;
  const float* src = ws + (int64_t)m * N + n;
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    int s = wave;
    for (; s + 3 * NW < nslice; s += 4 * NW) {
      const float4 v0 = *reinterpret_cast<const float4*>(src + s * slice), v1 = *reinterpret_cast<const float4*>(src + (s + NW) * slice),
                   v2 = *reinterpret_cast<const float4*>(src + (s + 2 * NW) * slice), v3 = *reinterpret_cast<const float4*>(src + (s + 3 * NW) * slice);
      acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
      acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
      acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
      acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
    }
    for (; s < nslice; s += NW) {
      const float4 v = *reinterpret_cast<const float4*>(src + s * slice);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave != 0 || !ok) return;
  float4* o = reinterpret_cast<float4*>(out + (int64_t)m * ldo + n);
  float4 t = *o;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const float4 v = part[w][lane];
    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
  }
  *o = t;
}
}  // namespace

extern "C" int64_t sa_gemm_splitk_workspace_bytes(int32_t M, int32_t N, int32_t split_k) {
  return (M > 0 && N > 0 && split_k > 1) ? (int64_t)split_k * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int sa_gemm_bf16(const SaGemmArgs* a, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SA_CHECK_ARG(a != nullptr, "sa_gemm_bf16: null args");
  if (a->colsum_out) {
    SA_CHECK_ARG(a->colsum_ws && a->split_k == 1 && a->N % 64 == 0, "sa_gemm_bf16: colsum_out needs colsum_ws, split_k == 1 and N %% 64 == 0");
    SA_CHECK_ARG((!a->out_bf16 || (a->ldo_bf16 % 8 == 0 && ((uintptr_t)a->out_bf16 & 15) == 0)) &&
                     (!a->aux_out || (a->ldaux % 8 == 0 && ((uintptr_t)a->aux_out & 15) == 0)),
                 "sa_gemm_bf16: colsum_out needs 16-byte aligned bf16 outputs with leading dimensions that are multiples of 8");
  }
  if (a->splitk_ws)
    SA_CHECK_ARG(a->split_k > 1 && a->N % 4 == 0 && ((uintptr_t)a->splitk_ws & 15) == 0, "sa_gemm_bf16: splitk_ws needs split_k > 1, N %% 4 == 0 and a 16-byte aligned workspace");
  const int rc = gemm_dispatch(a, stream);
  if (rc == 0 && a->splitk_ws) {
    const int64_t n4 = (int64_t)a->M * (a->N / 4);
    const int ksteps = (a->K + BK - 1) / BK, chunk = (ksteps + a->split_k - 1) / a->split_k;
    const int nslice = (ksteps + chunk - 1) / chunk;         // trailing slices with an empty K range wrote nothing
    if (nslice >= 32)
      hipLaunchKernelGGL(splitk_reduce_wide_kernel<8>, dim3((unsigned)((n4 + 63) / 64)), dim3(512), 0, stream, a->splitk_ws, nslice, a->M, a->N, a->out_f32, a->ldo_f32);
    else if (nslice >= 16)
      hipLaunchKernelGGL(splitk_reduce_wide_kernel<4>, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, stream, a->splitk_ws, nslice, a->M, a->N, a->out_f32, a->ldo_f32);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, a->splitk_ws, nslice, a->M, a->N, a->out_f32, a->ldo_f32);
    SA_LAUNCH_CHECK("sa_gemm_bf16(split-K reduce)");
  }
  if (rc != 0 || !a->colsum_out) return rc;
  hipLaunchKernelGGL(colsum_ws_reduce_kernel, dim3(a->N / 64), dim3(1024), 0, stream, a->colsum_ws, (a->M + 63) / 64, a->N, a->colsum_out, 0);
  SA_LAUNCH_CHECK("sa_gemm_bf16(colsum reduce)");
  return 0;
}

namespace {
int gemm_dispatch(const SaGemmArgs* a, hipStream_t stream) {
  SA_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0, "sa_gemm_bf16: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
  SA_CHECK_ARG(a->A && a->B, "sa_gemm_bf16: null operand");
  SA_CHECK_ARG(a->out_f32 || a->out_bf16, "sa_gemm_bf16: no output");
  SA_CHECK_ARG(a->lda % 8 == 0 && a->ldb % 8 == 0, "sa_gemm_bf16: lda/ldb must be multiples of 8 elements (16-byte rows)");
  SA_CHECK_ARG(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0, "sa_gemm_bf16: operands must be 16-byte aligned");
  if (a->a_kmajor || a->b_kmajor)
    SA_CHECK_ARG(a->K % BK == 0, "sa_gemm_bf16: K=%d must be a multiple of %d when an operand is k-major", a->K, BK);
  SA_CHECK_ARG(a->split_k >= 1, "sa_gemm_bf16: split_k must be >= 1");
  if (a->split_k > 1)
    SA_CHECK_ARG(a->out_f32 && !a->out_bf16 && !a->bias && !a->act && !a->residual && !a->row_group,
                 "sa_gemm_bf16: split_k > 1 supports only alpha and fp32 atomic accumulation");
  SA_CHECK_ARG(a->act >= 0 && a->act <= 4, "sa_gemm_bf16: act must be 0..4");
  if (a->act == 2 || a->act == 4) SA_CHECK_ARG(a->aux_in != nullptr, "sa_gemm_bf16: act=2/4 needs aux_in");
  if (a->act == 3) SA_CHECK_ARG(a->aux_out != nullptr, "sa_gemm_bf16: act=3 needs aux_out");
  if (a->out_f32) SA_CHECK_ARG(a->ldo_f32 % 4 == 0 && ((uintptr_t)a->out_f32 & 15) == 0, "sa_gemm_bf16: fp32 output must be 16-byte aligned rows");
  if (a->out_bf16) SA_CHECK_ARG(a->ldo_bf16 % 4 == 0 && ((uintptr_t)a->out_bf16 & 7) == 0, "sa_gemm_bf16: bf16 output rows must be 8-byte aligned");
  if (a->residual) SA_CHECK_ARG(a->ldr % 4 == 0 && ((uintptr_t)a->residual & 15) == 0, "sa_gemm_bf16: residual must be 16-byte aligned rows");
  if (a->bias) SA_CHECK_ARG(((uintptr_t)a->bias & 15) == 0, "sa_gemm_bf16: bias must be 16-byte aligned");
  if (a->aux_in || a->aux_out) SA_CHECK_ARG(a->ldaux % 4 == 0, "sa_gemm_bf16: ldaux must be a multiple of 4");

  GemmParams p;
  p.A = (const char*)a->A; p.B = (const char*)a->B;
  // extents in bytes: last row only spans its valid elements, so anything past it is out of range (zero fill)
  const int64_t a_rows = a->a_kmajor ? a->M : a->K, a_cols = a->a_kmajor ? a->K : a->M;
  const int64_t b_rows = a->b_kmajor ? a->N : a->K, b_cols = a->b_kmajor ? a->K : a->N;
  const int64_t a_bytes = ((a_rows - 1) * a->lda + a_cols) * 2, b_bytes = ((b_rows - 1) * a->ldb + b_cols) * 2;
  const int64_t lim = (int64_t)1 << 32;
  SA_CHECK_ARG((a_rows + 2 * BM) * a->lda * 2 < lim && (b_rows + 2 * BM) * a->ldb * 2 < lim,
               "sa_gemm_bf16: operand larger than the 4 GiB buffer-descriptor range");
  p.a_bytes = (uint32_t)a_bytes; p.b_bytes = (uint32_t)b_bytes;
  p.lda = (int)a->lda; p.ldb = (int)a->ldb;
  p.M = a->M; p.N = a->N; p.K = a->K; p.alpha = a->alpha;
  p.bias = a->bias; p.act = a->act;
  p.aux_in = (const bf16_t*)a->aux_in; p.aux_out = (bf16_t*)a->aux_out; p.ldaux = a->ldaux;
  p.residual = a->residual; p.ldr = a->ldr; p.res_mod = a->res_mod;
  p.out_f32 = a->out_f32; p.ldo_f32 = a->ldo_f32;
  p.out_bf16 = (bf16_t*)a->out_bf16; p.ldo_bf16 = a->ldo_bf16;
  p.row_group = a->row_group; p.split_k = a->split_k; p.accumulate = a->accumulate;
  p.colsum_ws = a->colsum_out ? a->colsum_ws : nullptr;
  p.split_ws = a->split_k > 1 ? a->splitk_ws : nullptr;
  p.tile_counter = nullptr;
  p.asum_out = a->asum_out; p.asum_ws = a->asum_ws; p.asum_lo = a->asum_skip_lo; p.asum_hi = a->asum_skip_hi;
  if (a->asum_out) {
    static const char* ws256 = getenv("SA_GEMM_WGRAD_STREAM256");
    const bool stream_path = a->split_k > 1 && !a->a_kmajor && !a->b_kmajor &&
                             (a->tile256 == 2 || (a->tile256 == 1 && !(ws256 && ws256[0] == '0') && (!a->splitk_ws || a->N % 4 == 0)));
    SA_CHECK_ARG(stream_path, "sa_gemm_bf16: asum_out is served by the streaming split-K kernels only (k-strided operands, split_k > 1, tile256 = 2 or 1)");
    SA_CHECK_ARG((a->asum_ws != nullptr) == (a->splitk_ws != nullptr), "sa_gemm_bf16: asum_ws goes with splitk_ws (both or neither)");
    SA_CHECK_ARG(a->asum_skip_lo >= 0 && a->asum_skip_hi >= a->asum_skip_lo, "sa_gemm_bf16: bad asum_skip range");
  }
  static const char* es_env = getenv("SA_GEMM_EPI_COMPACT");
  p.epi_kind = 0;
  {
    const bool al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; }(a->out_bf16) && [](const void* q) { return ((uintptr_t)q & 15) == 0; }(a->aux_out) &&
                      [](const void* q) { return ((uintptr_t)q & 7) == 0; }(a->aux_in);
    // the compact epilogues address their bf16 outputs / inputs through buffer descriptors (32-bit byte offsets, the rows of the last
    // tile included): matrices past that range take the general epilogue
    const int64_t lim32 = (int64_t)1 << 32;
    const bool desc_ok = (!a->out_bf16 || ((int64_t)a->M + 512) * a->ldo_bf16 * 2 < lim32) && (!(a->aux_in || a->aux_out) || ((int64_t)a->M + 512) * a->ldaux * 2 < lim32);
    const bool base_ok = !(es_env && es_env[0] == '0') && a->split_k == 1 && a->row_group == 0 && a->res_mod == 0 && !a->accumulate && a->N % 64 == 0 && al16 && desc_ok;
    const bool bf16_only = a->out_bf16 && !a->out_f32 && a->ldo_bf16 % 8 == 0 && !a->residual;
    if (base_ok && bf16_only && a->act == 0 && !a->aux_in && !a->aux_out && !a->colsum_out) p.epi_kind = 1;
    else if (base_ok && bf16_only && a->act == 1 && a->aux_out && a->ldaux % 8 == 0 && !a->colsum_out) p.epi_kind = 2;   // (not dispatched: spills)
    else if (base_ok && bf16_only && a->act == 3 && a->aux_out && a->ldaux % 8 == 0 && !a->colsum_out) p.epi_kind = 6;
    else if (base_ok && bf16_only && a->act == 1 && !a->aux_out && !a->aux_in && !a->colsum_out) p.epi_kind = 8;   // GELU only (no backward to come)
    else if (base_ok && a->out_f32 && !a->out_bf16 && a->residual && a->act == 0 && !a->aux_in && !a->aux_out && !a->colsum_out &&
             a->ldr % 4 == 0 && a->ldo_f32 % 4 == 0 && (((uintptr_t)a->residual | (uintptr_t)a->out_f32) & 15) == 0) p.epi_kind = 3;
    else if (base_ok && bf16_only && (a->act == 2 || a->act == 4) && a->aux_in && !a->bias && a->ldaux % 8 == 0 && ((uintptr_t)a->aux_in & 15) == 0)
      p.epi_kind = a->act == 2 ? 4 : 5;
  }
  static const char* rp_env = getenv("SA_GEMM_RING_PHASE");
  p.ring_phase = (rp_env && rp_env[0] == '0') ? 0 : 1;   // default on: 3-8 % on the dgrad shapes (SA_GEMM_RING_PHASE=0: uniform trickle)
  static const char* stg_env = getenv("SA_GEMM_STAGGER");
  p.stagger = stg_env ? atoi(stg_env) : 3;              // bit 0: late requests of wave row 1; bit 1: priority 1 for wave row 1 in the persistent kernel; bit 2: the same in the split-K 256^2 kernel (experiment)   // default on: measured 2-4 % on the forward shapes, 12-14 % on split-K wgrad (SA_GEMM_STAGGER=0 disables)
  static const char* gm256_env = getenv("SA_GEMM_GM256");
  p.gm256 = gm256_env ? atoi(gm256_env) : 4;
  if (p.gm256 < 1) p.gm256 = 1;
  static const char* nt_env = getenv("SA_GEMM_NT");
  p.nt_store = nt_env ? atoi(nt_env) : 1;          // SA_GEMM_NT=0 switches the streaming bf16 stores off (measured: +8 % launch time on
                                                    // the qkv / fc1 forward and fc2 dgrad shapes; fp32 outputs and read-once loads: no effect)
  p.tiles_m = (a->M + BM - 1) / BM; p.tiles_n = (a->N + BN - 1) / BN;
  static const char* gm_env = getenv("SA_GEMM_GM");
  p.gm = gm_env ? atoi(gm_env) : 8;
  if (p.gm < 1) p.gm = 1;

  // large dense problems go to the 256 x 256 kernel (half the operand bytes per FLOP through the CU's L2 path)
  static const char* force = getenv("SA_GEMM_TILE");
  static const char* bign_env = getenv("SA_GEMM_BIG_N");       // experiment knob: narrowest output that still takes the 256 x 256 kernels
  const int big_n = bign_env ? atoi(bign_env) : 192;           // d = 192 (ViT-T) outputs: 25 % of a 256-wide tile is padding, still 10-30 % faster
                                                               // than the 128^2 kernel with its general epilogue (scripts/bench_gemm.py, SA_BENCH_D=192)
  const bool big = a->split_k == 1 && a->M >= 1024 && a->N >= big_n && (int64_t)((a->M + 255) / 256) * ((a->N + 255) / 256) >= 128;
  // Measured on the ViT-B shapes (scripts/bench_gemm.py, random operands): the 128^2 kernel (mode 1, two 4-wave
  // workgroups per CU) beats the 256x128 three-stage (mode 3) and 256^2 (mode 2) variants at K = 768;
  // those stay selectable through SA_GEMM_TILE for experiments and are parity-tested.
  // default for large problems: mode 6, the persistent 256 x 256 kernel (measured 12 % faster than mode 1 over the ViT-B
  // forward + dgrad shapes: scripts/bench_gemm.py); small / ragged problems use the plain 128 x 128 kernel.
  // mode 8 (five-slot half-stage ring, requests trickled between MFMAs) measured 8-13 % faster than mode 6 on the dgrad (NN,
  // k-strided weight) shapes and equal on forward (NT): it is the default for NN only.
  // mode A (phased kernel, gemm_phase.hip): measured faster than mode 6 on the forward layout with the bias + residual -> fp32
  // epilogue (proj 150 vs 163 us, fc2 305 vs 343 us) and equal or slower elsewhere (scripts/bench_gemm.py): default there only.
  // ... and on the plain bias -> bf16 epilogue once the reduction is long (K >= 1536: 266 vs 285 us at K = 3072, 197 vs 211 at 2304; a tie at 768).
  static const char* k1_env = getenv("SA_GEMM_PHASE_K1");      // experiment knob: minimum K for mode A on epilogue kind 1 (0: never)
  const int k1_min = k1_env ? atoi(k1_env) : 1536;
  // ... and on narrow outputs (one column tile: N <= 256) from K = 512 on: ViT-T's fc1 / qkv data gradients 59 -> 51 us, 43 -> 41 us
  const bool nt_phase = a->a_kmajor && a->b_kmajor && a->K >= 2 * BK &&
                        (p.epi_kind == 3 || (p.epi_kind == 1 && k1_min > 0 && (a->K >= k1_min || (a->N <= 256 && a->K >= 512))));
  // thin reductions (K = 192: ViT-T) can stream through the weights-in-registers kernel (gemm_wreg.hip).  OPT-IN (SA_GEMM_WREG=1; 2 = and fail
  // when a K = 192 forward-layout launch is not covered, for the parity test): built in round 5, parity-green, measured equal or slower
  // than the tiled kernels on every ViT-T shape
  static const char* wreg_env = getenv("SA_GEMM_WREG");
  if (!force && wreg_env && (wreg_env[0] == '1' || wreg_env[0] == '2') && a->split_k == 1 && a->a_kmajor && a->b_kmajor && a->K == 192 && a->M >= 4096) {
    const int rc = p.epi_kind != 0 ? sagemm::launch_wreg(p, stream) : -1;
    if (rc >= 0) return rc;
    SA_CHECK_ARG(wreg_env[0] != '2', "sa_gemm_bf16: SA_GEMM_WREG=2 and this K = 192 launch is not covered by the weights-in-registers kernel (N=%d, epilogue kind %d)", a->N, p.epi_kind);
  }
  char mode = force ? force[0] : (big ? ((a->a_kmajor && !a->b_kmajor) ? '8' : (nt_phase ? 'A' : '6')) : '1');
  // mode P (gemm_pair.hip): 128 x 256 tiles, two 4-wave workgroups per CU -- one's epilogue under the other's main loop
  static const char* pair_env = getenv("SA_GEMM_PAIR");        // experiment knob: list of epilogue kinds that take mode P, e.g. "36"
  if (!force && big && pair_env && a->a_kmajor && a->b_kmajor && p.epi_kind > 0 && p.epi_kind < 10 && strchr(pair_env, '0' + p.epi_kind)) mode = 'P';
  if (mode == 'P' && a->split_k == 1) {
    if (a->a_kmajor && a->b_kmajor && big) {
      SA_CHECK_ARG((a_rows + 512) * a->lda * 2 < lim && (b_rows + 512) * a->ldb * 2 < lim, "sa_gemm_bf16: operand too large for the 256 tile");
      const int rc = sagemm::launch_pair(p, stream);
      if (rc >= 0) return rc;
    }
    mode = big ? ((a->a_kmajor && !a->b_kmajor) ? '8' : (nt_phase ? 'A' : '6')) : '1';
  }
  if (mode == 'A' && a->split_k == 1 && a->K >= 2 * BK && p.epi_kind != 0 && p.epi_kind != 2 && p.epi_kind != 4) {
    SA_CHECK_ARG((a_rows + 512) * a->lda * 2 < lim && (b_rows + 512) * a->ldb * 2 < lim, "sa_gemm_bf16: operand too large for the 256 tile");
    const int rc = sagemm::launch_phase(p, a->a_kmajor, a->b_kmajor, false, stream);
    if (rc >= 0) return rc;
    mode = (a->a_kmajor && !a->b_kmajor) ? '8' : '6';
  } else if (mode == 'A' && a->split_k == 1) {
    mode = big ? ((a->a_kmajor && !a->b_kmajor) ? '8' : '6') : '1';
  }
  if (mode == '8' && a->split_k == 1) {
    SA_CHECK_ARG((a_rows + 512) * a->lda * 2 < lim && (b_rows + 512) * a->ldb * 2 < lim, "sa_gemm_bf16: operand too large for the 256 tile");
    if (a->a_kmajor && a->b_kmajor) return launch256_ring<true, true, false>(p, stream);
    if (a->a_kmajor && !a->b_kmajor) return launch256_ring<true, false, false>(p, stream);
    if (!a->a_kmajor && a->b_kmajor) return launch256_ring<false, true, false>(p, stream);
    return launch256_ring<false, false, false>(p, stream);
  }
  if (mode == '6' && a->split_k == 1) {
    SA_CHECK_ARG((a_rows + 512) * a->lda * 2 < lim && (b_rows + 512) * a->ldb * 2 < lim, "sa_gemm_bf16: operand too large for the 256 tile");
    if (a->a_kmajor && a->b_kmajor) return launch256_persist<true, true>(p, stream);
    if (a->a_kmajor && !a->b_kmajor) return launch256_persist<true, false>(p, stream);
    if (!a->a_kmajor && a->b_kmajor) return launch256_persist<false, true>(p, stream);
    return launch256_persist<false, false>(p, stream);
  }
  if (a->split_k > 1 && a->tile256 == 2) {
    SA_CHECK_ARG(!a->a_kmajor && !a->b_kmajor, "sa_gemm_bf16: tile256 = 2 (the 192 x 192 streaming kernel) takes k-strided operands only");
    SA_CHECK_ARG((a_rows + 512) * a->lda * 2 < lim && (b_rows + 512) * a->ldb * 2 < lim, "sa_gemm_bf16: operand too large for the 192 tile");
    return sagemm::launch_stream(p, stream);
  }
  if (a->split_k > 1 && a->tile256) {
    SA_CHECK_ARG((a_rows + 512) * a->lda * 2 < lim && (b_rows + 512) * a->ldb * 2 < lim, "sa_gemm_bf16: operand too large for the 256 tile");
    // TN (weight gradients): the four-stage ring of gemm_stream.hip, default since round 5 (three 32 KiB stages in flight against one
    // 64 KiB stage: fc1 / fc2 282 -> 271 us, qkv 247 -> 243, proj 71 -> 75; ViT-B step 40.94 -> 40.27 ms; SA_GEMM_WGRAD_STREAM256=0: the
    // two-stage kernel below)
    static const char* wstream = getenv("SA_GEMM_WGRAD_STREAM256");
    if (!(wstream && wstream[0] == '0') && !a->a_kmajor && !a->b_kmajor && (!p.split_ws || a->N % 4 == 0)) return sagemm::launch_stream256(p, stream);
    static const char* wphase = getenv("SA_GEMM_WGRAD_PHASE");
    if (!p.split_ws && wphase && wphase[0] == '1' && !a->a_kmajor && !a->b_kmajor && (a->K + BK - 1) / BK / a->split_k >= 2)
      return sagemm::launch_phase(p, false, false, true, stream);
    static const char* wring = getenv("SA_GEMM_WGRAD_RING");
    if (!p.split_ws && wring && wring[0] == '1') {
      if (a->a_kmajor && a->b_kmajor) return launch256_ring<true, true, true>(p, stream);
      if (a->a_kmajor && !a->b_kmajor) return launch256_ring<true, false, true>(p, stream);
      if (!a->a_kmajor && a->b_kmajor) return launch256_ring<false, true, true>(p, stream);
      return launch256_ring<false, false, true>(p, stream);
    }
    if (a->a_kmajor && a->b_kmajor) return launch256<true, true, true>(p, stream);
    if (a->a_kmajor && !a->b_kmajor) return launch256<true, false, true>(p, stream);
    if (!a->a_kmajor && a->b_kmajor) return launch256<false, true, true>(p, stream);
    return launch256<false, false, true>(p, stream);
  }
  const bool swap = a->split_k == 1;
  const int sel = (a->a_kmajor ? 4 : 0) | (a->b_kmajor ? 2 : 0) | (swap ? 1 : 0);
  switch (sel) {
    case 7: return launch<true, true, true>(p, stream);
    case 6: return launch<true, true, false>(p, stream);
    case 5: return launch<true, false, true>(p, stream);
    case 4: return launch<true, false, false>(p, stream);
    case 3: return launch<false, true, true>(p, stream);
    case 2: return launch<false, true, false>(p, stream);
    case 1: return launch<false, false, true>(p, stream);
    default: return launch<false, false, false>(p, stream);
  }
}
}  // namespace

extern "C" int sa_gemm_wgrad_group(const SaGemmArgs* args, int32_t n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SA_CHECK_ARG(args != nullptr && n >= 1 && n <= sagemm::STREAM_MAX_PROBLEMS, "sa_gemm_wgrad_group: 1 .. %d products", sagemm::STREAM_MAX_PROBLEMS);
  sagemm::StreamGroup g = {};
  g.n = n; g.K = args[0].K; g.split_k = args[0].split_k; g.alpha = args[0].alpha;
  const bool det = args[0].splitk_ws != nullptr;
  const int64_t lim = (int64_t)1 << 32;
  for (int i = 0; i < n; ++i) {
    const SaGemmArgs* a = args + i;
    SA_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0 && a->A && a->B && a->out_f32, "sa_gemm_wgrad_group: product %d: empty problem or null pointer", i);
    SA_CHECK_ARG(!a->a_kmajor && !a->b_kmajor, "sa_gemm_wgrad_group: product %d: both operands must be k-strided ([K][M] and [K][N])", i);
    SA_CHECK_ARG(a->K == g.K && a->split_k == g.split_k && a->alpha == g.alpha && a->split_k > 1,
                 "sa_gemm_wgrad_group: product %d: K, alpha and split_k (> 1) must be the same in the whole group", i);
    SA_CHECK_ARG(!a->out_bf16 && !a->bias && !a->act && !a->residual && !a->row_group && !a->colsum_out,
                 "sa_gemm_wgrad_group: product %d: only alpha and fp32 accumulation are supported", i);
    SA_CHECK_ARG(a->lda % 8 == 0 && a->ldb % 8 == 0 && ((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0,
                 "sa_gemm_wgrad_group: product %d: operands must be 16-byte aligned with leading dimensions that are multiples of 8", i);
    SA_CHECK_ARG((a->splitk_ws != nullptr) == det, "sa_gemm_wgrad_group: splitk_ws must be set in every product or in none");
    if (det) SA_CHECK_ARG(a->N % 4 == 0 && a->ldo_f32 % 4 == 0 && (((uintptr_t)a->splitk_ws | (uintptr_t)a->out_f32) & 15) == 0,
                          "sa_gemm_wgrad_group: product %d: the workspace form needs N %% 4 == 0 and 16-byte aligned fp32 rows", i);
    SA_CHECK_ARG(((int64_t)a->K + 512) * a->lda * 2 < lim && ((int64_t)a->K + 512) * a->ldb * 2 < lim, "sa_gemm_wgrad_group: product %d: operand too large", i);
    sagemm::StreamProb& q = g.pr[i];
    q.A = (const char*)a->A; q.B = (const char*)a->B;
    q.a_bytes = (uint32_t)((((int64_t)a->K - 1) * a->lda + a->M) * 2);
    q.b_bytes = (uint32_t)((((int64_t)a->K - 1) * a->ldb + a->N) * 2);
    q.lda = (int)a->lda; q.ldb = (int)a->ldb; q.M = a->M; q.N = a->N;
    q.ws = a->splitk_ws; q.out = a->out_f32; q.ldo = a->ldo_f32;
    q.asum_out = a->asum_out; q.asum_ws = a->asum_ws; q.asum_lo = a->asum_skip_lo; q.asum_hi = a->asum_skip_hi;
    if (a->asum_out)
      SA_CHECK_ARG((a->asum_ws != nullptr) == det && a->asum_skip_lo >= 0 && a->asum_skip_hi >= a->asum_skip_lo,
                   "sa_gemm_wgrad_group: product %d: asum_ws goes with splitk_ws, and the skip range must be ordered", i);
  }
  const int tile = args[0].tile256 == 1 ? 256 : 192;       // (tile256 of the first product: 1 = the 256 x 256 ring for wide outputs)
  const int rc = sagemm::launch_stream_group(g, tile, stream);
  if (rc != 0 || !det) return rc;
  return sagemm::launch_stream_reduce(g, stream);
}

namespace {
// dst[c][r] = src[r][c] for a bf16 [R][C] matrix: 64 x 64 tiles through LDS (padded rows: no bank conflicts), 16-byte global accesses
__device__ __forceinline__ void transpose_tile(const bf16_t* __restrict__ src, int R, int Cn, bf16_t* __restrict__ dst, int tr, int tc,
                                               bf16_t (*tile)[66]) {
  for (int i = threadIdx.x; i < 64 * 8; i += 256) {                 // 64 rows x 8 chunks of 8 elements
    const int r = i >> 3, ch = i & 7;
    bf16x8 v;
    if (tr + r < R && tc + ch * 8 + 7 < Cn) v = *reinterpret_cast<const bf16x8*>(src + (int64_t)(tr + r) * Cn + tc + ch * 8);
    else
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (tr + r < R && tc + ch * 8 + e < Cn) ? src[(int64_t)(tr + r) * Cn + tc + ch * 8 + e] : f2bf(0.f);
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[r][ch * 8 + e] = v[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 8; i += 256) {
    const int c = i >> 3, ch = i & 7;                              // output row = source column
    if (tc + c >= Cn) continue;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[ch * 8 + e][c];
    bf16_t* o = dst + (int64_t)(tc + c) * R + tr + ch * 8;
    if (tr + ch * 8 + 7 < R) *reinterpret_cast<bf16x8*>(o) = v;
    else
#pragma unroll
      for (int e = 0; e < 8; ++e) if (tr + ch * 8 + e < R) o[e] = v[e];
  }
}

__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, int R, int Cn, bf16_t* __restrict__ dst) {
  __shared__ bf16_t tile[64][66];
  transpose_tile(src, R, Cn, dst, blockIdx.y * 64, blockIdx.x * 64, tile);
}

// Many matrices in ONE launch (all Linear weights after the optimiser step: 48 launches of ~5 us became one).  desc[m] = {src, dst,
// R | C << 32, first tile of matrix m}; block b finds its matrix by a uniform scan of the (few dozen) first-tile entries.
__global__ __launch_bounds__(256) void transpose_bf16_batch_kernel(const int64_t* __restrict__ desc, int n) {
  __shared__ bf16_t tile[64][66];
  const int b = blockIdx.x;
  int m = 0;
  while (m + 1 < n && desc[4 * (m + 1) + 3] <= b) ++m;
  const bf16_t* src = reinterpret_cast<const bf16_t*>(desc[4 * m]);
  bf16_t* dst = reinterpret_cast<bf16_t*>(desc[4 * m + 1]);
  const int R = (int)(desc[4 * m + 2] & 0xFFFFFFFF), Cn = (int)(desc[4 * m + 2] >> 32);
  const int t = b - (int)desc[4 * m + 3], tx = (Cn + 63) / 64;
  transpose_tile(src, R, Cn, dst, (t / tx) * 64, (t % tx) * 64, tile);
}
}  // namespace

extern "C" int sa_transpose_bf16(const void* src, int32_t R, int32_t Cn, void* dst, void* stream) {
  SA_CHECK_ARG(src && dst && R > 0 && Cn > 0, "sa_transpose_bf16: bad args");
  SA_CHECK_ARG(R % 8 == 0 && Cn % 8 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "sa_transpose_bf16: 16-byte aligned rows (R, C multiples of 8)");
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((Cn + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, R, Cn, (bf16_t*)dst);
  SA_LAUNCH_CHECK("sa_transpose_bf16");
  return 0;
}

extern "C" int sa_transpose_bf16_batch(const int64_t* desc_dev, int32_t n_matrices, int32_t n_tiles, void* stream) {
  SA_CHECK_ARG(desc_dev && n_matrices > 0 && n_tiles > 0, "sa_transpose_bf16_batch: bad args");
  hipLaunchKernelGGL(transpose_bf16_batch_kernel, dim3(n_tiles), dim3(256), 0, (hipStream_t)stream, desc_dev, n_matrices);
  SA_LAUNCH_CHECK("sa_transpose_bf16_batch");
  return 0;
}

extern "C" int sa_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  SA_CHECK_ARG(n >= 0 && (n == 0 || (src && dst)), "sa_cast_f32_to_bf16: bad args");
  if (n == 0) return 0;
  SA_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "sa_cast_f32_to_bf16: pointers must be 16-byte aligned");
  const int64_t want = (n + 8 * 256 - 1) / (8 * 256);
  const int grid = (int)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
  SA_LAUNCH_CHECK("sa_cast_f32_to_bf16");
  return 0;
}

extern "C" int64_t sa_colsum_workspace_bytes(int32_t M, int32_t N) {
  // at most 1024 row slabs of >= 64 rows (what sa_colsum_bf16 cuts the rows into), N floats each
  const int64_t slabs = (M + 63) / 64 < 1024 ? (M + 63) / 64 : 1024;
  return (M > 0 && N > 0) ? slabs * N * (int64_t)sizeof(float) : 0;
}

extern "C" int sa_colsum_bf16(const void* x, int64_t ld, int32_t M, int32_t N, float* out, int32_t accumulate, float* ws, int32_t n_ranges,
                              int64_t range_stride, void* stream) {
  SA_CHECK_ARG(x && out && M > 0 && N > 0 && n_ranges >= 1 && (n_ranges == 1 || range_stride >= N), "sa_colsum_bf16: bad args");
  // ws (sa_colsum_workspace_bytes x n_ranges): every block stores its row slab's partial sums and a second launch adds the slabs in slab
  // order -- bit-reproducible, and no float atomic anywhere; ws == NULL: one float atomic per column per block (order-dependent rounding).
  // n_ranges > 1: the same sums for the column ranges [z * range_stride, z * range_stride + N) of x into out + z * range_stride, in the
  // same two launches (the q and v bias gradients out of the packed dqkv, two d-wide ranges 2 d apart in both the matrix and the
  // [q | 0 | v] bias layout).
  if (!accumulate && !ws) {
    for (int z = 0; z < n_ranges; ++z)
      if (hipMemsetAsync(out + z * range_stride, 0, sizeof(float) * (size_t)N, (hipStream_t)stream) != hipSuccess) {
        sa_set_error("sa_colsum_bf16: memset failed");
        return 2;
      }
  }
  int gy;
  if ((N & 7) == 0 && (ld & 7) == 0 && ((uintptr_t)x & 15) == 0 && (range_stride & 7) == 0) {
    const int gx = (N + 255) / 256;
    // atomic form: two blocks per CU -- every block ends with one atomic per column, and those, not the 98 MB read, set the time beyond
    // that (measured on [63744, 768]: 512 blocks 18 us, 1024: 20, 2048: 24, 4096: 30); the workspace form ends in plain stores
    gy = ((ws ? 1024 : 512) / n_ranges + gx - 1) / gx;
    const int max_gy = (M + 63) / 64;
    if (gy > max_gy) gy = max_gy;
    if (gy > 1024) gy = 1024;
    const int rows_per_block = (((M + gy - 1) / gy) + 7) / 8 * 8;
    gy = (M + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(colsum_bf16_vec_kernel, dim3(gx, gy, n_ranges), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld, M, N, out, rows_per_block,
                       ws, range_stride);
  } else {
    const int gx = (N + 63) / 64;
    gy = (2048 / n_ranges + gx - 1) / gx;
    const int max_gy = (M + 63) / 64;
    if (gy > max_gy) gy = max_gy;
    if (gy > 1024) gy = 1024;
    const int rows_per_block = (M + gy - 1) / gy;
    gy = (M + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(gx, gy, n_ranges), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld, M, N, out, rows_per_block, ws,
                       range_stride);
  }
  SA_LAUNCH_CHECK("sa_colsum_bf16");
  if (ws) {
    hipLaunchKernelGGL(colsum_ws_reduce_kernel, dim3((N + 63) / 64, n_ranges), dim3(1024), 0, (hipStream_t)stream, ws, gy, N, out, accumulate ? 0 : 1,
                       range_stride);
    SA_LAUNCH_CHECK("sa_colsum_bf16(reduce)");
  }
  return 0;
}
