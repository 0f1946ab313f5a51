// Streaming split-K TN kernel for NARROW weight gradients (dW[N_out, K_out] = dY^T X with both extents a few hundred wide and the
// reduction a hundred thousand rows long: ViT-T, d = 192).  Own translation unit like gemm_phase.hip.
#include "gemm_common.h"

namespace {

// =====================================================================================================
// What bounds these launches: 2 * M * N * K flop over (M + N) * K * 2 bytes is 144 - 190 flop/B at N, M in {192, 576, 768} -- under the
// 312 flop/B ridge, so the operand rows should stream from HBM ONCE at the HBM rate and the matrix pipe idles three quarters of the time.
// The 128 x 128 kernel (two 4-wave workgroups per CU, one stage ahead, `vmcnt(0)` + barrier per K-step) moves 2.7 x the algorithmic
// bytes through L2 -> LDS (a 576 x 192 output is 5 x 2 tiles: dY is fetched twice, X five times) in bursts, and runs at 3.9 - 4.0 TB/s
// of algorithmic bytes; the 256 x 256 split-K kernel pads 192 to 256 and is no faster (scripts/bench_gemm.py, SA_BENCH_D=192).
//
// This kernel: a 192 x 192 output tile per workgroup (d = 192 outputs are 3 x 1, 1 x 1, 4 x 1, 1 x 4 tiles: no padding, the narrow
// operand is fetched by at most four workgroups that run side by side on one XCD), one 8-wave workgroup per CU, and a THREE-stage
// ring of K-steps in LDS: 48 KiB per stage (2 operands x three 8 KiB images of 64 k-rows x 64 columns), two stages = 96 KiB per CU
// always in flight, requests counted (`vmcnt(6)`: each wave sends six 1 KiB pieces per stage), ONE barrier per K-step.
//   iteration t:  wait until stage t has landed | barrier | request stage t + 2 into the slot stage t - 1 left | 36 MFMAs on stage t
// (the barrier both publishes stage t and proves every wave is past its reads of stage t - 1).
// Image = 64 k-rows x 128 B, 32-byte slots XOR-ed by bits 1 and 3 of the k-row: the eight rows {0..3, 8..11} a half-wave of a transposing
// read touches land on eight different 32-byte windows of the 256-byte bank row (the phased kernel's k-strided image).
// Waves 4 (rows) x 2 (columns), each 48 x 96 of the tile = 3 x 6 MFMA tiles, 72 accumulator registers.
// Output: split-K partials straight from the accumulators into the deterministic workspace (or fp32 atomics without one), the same
// slice layout and K-step chunks as the other split-K kernels, so sa_gemm_bf16's reduce launch serves it unchanged.
constexpr int ST_TILE = 192;
constexpr int ST_IMG = 64 * 64 * 2;                 // 8 KiB
constexpr int ST_OPER = 3 * ST_IMG;                 // 24 KiB: the 192 columns of one operand
constexpr int ST_STAGE = 2 * ST_OPER;               // 48 KiB
constexpr int ST_NSTAGE = 3;
constexpr int ST_LDS = ST_NSTAGE * ST_STAGE;        // 144 KiB

__device__ __forceinline__ int st_swz(int krow) { return ((krow >> 1) & 1) | (((krow >> 3) & 1) << 1); }

// byte offset (inside an operand's three images) of the first transposing read of a lane's fragment for the 16 columns from `sub`
// (k-step 0; the second read is 4 k-rows = 512 B on, k-step 1 is 32 k-rows = 4096 B on: neither changes the swizzle bits)
__device__ __forceinline__ int st_frag_off(int sub, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int krow = 8 * g + (i >> 2);
  const int col = (sub & 63) + 4 * (i & 3);
  return (sub >> 6) * ST_IMG + krow * 128 + (((col >> 4) ^ st_swz(krow)) << 5) + (col & 15) * 2;
}

__device__ __forceinline__ bf16x8 st_frag(const char* oper, int off) {
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(oper + off));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(oper + off + 512));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(512) void gemm_tn_stream_kernel(const sagemm::StreamGroup grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware remap as in the other kernels; the K slice is the slowest index, so the tiles of one slice -- the workgroups that share
  // the narrow operand's rows -- are neighbours on one XCD and fetch them from its L2 after the first miss
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int ks_id = lid / grp.ntiles;
  const int gtile = lid - ks_id * grp.ntiles;
  int pi = 0;                                                // the product this tile belongs to (a launch carries up to eight: see launch_stream_group)
#pragma unroll
  for (int i = 1; i < sagemm::STREAM_MAX_PROBLEMS; ++i)
    if (i < grp.n && gtile >= grp.pr[i].tile0) pi = i;
  const sagemm::StreamProb& p = grp.pr[pi];
  const int tile = gtile - p.tile0;
  const int tm = tile % p.tiles_m, tn = tile / p.tiles_m;
  const int m0 = tm * ST_TILE, n0 = tn * ST_TILE;

  const int ksteps = (grp.K + BK - 1) / BK;
  const int chunk = (ksteps + grp.split_k - 1) / grp.split_k;
  const int kt_begin = ks_id * chunk;
  const int nk = min(ksteps, kt_begin + chunk) - kt_begin;
  if (nk <= 0) return;                                       // (an empty trailing slice: the reduce skips it too)

  // ---- requests: waves 0..3 fetch operand A (the output's rows), waves 4..7 operand B; wave-instruction e of a wave is piece
  // 6 (wave & 3) + e of its operand's 24 (image = piece >> 3, k-rows 8 (piece & 7) + [0, 8)); lane = (k-row, 16-byte chunk)
  const bool isb = wave >= 4;
  const __amdgpu_buffer_rsrc_t rs = isb ? make_rsrc(p.B, p.b_bytes) : make_rsrc(p.A, p.a_bytes);
  const uint32_t ld2 = (uint32_t)(isb ? p.ldb : p.lda) * 2u;
  const int c0 = isb ? n0 : m0;
  uint32_t rq[6];
  int rq_lds[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) {
    const int piece = 6 * (wave & 3) + e;
    const int img = piece >> 3, krow = 8 * (piece & 7) + (lane >> 3);
    const int chunk16 = (lane & 7) ^ (st_swz(krow) << 1);
    rq[e] = (uint32_t)krow * ld2 + (uint32_t)(c0 + img * 64 + chunk16 * 8) * 2u;
    rq_lds[e] = (isb ? ST_OPER : 0) + piece * 1024;
  }
  auto request = [&](int slot, int kt, bool valid) {
    char* base = smem + slot * ST_STAGE;
    const uint32_t k_off = (uint32_t)kt * (uint32_t)BK * ld2;
#pragma unroll
    for (int e = 0; e < 6; ++e) lds_dma16<true>(rs, base + rq_lds[e], valid ? rq[e] + k_off : 0xFFFFFFF0u);
  };

  // ---- fragment offsets
  int fa_off[3], fb_off[6];
#pragma unroll
  for (int i = 0; i < 3; ++i) fa_off[i] = st_frag_off(48 * wm + 16 * i, lane);
#pragma unroll
  for (int j = 0; j < 6; ++j) fb_off[j] = ST_OPER + st_frag_off(96 * wn + 16 * j, lane);

  f32x4 acc[3][6];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // column sums of A (the bias gradient: StreamProb::asum_out) ride along as one more B fragment of ONES: the waves of column half 0 of the
  // tiles of output column 0 add three MFMAs per 18; every column of the result holds sum_k A[k][m]
  const bool do_asum = p.asum_out != nullptr && tn == 0 && wn == 0;
  f32x4 asum[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) asum[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8 ones = {(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};

  request(0, kt_begin, true);
  request(1, kt_begin + 1, 1 < nk);
  int slot = 0, fill = 2;
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // stage t is in (the six youngest pieces, stage t + 1, may still fly)
    __syncthreads();
    request(fill, kt_begin + t + 2, t + 2 < nk);
    const char* st = smem + slot * ST_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[3], fb[6];
#pragma unroll
      for (int i = 0; i < 3; ++i) fa[i] = st_frag(st, fa_off[i] + ks * 4096);
#pragma unroll
      for (int j = 0; j < 6; ++j) fb[j] = st_frag(st, fb_off[j] + ks * 4096);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      if (do_asum) {
#pragma unroll
        for (int i = 0; i < 3; ++i) asum[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, asum[i], 0, 0, 0);
      }
    }
    slot = slot == ST_NSTAGE - 1 ? 0 : slot + 1;
    fill = fill == ST_NSTAGE - 1 ? 0 : fill + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the out-of-range requests of the last two iterations)

  // ---- partials: lane owns rows m = 4 g + r of column n = c of each 16 x 16 tile
  const int g = lane >> 4, c = lane & 15;
  const int M = p.M, N = p.N;
  float* const ws = p.ws;
  float* const out = p.out;
  const int64_t ldo = p.ldo;
  const float alpha = grp.alpha;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int n = n0 + 96 * wn + 16 * j + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 48 * wm + 16 * i + 4 * g + r;
        if (m < M && n < N) {
          if (ws) ws[((int64_t)ks_id * M + m) * N + n] = alpha * acc[i][j][r];
          else atomicAdd(out + (int64_t)m * ldo + n, alpha * acc[i][j][r]);
        }
      }
    }
  if (do_asum && c == 0) {                                   // (column 0 of the ones product: lanes c = 0 hold rows 4 g + r)
    float* const aws = p.asum_ws;
    float* const aout = p.asum_out;
    const int lo = p.asum_lo, hi = p.asum_hi;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 48 * wm + 16 * i + 4 * g + r;
        if (m < M && (m < lo || m >= hi)) {
          if (aws) aws[(int64_t)ks_id * M + m] = alpha * asum[i][r];
          else atomicAdd(aout + m, alpha * asum[i][r]);
        }
      }
  }
}

// =====================================================================================================
// The same ring for WIDE weight gradients (d = 768 / 1024: MFMA-bound, 256 x 256 tile): four stages of 32-deep K-steps, 32 KiB each
// (per operand four images of 32 k-rows x 64 columns), THREE stages = 96 KiB in flight, `vmcnt(8)` (four pieces per wave and stage),
// one barrier per 32 MFMAs.  The k-strided layout keeps whole 512-byte row segments at BK = 32 (a k-major operand would be cut to 64-byte
// ones: DESIGN.md round 2, finding 3).  Waves 2 x 4, each 128 x 64 (8 x 4 MFMA tiles, swapped operand order: a lane owns four
// consecutive columns of a row, partials leave as float4).
constexpr int SW_IMG = 32 * 64 * 2;                 // 4 KiB
constexpr int SW_OPER = 4 * SW_IMG;                 // 16 KiB: 256 columns of one operand
constexpr int SW_STAGE = 2 * SW_OPER;               // 32 KiB
constexpr int SW_NSTAGE = 4;
constexpr int SW_LDS = SW_NSTAGE * SW_STAGE;        // 128 KiB

__device__ __forceinline__ int sw_frag_off(int sub, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int krow = 8 * g + (i >> 2);
  const int col = (sub & 63) + 4 * (i & 3);
  return (sub >> 6) * SW_IMG + krow * 128 + (((col >> 4) ^ st_swz(krow)) << 5) + (col & 15) * 2;
}

__global__ __launch_bounds__(512) void gemm_tn_stream256_kernel(const sagemm::StreamGroup grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  // K slice slowest: the tiles of one slice (of every product of the group) run side by side and share their operand slabs in L2;
  // inside a product the row tile runs fastest (a B column panel and all A panels of the slice: 12 slabs for 27 tiles at 2304 x 768)
  const int ks_id = lid / grp.ntiles;
  const int gtile = lid - ks_id * grp.ntiles;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < sagemm::STREAM_MAX_PROBLEMS; ++i)
    if (i < grp.n && gtile >= grp.pr[i].tile0) pi = i;
  const sagemm::StreamProb& p = grp.pr[pi];
  const int tile = gtile - p.tile0;
  const int tm = tile % p.tiles_m, tn = tile / p.tiles_m;
  const int m0 = tm * 256, n0 = tn * 256;

  const int ksteps = (grp.K + BK - 1) / BK;                  // slices are cut at the 64-row K-steps of the other split-K kernels
  const int chunk = (ksteps + grp.split_k - 1) / grp.split_k;
  const int kt_begin = ks_id * chunk;
  const int nk = 2 * (min(ksteps, kt_begin + chunk) - kt_begin);     // 32-deep stages
  if (nk <= 0) return;

  // ---- requests: waves 0..3 fetch operand A's image `wave`, waves 4..7 operand B's image `wave - 4`; piece e = k-rows 8 e + [0, 8)
  const bool isb = wave >= 4;
  const __amdgpu_buffer_rsrc_t rs = isb ? make_rsrc(p.B, p.b_bytes) : make_rsrc(p.A, p.a_bytes);
  const uint32_t ld2 = (uint32_t)(isb ? p.ldb : p.lda) * 2u;
  const int c0 = (isb ? n0 : m0) + (wave & 3) * 64;
  uint32_t rq[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int krow = 8 * e + (lane >> 3);
    const int chunk16 = (lane & 7) ^ (st_swz(krow) << 1);
    rq[e] = (uint32_t)krow * ld2 + (uint32_t)(c0 + chunk16 * 8) * 2u;
  }
  const int rq_lds = (isb ? SW_OPER : 0) + (wave & 3) * SW_IMG;
  const uint32_t k_base = (uint32_t)kt_begin * (uint32_t)BK * ld2;
  auto request = [&](int slot, int st, bool valid) {
    char* base = smem + slot * SW_STAGE + rq_lds;
    const uint32_t k_off = k_base + (uint32_t)st * 32u * ld2;
#pragma unroll
    for (int e = 0; e < 4; ++e) lds_dma16<true>(rs, base + e * 1024, valid ? rq[e] + k_off : 0xFFFFFFF0u);
  };

  int fa_off[8], fb_off[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) fa_off[i] = sw_frag_off(128 * wr + 16 * i, lane);
#pragma unroll
  for (int j = 0; j < 4; ++j) fb_off[j] = SW_OPER + sw_frag_off(64 * wc + 16 * j, lane);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // column sums of A through a B fragment of ones, as in the 192 kernel (wave column 0 of the tiles of output column 0: 8 MFMAs per 32)
  const bool do_asum = p.asum_out != nullptr && tn == 0 && wc == 0;
  f32x4 asum[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) asum[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16x8 ones = {(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};

  request(0, 0, true);
  request(1, 1, 1 < nk);
  request(2, 2, 2 < nk);
  int slot = 0, fill = 3;
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // stage t is in; stages t + 1 and t + 2 may still fly
    __syncthreads();
    request(fill, t + 3, t + 3 < nk);
    const char* st = smem + slot * SW_STAGE;
    bf16x8 fa[8], fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = st_frag(st, fb_off[j]);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = st_frag(st, fa_off[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    if (do_asum) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asum[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[i], asum[i], 0, 0, 0);
    }
    slot = slot == SW_NSTAGE - 1 ? 0 : slot + 1;
    fill = fill == SW_NSTAGE - 1 ? 0 : fill + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- partials: lane owns row 16 i + c, columns 16 j + 4 g + [0, 4) of its 128 x 64 block
  const int g = lane >> 4, c = lane & 15;
  const int M = p.M, N = p.N;
  float* const ws = p.ws;
  float* const out = p.out;
  const int64_t ldo = p.ldo;
  const float alpha = grp.alpha;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + 128 * wr + 16 * i + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 64 * wc + 16 * j + 4 * g;
      if (m < M && n < N) {
        if (ws) {
          *reinterpret_cast<float4*>(ws + ((int64_t)ks_id * M + m) * N + n) =
              make_float4(alpha * acc[i][j][0], alpha * acc[i][j][1], alpha * acc[i][j][2], alpha * acc[i][j][3]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < N) atomicAdd(out + (int64_t)m * ldo + n + r, alpha * acc[i][j][r]);
        }
      }
    }
  }
  if (do_asum && g == 0) {                                   // (swapped operands: every column of row 16 i + c holds the sum; lanes g = 0 store it)
    float* const aws = p.asum_ws;
    float* const aout = p.asum_out;
    const int lo = p.asum_lo, hi = p.asum_hi;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + 128 * wr + 16 * i + c;
      if (m < M && (m < lo || m >= hi)) {
        if (aws) aws[(int64_t)ks_id * M + m] = alpha * asum[i][0];
        else atomicAdd(aout + m, alpha * asum[i][0]);
      }
    }
  }
}

// deterministic split-K, second launch, for a whole group: out_i[m][n] += sum over slices (in slice order) of ws_i[s][m][n].  One
// block = 64 float4 columns of one product (blocks are dealt to the products in order); its eight waves take every eighth slice, their
// partial sums meet in LDS and are added in wave order (sa_gemm_bf16's splitk_reduce_wide_kernel, with a product table in front).
__global__ __launch_bounds__(512) void stream_reduce_kernel(const sagemm::StreamGroup grp, int nslice, int rblocks) {
  constexpr int NW = 8;
  __shared__ float4 part[NW][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x >= rblocks) {                          // the blocks behind the matrices: asum_out[m] += slice sums, 512 rows per block
    const int ab = (int)blockIdx.x - rblocks;
    int qi = -1;
#pragma unroll
    for (int i = 0; i < sagemm::STREAM_MAX_PROBLEMS; ++i)
      if (i < grp.n && grp.pr[i].ablock0 >= 0 && ab >= grp.pr[i].ablock0) qi = i;
    if (qi < 0) return;
    const sagemm::StreamProb& q = grp.pr[qi];
    const int m = (ab - q.ablock0) * 512 + (int)threadIdx.x;
    if (m >= q.M || (m >= q.asum_lo && m < q.asum_hi)) return;
    float s = 0.f;
    for (int sl = 0; sl < nslice; ++sl) s += q.asum_ws[(int64_t)sl * q.M + m];
    q.asum_out[m] += s;
    return;
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < sagemm::STREAM_MAX_PROBLEMS; ++i)
    if (i < grp.n && (int)blockIdx.x >= grp.pr[i].rblock0) pi = i;
  const sagemm::StreamProb& p = grp.pr[pi];
  const int M = p.M, N = p.N;
  const int64_t i4 = (int64_t)((int)blockIdx.x - p.rblock0) * 64 + lane;
  const int n4 = N >> 2;
  const bool ok = i4 < (int64_t)M * n4;
  const int m = ok ? (int)(i4 / n4) : 0, n = ok ? (int)(i4 - (int64_t)m * n4) * 4 : 0;
  const int64_t slice = (int64_t)M * N;
  const float* src = p.ws + (int64_t)m * N + n;
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
  float4* o = reinterpret_cast<float4*>(p.out + (int64_t)m * p.ldo + n);
  float4 t = *o;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const float4 v = part[w][lane];
    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
  }
  *o = t;
}

}  // namespace

// the streaming kernel (tile = 192 or 256) for `g.n` products that share the reduction (both operands k-strided, split_k > 1); fills in
// the tile bookkeeping of `g`.  0 on success, 2 on a launch error
int sagemm::launch_stream_group(StreamGroup& g, int tile, hipStream_t stream) {
  static bool cfg = false;
  if (!cfg) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_stream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_stream256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS) != hipSuccess) {
      sa_set_error("sa_gemm_bf16: 144 KiB of LDS per workgroup refused");
      return 2;
    }
    cfg = true;
  }
  int tiles = 0, rblocks = 0, ablocks = 0;
  for (int i = 0; i < g.n; ++i) {
    StreamProb& q = g.pr[i];
    q.tiles_m = (q.M + tile - 1) / tile;
    q.tile0 = tiles;
    tiles += q.tiles_m * ((q.N + tile - 1) / tile);
    q.rblock0 = rblocks;
    rblocks += (int)(((int64_t)q.M * (q.N / 4) + 63) / 64);
    q.ablock0 = (q.asum_out && q.asum_ws) ? ablocks : -1;
    if (q.ablock0 >= 0) ablocks += (q.M + 511) / 512;
  }
  g.ntiles = tiles;
  if (tile == ST_TILE) hipLaunchKernelGGL(gemm_tn_stream_kernel, dim3((unsigned)(tiles * g.split_k)), dim3(512), ST_LDS, stream, g);
  else hipLaunchKernelGGL(gemm_tn_stream256_kernel, dim3((unsigned)(tiles * g.split_k)), dim3(512), SW_LDS, stream, g);
  SA_LAUNCH_CHECK("sa_gemm_bf16(streaming split-K)");
  return 0;
}

// the reduce launch of a group whose partials went to workspaces (every product's `ws` set)
int sagemm::launch_stream_reduce(const StreamGroup& g, hipStream_t stream) {
  const int ksteps = (g.K + BK - 1) / BK, chunk = (ksteps + g.split_k - 1) / g.split_k;
  const int nslice = (ksteps + chunk - 1) / chunk;           // trailing slices with an empty K range wrote nothing
  const StreamProb& last = g.pr[g.n - 1];
  const int rblocks = last.rblock0 + (int)(((int64_t)last.M * (last.N / 4) + 63) / 64);
  int ablocks = 0;
  for (int i = 0; i < g.n; ++i)
    if (g.pr[i].ablock0 >= 0) ablocks = g.pr[i].ablock0 + (g.pr[i].M + 511) / 512;
  hipLaunchKernelGGL(stream_reduce_kernel, dim3((unsigned)(rblocks + ablocks)), dim3(512), 0, stream, g, nslice, rblocks);
  SA_LAUNCH_CHECK("sa_gemm_wgrad_group(split-K reduce)");
  return 0;
}

namespace {
int launch_stream_single(const GemmParams& p, int tile, hipStream_t stream) {
  sagemm::StreamGroup g = {};
  g.n = 1; g.K = p.K; g.split_k = p.split_k; g.alpha = p.alpha;
  sagemm::StreamProb& q = g.pr[0];
  q.A = p.A; q.B = p.B; q.a_bytes = p.a_bytes; q.b_bytes = p.b_bytes; q.lda = p.lda; q.ldb = p.ldb; q.M = p.M; q.N = p.N;
  q.ws = p.split_ws; q.out = p.out_f32; q.ldo = p.ldo_f32;
  q.asum_out = p.asum_out; q.asum_ws = p.asum_ws; q.asum_lo = p.asum_lo; q.asum_hi = p.asum_hi;
  const int rc = sagemm::launch_stream_group(g, tile, stream);
  // (a single product's matrix partials are summed by sa_gemm_bf16's reduce launch; its row sums need this one)
  if (rc == 0 && q.asum_out && q.asum_ws) {
    const int ksteps = (g.K + BK - 1) / BK, chunk = (ksteps + g.split_k - 1) / g.split_k;
    const int nslice = (ksteps + chunk - 1) / chunk;
    hipLaunchKernelGGL(stream_reduce_kernel, dim3((unsigned)((q.M + 511) / 512)), dim3(512), 0, stream, g, nslice, 0);
    SA_LAUNCH_CHECK("sa_gemm_bf16(row-sum reduce)");
  }
  return rc;
}
}  // namespace

int sagemm::launch_stream(GemmParams p, hipStream_t stream) { return launch_stream_single(p, ST_TILE, stream); }
// the 256 x 256 form; with a workspace N must be a multiple of 4 (float4 partial rows)
int sagemm::launch_stream256(GemmParams p, hipStream_t stream) { return launch_stream_single(p, 256, stream); }
