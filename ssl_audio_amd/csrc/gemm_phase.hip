// Phased persistent 256 x 256 x 64 bf16 MFMA GEMM for gfx950 (see the comment above the kernel).  Own translation unit: the
// kernel family is 12 instantiations of a 250-VGPR kernel and dominates the build time of the library.
#include "gemm_common.h"

namespace {

// =====================================================================================================
// Phased persistent 256 x 256 x 64 kernel ("mode A"; forward NT, dgrad NN, split-K wgrad TN).
//
// Two waves live on every SIMD (waves w and w + 4: wave row 0 = waves 0..3 owns tile rows 0..127, wave row 1 rows 128..255).
// A K-tile is consumed in FOUR phases, one 64 x 32 quadrant of the wave's 128 x 64 output each; a phase is
//     [L: fragment reads of this quadrant + 2 LDS-DMA requests + counted wait] s_barrier [M: 16 MFMAs] s_barrier
// and wave row 1 runs ONE barrier interval behind row 0, so that in every interval one wave of each SIMD is in its MFMA segment
// while its partner reads fragments and issues requests: the matrix pipe never waits for the LDS reads that open a quadrant, and
// the request stream is a steady 2 KiB per wave and interval instead of a burst per K-step.
//
// LDS: two K-tile buffers of eight 8 KiB IMAGES (64 rows x 128 B) + 4 KiB of epilogue scratch per wave = 160 KiB.
//   image 2r+0 / 2r+1: A_lo / A_hi of wave row r  = tile rows r*128 + [0,64) / + [64,128)      (read in P1 / P3, by row r only)
//   image 4+2h / 5+2h: B_lo / B_hi of column half h = for both wave columns of the half, their columns [0,32) / [32,64) (P1 / P2)
// An image dies phase by phase, so it is re-requested for K-tile s+2 as soon as its last reader is past it:
//   phase of K-tile s      reads                    requests (2 instructions per wave = one image pair per interval)
//   P1                     B_lo(s), A_lo(s)         A_hi(s+1)
//   P2                     B_hi(s)                  A_lo(s+2)
//   P3                     A_hi(s)                  B_lo(s+2)
//   P4                     -                        B_hi(s+2)
// With exactly one request pair per phase, `s_waitcnt vmcnt(10)` after the pair leaves the five youngest pieces in flight
// (40 KiB per wave row, 80 KiB per CU) and proves that everything the NEXT phase reads has landed; the barrier that ends the
// L segment publishes it to the other waves (read one phase after the wait: cdna_hip_programming.md "Pipelining across
// barriers").  WAR: every request goes out at least one full barrier interval after the last read of its image completed
// (the stagger costs row 0 one extra phase for the shared B images; the table above already includes it).
// The K-tile stream runs across tile (and split-K unit) borders; past its end the requests go out of range (zero fill, no
// traffic, same count).  The epilogue scratch is private to the wave and outside the pipeline, so the stream never stops.
constexpr int IMG_BYTES = 8192;
constexpr int PH_BUF = 8 * IMG_BYTES;
constexpr int PH_LDS = 2 * PH_BUF + 8 * 4096;

// 32-byte-slot XOR of a k-strided image row: the 8 rows one half-wave of a transposing read touches ({0..3, 8..11} + 4t + 16u)
// land on 8 distinct 32-byte windows of the 256-byte bank row (rows 2 apart and 8 apart get different slots)
__device__ __forceinline__ int ph_swz(int krow) { return ((krow >> 1) & 1) | (((krow >> 3) & 1) << 1); }

template <bool KMAJOR>
__device__ __forceinline__ bf16x8 ph_frag(const char* img, int sub0, int kstep, int lane) {
  if constexpr (KMAJOR) {
    return load_frag<true>(img, sub0, kstep, lane);          // 64 rows x 128 B, chunk ^ (row & 7): the 128^2 tile's geometry
  } else {
    const int i = lane & 15, g = lane >> 4;
    const int krow = kstep * 32 + 8 * g + (i >> 2);
    const int col = sub0 + 4 * (i & 3);
    const int off = krow * 128 + (((col >> 4) ^ ph_swz(krow)) << 5) + (col & 15) * 2;    // ph_swz(krow) == ph_swz(krow + 4)
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + off));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + off + 512));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
}

#ifdef SA_PHASE_STAMPS
__device__ unsigned long long sa_phase_prof[4][8];
#endif

template <bool A_KM, bool B_KM, bool SPLIT, int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_phase_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
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
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);

  int u = lid;
  if (u >= nunits) return;
  int m0, n0, kt0, nk;
  unit(u, m0, n0, kt0, nk);

  // ---- per-lane parts of the request addresses (instruction j = 0, 1 of this wave's pair: image rows 16 wc + 8 j + (lane >> 3))
  // k-major image row = operand row, 16-byte chunk XOR (row & 7); k-strided image row = k, 32-byte slot XOR ph_swz(k).
  // A B image holds the two wave columns' 32-column groups: image rows / columns 32..63 are 64 operand columns after 0..31.
  uint32_t la[2], lb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r8 = lane >> 3, c16 = lane & 7;
    if constexpr (A_KM) la[j] = (uint32_t)(r8 * p.lda + ((c16 ^ r8) << 3)) * 2u;
    else {
      const int col = ((((c16 >> 1) ^ (((r8 >> 1) & 1) | (j << 1))) << 1) | (c16 & 1)) << 3;
      la[j] = (uint32_t)(r8 * p.lda + col) * 2u;
    }
    if constexpr (B_KM) lb[j] = (uint32_t)(r8 * p.ldb + ((c16 ^ r8) << 3)) * 2u;
    else {
      const int col = ((((c16 >> 1) ^ (((r8 >> 1) & 1) | (j << 1))) << 1) | (c16 & 1)) << 3;
      lb[j] = (uint32_t)(r8 * p.ldb + col + (col >= 32 ? 32 : 0)) * 2u;
    }
  }
  const int wrow = 16 * wc;                                  // first image row of this wave's request pair
  const int bsplit = (B_KM && wc >= 2) ? 32 : 0;             // k-major B image rows 32..63 = the second wave column's group
  // request pair for one image: `img` in the buffer, operand base index `base` (row of a k-major operand / column of a k-strided one)
  auto req_a = [&](char* img, int base, int k0, bool valid) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      uint32_t voff = A_KM ? la[j] + (uint32_t)((base + wrow + 8 * j) * p.lda + k0) * 2u
                           : la[j] + (uint32_t)((k0 + wrow + 8 * j) * p.lda + base) * 2u;
      voff = valid ? voff : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, LDS_PTR(img + (2 * wc + j) * 1024), 16, voff, 0, 0, 0);
    }
  };
  auto req_b = [&](char* img, int base, int k0, bool valid) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      uint32_t voff = B_KM ? lb[j] + (uint32_t)((base + wrow + bsplit + 8 * j) * p.ldb + k0) * 2u
                           : lb[j] + (uint32_t)((k0 + wrow + 8 * j) * p.ldb + base) * 2u;
      voff = valid ? voff : 0xFFFFFFF0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, LDS_PTR(img + (2 * wc + j) * 1024), 16, voff, 0, 0, 0);
    }
  };
  // this wave's images: it REQUESTS A_x of its own row and B_x of column half `wr`; it READS A_x of its row and B_x of half wc >> 1
  const int ia_lo = (2 * wr) * IMG_BYTES, ia_hi = (2 * wr + 1) * IMG_BYTES;
  const int qb_lo = (4 + 2 * wr) * IMG_BYTES, qb_hi = (5 + 2 * wr) * IMG_BYTES;
  const int rb_lo = (4 + 2 * (wc >> 1)) * IMG_BYTES, rb_hi = (5 + 2 * (wc >> 1)) * IMG_BYTES;
  const int bsub = (wc & 1) * 32;                            // this wave's 32 rows / columns inside a B image

  // ---- stream cursor: position of K-tile s + 2 (c2) and s + 1 (c1) while K-tile s is consumed
  struct Pos { int m0, n0, k0, valid; };
  int cu = u, cm0 = m0, cn0 = n0, ck = kt0, ckend = kt0 + nk;
  // dynamic hand-out (p.tile_counter, set by the host only when every unit has >= 6 K-tiles): thread 0 draws the ticket of the NEXT
  // unit during the first K-tile of a unit, publishes it through the first word of wave 0's (idle) epilogue scratch during the second
  // and every wave has it in an SGPR before the cursor -- three K-tiles ahead of the compute -- crosses into that unit
  const bool dyn = p.tile_counter != nullptr;
  int* const ticket_word = reinterpret_cast<int*>(smem + 2 * PH_BUF);
  int unext = 0x7fffffff;                                    // dynamic mode: the drawn id of the next unit (until drawn: none)
  auto take = [&]() {                                        // current cursor position, then advance by one K-tile
    Pos q{cm0, cn0, ck * BK, cu < nunits};
    if (q.valid && ++ck == ckend) {
      cu = dyn ? unext : cu + nwg;
      if (cu < nunits) {
        int nk2;
        unit(cu, cm0, cn0, ck, nk2);
        ckend = ck + nk2;
      }
    }
    return q;
  };
  auto issue_a_lo = [&](const Pos& q, int buf) { req_a(smem + buf * PH_BUF + ia_lo, q.m0 + wr * 128, q.k0, q.valid); };
  auto issue_a_hi = [&](const Pos& q, int buf) { req_a(smem + buf * PH_BUF + ia_hi, q.m0 + wr * 128 + 64, q.k0, q.valid); };
  auto issue_b_lo = [&](const Pos& q, int buf) { req_b(smem + buf * PH_BUF + qb_lo, q.n0 + wr * 128, q.k0, q.valid); };
  auto issue_b_hi = [&](const Pos& q, int buf) { req_b(smem + buf * PH_BUF + qb_hi, q.n0 + wr * 128 + 32, q.k0, q.valid); };

  // prologue: the seven pieces the steady state would have requested before (s = 0, P1), in its order
  Pos c0 = take();
  Pos c1 = take();
  issue_a_lo(c0, 0); issue_b_lo(c0, 0); issue_b_hi(c0, 0); issue_a_hi(c0, 0);
  issue_a_lo(c1, 1); issue_b_lo(c1, 1); issue_b_hi(c1, 1);
  Pos c2 = take();
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");         // A_lo(0), B_lo(0) have landed
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();                // wave row 1 runs one interval behind from here on

  // SA_PHASE_STAMPS (scripts/microbench/gemm_phase_prof.hip only; never in the library): cycle stamps of every wave 0 / wave 4
  // per segment -- [0] L issue (reads + requests), [1] vmcnt wait, [2] barrier 1, [3] lgkmcnt wait, [4] MFMA issue, [5] barrier 2
#ifdef SA_PHASE_STAMPS
  unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter(), tnow;
  const unsigned long long tbegin = tprev;
#define PH_T(k) { __builtin_amdgcn_sched_barrier(0); tnow = __builtin_readcyclecounter(); pt[k] += tnow - tprev; tprev = tnow; __builtin_amdgcn_sched_barrier(0); }
#else
#define PH_T(k)
#endif
#define SA_MM(FA, FB, ACC) (SPLIT ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA, FB, ACC, 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB, FA, ACC, 0, 0, 0))
#define SA_L_END()                                             \
  PH_T(0)                                                      \
  if (fresh) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(10 + EPI_STORES) : "memory");   \
  else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");       \
  __builtin_amdgcn_sched_barrier(0);                           \
  PH_T(1)                                                      \
  __builtin_amdgcn_s_barrier();                                \
  PH_T(2)                                                      \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           \
  __builtin_amdgcn_sched_barrier(0);                           \
  PH_T(3)                                                      \
  __builtin_amdgcn_s_setprio(1);
#define SA_M_END()                                             \
  __builtin_amdgcn_s_setprio(0);                               \
  __builtin_amdgcn_sched_barrier(0);                           \
  PH_T(4)                                                      \
  __builtin_amdgcn_s_barrier();                                \
  __builtin_amdgcn_sched_barrier(0);                           \
  PH_T(5)
#define SA_QUAD(AI, BJ, FBX)                                                                       \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                 \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[AI + i][BJ + j] = SA_MM(fa[i][ks], FBX[j][ks], acc[AI + i][BJ + j]);

  // vmcnt counts stores too, in issue order: right after an epilogue the counted wait would first drain the tile's stores
  // (an HBM round trip) although everything the next phases read was requested BEFORE them.  For five phases after the epilogue
  // of a FULL tile (its store count is then exact) the wait leaves those stores outstanding as well.
  constexpr int EPI_STORES = SPLIT ? 0 : (EPI == 1 ? 16 : (EPI == 3 || EPI == 6) ? 32 : 0);
  bool after_full = false;                                   // the previous unit of this workgroup stored a full 256 x 256 tile
  int sidx = 0;                                              // stream index of the K-tile being consumed (buffer = sidx & 1)
  while (true) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    int ticket = 0;
    for (int kt = 0; kt < nk; ++kt, ++sidx) {
      const int buf = sidx & 1;
      const char* cur = smem + buf * PH_BUF;
      bool fresh = EPI_STORES > 0 && after_full && kt <= 1;
      if (dyn && kt == 0 && threadIdx.x == 0) ticket = atomicAdd(p.tile_counter, 1);
      if (dyn && kt == 1 && threadIdx.x == 0) *ticket_word = nwg + ticket;
      // ---- P1: quadrant (A_lo, B_lo)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fb0[j][ks] = ph_frag<B_KM>(cur + rb_lo, bsub + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[i][ks] = ph_frag<A_KM>(cur + ia_lo, i * 16, ks, lane);
      issue_a_hi(c1, buf ^ 1);
      SA_L_END()
      SA_QUAD(0, 0, fb0)
      SA_M_END()
      fresh = fresh && kt == 0;
      // ---- P2: quadrant (A_lo, B_hi)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fb1[j][ks] = ph_frag<B_KM>(cur + rb_hi, bsub + j * 16, ks, lane);
      issue_a_lo(c2, buf);
      SA_L_END()
      SA_QUAD(0, 2, fb1)
      SA_M_END()
      // ---- P3: quadrant (A_hi, B_hi)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fa[i][ks] = ph_frag<A_KM>(cur + ia_hi, i * 16, ks, lane);
      issue_b_lo(c2, buf);
      SA_L_END()
      SA_QUAD(4, 2, fb1)
      SA_M_END()
      // ---- P4: quadrant (A_hi, B_lo)
      issue_b_hi(c2, buf);
      SA_L_END()
      SA_QUAD(4, 0, fb0)
      SA_M_END()
      if (dyn && kt == 1) unext = __builtin_amdgcn_readfirstlane(*ticket_word);     // (six barriers after thread 0's write)
      c1 = c2;
      c2 = take();
    }
    // ---- epilogue (wave-private 4 KiB scratch outside the pipeline; the request stream of the next unit is already running)
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
      char* wl = smem + 2 * PH_BUF + wave * 4096;
      float cs[4][4];
      const int nb = n0 + wc * 64;
      float4 bias4[4];                                   // once per tile: later loads would queue behind the tile's stores
      const bool has_bias = (EPI == 1 || EPI == 3 || EPI == 6) && p.bias != nullptr && nb < p.N;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bias4[j] = has_bias ? *reinterpret_cast<const float4*>(p.bias + nb + j * 16 + 4 * (lane >> 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
      uint4 auxv[(EPI == 5) ? 16 : 1];                   // kind 5: the tile's whole aux_in block of this wave (128 rows x 64 columns)
      if constexpr (EPI == 5) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = m0 + wr * 128 + q * 8 + (lane >> 3);
          auxv[q] = (m < p.M && nb < p.N) ? *reinterpret_cast<const uint4*>(p.aux_in + (int64_t)m * p.ldaux + nb + (lane & 7) * 8) : make_uint4(0u, 0u, 0u, 0u);
        }
      }
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        f32x4(&blk)[2][4] = reinterpret_cast<f32x4(&)[2][4]>(acc[2 * h]);
        const int mb = m0 + wr * 128 + h * 32;
        const uint4* pre = (EPI == 5) ? &auxv[4 * h] : nullptr;
        if (h & 1) wave_epilogue_compact<true, EPI, 2, 2>(p, blk, mb, nb, wl, lane, cs, bias4, pre);
        else wave_epilogue_compact<true, EPI, 2, 1>(p, blk, mb, nb, wl, lane, cs, bias4, pre);
      }
    }
    PH_T(6)
    after_full = m0 + 256 <= p.M && n0 + 256 <= p.N;
    u = dyn ? unext : u + nwg;
    if (u >= nunits) break;
    unit(u, m0, n0, kt0, nk);
  }
#ifdef SA_PHASE_STAMPS
  if ((blockIdx.x == 0 || blockIdx.x == 133) && (threadIdx.x & 255) == 0) {     // waves 0 and 4 of two workgroups
    unsigned long long* o = sa_phase_prof[(blockIdx.x == 0 ? 0 : 2) + wr];
    for (int k = 0; k < 7; ++k) o[k] = pt[k];
    o[7] = __builtin_readcyclecounter() - tbegin;
  }
#endif
  if (wr == 0) __builtin_amdgcn_s_barrier();                // pairs with row 1's extra barrier
#undef SA_MM
#undef PH_T
#undef SA_L_END
#undef SA_M_END
#undef SA_QUAD
}

template <bool A_KM, bool B_KM, bool SPLIT, int EPI>
int launch256_phase_one(const GemmParams& p, hipStream_t stream, int slots) {
  static bool cfg = false;
  if (!cfg) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256_phase_kernel<A_KM, B_KM, SPLIT, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            PH_LDS) != hipSuccess) {
      sa_set_error("sa_gemm_bf16: 160 KiB of LDS per workgroup refused");
      return 2;
    }
    cfg = true;
  }
  const int nunits = p.tiles_m * p.tiles_n * p.split_k;
  const dim3 grid(nunits < budget_slots(slots) ? nunits : budget_slots(slots));
  GemmParams q = p;
  const int ksteps_all = (p.K + BK - 1) / BK;
  // the shortest K slice: slices are chunk = ceil(ksteps / split) K-tiles long, the LAST one holds what is left (the floor
  // overstated it, e.g. 100 K-tiles in 13 slices: floor 7, last slice 4 -- the cursor then reads the next ticket before it is drawn)
  const int chunk_nk = (ksteps_all + p.split_k - 1) / p.split_k;
  const int min_nk = SPLIT ? ksteps_all - (p.split_k - 1) * chunk_nk : ksteps_all;
  q.tile_counter = (min_nk >= 6 && nunits > (int)grid.x) ? sagemm::next_tile_counter(stream) : nullptr;
  hipLaunchKernelGGL((gemm256_phase_kernel<A_KM, B_KM, SPLIT, EPI>), grid, dim3(512), PH_LDS, stream, q);
  SA_LAUNCH_CHECK("sa_gemm_bf16(256 phased)");
  return 0;
}

// returns -1 when this problem is not one the phased kernel covers (the caller then takes the general 256^2 kernels)
template <bool A_KM, bool B_KM, bool SPLIT>
int launch256_phase(GemmParams p, hipStream_t stream) {
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  const int ksteps_all = (p.K + BK - 1) / BK;
  if (SPLIT) {                                          // no empty K slices: both cursors walk the same unit list
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
    slots = prop.multiProcessorCount;
  }
  if constexpr (SPLIT) {
    return launch256_phase_one<A_KM, B_KM, true, 0>(p, stream, slots);
  } else {
    switch (p.epi_kind) {
      case 1: return launch256_phase_one<A_KM, B_KM, false, 1>(p, stream, slots);
      case 3: return launch256_phase_one<A_KM, B_KM, false, 3>(p, stream, slots);
      case 5: return launch256_phase_one<A_KM, B_KM, false, 5>(p, stream, slots);
      case 6: return launch256_phase_one<A_KM, B_KM, false, 6>(p, stream, slots);
      default: return -1;
    }
  }
}


}  // namespace

int sagemm::launch_phase(GemmParams p, bool a_kmajor, bool b_kmajor, bool split, hipStream_t stream) {
  if (split) {
    if (!a_kmajor && !b_kmajor) return launch256_phase<false, false, true>(p, stream);
    return -1;
  }
  if (a_kmajor && b_kmajor) return launch256_phase<true, true, false>(p, stream);
  if (a_kmajor && !b_kmajor) return launch256_phase<true, false, false>(p, stream);
  return -1;
}
