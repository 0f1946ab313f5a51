// Weights-in-registers streaming kernel for THIN forward-layout products (Y = X W^T with K = 192: ViT-T's qkv / proj / fc1 forward and its
// fc2 / proj data gradients, models/mae.py:106-131,155-163 at embed_dim 192).  Own translation unit like gemm_phase.hip / gemm_stream.hip.
// OPT-IN (SA_GEMM_WREG=1): parity-green, measured equal or slower than the tiled kernels -- what bounds it is in DESIGN.md section 6,
// round 5, item 11 (epilogue VALU + MFMA + store time ADD with eight waves per CU), next to the store-data hazard its first build hit.
#include "gemm_common.h"

namespace {

// =====================================================================================================
// What bounds these launches: 2 * M * N * 192 flop over M * (192 + N) * 2 bytes (+ the epilogue's second output / residual / aux rows) is
// 40 - 150 flop/B -- far under the 312 flop/B ridge: the activation rows should stream from HBM ONCE, the outputs leave once, and the
// matrix pipe idles four fifths of the time.  The 256 x 256 kernels run them at 4.0 - 4.5 TB/s of algorithmic bytes: three K-steps per
// tile, so a workgroup alternates between fetching, 96 MFMAs and an epilogue that writes more bytes than the main loop read.
//
// This kernel turns the tiling round.  The WEIGHT is the small operand (N x 192 bf16 = 72 - 288 KiB) and a CU's register file is
// 512 KiB: every wave keeps a 96-column x 192-deep block of W as MFMA A-fragments in 144 registers for the whole launch (8 waves x 96
// columns = N = 768; 6 waves for N = 576; for N = 192 two column blocks x four row groups), and the ACTIVATION rows stream through a
// four-slot LDS ring of 64-row stages (three 8 KiB k-major images of 64 k each, the 128-byte-row image with the 16-byte chunk XOR of the
// other kernels: conflict-free ds_read_b128 fragments).  Per 16-row tile a wave reads six X fragments and issues 36 MFMAs against its
// resident W; the accumulators (swapped operand order: a lane owns ONE row and consecutive columns) go straight to the epilogue -- no
// LDS staging: the column-to-fragment-row map of W is chosen so that a lane's two accumulators of a tile PAIR are 8 consecutive
// columns (tile 2p row i <-> column 32p + 8 (i >> 2) + (i & 3), tile 2p + 1 the same + 4): 16-byte bf16 / 32-byte fp32 stores, four
// lanes = one 64- / 128-byte row segment.  One barrier per 64 rows; the waves of a workgroup never exchange data, so one wave's
// epilogue (VALU, stores) runs under another's MFMAs and fetches by itself.
//   iteration i:  wait until stage i has landed | barrier | T x { 36 MFMAs | request next tile's epilogue inputs | (last tile:)
//                 request stage i + 3 into the slot stage i - 1 left | epilogue + stores }
// vmcnt is ONE in-order queue of loads, LDS-DMA requests and stores: the counted wait for stage i allows exactly the operations issued
// after its requests (a lower bound, all static: invalid requests and out-of-range rows are issued out of bounds, never skipped).
// -DWR_EXP=<bits> (A/B builds only, scripts/ab_build.sh): timing experiments that break the results -- 1: no output stores, 2: no LDS reads / MFMAs,
// 4: no GELU arithmetic, 8: no counted wait at the top of an iteration
#ifndef WR_EXP
#define WR_EXP 0
#endif
constexpr int WR_K = 192;
constexpr int WR_IMG = 64 * 128;                    // 8 KiB: 64 rows x 64 k
constexpr int WR_STAGE = 3 * WR_IMG;                // 24 KiB: 64 rows x 192 k
constexpr int WR_D = 4;
constexpr int WR_LDS = WR_D * WR_STAGE;             // 96 KiB (+ N floats of bias)

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int N_>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_ > 63 ? 63 : N_) : "memory"); }

__device__ __forceinline__ uint32_t wr_cvt_pk(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2)); }
__device__ __forceinline__ u32x4 wr_pack8(const float (&v)[8]) {
  return u32x4{wr_cvt_pk(v[0], v[1]), wr_cvt_pk(v[2], v[3]), wr_cvt_pk(v[4], v[5]), wr_cvt_pk(v[6], v[7])};
}
template <bool NT>
__device__ __forceinline__ void wr_store16(u32x4 val, __amdgpu_buffer_rsrc_t rs, uint32_t voff, int soff) {
#if WR_EXP & 1
  asm volatile("" ::"v"(val), "v"(voff));
  return;
#endif
  // (the constant is added to the per-lane offset: the compiler moves it into the instruction's immediate field; passed as the
  // builtin's scalar offset it is materialised in an SGPR)
  if constexpr (NT) __builtin_amdgcn_raw_buffer_store_b128(val, rs, voff + (uint32_t)soff, 0, 2);
  else __builtin_amdgcn_raw_buffer_store_b128(val, rs, voff + (uint32_t)soff, 0, 0);
}

// KIND: the compact epilogue kinds of gemm_common.h (1 bias -> bf16 | 3 bias + residual -> fp32 | 5 x aux, column sums -> bf16 |
// 6 GELU pair -> two bf16 | 8 GELU -> bf16).  NB = N / 96 column blocks (2, 6, 8).
template <int KIND, int NB, bool NT>
__global__ __launch_bounds__(64 * (NB == 2 ? 8 : NB)) void gemm_wreg_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RG = NB == 2 ? 4 : 1;               // row groups: N = 192 leaves room for four waves per column block
  constexpr int NW = NB * RG;
  constexpr int T = 4 / RG;                         // 16-row tiles per wave and stage
  constexpr int PW = 24 / NW;                       // 1 KiB LDS-DMA pieces per wave and stage
  constexpr int L = KIND == 3 ? 6 : (KIND == 5 ? 3 : 0);      // epilogue input loads per tile
  constexpr int S = (KIND == 3 || KIND == 6) ? 6 : 3;         // stores per tile
  constexpr int ITER_MIN = T * (L + S) + PW;        // vector-memory operations a wave issues per iteration (the column-sum stores not counted: lower bound)
  static_assert(KIND != 5 || RG == 1, "the fused column sums want whole 64-row stages per wave");

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nb = wave % NB, rg = wave / NB;
  const int g = lane >> 4, c = lane & 15;
  float* const bias_s = reinterpret_cast<float*>(smem + WR_LDS);

  const int nstages = (p.M + 63) >> 6;
  const int G = gridDim.x;                          // <= nstages
  const int n_it = (nstages - (int)blockIdx.x + G - 1) / G;

  // ---- activation requests: piece q of a stage's 24 = image q >> 3, rows 8 (q & 7) + [0, 8); lane = (row, 16-byte slot)
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
  const uint32_t lda2 = (uint32_t)p.lda * 2u;
  uint32_t rq[PW];
#pragma unroll
  for (int e = 0; e < PW; ++e) {
    const int q = wave * PW + e;
    const int row = 8 * (q & 7) + (lane >> 3);
    const int chunk = (lane & 7) ^ (row & 7);
    rq[e] = (uint32_t)row * lda2 + (uint32_t)((q >> 3) * 64 + chunk * 8) * 2u;
  }
  auto request = [&](int slot, int stage, bool valid) {
    char* base = smem + slot * WR_STAGE + wave * PW * 1024;
    const uint32_t off = (uint32_t)stage * 64u * lda2;
#pragma unroll
    for (int e = 0; e < PW; ++e) lds_dma16<true>(ra, base + e * 1024, valid ? rq[e] + off : 0xFFFFFFF0u);
  };
  // PIPE (the kinds without epilogue inputs: 1, 6, 8): the MFMAs of tile n + 1 are issued BETWEEN the epilogue instructions of tile n, out of
  // a second accumulator set.  Measured on the straight form: a launch costs MFMA time + epilogue VALU time + store time (qkv forward
  // 48 us = 17 + 20 + 11; fc1 forward 113 = 18 + 67 + 28) -- the two waves of a SIMD run in lock-step and want the same pipe at the same time.
  constexpr bool PIPE = L == 0;
  request(0, blockIdx.x, true);
  request(1, blockIdx.x + G, 1 < n_it);
  request(2, blockIdx.x + 2 * G, 2 < n_it);
  if constexpr (PIPE) request(3, blockIdx.x + 3 * G, 3 < n_it);

  // ---- the resident weight block: wf[t][kk] = W[n(t, c)][32 kk + 8 g .. + 7]
  bf16x8 wf[6][6];
  {
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const int n = nb * 96 + 32 * (t >> 1) + 8 * (c >> 2) + 4 * (t & 1) + (c & 3);
#pragma unroll
      for (int kk = 0; kk < 6; ++kk)
        wf[t][kk] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rb, ((uint32_t)n * (uint32_t)p.ldb + (uint32_t)(32 * kk + 8 * g)) * 2u, 0, 0));
    }
  }
  if constexpr (KIND != 5)
    for (int n = threadIdx.x; n < p.N; n += 64 * NW) bias_s[n] = p.bias ? p.bias[n] : 0.f;

  // ---- epilogue addressing: lane = row c of the tile, columns nb * 96 + 32 pp + 8 g + [0, 8) of pair pp
  const float alpha = p.alpha;
  const int col0 = nb * 96;
  [[maybe_unused]] __amdgpu_buffer_rsrc_t ro, rx, rr;
  [[maybe_unused]] uint32_t vo = 0, vx = 0, vr = 0, ldo2 = 0, ldx2 = 0, ldr4 = 0;
  if constexpr (KIND == 3) {
    ro = make_rsrc(p.out_f32, (uint32_t)((((int64_t)p.M - 1) * p.ldo_f32 + p.N) * 4));
    rr = make_rsrc(p.residual, (uint32_t)((((int64_t)p.M - 1) * p.ldr + p.N) * 4));
    ldo2 = (uint32_t)p.ldo_f32 * 4u; ldr4 = (uint32_t)p.ldr * 4u;
    vo = (uint32_t)c * ldo2 + (uint32_t)(col0 + 8 * g) * 4u;
    vr = (uint32_t)c * ldr4 + (uint32_t)(col0 + 8 * g) * 4u;
  } else {
    ro = make_rsrc(p.out_bf16, (uint32_t)((((int64_t)p.M - 1) * p.ldo_bf16 + p.N) * 2));
    ldo2 = (uint32_t)p.ldo_bf16 * 2u;
    vo = (uint32_t)c * ldo2 + (uint32_t)(col0 + 8 * g) * 2u;
    if constexpr (KIND == 5 || KIND == 6) {
      rx = make_rsrc(KIND == 5 ? (const void*)p.aux_in : (const void*)p.aux_out, (uint32_t)((((int64_t)p.M - 1) * p.ldaux + p.N) * 2));
      ldx2 = (uint32_t)p.ldaux * 2u;
      vx = (uint32_t)c * ldx2 + (uint32_t)(col0 + 8 * g) * 2u;
    }
  }
  // epilogue inputs of one tile (kind 3: the residual rows, 6 x 16 B; kind 5: the aux rows, 3 x 16 B), requested one tile ahead
  constexpr int NIN = KIND == 3 ? 6 : (KIND == 5 ? 3 : 1);
  u32x4 in_cur[NIN], in_next[NIN];
  // Every offset is per-lane offset + a COMPILE-TIME constant (the instruction's immediate field), never an SGPR:
  //  * hipcc keeps two wait states between a store of more than 8 bytes and a VALU write of its data registers -- but only when the store's
  //    scalar-offset field holds no register; with an SGPR there it assumes the hardware needs none and pads nothing.  On gfx950 that
  //    did not hold: with the column block's offset in an SGPR, `buffer_store_dwordx4 v[166:169], .., s9 offen` followed by
  //    `v_pk_mul ..; v_and_b32 v167, ..` stored the NEW v167 for lanes 12 - 15 of every row -- dword 1 of the lanes whose data is read
  //    last (kind 5: 1.2 % of the outputs wrong while the column sums of the same values were right).  Without a register in that
  //    field the compiler's own rule applies (`store; v_pk_mul; s_nop 0; v_and`).
  //  * rows past M -- of the last stage, and of stages past the end that are requested ahead -- fall out of the resource's range.
  auto load_inputs = [&](int rowbase, u32x4 (&in)[NIN]) {
    if constexpr (KIND == 3) {
      const uint32_t vt = vr + (uint32_t)rowbase * ldr4;
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) {
        in[2 * pp] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, vt + pp * 128u, 0, 0));
        in[2 * pp + 1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, vt + (pp * 128u + 16u), 0, 0));
      }
    } else if constexpr (KIND == 5) {
      const uint32_t vt = vx + (uint32_t)rowbase * ldx2;
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) in[pp] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, vt + pp * 64u, 0, 0));
    }
  };
  load_inputs((int)blockIdx.x * 64 + 16 * (RG == 1 ? 0 : rg), in_cur);

  // X fragment offsets inside a stage: row c of the tile, 16-byte chunk 4 ks + g of the image, XOR-ed with the row
  const int xoff0 = c * 128 + ((g ^ (c & 7)) << 4), xoff1 = c * 128 + (((4 + g) ^ (c & 7)) << 4);

  [[maybe_unused]] float cs[3][8];
  if constexpr (KIND == 5) {
#pragma unroll
    for (int pp = 0; pp < 3; ++pp)
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[pp][e] = 0.f;
  }

  // the prologue (three stages, the weights, the first tile's inputs) has landed.  The weight registers are then made opaque: the compiler
  // must not carry "this load may still be in flight" into the loop -- it sank the loads below a plain wait and protected their first
  // use with counted waits INSIDE the loop (vmcnt(8) once per iteration: the whole queue drained every 64 rows)
  wait_vm<0>();
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) asm volatile("" : "+v"(wf[t][kk]));
  if constexpr (PIPE) {
    // VALU instructions of this tile's epilogue per MFMA of the next tile
    constexpr int VPM = KIND == 6 ? 8 : (KIND == 8 ? 6 : 1);
    // One pipeline step: the 36 MFMAs of the NEXT tile (X fragments of rows [16 tn, 16 tn + 16) of the stage at `stg`, into `an`) in six groups of
    // six, and between them the six chunks of THIS tile's epilogue (accumulators `ac`; chunk h = columns 4 (h & 1) + [0, 4) of pair h >> 1).
    // Every group is its own scheduling region (sched_barrier) with an MFMA : VALU pattern inside it.
    auto step = [&](f32x4 (&an)[6], f32x4 (&ac)[6], const char* stg, int tn, int rowbase, bool first_only) {
      const char* xb = stg + tn * 2048;
#pragma unroll
      for (int t = 0; t < 6; ++t) an[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) {
        // (pair granularity: eight independent GELU chains per region -- with four, the dependent packed-fp32 chains of the polynomial
        // left the VALU waiting on its own results: 130 us against 113 for fc1 forward)
        const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(xb + pp * WR_IMG + xoff0);
        const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(xb + pp * WR_IMG + xoff1);
#pragma unroll
        for (int t = 0; t < 6; ++t) an[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][2 * pp], x0, an[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 6; ++t) an[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][2 * pp + 1], x1, an[t], 0, 0, 0);
        if (!first_only) {
          float v[8], dy[8];
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_s + col0 + 32 * pp + 8 * g);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias_s + col0 + 32 * pp + 8 * g + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = fmaf(ac[2 * pp][e], alpha, b0[e]);
            v[4 + e] = fmaf(ac[2 * pp + 1][e], alpha, b1[e]);
          }
          if constexpr (KIND == 6 || KIND == 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float y;
              gelu_pair(v[e], y, dy[e]);
              v[e] = y;
            }
            if constexpr (KIND == 6) wr_store16<NT>(wr_pack8(dy), rx, vx + (uint32_t)rowbase * ldx2, pp * 64);
          }
          wr_store16<NT>(wr_pack8(v), ro, vo + (uint32_t)rowbase * ldo2, pp * 64);
#pragma unroll
          for (int q = 0; q < 12; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);                 // VPM VALU
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    __syncthreads();                                 // stage 0 and the bias table are in
    f32x4 accA[6], accB[6];
    step(accA, accB, smem, RG == 1 ? 0 : rg, 0, true);
#pragma clang loop unroll(disable)
    for (int i = 0; i < n_it; ++i) {
      const int sidx = (int)blockIdx.x + i * G;
#pragma unroll
      for (int j = 0; j < T; ++j) {
        const int tj = RG == 1 ? j : rg;
        const int rowbase = sidx * 64 + 16 * tj;
        if (j == T - 1) {
          // the next tile opens stage i + 1: requested at the barrier of stage i - 2 (the first four in the prologue); since then this wave
          // has issued at least 3 T S + 2 PW vector-memory operations.  Past the barrier nobody reads stage i any more: its slot takes stage i + 4.
#if !(WR_EXP & 8)
          wait_vm<3 * T * S + 2 * PW>();
#endif
          __syncthreads();
          request(i & 3, sidx + 4 * G, i + 4 < n_it);
        }
        const char* stg = smem + ((j == T - 1 ? i + 1 : i) & 3) * WR_STAGE;
        const int tn = RG == 1 ? (j + 1) % T : rg;
        // (tiles alternate between the two accumulator sets; T is even or 1 -- with T = 1 the sets swap per iteration through a copy)
        if ((T == 1) || !(j & 1)) step(accB, accA, stg, tn, rowbase, false);
        else step(accA, accB, stg, tn, rowbase, false);
        if constexpr (T == 1) {
#pragma unroll
          for (int t = 0; t < 6; ++t) accA[t] = accB[t];
        }
      }
    }
    wait_vm<0>();
    return;
  }
#pragma clang loop unroll(disable)
  for (int i = 0; i < n_it; ++i) {
#if !(WR_EXP & 8)
    if (i > 0) wait_vm<S + 2 * ITER_MIN>();
#endif
             // stage i was requested in iteration i - 3, ahead of that iteration's last stores
    __syncthreads();
    const int slot = i & 3;
    const int sidx = (int)blockIdx.x + i * G;
    const char* st = smem + slot * WR_STAGE;
#pragma unroll
    for (int j = 0; j < T; ++j) {
      const int tj = RG == 1 ? j : rg;
      const int rowbase = sidx * 64 + 16 * tj;
      f32x4 acc[6];
#pragma unroll
      for (int t = 0; t < 6; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const char* xb = st + tj * 2048;
      // (one X fragment ahead, no further: left to itself the scheduler hoists all six reads -- 24 registers this kernel does not have)
      bf16x8 xf = *reinterpret_cast<const bf16x8*>(xb + xoff0);
#pragma unroll
      for (int kk = 0; kk < ((WR_EXP & 2) ? 1 : 6); ++kk) {
        bf16x8 xn = xf;
        if (kk < 5) xn = *reinterpret_cast<const bf16x8*>(xb + ((kk + 1) >> 1) * WR_IMG + (((kk + 1) & 1) ? xoff1 : xoff0));
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][kk], xf, acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        xf = xn;
      }
      // the next tile's inputs (the next stage's first tile after the last one), then -- behind them in the queue -- the stage after next's rows
      if constexpr (L > 0) load_inputs(j + 1 < T ? rowbase + 16 : (sidx + G) * 64 + 16 * (RG == 1 ? 0 : rg), in_next);
      if (j == T - 1) request((i + 3) & 3, sidx + 3 * G, i + 3 < n_it);

      // ---- epilogue of tile j.  (Its inputs are made opaque here: the bf16 -> fp32 widening of the NEXT tile's aux rows was otherwise
      // hoisted to right behind their loads -- a wait for requests just issued, and twelve more live registers.)
      if constexpr (L > 0) {
#pragma unroll
        for (int q = 0; q < NIN; ++q) asm volatile("" : "+v"(in_cur[q]));
      }
#pragma unroll
      for (int pp = 0; pp < 3; ++pp) {
        float v[8];
        if constexpr (KIND == 5) {
          const bf16x8 h = __builtin_bit_cast(bf16x8, in_cur[pp]);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] = acc[2 * pp + (e >> 2)][e & 3] * alpha * bf2f(h[e]);
            cs[pp][e] += v[e];                       // rows past M: zero operand rows times a zero-filled aux row
          }
        } else {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias_s + col0 + 32 * pp + 8 * g);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias_s + col0 + 32 * pp + 8 * g + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = fmaf(acc[2 * pp][e], alpha, b0[e]);
            v[4 + e] = fmaf(acc[2 * pp + 1][e], alpha, b1[e]);
          }
        }
        if constexpr (KIND == 3) {
          const f32x4 r0 = __builtin_bit_cast(f32x4, in_cur[2 * pp]), r1 = __builtin_bit_cast(f32x4, in_cur[2 * pp + 1]);
          const uint32_t vt = vo + (uint32_t)rowbase * ldo2;
          const f32x4 o0 = {v[0] + r0[0], v[1] + r0[1], v[2] + r0[2], v[3] + r0[3]};
          const f32x4 o1 = {v[4] + r1[0], v[5] + r1[1], v[6] + r1[2], v[7] + r1[3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), ro, vt + pp * 128u, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), ro, vt + (pp * 128u + 16u), 0, 0);
        } else {
          const int so = pp * 64;
          if constexpr (KIND == 6 || KIND == 8) {
            float dy[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float y;
#if WR_EXP & 4
              y = v[e] * 0.5f; dy[e] = v[e] + 1.0f;
#else
              gelu_pair(v[e], y, dy[e]);
#endif
              v[e] = y;
            }
            if constexpr (KIND == 6) wr_store16<NT>(wr_pack8(dy), rx, vx + (uint32_t)rowbase * ldx2, so);
          }
          wr_store16<NT>(wr_pack8(v), ro, vo + (uint32_t)rowbase * ldo2, so);
        }
      }
      if constexpr (L > 0) {
#pragma unroll
        for (int q = 0; q < NIN; ++q) in_cur[q] = in_next[q];
      }
    }
    if constexpr (KIND == 5) {
      // fc1's bias gradient: the column sums of this 64-row stage, one row of the workspace (colsum_ws_reduce_kernel adds the rows in
      // order).  Buffer stores from an SGPR resource: a 64-bit per-lane pointer kept across the loop was spilled, and its reload's
      // `vmcnt(0)` drained the whole queue once per stage.  Lanes c != 0 store out of range.
      if (p.colsum_ws) {
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(p.colsum_ws, (uint32_t)((int64_t)nstages * p.N * 4));
        const uint32_t vc = c == 0 ? ((uint32_t)sidx * (uint32_t)p.N + (uint32_t)(col0 + 8 * g)) * 4u : 0xFFFFF000u;
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) {
          float t8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { t8[e] = row16_sum(cs[pp][e]); cs[pp][e] = 0.f; }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{t8[0], t8[1], t8[2], t8[3]}), rc, vc + pp * 128u, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{t8[4], t8[5], t8[6], t8[7]}), rc, vc + (pp * 128u + 16u), 0, 0);
        }
      }
    }
  }
  wait_vm<0>();                                      // (the out-of-range requests of the last iterations)
}

template <int KIND, int NB>
int launch_one(const GemmParams& p, int grid, hipStream_t stream) {
  constexpr int threads = 64 * (NB == 2 ? 8 : NB);
  const int lds = WR_LDS + p.N * (int)sizeof(float);
  static bool cfg = false;
  if (!cfg) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wreg_kernel<KIND, NB, true>), hipFuncAttributeMaxDynamicSharedMemorySize, WR_LDS + 768 * 4) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wreg_kernel<KIND, NB, false>), hipFuncAttributeMaxDynamicSharedMemorySize, WR_LDS + 768 * 4) != hipSuccess) {
      sa_set_error("sa_gemm_bf16(weights in registers): %d bytes of LDS refused", WR_LDS + 768 * 4);
      return 2;
    }
    cfg = true;
  }
  if (p.nt_store) hipLaunchKernelGGL((gemm_wreg_kernel<KIND, NB, true>), dim3(grid), dim3(threads), lds, stream, p);
  else hipLaunchKernelGGL((gemm_wreg_kernel<KIND, NB, false>), dim3(grid), dim3(threads), lds, stream, p);
  SA_LAUNCH_CHECK("sa_gemm_bf16(weights in registers)");
  return 0;
}

}  // namespace

// -1: not covered (the caller falls back to the tiled kernels)
int sagemm::launch_wreg(GemmParams p, hipStream_t stream) {
  if (p.K != WR_K || p.split_k != 1 || p.M < 64) return -1;
  const int nb = p.N == 192 ? 2 : (p.N == 576 ? 6 : (p.N == 768 ? 8 : 0));
  if (!nb) return -1;
  const int kind = p.epi_kind;
  const bool have = (kind == 1) || (kind == 3 && nb == 2) || ((kind == 5 || kind == 6 || kind == 8) && nb == 8);
  if (!have) return -1;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      sa_set_error("sa_gemm_bf16: cannot query the device");
      return 2;
    }
    cus = prop.multiProcessorCount;
  }
  const int nstages = (p.M + 63) / 64;
  const int slots = budget_slots(cus);
  const int grid = nstages < slots ? nstages : slots;
  // 32-bit byte offsets reach four stages past the end (requests and input loads of stages that do not exist go out of range, never wrap)
  const int64_t rows = (int64_t)p.M + 64 * (4 * (int64_t)grid + 1), lim = (int64_t)1 << 32;
  if (rows * p.lda * 2 >= lim) return -1;
  if (kind == 3 && (rows * p.ldo_f32 * 4 >= lim || rows * p.ldr * 4 >= lim)) return -1;
  if (kind != 3 && rows * p.ldo_bf16 * 2 >= lim) return -1;
  if ((kind == 5 || kind == 6) && rows * p.ldaux * 2 >= lim) return -1;
  if ((int64_t)p.N * p.ldb * 2 >= lim) return -1;
#define SA_WREG_CASE(KD, NBV) if (kind == KD && nb == NBV) return launch_one<KD, NBV>(p, grid, stream);
  SA_WREG_CASE(1, 2) SA_WREG_CASE(1, 6) SA_WREG_CASE(1, 8)
  SA_WREG_CASE(3, 2)
  SA_WREG_CASE(5, 8) SA_WREG_CASE(6, 8) SA_WREG_CASE(8, 8)
#undef SA_WREG_CASE
  return -1;
}
