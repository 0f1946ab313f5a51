// Shared pieces of the bf16 MFMA GEMM translation units (gemm_bf16.hip: dispatch + the general kernels; gemm_phase.hip: the
// phased persistent 256 x 256 kernel; gemm_stream.hip: the 192 x 192 streaming split-K kernel): problem description, operand staging, fragment reads and the fused epilogues.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "../../include/ssl_audio_hip.h"

namespace sagemm {
// CUs the persistent kernels may occupy (0 = all).  Under data parallelism RCCL's collective kernels hold a few CUs for the
// length of an all-reduce; a persistent grid sized to the whole chip would then run its last workgroups as a second wave and
// take twice as long, so the host reserves those CUs (sa_set_cu_budget) instead.
extern int g_cu_budget;
inline int budget_slots(int cus, int per_cu = 1) {
  const int use = (g_cu_budget > 0 && g_cu_budget < cus) ? g_cu_budget : cus;
  return use * per_cu;
}

struct GemmParams {
  const char* A; const char* B;
  uint32_t a_bytes, b_bytes;       // buffer extents for the hardware bounds check
  int lda, ldb;
  int M, N, K;
  float alpha;
  const float* bias;
  int act;
  const bf16_t* aux_in; bf16_t* aux_out; int64_t ldaux;
  const float* residual; int64_t ldr; int res_mod;
  float* out_f32; int64_t ldo_f32;
  bf16_t* out_bf16; int64_t ldo_bf16;
  int row_group;
  int split_k; int accumulate;
  float* split_ws;   // deterministic split-K: slice s stores alpha * partial into split_ws[(s * M + m) * N + n] instead of atomically adding to out_f32
  int tiles_m, tiles_n;
  int gm;   // row-panels per tile group (L2 locality knob)
  int gm256;         // row-panels per tile group in the 256^2 kernels
  int stagger;       // persistent 256^2 kernel: wave row 1 requests its LDS-DMA share mid-step
  int ring_phase;    // ring kernel: requests phased by wave row
  int epi_kind;      // 1..4: one of the compact epilogues applies (wave_epilogue_compact); 0: general epilogue
  int nt_store;      // bf16 outputs with non-temporal stores
  float* colsum_ws;  // [ceil(M/64)][N] partial column sums of the fp32 epilogue result (bias gradient of the producing Linear), or null
  float* asum_out; float* asum_ws; int asum_lo, asum_hi;   // streaming split-K kernels: column sums of A (see StreamProb)
  int* tile_counter; // dynamic tile hand-out of the persistent kernels: a zeroed device word; workgroup b starts on tile b and draws every
                     // later tile as gridDim.x + atomicAdd(tile_counter, 1).  null: static striding (tile b, b + grid, ...)
};

// A zeroed ticket word for one launch (hipMemsetAsync on `stream`, capturable), or null when dynamic hand-out is off.
int* next_tile_counter(hipStream_t stream);


// gemm_phase.hip: launches the phased kernel for (layout, split) if it covers the problem (compact epilogue kinds 1/3/5/6 or split-K
// atomics); returns -1 when it does not, 0 on success, 2 on a launch error
int launch_phase(GemmParams p, bool a_kmajor, bool b_kmajor, bool split, hipStream_t stream);
// gemm_stream.hip: the 192 x 192 streaming split-K kernel (both operands k-strided, split_k > 1); 0 on success, 2 on a launch error.
// One launch serves up to STREAM_MAX_PROBLEMS products over the SAME reduction (a transformer block's four weight gradients): the
// workgroups -- and with them the fp32 partial tiles the split costs, one per workgroup -- are shared out over all of them.
constexpr int STREAM_MAX_PROBLEMS = 8;
struct StreamProb {
  const char* A; const char* B;    // [K][M], [K][N] bf16 (k-strided)
  float* ws;                       // [split_k][M][N] partials, or null: fp32 atomics into out
  float* out; int64_t ldo;
  uint32_t a_bytes, b_bytes;
  int lda, ldb, M, N;
  int tile0, tiles_m, rblock0;     // filled by launch_stream_group: first tile / first reduce block of this product
  float* asum_out; float* asum_ws; // optional: asum_out[m] += sum_k A[k][m] (bias gradient), partials [split_k][M] in the workspace form
  int asum_lo, asum_hi;            // rows [asum_lo, asum_hi) are left alone
  int ablock0;                     // filled by launch_stream_group: first reduce block of the row sums (512 rows per block)
};
struct StreamGroup {
  StreamProb pr[STREAM_MAX_PROBLEMS];
  int n, ntiles, K, split_k;
  float alpha;
};
int launch_stream(GemmParams p, hipStream_t stream);
int launch_stream_group(StreamGroup& g, int tile /* 192 or 256 */, hipStream_t stream);
int launch_stream_reduce(const StreamGroup& g, hipStream_t stream);
int launch_stream256(GemmParams p, hipStream_t stream);
// gemm_pair.hip: the 128 x 256 kernel with two workgroups per CU (both operands k-major, compact epilogue kinds 1/3/5/6/8); -1: not covered
int launch_pair(GemmParams p, hipStream_t stream);
// gemm_wreg.hip: the weights-in-registers streaming kernel for thin forward-layout products (K = 192, N in {192, 576, 768}, compact
// epilogue kinds 1 / 3 / 5 / 6 / 8: ViT-T's qkv / proj / fc1 forward, fc2 / proj data gradients); -1: not covered
int launch_wreg(GemmParams p, hipStream_t stream);
}  // namespace sagemm
using sagemm::GemmParams;
using sagemm::budget_slots;

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand tile
constexpr int NTHREADS = 256;

// 16-byte-chunk XOR for a k-strided tile row: rows {0..3, 8..11} (one 32-lane half of a transposing
// read) land on 8 distinct 32-byte slots of the 256-byte bank row.
__device__ __forceinline__ int ks_swz(int krow) { return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 1; }

// One LDS-DMA request (1 KiB: 16 bytes per lane, lane-linear in LDS from `lds`).  HIDDEN = true issues it through inline assembly:
// hipcc puts an `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 (the transposing read of the k-strided operand layouts) that
// follows a builtin LDS-DMA, because it cannot tell that the request targets the OTHER stage -- which drains the K-tile just requested
// before the current one is consumed, i.e. serialises what the double buffer is there to overlap (plain ds_read_b128 is not affected).
// A request the compiler does not see gets no such wait; the kernels' own `s_waitcnt vmcnt` + barrier still order it against the
// reads of its stage, and the compiler's counted waits for other memory operations can only over-wait (vmcnt retires in order).
template <bool HIDDEN>
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, uint32_t voff) {
  if constexpr (HIDDEN) {
    // (the low 32 bits of a generic pointer into LDS are the LDS byte address: the shared aperture lives in the high half)
    const uint32_t addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(addr), "v"(voff), "s"(rsrc) : "m0");
  } else {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds), 16, voff, 0, 0, 0);
  }
}

// ---- HBM -> LDS staging of one operand tile ---------------------------------------------------------
// k-major operand: tile = 128 rows x 64 k (128 B rows).  16 wave-instructions of 1 KiB (8 rows each).
template <bool KMAJOR, int NWAVES = 4, bool HIDDEN = false>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int ld, int row0_or_col0, int k0,
                                           int wave, int lane) {
  constexpr int PER_WAVE = 16 / NWAVES;
  if constexpr (KMAJOR) {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int q = wave * PER_WAVE + i;
      const int row = q * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ (row & 7);
      const uint32_t voff = ((uint32_t)(row0_or_col0 + row) * (uint32_t)ld + (uint32_t)(k0 + chunk * 8)) * 2u;
      lds_dma16<HIDDEN>(rsrc, lds_tile + q * 1024, voff);
    }
  } else {
    // k-strided operand: tile = 64 k-rows x 128 cols (256 B rows).  16 wave-instructions (4 k-rows each).
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int q = wave * PER_WAVE + i;
      const int krow = q * 4 + (lane >> 4);
      const int chunk = (lane & 15) ^ ks_swz(krow);
      const uint32_t voff = ((uint32_t)(k0 + krow) * (uint32_t)ld + (uint32_t)(row0_or_col0 + chunk * 8)) * 2u;
      lds_dma16<HIDDEN>(rsrc, lds_tile + q * 1024, voff);
    }
  }
}

// one wave-instruction (number q of 16) of stage_tile, with a validity flag: lets a kernel space its LDS-DMA requests out between
// MFMAs; an invalid request goes out of range (zero fill, no traffic) so vmcnt bookkeeping stays exact
template <bool KMAJOR, bool HIDDEN = false>
__device__ __forceinline__ void stage_one_v(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int ld, int row0_or_col0, int k0, int q, int lane, bool valid) {
  uint32_t voff;
  if constexpr (KMAJOR) {
    const int row = q * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (row & 7);
    voff = ((uint32_t)(row0_or_col0 + row) * (uint32_t)ld + (uint32_t)(k0 + chunk * 8)) * 2u;
  } else {
    const int krow = q * 4 + (lane >> 4);
    const int chunk = (lane & 15) ^ ks_swz(krow);
    voff = ((uint32_t)(k0 + krow) * (uint32_t)ld + (uint32_t)(row0_or_col0 + chunk * 8)) * 2u;
  }
  voff = valid ? voff : 0xFFFFFFF0u;
  lds_dma16<HIDDEN>(rsrc, lds_tile + q * 1024, voff);
}

// ---- LDS -> register fragment for one 16-wide sub-tile and one 32-deep k-step -------------------------
// Returns the 8 bf16 a lane feeds to v_mfma_f32_16x16x32_bf16: element j <-> k = 8*(lane>>4) + j, for
// output index (row of A / column of B) sub0 + (lane & 15).
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 load_frag(const char* lds_tile, int sub0, int kstep, int lane) {
  if constexpr (KMAJOR) {
    const int r = sub0 + (lane & 15);
    const int kq = kstep * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds_tile + r * 128 + ((kq ^ (r & 7)) << 4));
  } else {
    const int i = lane & 15, g = lane >> 4;
    const int krow = kstep * 32 + 8 * g + (i >> 2);
    const int col = sub0 + 4 * (i & 3);
    const int off = (((col >> 3) ^ ks_swz(krow)) << 4) + ((col >> 2) & 1) * 8;  // ks_swz(krow) == ks_swz(krow + 4)
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_tile + krow * 256 + off));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_tile + (krow + 4) * 256 + off));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
}

// ---- epilogue for one accumulator fragment: lane owns row m and columns n..n+3 (swapped MFMA operand order)
__device__ __forceinline__ void epilogue_store(const GemmParams& p, const f32x4& a, int m, int n, int64_t orow, int64_t rrow) {
  if (n >= p.N) return;
  if (n + 3 >= p.N) {
    for (int r = 0; r < 4 && n + r < p.N; ++r) {
      float x = a[r] * p.alpha;
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
    return;
  }
  float v[4] = {a[0] * p.alpha, a[1] * p.alpha, a[2] * p.alpha, a[3] * p.alpha};
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

// ---- wave-private staged epilogue for a 64 x 64 wave tile (acc[4][4], swapped operand order: lane = row c, cols 4g..)
// A row-per-lane bf16 epilogue is 16 x 8-byte stores per lane, each instruction touching 16 rows x 32 B: the store
// tail is ISSUE-bound (cdna_hip_programming.md T21).  Staging the tile through 9 KiB of this wave's LDS turns it
// into 8 x 16-byte stores per lane that write whole 128-byte row segments.  fp32 outputs already store 16 B per lane.
constexpr int EPI_STRIDE = 144;                 // bytes per staged row: 64 bf16 + 16 B pad (16-byte aligned rows)
// XOR key of the swizzled (128-byte row) scratch.  Its row-per-lane side (ds_write_b64 / ds_read_b64 of 8 bytes per lane) is a 2-way bank
// conflict by construction, whatever the key: a ds_write_b64 is serviced in groups of 16 consecutive lanes with 32 banks of 4 bytes
// (MI355X_MICROARCH.md, LDS table) -- here 16 ROWS at one 8-byte column piece, and rows 128 bytes apart share their banks, so 16 pieces
// need 16 distinct 8-byte slots of a 128-byte window, but a chunk XOR leaves the piece's half (g & 1) fixed: 8 slots.  Spreading them
// over 16 slots means swapping the two halves of a 16-byte chunk for half of the rows, which the 16-byte read side would have to undo
// with 4 v_cndmask per read -- in an epilogue that is VALU-bound.  Round 5 measured the conflicts as free: a key that removes the
// conflict under 64-bank, 32-lane servicing ((row >> 1) & 7) left both SQ_LDS_BANK_CONFLICT (3.825e6 per launch set) and the block
// time (2.794 / 2.812 vs 2.808 / 2.797 ms) where they were (VERDICT r4 #4a).
__device__ __forceinline__ int epi_key(int row) { return row & 7; }
constexpr int EPI_BYTES = 64 * EPI_STRIDE;      // 9216 B per wave

// SWZ = false: padded rows (144 B, 9 KiB per wave).  SWZ = true: 128-byte rows with the 16-byte chunk XOR-ed by epi_key(row),
// exactly 8 KiB per wave -- four waves fit one 32 KiB pipeline stage (the persistent kernel stages its epilogue in the
// stage it has just finished reading while the other stage already receives the next tile).
// NI = 16-row groups handled per call: 4 (a 64-row block, 8 KiB of scratch) or 2 (32 rows, 4 KiB: the phased kernel, whose
// pipeline owns 128 of the 160 KiB of LDS).
template <bool SWZ, int XF = 0, int NI = 4>   // XF = 1: store GELU'(v) and REPLACE v by GELU(v) (forward fc1 with act = 3; one exp + one rcp for both); XF = 2: store GELU(v)
__device__ __forceinline__ void staged_store_bf16(const GemmParams& p, f32x4 (&v)[NI][4], char* wlds, bf16_t* dst, int64_t ld,
                                                  int m_base, int n_base, bool remap, int lane) {
  const int g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x4 h = {f2bf(v[i][j][0]), f2bf(v[i][j][1]), f2bf(v[i][j][2]), f2bf(v[i][j][3])};
      if constexpr (XF == 1) {
        float dy[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float y;
          gelu_pair(v[i][j][r], y, dy[r]);
          v[i][j][r] = y;
        }
        h = bf16x4{f2bf(dy[0]), f2bf(dy[1]), f2bf(dy[2]), f2bf(dy[3])};
      } else if constexpr (XF == 2) {
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float dy;
          gelu_pair(v[i][j][r], y[r], dy);                  // (the derivative's two extra operations are dead code here)
        }
        h = bf16x4{f2bf(y[0]), f2bf(y[1]), f2bf(y[2]), f2bf(y[3])};
      }
      const int row = i * 16 + c;
      if constexpr (SWZ)
        *reinterpret_cast<bf16x4*>(wlds + row * 128 + (((j * 2 + (g >> 1)) ^ epi_key(row)) << 4) + (g & 1) * 8) = h;
      else
        *reinterpret_cast<bf16x4*>(wlds + row * EPI_STRIDE + (j * 16 + 4 * g) * 2) = h;
    }
  const int ch = lane & 7;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  if (!(remap && p.row_group > 0)) {
    // Buffer stores: ONE per-lane offset for the whole call, the row block's offset in an SGPR, rows past M dropped by the resource's
    // bounds check -- no per-store 64-bit address arithmetic, compares or exec masks (they were 40 % of this epilogue's VALU issue).
    // (the host checks that the output spans less than 4 GiB, as it does for the operands)
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(dst, (uint32_t)((((int64_t)p.M - 1) * ld + p.N) * 2));
    const uint32_t ld2 = (uint32_t)ld * 2u;
    // The row block's offset travels in the PER-LANE offset (one v_add per store), not in the instruction's scalar-offset field: hipcc keeps
    // two wait states between a 16-byte store and a VALU write of its data registers only when that field holds NO register -- with an
    // SGPR there it assumes the hardware needs none, and on gfx950 that measurably did not hold (gemm_wreg.hip's first build stored a
    // later value for lanes 12 - 15: DESIGN.md section 6, round 5, item 11).  These stores were never caught doing it
    // (tests/test_kernels_gpu.py::test_gemm_elementwise_at_step_shapes), but the ISA had the unprotected pattern at 94 sites
    // (scripts/diag/scan_store_hazard.py); in this form the compiler's own rule pads them.
    // -DSA_STORE_SOFF=1 brings the scalar-offset form back (A/B builds).
#ifndef SA_STORE_SOFF
#define SA_STORE_SOFF 0
#endif
    uint32_t voff = (uint32_t)(lane >> 3) * ld2 + (uint32_t)ch * 16u;
    uint32_t soff = (uint32_t)m_base * ld2 + (uint32_t)n_base * 2u;
#if !SA_STORE_SOFF
    voff += soff;
    soff = 0;
#endif
    const bool nt = p.nt_store != 0;
#pragma unroll
    for (int it = 0; it < 2 * NI; ++it) {
      const int r = it * 8 + (lane >> 3);
      const uint4 q = SWZ ? *reinterpret_cast<const uint4*>(wlds + r * 128 + ((ch ^ epi_key(r)) << 4))
                          : *reinterpret_cast<const uint4*>(wlds + r * EPI_STRIDE + ch * 16);
      const u32x4 val = {q.x, q.y, q.z, q.w};
      // streaming stores: a tile round writes 4 MiB per XCD, i.e. the whole L2, and would evict the operand panels
#if SA_STORE_SOFF
      if (nt) __builtin_amdgcn_raw_buffer_store_b128(val, rs, voff, soff, 2);
      else __builtin_amdgcn_raw_buffer_store_b128(val, rs, voff, soff, 0);
      soff += 8u * ld2;
#else
      if (nt) __builtin_amdgcn_raw_buffer_store_b128(val, rs, voff, 0, 2);
      else __builtin_amdgcn_raw_buffer_store_b128(val, rs, voff, 0, 0);
      voff += 8u * ld2;
#endif
    }
    return;
  }
#pragma unroll
  for (int it = 0; it < 2 * NI; ++it) {
    const int r = it * 8 + (lane >> 3);
    const int m = m_base + r;
    const uint4 val = SWZ ? *reinterpret_cast<const uint4*>(wlds + r * 128 + ((ch ^ epi_key(r)) << 4))
                          : *reinterpret_cast<const uint4*>(wlds + r * EPI_STRIDE + ch * 16);
    if (m < p.M) {
      const int64_t orow = (int64_t)m + m / p.row_group + 1;
      *reinterpret_cast<uint4*>(dst + orow * ld + n_base + ch * 8) = val;
    }
  }
}

// Sum over the 16 lanes of a DPP row (lanes with equal lane >> 4), result in every lane: four v_add_f32 with a DPP operand -- lane ^ 1,
// lane ^ 2 inside a quad, then the mirrored half row and the mirrored row, whose lanes already hold their quad's / half row's sum.
// Same additions in the same order as the xor butterfly `t += __shfl_xor(t, o)` for o = 1, 2, 4, 8 (bit-identical), without its four
// ds_bpermute round trips through the LDS pipe.
__device__ __forceinline__ float row16_sum(float t) {
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x141, 0xF, 0xF, true));   // row_half_mirror
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x140, 0xF, 0xF, true));   // row_mirror
  return t;
}

// Mirror of staged_store_bf16 for an epilogue INPUT: a row-per-lane read of aux_in touches 16 rows x 32 B per instruction (a quarter
// of every 128-byte line it pulls in); going through the wave's LDS scratch reads whole 128-byte row segments instead.  Swizzled layout only.
template <int NI = 4>
__device__ __forceinline__ void staged_load_bf16(const GemmParams& p, char* wlds, const bf16_t* src, int64_t ld, int m_base, int n_base, int lane,
                                                 const uint4* pre = nullptr) {
  // pre: the 2 * NI row segments of this lane, already loaded by the caller (all of a tile's epilogue inputs are requested in one
  // batch before its first store: a load issued after a store waits for that store in the in-order vmcnt queue)
  const int ch = lane & 7;
#pragma unroll
  for (int it = 0; it < 2 * NI; ++it) {
    const int r = it * 8 + (lane >> 3);
    const int m = m_base + r;
    uint4 val = make_uint4(0u, 0u, 0u, 0u);
    if (pre) val = pre[it];
    else if (m < p.M) val = *reinterpret_cast<const uint4*>(src + (int64_t)m * ld + n_base + ch * 8);
    *reinterpret_cast<uint4*>(wlds + r * 128 + ((ch ^ epi_key(r)) << 4)) = val;
  }
}

// Compact epilogues for the four combinations the ViT blocks use (whole 64-column block inside N, aligned outputs).  The general
// epilogue below is thousands of instructions of mostly-untaken paths with spilled scalars; a once-per-tile walk through it was
// measured at 15 k cycles per 256 x 256 tile even for a bias-only epilogue, against 9 k for the compact path.
//   kind 1: alpha * acc + bias -> bf16                                   (qkv forward, every dgrad without an activation)
//   kind 2: alpha * acc + bias -> aux_out (bf16), GELU -> bf16           (fc1 forward)
//   kind 3: alpha * acc + bias + residual -> fp32                        (proj / fc2 forward)
//   kind 4: alpha * acc * GELU'(aux_in) -> bf16 (+ column sums)          (fc2 dgrad, act 2: aux holds the pre-activation)
//   kind 5: alpha * acc * aux_in -> bf16 (+ column sums)                 (fc2 dgrad, act 4: aux already holds GELU')
//   kind 6: alpha * acc + bias -> aux_out = GELU'(.), bf16 = GELU(.)     (fc1 forward, act 3)
//   kind 8: alpha * acc + bias -> bf16 = GELU(.)                          (fc1 forward WITHOUT a backward to come: act 1, no aux -- the
//           EMA target network of main_bt_byol.py:97-101, encode_vit / the HEAR wrappers; until round 5 the general epilogue)
// NI / CSM: see staged_store_bf16; with NI = 2 a 64-row block is two calls and the fused column sums travel between them in
// `csacc` (CSM 1: first half, keep the sums; CSM 2: second half, add and write the workspace row; CSM 0: whole block at once).
template <bool SWZ, int kind, int NI = 4, int CSM = 0>
__device__ __forceinline__ void wave_epilogue_compact(const GemmParams& p, f32x4 (&acc)[NI][4], int m_base, int n_base, char* wlds, int lane,
                                                      float (*csacc)[4] = nullptr, const float4* bias4 = nullptr, const uint4* aux_pre = nullptr) {
  // bias4: the four float4 of this lane's bias columns, loaded ONCE per tile by the caller.  A load inside this function sits
  // behind the previous call's stores in the in-order vmcnt queue, so waiting for it drains them (an HBM round trip per call).
  if (n_base >= p.N) return;                    // N is a multiple of 64 here: a 64-column block is wholly inside or wholly outside
  const int g = lane >> 4, c = lane & 15;
  const float alpha = p.alpha;
  if constexpr (kind != 4 && kind != 5) {   // (kinds 1, 2, 3, 6 start from alpha * acc + bias)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias4) b = bias4[j];
      else if (p.bias) b = *reinterpret_cast<const float4*>(p.bias + n_base + j * 16 + 4 * g);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        acc[i][j][0] = acc[i][j][0] * alpha + b.x; acc[i][j][1] = acc[i][j][1] * alpha + b.y;
        acc[i][j][2] = acc[i][j][2] * alpha + b.z; acc[i][j][3] = acc[i][j][3] * alpha + b.w;
      }
    }
  }
  if constexpr (kind == 1) {
    staged_store_bf16<SWZ, 0, NI>(p, acc, wlds, p.out_bf16, p.ldo_bf16, m_base, n_base, false, lane);
  } else if constexpr (kind == 8) {
    staged_store_bf16<SWZ, 2, NI>(p, acc, wlds, p.out_bf16, p.ldo_bf16, m_base, n_base, false, lane);
  } else if constexpr (kind == 6) {          // fc1 forward, act 3: aux <- GELU'(v), acc <- GELU(v) in one pass, then the activation
    staged_store_bf16<SWZ, 1, NI>(p, acc, wlds, p.aux_out, p.ldaux, m_base, n_base, false, lane);
    staged_store_bf16<SWZ, 0, NI>(p, acc, wlds, p.out_bf16, p.ldo_bf16, m_base, n_base, false, lane);
  } else if constexpr (kind == 2) {
    staged_store_bf16<SWZ, 0, NI>(p, acc, wlds, p.aux_out, p.ldaux, m_base, n_base, false, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = gelu_f(acc[i][j][r]);
      __builtin_amdgcn_sched_barrier(0);               // 16 activations at a time: 64 interleaved erf chains spill
    }
    staged_store_bf16<SWZ, 0, NI>(p, acc, wlds, p.out_bf16, p.ldo_bf16, m_base, n_base, false, lane);
  } else if constexpr (kind == 3) {
    // residual and output may alias as far as the compiler knows, so a load written after a store stays after it -- and waiting
    // for that load (the youngest op, vmcnt(0)) then waits for the store as well: 32 serialised HBM round trips per tile.  All
    // eight residual loads of a 32-row group go out first, then the eight stores.
#pragma unroll
    for (int i0 = 0; i0 < NI; i0 += 2) {
      float4 rv[2][4];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int m = m_base + (i0 + ii) * 16 + c;
        const float* rrow = p.residual + (int64_t)(m < p.M ? m : 0) * p.ldr + n_base + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) rv[ii][j] = *reinterpret_cast<const float4*>(rrow + j * 16);
      }
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = i0 + ii;
        const int m = m_base + i * 16 + c;
        if (m >= p.M) continue;
        float* orow = p.out_f32 + (int64_t)m * p.ldo_f32 + n_base + 4 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<float4*>(orow + j * 16) = make_float4(acc[i][j][0] + rv[ii][j].x, acc[i][j][1] + rv[ii][j].y,
                                                                  acc[i][j][2] + rv[ii][j].z, acc[i][j][3] + rv[ii][j].w);
      }
    }
  } else {
    static_assert(SWZ, "the compact epilogues stage through the swizzled 8 KiB scratch");
    staged_load_bf16<NI>(p, wlds, p.aux_in, p.ldaux, m_base, n_base, lane, aux_pre);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int row = i * 16 + c;                       // rows past M were staged as zeros: their products are zero, nothing is stored
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x4 h = *reinterpret_cast<const bf16x4*>(wlds + row * 128 + (((j * 2 + (g >> 1)) ^ epi_key(row)) << 4) + (g & 1) * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = acc[i][j][r] * alpha * (kind == 5 ? bf2f(h[r]) : dgelu_f(bf2f(h[r])));
      }
    }
    if (p.colsum_ws && (m_base & ~63) < p.M) {       // (the 64-row block owns one workspace row; with NI = 2 both halves take part)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float cs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = 0.f;
#pragma unroll
          for (int i = 0; i < NI; ++i) t += acc[i][j][r];      // rows past M: zero operand rows times a zero-filled aux row, exactly 0
          t = row16_sum(t);
          if constexpr (CSM == 2) t += csacc[j][r];
          cs[r] = t;
          if constexpr (CSM == 1) csacc[j][r] = t;
        }
        if constexpr (CSM != 1)
          if (c == 0) *reinterpret_cast<float4*>(p.colsum_ws + (int64_t)(m_base >> 6) * p.N + n_base + j * 16 + 4 * g) = make_float4(cs[0], cs[1], cs[2], cs[3]);
      }
    }
    staged_store_bf16<SWZ, 0, NI>(p, acc, wlds, p.out_bf16, p.ldo_bf16, m_base, n_base, false, lane);
  }
}

template <bool SWZ = false>
__device__ __forceinline__ void wave_epilogue_64x64(const GemmParams& p, f32x4 (&acc)[4][4], int m_base, int n_base, char* wlds, int lane) {
  const int g = lane >> 4, c = lane & 15;
  const bool fast = (n_base + 64 <= p.N) &&
                    (!p.out_bf16 || ((p.ldo_bf16 & 7) == 0 && ((uintptr_t)p.out_bf16 & 15) == 0)) &&
                    (!p.aux_out || ((p.ldaux & 7) == 0 && ((uintptr_t)p.aux_out & 15) == 0));
  if (!fast) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m_base + i * 16 + c;
      if (m >= p.M) continue;
      const int64_t orow = p.row_group > 0 ? (int64_t)m + m / p.row_group + 1 : (int64_t)m;
      const int64_t rrow = p.res_mod > 0 ? (int64_t)(m % p.res_mod) : (int64_t)m;
#pragma unroll
      for (int j = 0; j < 4; ++j) epilogue_store(p, acc[i][j], m, n_base + j * 16 + 4 * g, orow, rrow);
    }
    return;
  }
  // v = alpha * acc + bias
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) b = *reinterpret_cast<const float4*>(p.bias + n_base + j * 16 + 4 * g);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i][j][0] = acc[i][j][0] * p.alpha + b.x; acc[i][j][1] = acc[i][j][1] * p.alpha + b.y;
      acc[i][j][2] = acc[i][j][2] * p.alpha + b.z; acc[i][j][3] = acc[i][j][3] * p.alpha + b.w;
    }
  }
  if (p.act == 3) {                                  // (act = 3 always comes with aux_out)
    staged_store_bf16<SWZ, 1>(p, acc, wlds, p.aux_out, p.ldaux, m_base, n_base, false, lane);
  } else if (p.act == 1) {
    if (p.aux_out) staged_store_bf16<SWZ>(p, acc, wlds, p.aux_out, p.ldaux, m_base, n_base, false, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = gelu_f(acc[i][j][r]);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m_base + i * 16 + c;
    if (m >= p.M) continue;
    const int64_t orow = p.row_group > 0 ? (int64_t)m + m / p.row_group + 1 : (int64_t)m;
    const int64_t rrow = p.res_mod > 0 ? (int64_t)(m % p.res_mod) : (int64_t)m;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n_base + j * 16 + 4 * g;
      if (p.act == 2 || p.act == 4) {
        const bf16x4 h = *reinterpret_cast<const bf16x4*>(p.aux_in + (int64_t)m * p.ldaux + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] *= p.act == 4 ? bf2f(h[r]) : dgelu_f(bf2f(h[r]));
      }
      if (p.residual) {
        const float4 rv = *reinterpret_cast<const float4*>(p.residual + rrow * p.ldr + n);
        acc[i][j][0] += rv.x; acc[i][j][1] += rv.y; acc[i][j][2] += rv.z; acc[i][j][3] += rv.w;
      }
      if (p.out_f32) {
        float* o = p.out_f32 + orow * p.ldo_f32 + n;
        if (p.accumulate) {
          const float4 old = *reinterpret_cast<const float4*>(o);
          acc[i][j][0] += old.x; acc[i][j][1] += old.y; acc[i][j][2] += old.z; acc[i][j][3] += old.w;
        }
        *reinterpret_cast<float4*>(o) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // keep each 16-row group's loads/stores together: hoisting all 16 residual loads spills
  }
  if (p.colsum_ws && m_base < p.M) {     // (a 64-row block wholly past M has no workspace row)
    // column sums of this 64 x 64 block over its valid rows: 4 adds in registers, a 16-lane butterfly, one row of the workspace
    float cs[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) t += (m_base + i * 16 + c < p.M) ? acc[i][j][r] : 0.f;
        t = row16_sum(t);
        cs[j][r] = t;
      }
    if (c == 0) {
      float* w = p.colsum_ws + (int64_t)(m_base >> 6) * p.N + n_base + 4 * g;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(w + j * 16) = make_float4(cs[j][0], cs[j][1], cs[j][2], cs[j][3]);
    }
  }
  if (p.nt_store == 3) {                                 // timing experiment: no staging, no stores
    asm volatile("" ::"v"(acc[0][0]), "v"(acc[3][3]), "v"(acc[1][2]), "v"(acc[2][1]));
    return;
  }
  if (p.out_bf16) staged_store_bf16<SWZ>(p, acc, wlds, p.out_bf16, p.ldo_bf16, m_base, n_base, true, lane);
}

}  // namespace
