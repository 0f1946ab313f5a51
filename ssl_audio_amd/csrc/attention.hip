// Fused multi-head self-attention forward / backward for the ViT encoder (head_dim 64, N <= 512 tokens).
// Replaces models/mae.py:133-138: softmax(q k^T * hd^-0.5) v, reading q/k/v straight out of the packed
// [rows, 3*C] qkv activation (no permute copies) and writing [rows, C] in the layout `proj` consumes.
//
// One workgroup (4 waves) per (sequence, head).  A 10 s clip is N = 249 tokens, so a whole head's K and V
// (2 x 32 KiB bf16) sit in LDS and the softmax of a query row is exact (no online rescaling): each wave
// takes 16-query tiles, keeps S^T = K Q^T for all keys in registers (query on the lane, keys in registers:
// row max / sum are in-lane plus two cross-lane steps), and feeds P^T straight back as the MFMA operand of
// O^T = V^T P^T (cdna_hip_programming.md §3 "accumulator tile as the next MFMA's operand").  V is consumed
// through ds_read_b64_tr_b16, so no transposed copy exists anywhere.
// Backward recomputes P from the saved log-sum-exp: pass A (wave owns query tiles) produces dQ, pass B
// (wave owns key tiles) produces dK and dV, so no gradient is ever summed across waves or workgroups.
// Two instantiations: NMAX = 256 (the 16 x 16-patch ViTs at 10 s: 249 tokens; 64 KiB of LDS, two workgroups per CU) and
// NMAX = 512 (the 16 x 8-patch ConvStem ViTs at 10 s: 501 tokens; the head's K and V are 128 KiB, one workgroup per CU, and the
// exact softmax keeps 512 scores per query in 128 registers).
#include <stdlib.h>
#include "common.h"
#include <type_traits>
#include "../../include/ssl_audio_hip.h"

// -DSA_ATTN_DBG=1 (A/B builds only, scripts/ab_build.sh + scripts/diag/attn_timeline.py): waves 0 and 7 of every backward workgroup leave
// cycle stamps of their phases and the id of the CU they ran on in a buffer handed in through sa_attn_dbg_set.
#ifndef SA_ATTN_DBG
#define SA_ATTN_DBG 0
#endif
#if SA_ATTN_DBG
__device__ unsigned long long* sa_attn_dbg_ptr = nullptr;
extern "C" int sa_attn_dbg_set(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(sa_attn_dbg_ptr), &buf, sizeof(buf)); }
#define ATTN_STAMP(k) { __builtin_amdgcn_sched_barrier(0); dbg_t[k] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#else
#define ATTN_STAMP(k)
#endif

namespace {

constexpr int HD = 64;          // head dim
constexpr int NMAX_ALL = 512;   // max tokens per sequence over all instantiations
constexpr int NW_FWD = 8;           // waves per workgroup, forward (64 KiB of LDS -> two workgroups = 16 waves per CU)
constexpr int NW_BWD = 8;           // backward: 8 waves per workgroup, two workgroups per CU (66 KiB of LDS each)

// ---- LDS image: [rows][64] bf16, 128-byte rows, 16-byte chunk c of row r stored at chunk c ^ (r & 7)
__device__ __forceinline__ int img_off(int row, int col) {  // col in elements, multiple of 4
  return row * 128 + ((((col >> 3) ^ (row & 7))) << 4) + ((col >> 2) & 1) * 8;
}

// stage rows [0, nrows) of a strided global matrix (row stride ld elements, starting column col0) into an image
__device__ __forceinline__ void stage_rows(__amdgpu_buffer_rsrc_t rsrc, char* img, int ld, int col0, int nrows, int wave, int lane, int nwaves) {
  const int ninstr = (nrows + 7) >> 3;
  for (int q = wave; q < ninstr; q += nwaves) {
    const int row = q * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (row & 7);
    const uint32_t voff = ((uint32_t)row * (uint32_t)ld + (uint32_t)(col0 + chunk * 8)) * 2u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(img + q * 1024), 16, voff, 0, 0, 0);
  }
}

// row read: lane gets M[row0 + (lane&15)][32*ks + 8*(lane>>4) .. +7]
__device__ __forceinline__ bf16x8 row_frag(const char* img, int row0, int ks, int lane) {
  const int r = row0 + (lane & 15);
  const int kq = ks * 4 + (lane >> 4);
  return *reinterpret_cast<const bf16x8*>(img + r * 128 + ((kq ^ (r & 7)) << 4));
}

// transposing read: lane gets M[rows][col0 + (lane&15)] for rows {lo + 4g' .. } -- precisely:
// element j < 4: row lo + 4*(lane>>4) + j ; element j >= 4: row hi + 4*(lane>>4) + (j - 4)
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int lo, int hi, int col0, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int q = i >> 2, col = col0 + 4 * (i & 3);
  const int r0 = lo + 4 * g + q, r1 = hi + 4 * g + q;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + img_off(r0, col)));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + img_off(r1, col)));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// row fragment straight from global memory (same lane mapping as row_frag): M[row0 + (lane&15)][col0 + 32*ks + 8*(lane>>4) .. +7]
__device__ __forceinline__ bf16x8 row_frag_global(__amdgpu_buffer_rsrc_t rsrc, int ld, int row0, int col0, int ks, int lane) {
  const uint32_t voff = ((uint32_t)(row0 + (lane & 15)) * (uint32_t)ld + (uint32_t)(col0 + 32 * ks + 8 * (lane >> 4))) * 2u;
  auto raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
  return __builtin_bit_cast(bf16x8, raw);
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 r = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
  return r;
}

// two fp32 -> one dword of two bf16 (v_cvt_pk_bf16_f32); building fragments from explicit pairs keeps the compiler from pairing
// neighbouring elements of different registers and re-aligning them with v_perm / v_alignbit afterwards
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) bf16_t bf16x2;
__device__ __forceinline__ uint32_t cvt_pk(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2)); }
__device__ __forceinline__ uint32_t cvt_pk(f32x2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define MFMA16(x, y, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((x), (y), (c), 0, 0, 0)

// operand prescale: the backward folds scale * log2(e) into the register-resident Q (pass A) / K (pass B) fragment, so the scores
// come out of the MFMA chain in log2 units and, with -lse as the chain's initial accumulator, p = exp2(acc) needs no further VALU
__device__ __forceinline__ bf16x8 scale_frag(const bf16x8& x, float c) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = f2bf(bf2f(x[j]) * c);
  return r;
}

// =====================================================================================================
// FULL: the sequence fills every 16-key tile and 32-key step of the instantiation (N > NMAX - 16: 249 of 256, 501 of 512 -- the 10 s
// clips), so the per-tile "is this tile in range" tests are compile-time true: in the unrolled loops they had become ~180 selects and
// moves per query tile (a third of the kernel's VALU issue) next to the ~320 the softmax itself needs.
template <int NMAX, bool FULL>
__global__ __launch_bounds__(64 * NW_FWD, NMAX == 256 ? 4 : 2) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, int64_t total_rows, int ld, int C, int H, int N,
                                                          int nq, float scale, bf16_t* __restrict__ out, int ldo, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IMG = NMAX * HD * 2;      // bytes of one [NMAX][64] bf16 LDS image
  constexpr int NKT = NMAX / 16, NKS = NMAX / 32;
  char* Kimg = smem;
  char* Vimg = smem + IMG;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = blockIdx.x / H, h = blockIdx.x % H;
  const int64_t row_base = (int64_t)s * N;
  const bf16_t* base = qkv + row_base * ld;
  const uint32_t bytes = (uint32_t)(((total_rows - row_base - 1) * ld + 3 * C) * 2);
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(base, bytes);

  const int nkt = FULL ? NKT : (N + 15) >> 4;           // 16-key tiles
  const int nks = FULL ? NKS : (N + 31) >> 5;           // 32-key steps for P.V
  const int g = lane >> 4, c = lane & 15;
  const int nqt = (nq + 15) >> 4;            // only the first nq queries of each sequence are wanted (nq = N normally)
  // Q fragments of a query tile (Y operand: k = d), one tile ahead: the first tile's are requested in front of the K / V staging, the
  // next tile's as soon as the current tile's scores are done with the registers -- a query tile no longer starts with a global round trip
  auto load_q = [&](int qt, bf16x8 (&q)[2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint32_t voff = ((uint32_t)(qt * 16 + c) * (uint32_t)ld + (uint32_t)(h * HD + 32 * ks + 8 * g)) * 2u;
      q[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
    }
  };
#if SA_ATTN_DBG
  unsigned long long dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  ATTN_STAMP(0)
  bf16x8 qf[2];
  if (wave < nqt) load_q(wave, qf);
  stage_rows(rs, Kimg, ld, C + h * HD, nks * 32, wave, lane, NW_FWD);
  stage_rows(rs, Vimg, ld, 2 * C + h * HD, nks * 32, wave, lane, NW_FWD);
  ATTN_STAMP(1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  ATTN_STAMP(2)

  // accumulator start of the last key tile: -inf on keys >= N, else 0 -- rebuilt per query tile from ONE live register (the lane's
  // first padded element; four live registers here were spilled in the FULL instantiation, which runs at the 128-register cap)
  int pad_from = N - (nkt - 1) * 16 - 4 * g;
  for (int qt = wave; qt < nqt; qt += NW_FWD) {
    const int query = qt * 16 + c;
    // S^T tiles: st[kt][r] = score(key = 16*kt + 4*g + r, query)
    f32x4 st[NKT];
    float mx = -INFINITY;
    asm volatile("" : "+v"(pad_from));                 // opaque: keeps the selects below inside the query-tile loop
    f32x4 kb;
#pragma unroll
    for (int r = 0; r < 4; ++r) kb[r] = (r >= pad_from) ? -INFINITY : 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (FULL || kt < nkt) {
        if (kt == nkt - 1) st[kt] = kb;                // padding keys start at -inf: the MFMA accumulate keeps them there
        st[kt] = MFMA16(row_frag(Kimg, kt * 16, 0, lane), qf[0], st[kt]);
        st[kt] = MFMA16(row_frag(Kimg, kt * 16, 1, lane), qf[1], st[kt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);   // raw scores; the (positive) scale is folded into the exponent
      }
    }
    if (qt + NW_FWD < nqt) load_q(qt + NW_FWD, qf);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    const float c2 = scale * 1.44269504088896340736f;            // exp(scale * (s - max)) = exp2(c2 * s - c2 * max): one fma + v_exp
    const float mb = mx * c2;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (FULL || kt < nkt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(st[kt][r], c2, -mb));   // exp2(-inf) = 0 for masked keys
          st[kt][r] = p;
          sum += p;
        }
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    // O^T[d][query] = sum_key V^T[d][key] P^T[key][query]
    f32x4 ot[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ps = 0; ps < NKS; ++ps) {
      if (FULL || ps < nks) {
        const bf16x8 pf = pack8(st[2 * ps], st[2 * ps + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ot[dt] = MFMA16(tr_frag(Vimg, 32 * ps, 32 * ps + 16, dt * 16, lane), pf, ot[dt]);
      }
    }
    if (query < nq) {
      const float inv = 1.f / sum;
      bf16_t* orow = out + (row_base + query) * ldo + h * HD;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 o = {f2bf(ot[dt][0] * inv), f2bf(ot[dt][1] * inv), f2bf(ot[dt][2] * inv), f2bf(ot[dt][3] * inv)};
        *reinterpret_cast<bf16x4*>(orow + dt * 16 + 4 * g) = o;
      }
      if (g == 0 && lse) lse[((int64_t)s * H + h) * N + query] = mx * scale + __logf(sum);
    }
  }
  // (One workgroup per head, NOT a persistent walk: the forward's per-CU timeline -- scripts/diag/attn_timeline.py ... fwd,
  // profiles/r05_attn_timeline.txt -- shows 1.57 workgroups resident on average, a slot empty ~6.5 k cycles of a 29 k-cycle period; a
  // two-per-CU grid walking the heads with a barrier per head measured 107 us against 98 - 99 us for this form: the walk's workgroups
  // load and compute in lock-step per CU.)
#if SA_ATTN_DBG
  ATTN_STAMP(5)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATTN_STAMP(6)
  dbg_t[3] = dbg_t[4] = dbg_t[2];        // (no passes A / B here: the timeline script's columns 2 .. 5 are one compute phase)
  if (sa_attn_dbg_ptr && lane == 0 && (wave == 0 || wave == NW_FWD - 1)) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    unsigned long long* o_ = sa_attn_dbg_ptr + ((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 8;
    for (int k = 0; k < 7; ++k) o_[k] = dbg_t[k];
    o_[7] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
}

// =====================================================================================================
// Backward.  Every byte of a head is read from global memory exactly once:
//   * K and V rows go to two LDS images (LDS-DMA); each wave reads Q, dO and O rows of ITS query tiles as 16-byte row fragments --
//     delta = rowsum(dO * O) falls out of those fragments (reduced over the four lanes that share a row), so there is no separate
//     statistics pass and pass A (dQ, wave owns query tiles) starts from registers;
//   * before the images are given up, each wave pulls the K, V fragments of ITS key tiles out of them (pass B's register operand);
//   * the waves then write their Q / dO fragments into the same LDS (the Q and dO images of pass B: dK, dV, wave owns key tiles).
// The version before this one re-staged Q / dO from global and fetched the pass-B fragments from global as well: 320 KB of reads
// per head instead of 160, and with the compute loops removed it still took 70 % of its time -- the kernel was bound by that traffic.
template <int NMAX>
__global__ __launch_bounds__(64 * NW_BWD, NMAX == 256 ? 4 : 2) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, int64_t total_rows, int ld, int C, int H, int N_arg,
                                                          int nq_arg, float scale, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                          int ldo, const float* __restrict__ lse, bf16_t* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int IMG = NMAX * HD * 2;                 // NMAX = 256: 2 x 32 KiB + statistics = 66 KiB -> two workgroups per CU
  constexpr int NT = NMAX / (16 * NW_BWD);           // 16-row tiles per wave (2 or 4): tile j of wave w is w + j * NW_BWD
  char* Kimg = smem;
  char* Vimg = smem + IMG;
  char* Qimg = smem;                                 // pass B reuses the two images
  char* Dimg = smem + IMG;                           // dO
  float* lse_s = reinterpret_cast<float*>(smem + 2 * IMG);   // -lse in log2 units / -delta per query: the initial accumulators of
  float* del_s = lse_s + NMAX;                                // the S and dP chains of pass B
  const int lane_id = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // The body is a loop over (sequence, head) units so that the grid can be persistent (SA_ATTN_BWD_PERSIST=1: two workgroups per CU walk
  // the list with the stride of the grid).  Measured in round 5 (scripts/diag/attn_timeline.py, profiles/r05_attn_timeline.txt): with one
  // workgroup per head a CU's slot stays empty for 5 - 8 k cycles between a workgroup's last store and its successor's entry -- 5 k of
  // them because wave 7 ends that much later than wave 0 (LDS is released when the LAST wave ends), 2 - 3 k dispatch -- which the
  // persistent walk only trades for the same skew at its loop barrier, while its static share (6 heads per workgroup, lives spread
  // 47 - 97 k cycles) ends on the slowest workgroup: 340 - 344 us against 307 - 312 us for this kernel with one workgroup per head
  // (HEAD's kernel: 317 - 324 us, same box).  Default: one head per workgroup -- the loop then runs once.
  const int n_units = (int)(total_rows / N_arg) * H;
#pragma clang loop unroll(disable)
  for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
  // Everything derived from the lane id (LDS offsets of the fragment reads, ...) is re-derived per head from an opaque copy: hoisted out of
  // the loop those values stay live across the prologue's 48 registers of Q / dO / O fragments and the kernel spills (it runs at the
  // 128-register cap of four waves per SIMD: 300 bytes of scratch per lane and 443 us instead of 317 when the compiler was left to it).
  int lane = lane_id;
  asm volatile("" : "+v"(lane));
  const int g = lane >> 4, c = lane & 15;
  // ... and so are the sequence length and the query count: as loop invariants every `tile < count` test of the unrolled passes is
  // hoisted into its own SGPR pair (~25 of them), the scalar file overflows and the prologue's fragment loads end up in scratch
  int N = N_arg, nq = nq_arg;
  asm volatile("" : "+s"(N), "+s"(nq));
  const int s = unit / H, h = unit - s * H;
  const int64_t row_base = (int64_t)s * N;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv + row_base * ld, (uint32_t)(((total_rows - row_base - 1) * ld + 3 * C) * 2));
  const __amdgpu_buffer_rsrc_t rd = make_rsrc(dout + row_base * ldo, (uint32_t)(((total_rows - row_base - 1) * ldo + C) * 2));
  const __amdgpu_buffer_rsrc_t ro = make_rsrc(o + row_base * ldo, (uint32_t)(((total_rows - row_base - 1) * ldo + C) * 2));

  const int nkt = (N + 15) >> 4;                       // 16-row tiles holding real rows
  const int nks = (N + 31) >> 5;                       // 32-row steps: tiles [0, 2 * nks) exist in the images (rows >= N hold finite junk)
  const int nqt = (nq + 15) >> 4, nqs = (nq + 31) >> 5;   // queries >= nq carry no upstream gradient (CLS-only last block)
  const float c2 = scale * 1.44269504088896340736f;    // p = exp(scale*s - lse) = exp2(c2*s - lse*log2e)
#if SA_ATTN_DBG
  unsigned long long dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  ATTN_STAMP(0)

  stage_rows(rs, Kimg, ld, C + h * HD, nks * 32, wave, lane, NW_BWD);
  stage_rows(rs, Vimg, ld, 2 * C + h * HD, nks * 32, wave, lane, NW_BWD);
  // own query tiles: Q, dO (kept until they become the pass-B images) and O (only for delta).  Rows past N read the next
  // sequence or zeros -- finite either way, and every use of them is multiplied by p = 0 or never stored.
  // (every conditionally loaded array starts defined: inside the head loop an undefined path would be merged with the PREVIOUS head's
  // registers, i.e. 64 fragment registers would stay live across the whole loop body)
  bf16x8 qf[NT][2] = {}, df[NT][2] = {};
  float nl[NT] = {}, nd[NT] = {};                      // -lse (log2 units), -delta of query tile j, row c
  {
    bf16x8 of[NT][2] = {};
    float ls[NT] = {};
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int t = wave + j * NW_BWD;
      if (t < 2 * nks) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          qf[j][ks] = row_frag_global(rs, ld, t * 16, h * HD, ks, lane);
          df[j][ks] = row_frag_global(rd, ldo, t * 16, h * HD, ks, lane);
          of[j][ks] = row_frag_global(ro, ldo, t * 16, h * HD, ks, lane);
        }
        const int query = t * 16 + c;
        ls[j] = lse[((int64_t)s * H + h) * N + (query < nq ? query : 0)];
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int t = wave + j * NW_BWD;
      if (t < 2 * nks) {
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) dl += bf2f(df[j][ks][e]) * bf2f(of[j][ks][e]);
        dl += __shfl_xor(dl, 16, 64);                  // the four lanes g = 0..3 of a row hold 16 of its 64 columns each
        dl += __shfl_xor(dl, 32, 64);
        const int query = t * 16 + c;
        nl[j] = query < nq ? -ls[j] * 1.44269504088896340736f : -INFINITY;   // -inf: every p of an un-queried row is exactly 0
        nd[j] = query < nq ? -dl : 0.f;
        if (g == 0) { lse_s[query] = nl[j]; del_s[query] = nd[j]; }
      }
    }
  }
  ATTN_STAMP(1)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  ATTN_STAMP(2)

  // per-lane LDS offsets, loop invariant: tile bases are multiples of 16 rows, so (row & 7) never depends on the tile
  const int rf0 = c * 128 + ((g ^ (c & 7)) << 4), rf1 = c * 128 + (((4 + g) ^ (c & 7)) << 4);       // row_frag, k-step 0 / 1
  int trf[4];                                                                                        // tr_frag, d-tile 0..3
  {
    const int rr0 = 4 * g + (c >> 2), pp = c & 3;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) trf[dt] = rr0 * 128 + ((((dt * 2) + (pp >> 1)) ^ (rr0 & 7)) << 4) + (pp & 1) * 8;
  }
  auto RF = [&](const char* img, int tile, int ks) { return *reinterpret_cast<const bf16x8*>(img + tile * 2048 + (ks ? rf1 : rf0)); };
  auto TR = [&](const char* img, int step, int dt) {
    const char* b = img + step * 4096 + trf[dt];
    s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b));
    s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + 2048));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };

  // ------------------------------------------------------------------ pass A: dQ (wave owns query tiles)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int qt = wave + j * NW_BWD;
    if (qt >= nkt) continue;
    const int query = qt * 16 + c;
    f32x4 acc[4];
    float z0 = 0.f;
    asm volatile("" : "+v"(z0));                             // (an opaque zero: constant zero vectors get hoisted out of the head loop and held in 48 registers)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{z0, z0, z0, z0};
    if (qt < nqt) {                                          // dQ of an un-queried tile is exactly zero
      // Q prescaled by scale * log2(e): with -lse as the chain's initial accumulator the scores come out as S' = c2 s - lse, and
      // dP' = dP - delta likewise, so p = exp2(S') and dS = p * dP' need no further VALU
      const bf16x8 qs0 = scale_frag(qf[j][0], c2), qs1 = scale_frag(qf[j][1], c2);
      const bf16x8 df0 = df[j][0], df1 = df[j][1];
      const float nlj = nl[j];
      const f32x4 s_init = {nlj, nlj, nlj, nlj}, d_init = {nd[j], nd[j], nd[j], nd[j]};
      // one 32-key step.  Only the last step can hold keys >= N (they start at -inf so that p = 0); the others are unrolled with
      // compile-time step numbers, so every LDS address is the lane's base register + an immediate.
      auto step = [&](int ps, auto masked) {
        u32x4 dsw;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int kt = 2 * ps + u;
          f32x4 sv = s_init, dp;
          if constexpr (decltype(masked)::value) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sv[r] = (kt * 16 + 4 * g + r >= N) ? -INFINITY : nlj;
          }
          sv = MFMA16(RF(Kimg, kt, 0), qs0, sv);
          sv = MFMA16(RF(Kimg, kt, 1), qs1, sv);
          dp = MFMA16(RF(Vimg, kt, 0), df0, d_init);
          dp = MFMA16(RF(Vimg, kt, 1), df1, dp);
          const f32x2 p01 = {__builtin_amdgcn_exp2f(sv[0]), __builtin_amdgcn_exp2f(sv[1])}, p23 = {__builtin_amdgcn_exp2f(sv[2]), __builtin_amdgcn_exp2f(sv[3])};
          dsw[2 * u] = cvt_pk(p01 * f32x2{dp[0], dp[1]});
          dsw[2 * u + 1] = cvt_pk(p23 * f32x2{dp[2], dp[3]});
        }
        const bf16x8 dsf = __builtin_bit_cast(bf16x8, dsw);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[dt] = MFMA16(TR(Kimg, ps, dt), dsf, acc[dt]);
      };
#pragma unroll
      for (int ps = 0; ps < NMAX / 32 - 1; ++ps) {
        if (ps >= nks - 1) continue;
        step(ps, std::false_type{});
      }
      step(nks - 1, std::true_type{});
    }
    if (query < N) {
      bf16_t* drow = dqkv + (row_base + query) * ld + h * HD;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)     // the softmax scale is applied once, to the accumulator
        *reinterpret_cast<u32x2*>(drow + dt * 16 + 4 * g) = u32x2{cvt_pk(acc[dt][0] * scale, acc[dt][1] * scale), cvt_pk(acc[dt][2] * scale, acc[dt][3] * scale)};
    }
  }

  ATTN_STAMP(3)
  // ------------------------------------------------------------------ hand-over: K, V fragments out of the images, Q, dO fragments in
  bf16x8 kf[NT][2] = {}, vf[NT][2] = {};
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int kt = wave + j * NW_BWD;
    if (kt < nkt) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) { kf[j][ks] = RF(Kimg, kt, ks); vf[j][ks] = RF(Vimg, kt, ks); }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                   // every wave is done reading the K / V images
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int t = wave + j * NW_BWD;
    if (t < 2 * nks) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        *reinterpret_cast<bf16x8*>(Qimg + t * 2048 + (ks ? rf1 : rf0)) = qf[j][ks];
        *reinterpret_cast<bf16x8*>(Dimg + t * 2048 + (ks ? rf1 : rf0)) = df[j][ks];
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  ATTN_STAMP(4)

  // ------------------------------------------------------------------ pass B: dK, dV (wave owns key tiles)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int kt = wave + j * NW_BWD;
    if (kt >= nkt) continue;
    const int key = kt * 16 + c;
    const bf16x8 kf0 = scale_frag(kf[j][0], c2), kf1 = scale_frag(kf[j][1], c2);
    const bf16x8 vf0 = vf[j][0], vf1 = vf[j][1];
    f32x4 dk[4], dv[4];
    float z1 = 0.f;
    asm volatile("" : "+v"(z1));
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{z1, z1, z1, z1};
#pragma unroll
    for (int qs = 0; qs < NMAX / 32; ++qs) {
      if (qs >= nqs) continue;
      u32x4 pw, dsw;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qt = 2 * qs + u;
        // the chains start from -lse / -delta of this lane's 4 queries (stored negated in LDS: the loads ARE the C operands).  Keys
        // >= N of the last tile need no mask here: a key column only ever reaches its own dK / dV rows, which are not stored.
        f32x4 sv = *reinterpret_cast<const f32x4*>(lse_s + qt * 16 + 4 * g);
        f32x4 dp = *reinterpret_cast<const f32x4*>(del_s + qt * 16 + 4 * g);
        sv = MFMA16(RF(Qimg, qt, 0), kf0, sv);   // D[query = 4g + r][key = c]
        sv = MFMA16(RF(Qimg, qt, 1), kf1, sv);
        dp = MFMA16(RF(Dimg, qt, 0), vf0, dp);
        dp = MFMA16(RF(Dimg, qt, 1), vf1, dp);
        const f32x2 p01 = {__builtin_amdgcn_exp2f(sv[0]), __builtin_amdgcn_exp2f(sv[1])}, p23 = {__builtin_amdgcn_exp2f(sv[2]), __builtin_amdgcn_exp2f(sv[3])};
        pw[2 * u] = cvt_pk(p01);
        pw[2 * u + 1] = cvt_pk(p23);
        dsw[2 * u] = cvt_pk(p01 * f32x2{dp[0], dp[1]});
        dsw[2 * u + 1] = cvt_pk(p23 * f32x2{dp[2], dp[3]});
      }
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pw), dsf = __builtin_bit_cast(bf16x8, dsw);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dv[dt] = MFMA16(TR(Dimg, qs, dt), pf, dv[dt]);    // dV^T[d][key]
        dk[dt] = MFMA16(TR(Qimg, qs, dt), dsf, dk[dt]);   // dK^T[d][key]
      }
    }
    if (key < N) {
      bf16_t* krow = dqkv + (row_base + key) * ld + C + h * HD;
      bf16_t* vrow = krow + C;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *reinterpret_cast<u32x2*>(krow + dt * 16 + 4 * g) = u32x2{cvt_pk(dk[dt][0] * scale, dk[dt][1] * scale), cvt_pk(dk[dt][2] * scale, dk[dt][3] * scale)};
        *reinterpret_cast<u32x2*>(vrow + dt * 16 + 4 * g) = u32x2{cvt_pk(dv[dt][0], dv[dt][1]), cvt_pk(dv[dt][2], dv[dt][3])};
      }
    }
  }
#if SA_ATTN_DBG
  ATTN_STAMP(5)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATTN_STAMP(6)
  if (sa_attn_dbg_ptr && lane == 0 && (wave == 0 || wave == NW_BWD - 1)) {
    // HW_REG_HW_ID (4): wave [3:0] simd [5:4] pipe [7:6] cu [11:8] sh [12] se [15:13];  HW_REG_XCC_ID (20): xcc [3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    unsigned long long* o_ = sa_attn_dbg_ptr + ((size_t)unit * 2 + (wave ? 1 : 0)) * 8;
    for (int k = 0; k < 7; ++k) o_[k] = dbg_t[k];
    o_[7] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
  __syncthreads();                                   // the next head's K / V rows replace the Q / dO images pass B has been reading
  }
}

}  // namespace

static int attn_check(const char* who, const void* qkv, int64_t rows, int64_t ld, int C, int H, int N) {
  SA_CHECK_ARG(qkv && rows > 0, "%s: null/empty input", who);
  SA_CHECK_ARG(H > 0 && C == H * HD, "%s: head_dim must be %d (C=%d, H=%d)", who, HD, C, H);
  SA_CHECK_ARG(N > 0 && N <= NMAX_ALL, "%s: sequence length %d outside [1, %d]", who, N, NMAX_ALL);
  SA_CHECK_ARG(rows % N == 0, "%s: rows=%lld not a multiple of N=%d", who, (long long)rows, N);
  SA_CHECK_ARG(ld >= 3 * C && ld % 8 == 0 && ((uintptr_t)qkv & 15) == 0, "%s: qkv must have 16-byte aligned rows", who);
  SA_CHECK_ARG((rows + NMAX_ALL) * ld * 2 < ((int64_t)1 << 32), "%s: qkv larger than the 4 GiB buffer-descriptor range", who);
  return 0;
}

extern "C" int sa_attention_fwd(const void* qkv, int64_t rows, int64_t ld, int32_t C, int32_t H, int32_t N, int32_t n_query, float scale,
                                void* out, int64_t ldo, float* lse, void* stream) {
  if (n_query <= 0 || n_query > N) n_query = N;
  if (attn_check("sa_attention_fwd", qkv, rows, ld, C, H, N)) return 1;
  SA_CHECK_ARG(out && ldo >= C && ldo % 4 == 0, "sa_attention_fwd: bad output");
  const int S = (int)(rows / N);
  if (N <= 256) {
    static bool configured = false;
    if (!configured) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * HD * 2);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * HD * 2);
      configured = true;
    }
    if (N > 256 - 16)
      hipLaunchKernelGGL((attn_fwd_kernel<256, true>), dim3(S * H), dim3(64 * NW_FWD), 2 * 256 * HD * 2, (hipStream_t)stream, (const bf16_t*)qkv, rows, (int)ld, C,
                         H, N, n_query, scale, (bf16_t*)out, (int)ldo, lse);
    else
      hipLaunchKernelGGL((attn_fwd_kernel<256, false>), dim3(S * H), dim3(64 * NW_FWD), 2 * 256 * HD * 2, (hipStream_t)stream, (const bf16_t*)qkv, rows, (int)ld, C,
                         H, N, n_query, scale, (bf16_t*)out, (int)ldo, lse);
  } else {
    static bool configured = false;
    if (!configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<512, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * HD * 2) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<512, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * HD * 2) != hipSuccess) {
        sa_set_error("sa_attention_fwd: 128 KiB of LDS per workgroup refused");
        return 2;
      }
      configured = true;
    }
    if (N > 512 - 16)
      hipLaunchKernelGGL((attn_fwd_kernel<512, true>), dim3(S * H), dim3(64 * NW_FWD), 2 * 512 * HD * 2, (hipStream_t)stream, (const bf16_t*)qkv, rows, (int)ld, C,
                         H, N, n_query, scale, (bf16_t*)out, (int)ldo, lse);
    else
      hipLaunchKernelGGL((attn_fwd_kernel<512, false>), dim3(S * H), dim3(64 * NW_FWD), 2 * 512 * HD * 2, (hipStream_t)stream, (const bf16_t*)qkv, rows, (int)ld, C,
                         H, N, n_query, scale, (bf16_t*)out, (int)ldo, lse);
  }
  SA_LAUNCH_CHECK("sa_attention_fwd");
  return 0;
}

// workgroups of the persistent backward: `per_cu` per CU (what its LDS allows), never more than there are heads
static int attn_bwd_grid(int units, int per_cu) {
  static int cus = 0;
  if (!cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  static const char* env = getenv("SA_ATTN_BWD_PERSIST");          // (1: persistent walk, measured slower: see the kernel's comment)
  if (!(env && env[0] == '1')) return units;
  return units < per_cu * cus ? units : per_cu * cus;
}

extern "C" int sa_attention_bwd(const void* qkv, int64_t rows, int64_t ld, int32_t C, int32_t H, int32_t N, int32_t n_query, float scale,
                                const void* out, const void* dout, int64_t ldo, const float* lse, void* dqkv, void* stream) {
  if (n_query <= 0 || n_query > N) n_query = N;
  if (attn_check("sa_attention_bwd", qkv, rows, ld, C, H, N)) return 1;
  SA_CHECK_ARG(out && dout && lse && dqkv && ldo >= C && ldo % 8 == 0, "sa_attention_bwd: bad args");
  SA_CHECK_ARG(((uintptr_t)dout & 15) == 0 && ((uintptr_t)out & 15) == 0, "sa_attention_bwd: out/dout must be 16-byte aligned");
  const int S = (int)(rows / N);
  if (N <= 256) {
    constexpr int lds = 2 * 256 * HD * 2 + 2 * 256 * (int)sizeof(float);
    static bool configured = false;
    if (!configured) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      configured = true;
    }
    hipLaunchKernelGGL(attn_bwd_kernel<256>, dim3(attn_bwd_grid(S * H, 2)), dim3(64 * NW_BWD), lds, (hipStream_t)stream, (const bf16_t*)qkv, rows, (int)ld, C, H, N, n_query,
                       scale, (const bf16_t*)out, (const bf16_t*)dout, (int)ldo, lse, (bf16_t*)dqkv);
  } else {
    constexpr int lds = 2 * 512 * HD * 2 + 2 * 512 * (int)sizeof(float);
    static bool configured = false;
    if (!configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
        sa_set_error("sa_attention_bwd: 132 KiB of LDS per workgroup refused");
        return 2;
      }
      configured = true;
    }
    hipLaunchKernelGGL(attn_bwd_kernel<512>, dim3(attn_bwd_grid(S * H, 1)), dim3(64 * NW_BWD), lds, (hipStream_t)stream, (const bf16_t*)qkv, rows, (int)ld, C, H, N, n_query,
                       scale, (const bf16_t*)out, (const bf16_t*)dout, (int)ldo, lse, (bf16_t*)dqkv);
  }
  SA_LAUNCH_CHECK("sa_attention_bwd");
  return 0;
}
