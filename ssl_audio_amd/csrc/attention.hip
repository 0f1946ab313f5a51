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
template <int NMAX>
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

  const int nkt = (N + 15) >> 4;           // 16-key tiles
  const int nks = (N + 31) >> 5;           // 32-key steps for P.V
  stage_rows(rs, Kimg, ld, C + h * HD, nks * 32, wave, lane, NW_FWD);
  stage_rows(rs, Vimg, ld, 2 * C + h * HD, nks * 32, wave, lane, NW_FWD);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int g = lane >> 4, c = lane & 15;
  const int nqt = (nq + 15) >> 4;            // only the first nq queries of each sequence are wanted (nq = N normally)
  f32x4 kbias;                               // accumulator start of the last key tile: -inf on keys >= N, else 0
#pragma unroll
  for (int r = 0; r < 4; ++r) kbias[r] = ((nkt - 1) * 16 + 4 * g + r >= N) ? -INFINITY : 0.f;
  for (int qt = wave; qt < nqt; qt += NW_FWD) {
    // Q fragments for this lane's query (Y operand: k = d)
    const int query = qt * 16 + c;
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint32_t voff = ((uint32_t)query * (uint32_t)ld + (uint32_t)(h * HD + 32 * ks + 8 * g)) * 2u;
      auto raw = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
      qf[ks] = __builtin_bit_cast(bf16x8, raw);
    }
    // S^T tiles: st[kt][r] = score(key = 16*kt + 4*g + r, query)
    f32x4 st[NKT];
    float mx = -INFINITY;
    f32x4 kb = kbias;                                  // opaque copy: keeps the 16 per-tile selects below from being hoisted
    asm volatile("" : "+v"(kb));                       // out of the query-tile loop (64 live VGPRs -> spills)
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kt < nkt) {
        if (kt == nkt - 1) st[kt] = kb;                // padding keys start at -inf: the MFMA accumulate keeps them there
        st[kt] = MFMA16(row_frag(Kimg, kt * 16, 0, lane), qf[0], st[kt]);
        st[kt] = MFMA16(row_frag(Kimg, kt * 16, 1, lane), qf[1], st[kt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kt][r]);   // raw scores; the (positive) scale is folded into the exponent
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    const float c2 = scale * 1.44269504088896340736f;            // exp(scale * (s - max)) = exp2(c2 * s - c2 * max): one fma + v_exp
    const float mb = mx * c2;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
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
      if (ps < nks) {
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
}

// =====================================================================================================
template <int NMAX>
__global__ __launch_bounds__(64 * NW_BWD, NMAX == 256 ? 4 : 2) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, int64_t total_rows, int ld, int C, int H, int N,
                                                          int nq, float scale, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                          int ldo, const float* __restrict__ lse, bf16_t* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Two images at a time (NMAX = 256: 2 x 32 KiB, 66 KiB with the statistics -> two workgroups per CU): K,V during pass A, then the
  // same LDS is re-staged with Q,dO for pass B.  The operand that is NOT in LDS is read as 16-byte row fragments from global.
  constexpr int IMG = NMAX * HD * 2;
  char* Kimg = smem;
  char* Vimg = smem + IMG;
  char* Qimg = smem;                                 // pass B reuses the two images
  char* Dimg = smem + IMG;                           // dO
  float* lse_s = reinterpret_cast<float*>(smem + 2 * IMG);
  float* del_s = lse_s + NMAX;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = blockIdx.x / H, h = blockIdx.x % H;
  const int64_t row_base = (int64_t)s * N;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(qkv + row_base * ld, (uint32_t)(((total_rows - row_base - 1) * ld + 3 * C) * 2));
  const __amdgpu_buffer_rsrc_t rd = make_rsrc(dout + row_base * ldo, (uint32_t)(((total_rows - row_base - 1) * ldo + C) * 2));

  const int nkt = (N + 15) >> 4;
  const int nks = (N + 31) >> 5;
  const int nrows = nks * 32;
  // The operand a wave owns (Q, dO of its query tile in pass A; K, V of its key tile in pass B) comes straight from global as four
  // 16-byte row fragments.  Each tile's fragments are requested one tile ahead -- the first ones here, ahead of the staging -- so
  // that no tile starts with an exposed L2 / HBM round trip.
  struct Frag4 { bf16x8 a0, a1, b0, b1; };
  auto load_qd = [&](int qt) {
    return Frag4{row_frag_global(rs, ld, qt * 16, h * HD, 0, lane), row_frag_global(rs, ld, qt * 16, h * HD, 1, lane),
                 row_frag_global(rd, ldo, qt * 16, h * HD, 0, lane), row_frag_global(rd, ldo, qt * 16, h * HD, 1, lane)};
  };
  auto load_kv = [&](int kt) {
    return Frag4{row_frag_global(rs, ld, kt * 16, C + h * HD, 0, lane), row_frag_global(rs, ld, kt * 16, C + h * HD, 1, lane),
                 row_frag_global(rs, ld, kt * 16, 2 * C + h * HD, 0, lane), row_frag_global(rs, ld, kt * 16, 2 * C + h * HD, 1, lane)};
  };
  Frag4 nxt = load_qd(wave);
  stage_rows(rs, Kimg, ld, C + h * HD, nrows, wave, lane, NW_BWD);
  stage_rows(rs, Vimg, ld, 2 * C + h * HD, nrows, wave, lane, NW_BWD);
  // delta[q] = sum_d dO[q][d] * O[q][d], lse -> LDS.  Eight lanes share a query row (8 x 16 B = the row's 128 bytes of one head), so
  // a wave instruction reads 8 whole row segments instead of 64 scattered 16-byte pieces (that version cost a quarter of the kernel).
  {
    const int sub = lane >> 3, ch = lane & 7;
    constexpr int NIT = NMAX / (NW_BWD * 8);           // 4 row groups per wave: all 12 loads go out before the first is consumed
    bf16x8 a[NIT], b[NIT];
    float ls[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = (it * NW_BWD + wave) * 8 + sub;
      const int qc = q < nq ? q : 0;                    // clamp: the row is read but its result discarded
      a[it] = *reinterpret_cast<const bf16x8*>(o + (row_base + qc) * ldo + h * HD + ch * 8);
      b[it] = *reinterpret_cast<const bf16x8*>(dout + (row_base + qc) * ldo + h * HD + ch * 8);
      ls[it] = lse[((int64_t)s * H + h) * N + qc];
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = (it * NW_BWD + wave) * 8 + sub;
      float dl = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) dl += bf2f(a[it][j]) * bf2f(b[it][j]);
      dl += __shfl_xor(dl, 1, 64);
      dl += __shfl_xor(dl, 2, 64);
      dl += __shfl_xor(dl, 4, 64);
      if (ch == 0) {
        // both stored NEGATED: they are the initial accumulators of the S and dP chains.  queries >= nq: -lse = -inf makes every p exactly 0
        del_s[q] = q < nq ? -dl : 0.f;
        lse_s[q] = q < nq ? -ls[it] * 1.44269504088896340736f : -INFINITY;           // log2 units for v_exp
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int g = lane >> 4, c = lane & 15;
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
  const float c2 = scale * 1.44269504088896340736f;    // p = exp(scale*s - lse) = exp2(c2*s - lse*log2e)

  // ------------------------------------------------------------------ pass A: dQ (wave owns query tiles)
  const int nqt = (nq + 15) >> 4, nqs = (nq + 31) >> 5;   // queries >= nq carry no upstream gradient (CLS-only last block)
  for (int qt = wave; qt < nkt; qt += NW_BWD) {
    const int query = qt * 16 + c;
    const Frag4 cur = nxt;
    if (qt + NW_BWD < nkt) nxt = load_qd(qt + NW_BWD);
    if (qt >= nqt) {                                       // dQ of an un-queried tile is exactly zero
      if (query < N) {
        bf16_t* drow = dqkv + (row_base + query) * ld + h * HD;
        const bf16x4 z = {f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<bf16x4*>(drow + dt * 16 + 4 * g) = z;
      }
      continue;
    }
    const bf16x8 qf0 = scale_frag(cur.a0, c2), qf1 = scale_frag(cur.a1, c2);
    const bf16x8 df0 = cur.b0, df1 = cur.b1;
    // initial accumulators of the two chains (loop invariant, the MFMA's C operand): S' = c2 s - lse, dP' = dP - delta
    const float nl = lse_s[query], nd = del_s[query];      // -lse (log2 units), -delta
    const f32x4 s_init = {nl, nl, nl, nl}, d_init = {nd, nd, nd, nd};
    f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
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
          for (int r = 0; r < 4; ++r) sv[r] = (kt * 16 + 4 * g + r >= N) ? -INFINITY : nl;
        }
        sv = MFMA16(RF(Kimg, kt, 0), qf0, sv);
        sv = MFMA16(RF(Kimg, kt, 1), qf1, sv);
        dp = MFMA16(RF(Vimg, kt, 0), df0, d_init);
        dp = MFMA16(RF(Vimg, kt, 1), df1, dp);
        // p = exp2(S'): 0 for padding keys (-inf) and un-queried rows (-lse = -inf); dS = p * dP'; the softmax scale is applied once, to the accumulator
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
    if (query < N) {
      bf16_t* drow = dqkv + (row_base + query) * ld + h * HD;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 v = {f2bf(acc[dt][0] * scale), f2bf(acc[dt][1] * scale), f2bf(acc[dt][2] * scale), f2bf(acc[dt][3] * scale)};
        *reinterpret_cast<bf16x4*>(drow + dt * 16 + 4 * g) = v;
      }
    }
  }

  // ------------------------------------------------------------------ pass B: dK, dV (wave owns key tiles)
  nxt = load_kv(wave);
  __syncthreads();                                   // every wave is done reading the K / V images
  stage_rows(rs, Qimg, ld, h * HD, nqs * 32, wave, lane, NW_BWD);
  stage_rows(rd, Dimg, ldo, h * HD, nqs * 32, wave, lane, NW_BWD);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = wave; kt < nkt; kt += NW_BWD) {
    const int key = kt * 16 + c;
    const bool kvalid = key < N;
    const Frag4 cur = nxt;
    if (kt + NW_BWD < nkt) nxt = load_kv(kt + NW_BWD);
    const bf16x8 kf0 = scale_frag(cur.a0, c2), kf1 = scale_frag(cur.a1, c2);
    const bf16x8 vf0 = cur.b0, vf1 = cur.b1;
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dk[dt] = dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
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
    if (kvalid) {
      bf16_t* krow = dqkv + (row_base + key) * ld + C + h * HD;
      bf16_t* vrow = krow + C;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 a = {f2bf(dk[dt][0] * scale), f2bf(dk[dt][1] * scale), f2bf(dk[dt][2] * scale), f2bf(dk[dt][3] * scale)};
        bf16x4 b = {f2bf(dv[dt][0]), f2bf(dv[dt][1]), f2bf(dv[dt][2]), f2bf(dv[dt][3])};
        *reinterpret_cast<bf16x4*>(krow + dt * 16 + 4 * g) = a;
        *reinterpret_cast<bf16x4*>(vrow + dt * 16 + 4 * g) = b;
      }
    }
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
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * HD * 2);
      configured = true;
    }
    hipLaunchKernelGGL(attn_fwd_kernel<256>, dim3(S * H), dim3(64 * NW_FWD), 2 * 256 * HD * 2, (hipStream_t)stream, (const bf16_t*)qkv, rows, (int)ld, C, H,
                       N, n_query, scale, (bf16_t*)out, (int)ldo, lse);
  } else {
    static bool configured = false;
    if (!configured) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * HD * 2) != hipSuccess) {
        sa_set_error("sa_attention_fwd: 128 KiB of LDS per workgroup refused");
        return 2;
      }
      configured = true;
    }
    hipLaunchKernelGGL(attn_fwd_kernel<512>, dim3(S * H), dim3(64 * NW_FWD), 2 * 512 * HD * 2, (hipStream_t)stream, (const bf16_t*)qkv, rows, (int)ld, C, H,
                       N, n_query, scale, (bf16_t*)out, (int)ldo, lse);
  }
  SA_LAUNCH_CHECK("sa_attention_fwd");
  return 0;
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
    hipLaunchKernelGGL(attn_bwd_kernel<256>, dim3(S * H), dim3(64 * NW_BWD), lds, (hipStream_t)stream, (const bf16_t*)qkv, rows, (int)ld, C, H, N, n_query,
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
    hipLaunchKernelGGL(attn_bwd_kernel<512>, dim3(S * H), dim3(64 * NW_BWD), lds, (hipStream_t)stream, (const bf16_t*)qkv, rows, (int)ld, C, H, N, n_query,
                       scale, (const bf16_t*)out, (const bf16_t*)dout, (int)ldo, lse, (bf16_t*)dqkv);
  }
  SA_LAUNCH_CHECK("sa_attention_bwd");
  return 0;
}
