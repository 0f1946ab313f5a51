// LayerNorm forward / backward for the ViT residual stream (fp32 stream, bf16 GEMM operands).
// HBM-bound: one wave per row, the row lives in registers (D <= 2048), 16-byte loads, wave reductions.
// Replaces nn.LayerNorm(eps=1e-6) at models/mae.py:149,153,208 (+ decoder_norm :227).
#include "common.h"
#include "../../include/ssl_audio_hip.h"

namespace {

constexpr int MAXV_LIMIT = 8;  // float4 per lane -> D <= 64 * 4 * 8 = 2048 (kernels are instantiated for 1, 2, 3, 4, 8)

// Sum over the LPR lanes that share a row (LPR = 64: the whole wave; 32 / 16: two / four rows per wave, the narrow-model layouts).
// 16 lanes are one DPP row: four v_add_f32 with a DPP operand (quad xor 1, xor 2, mirrored half row, mirrored row).
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (LPR == 64) {
    return wave_sum(v);
  } else {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    if constexpr (LPR == 32) v += __shfl_xor(v, 16, 64);
    return v;
  }
}

// y = (x - mean) * rstd * gamma + beta ; writes bf16 and/or fp32, saves mean / rstd per row
// RPW rows per wave (1, 2 or 4): a row is spread over LPR = 64 / RPW lanes, lane `sub` owning the float4 columns sub + i * LPR.  Narrow
// rows (D = 192: 48 float4) left a quarter of the wave idle and one 768-byte row in flight per wave; four rows per wave use every lane
// and keep four rows' loads in flight.
template <int MAXV, int RPW>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y_bf16,
                                                     float* __restrict__ y_f32, int64_t ldy, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int M, int D, float eps) {
  constexpr int LPR = 64 / RPW;
  const int lane = threadIdx.x & 63, sub = lane & (LPR - 1), rsel = lane / LPR;
  const int nv = D >> 2;  // float4 per row
  for (int rb = blockIdx.x * 4 + (threadIdx.x >> 6); rb * RPW < M; rb += gridDim.x * 4) {
    const int row = rb * RPW + rsel;
    const bool live = row < M;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)(live ? row : 0) * ldx);
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = sub + i * LPR;
      if (c < nv) {
        v[i] = xr[c];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
    }
    const float mean = group_sum<LPR>(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = sub + i * LPR;
      if (c < nv) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float rstd = rsqrtf(group_sum<LPR>(q) / (float)D + eps);
    if (sub == 0 && live) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = sub + i * LPR;
      if (c < nv && live) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[c];
        const float4 b = reinterpret_cast<const float4*>(beta)[c];
        float4 o;
        o.x = (v[i].x - mean) * rstd * g.x + b.x;
        o.y = (v[i].y - mean) * rstd * g.y + b.y;
        o.z = (v[i].z - mean) * rstd * g.z + b.z;
        o.w = (v[i].w - mean) * rstd * g.w + b.w;
        if (y_f32) reinterpret_cast<float4*>(y_f32 + (int64_t)row * ldy)[c] = o;
        if (y_bf16) {
          bf16x4 h = {f2bf(o.x), f2bf(o.y), f2bf(o.z), f2bf(o.w)};
          reinterpret_cast<bf16x4*>(y_bf16 + (int64_t)row * ldy)[c] = h;
        }
      }
    }
  }
}

// dx = dres + rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat));  dgamma += sum dy*xhat ; dbeta += sum dy
// dy arrives as bf16 (from a dgrad GEMM) or fp32 (from the loss side).  Each lane owns fixed columns, so its
// dgamma/dbeta partials stay in registers over all rows the block visits; one LDS reduce + a workspace row at the end.
template <typename DY, int MAXV, int RPW>
__global__ __launch_bounds__(256, (MAXV <= 4 ? 4 : 1)) void ln_bwd_kernel(const DY* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                     const float* __restrict__ rstd_in, const float* __restrict__ dres, int64_t lddres,
                                                     float* __restrict__ dx_f32, bf16_t* __restrict__ dx_bf16, int64_t lddx,
                                                     float* __restrict__ ws, int M, int D) {
  constexpr int LPR = 64 / RPW;                          // lanes per row (see ln_fwd_kernel)
  __shared__ float red[4 * 64 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & (LPR - 1), rsel = lane / LPR;
  const int nv = D >> 2;
  float4 ag[MAXV], ab[MAXV], ax[MAXV];   // column partials: dgamma, dbeta, and sum of the OUTPUT dx (= bias gradient of the
#pragma unroll                           // Linear whose output gradient dx is: proj for LN2, the previous block's fc2 for LN1)
  for (int i = 0; i < MAXV; ++i) ag[i] = ab[i] = ax[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int rb = blockIdx.x * 4 + wave; rb * RPW < M; rb += gridDim.x * 4) {
    const int row = rb * RPW + rsel;
    const bool live = row < M;
    const int rowc = live ? row : 0;
    const float mean = mean_in[rowc], rstd = rstd_in[rowc];
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)rowc * ldx);
    // From four float4 columns per lane on (D = 1024, ViT-L) a row keeps dy between the two sweeps as the two dwords it was loaded as
    // (bf16) and re-reads gamma from L1 in the second sweep: 6 instead of 8 registers per column take the kernel from 136 to 124
    // registers, i.e. from three to four waves per SIMD -- 259 -> 185 us at [63 744, 1024].  Narrower rows already run four waves per
    // SIMD and the recomputation costs them 2 - 7 % (measured), so they keep the fp32 product.
    constexpr bool PACKED = sizeof(DY) == 2 && MAXV >= 4;
    float4 xh[MAXV], gd[PACKED ? 1 : MAXV];
    bf16x4 dh[PACKED ? MAXV : 1];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = sub + i * LPR;
      if (c < nv && live) {
        const float4 xv = xr[c];
        float4 d;
        if constexpr (sizeof(DY) == 2) {
          const bf16x4 h = reinterpret_cast<const bf16x4*>(dy + (int64_t)row * lddy)[c];
          d = make_float4(bf2f(h[0]), bf2f(h[1]), bf2f(h[2]), bf2f(h[3]));
        } else {
          d = reinterpret_cast<const float4*>(dy + (int64_t)row * lddy)[c];
        }
        const float4 g = reinterpret_cast<const float4*>(gamma)[c];
        xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        ag[i].x += d.x * xh[i].x; ag[i].y += d.y * xh[i].y; ag[i].z += d.z * xh[i].z; ag[i].w += d.w * xh[i].w;
        ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
        const float4 gdi = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
        if constexpr (PACKED) dh[i] = reinterpret_cast<const bf16x4*>(dy + (int64_t)row * lddy)[c];
        else gd[i] = gdi;
        s1 += (gdi.x + gdi.y) + (gdi.z + gdi.w);
        s2 += (gdi.x * xh[i].x + gdi.y * xh[i].y) + (gdi.z * xh[i].z + gdi.w * xh[i].w);
      }
    }
    const float m1 = group_sum<LPR>(s1) / (float)D, m2 = group_sum<LPR>(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = sub + i * LPR;
      if (c < nv && live) {
        float4 gdi;
        if constexpr (PACKED) {
          const float4 g = reinterpret_cast<const float4*>(gamma)[c];
          gdi = make_float4(bf2f(dh[i][0]) * g.x, bf2f(dh[i][1]) * g.y, bf2f(dh[i][2]) * g.z, bf2f(dh[i][3]) * g.w);
        } else {
          gdi = gd[i];
        }
        float4 o;
        o.x = rstd * (gdi.x - m1 - xh[i].x * m2);
        o.y = rstd * (gdi.y - m1 - xh[i].y * m2);
        o.z = rstd * (gdi.z - m1 - xh[i].z * m2);
        o.w = rstd * (gdi.w - m1 - xh[i].w * m2);
        if (dres) {
          const float4 r = reinterpret_cast<const float4*>(dres + (int64_t)row * lddres)[c];
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (dx_f32) reinterpret_cast<float4*>(dx_f32 + (int64_t)row * lddx)[c] = o;
        if (dx_bf16) {
          bf16x4 h = {f2bf(o.x), f2bf(o.y), f2bf(o.z), f2bf(o.w)};
          reinterpret_cast<bf16x4*>(dx_bf16 + (int64_t)row * lddx)[c] = h;
        }
        ax[i].x += o.x; ax[i].y += o.y; ax[i].z += o.z; ax[i].w += o.w;
      }
    }
  }
  if (ws == nullptr) return;
  // Column partials leave the block through a workspace ([kind][block][D]); ln_bwd_reduce_kernel sums them.  (Atomics on
  // 3*D addresses from ~1000 blocks are resolved memory-side across the 8 XCDs and cost more than the whole streaming pass.)
  // The block's 4 waves x RPW row groups each hold a partial of column chunk (sub, i): they meet in LDS, LPR columns at a time.
  const int nblk = gridDim.x;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = sub + i * LPR;
    if (i * LPR >= nv) break;
#pragma unroll
    for (int kind = 0; kind < 3; ++kind) {
      __syncthreads();
      reinterpret_cast<float4*>(red)[wave * 64 + lane] = kind == 0 ? ag[i] : kind == 1 ? ab[i] : ax[i];
      __syncthreads();
      if (wave == 0 && rsel == 0 && c < nv) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int r = 0; r < RPW; ++r) {
            const float4 u = reinterpret_cast<float4*>(red)[w * 64 + r * LPR + sub];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
          }
        reinterpret_cast<float4*>(ws + ((int64_t)kind * nblk + blockIdx.x) * D)[c] = t;
      }
    }
  }
}

// out[kind][col] += sum over blocks of ws[kind][block][col]; 16 waves split the blocks, lanes are columns
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* __restrict__ ws, int nblk, int D, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ dxsum) {
  __shared__ float red[16][64];
  const int kind = blockIdx.y;
  float* out = kind == 0 ? dgamma : kind == 1 ? dbeta : dxsum;
  if (out == nullptr) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const float* base = ws + (int64_t)kind * nblk * D + col;
  float acc = 0.f;
  if (col < D) {
    int p = wave;
    for (; p + 7 * 16 < nblk; p += 8 * 16) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(int64_t)(p + u * 16) * D];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; p < nblk; p += 16) acc += base[(int64_t)p * D];
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && col < D) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][lane];
    out[col] += t;
  }
}

inline int ln_rpw(int D) {   // rows per wave: the widest layout whose LPR = 64 / RPW lanes still hold a row in <= 4 float4 each
  const int nv = D / 4;
  return nv <= 16 * 4 && nv % 16 == 0 && nv <= 64 ? 4 : (nv <= 32 * 4 && nv % 32 == 0 && nv <= 128 ? 2 : 1);
}

inline int ln_bwd_grid(int M, int D = 2048) {   // fewer, fatter blocks than forward: every block ends with a 3*D partial
  const int g = (M + 4 * ln_rpw(D) - 1) / (4 * ln_rpw(D));
  return g > 1024 ? 1024 : g;
}

inline int ln_grid(int M, int D) {
  const int want = (M + 4 * ln_rpw(D) - 1) / (4 * ln_rpw(D));
  return want < 2048 ? want : 2048;
}

}  // namespace

extern "C" int sa_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                                int64_t ldy, float* mean, float* rstd, int32_t M, int32_t D, float eps, void* stream) {
  SA_CHECK_ARG(x && gamma && beta && (y_bf16 || y_f32), "sa_layernorm_fwd: null pointer");
  SA_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * MAXV_LIMIT, "sa_layernorm_fwd: D=%d must be a multiple of 4 and <= %d", D, 64 * 4 * MAXV_LIMIT);
  SA_CHECK_ARG(ldx % 4 == 0 && ldy % 4 == 0, "sa_layernorm_fwd: leading dims must be multiples of 4");
  const int rpw = ln_rpw(D);
  const int nv = (D / 4 + 64 / rpw - 1) / (64 / rpw);        // float4 per lane
#define SA_LN_FWD(V, R) hipLaunchKernelGGL((ln_fwd_kernel<V, R>), dim3(ln_grid(M, D)), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, \
                                           (bf16_t*)y_bf16, y_f32, ldy, mean, rstd, M, D, eps)
  if (rpw == 4) { if (nv <= 1) SA_LN_FWD(1, 4); else if (nv == 2) SA_LN_FWD(2, 4); else if (nv == 3) SA_LN_FWD(3, 4); else SA_LN_FWD(4, 4); }
  else if (rpw == 2) { if (nv <= 1) SA_LN_FWD(1, 2); else if (nv == 2) SA_LN_FWD(2, 2); else if (nv == 3) SA_LN_FWD(3, 2); else SA_LN_FWD(4, 2); }
  else if (nv <= 1) SA_LN_FWD(1, 1); else if (nv == 2) SA_LN_FWD(2, 1); else if (nv == 3) SA_LN_FWD(3, 1); else if (nv == 4) SA_LN_FWD(4, 1); else SA_LN_FWD(8, 1);
#undef SA_LN_FWD
  SA_LAUNCH_CHECK("sa_layernorm_fwd");
  return 0;
}

extern "C" int64_t sa_layernorm_bwd_workspace_bytes(int32_t M, int32_t D) {
  return (int64_t)3 * ln_bwd_grid(M, D) * D * (int64_t)sizeof(float);
}

extern "C" int sa_layernorm_bwd(const void* dy, int32_t dy_is_bf16, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                                const float* mean, const float* rstd, const float* dres, int64_t lddres, float* dx_f32, void* dx_bf16,
                                int64_t lddx, float* dgamma, float* dbeta, float* dxsum, float* workspace, int32_t M, int32_t D, void* stream) {
  SA_CHECK_ARG(dy && x && gamma && mean && rstd && (dx_f32 || dx_bf16), "sa_layernorm_bwd: null pointer");
  SA_CHECK_ARG(M > 0 && D > 0 && D % 4 == 0 && D <= 64 * 4 * MAXV_LIMIT, "sa_layernorm_bwd: D=%d must be a multiple of 4 and <= %d", D, 64 * 4 * MAXV_LIMIT);
  SA_CHECK_ARG(ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && (!dres || lddres % 4 == 0), "sa_layernorm_bwd: leading dims must be multiples of 4");
  const bool sums = dgamma || dbeta || dxsum;
  SA_CHECK_ARG(!sums || workspace, "sa_layernorm_bwd: column sums need a workspace of sa_layernorm_bwd_workspace_bytes(M, D)");
  const int grid = ln_bwd_grid(M, D);
  float* ws = sums ? workspace : nullptr;
  const int rpw = ln_rpw(D);
  const int nv = (D / 4 + 64 / rpw - 1) / (64 / rpw);        // float4 per lane
#define SA_LN_BWD(T, V, R) hipLaunchKernelGGL((ln_bwd_kernel<T, V, R>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)dy, lddy, x, ldx, \
                                              gamma, mean, rstd, dres, lddres, dx_f32, (bf16_t*)dx_bf16, lddx, ws, M, D)
#define SA_LN_BWD_R(T, R) do { if (nv <= 1) SA_LN_BWD(T, 1, R); else if (nv == 2) SA_LN_BWD(T, 2, R); else if (nv == 3) SA_LN_BWD(T, 3, R); \
                               else SA_LN_BWD(T, 4, R); } while (0)
#define SA_LN_BWD_V(T) do { if (rpw == 4) SA_LN_BWD_R(T, 4); else if (rpw == 2) SA_LN_BWD_R(T, 2); else if (nv <= 4) SA_LN_BWD_R(T, 1); \
                            else SA_LN_BWD(T, 8, 1); } while (0)
  if (dy_is_bf16) SA_LN_BWD_V(bf16_t); else SA_LN_BWD_V(float);
#undef SA_LN_BWD_V
#undef SA_LN_BWD_R
#undef SA_LN_BWD
  SA_LAUNCH_CHECK("sa_layernorm_bwd");
  if (sums) {
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((D + 63) / 64, 3), dim3(1024), 0, (hipStream_t)stream, ws, grid, D, dgamma, dbeta, dxsum);
    SA_LAUNCH_CHECK("sa_layernorm_bwd(reduce)");
  }
  return 0;
}
