// The Barlow Twins loss term as FIVE launches around its three collectives (SURVEY.md §2.3 "K7"; utils/loss.py:15-30):
//
//   forward   sa_bt_stats2        column (mean, M2) of both views                                  -> [all-gather of the packed statistics]
//             sa_bt_corr          cross-rank combine (Chan) + running buffers + normalise + c = z1n^T z2n / n   -> [all-reduce of c]
//             sa_bt_loss_grad     loss and G = dL/dc                                                (bn_loss.hip)
//   backward  sa_bt_bwd_products  dz1n = z2n G^T / n, dz2n = z1n G / n and their BatchNorm column sums          -> [all-reduce of the sums]
//             sa_bt_bwd_apply     dz of both views
//
// against the fifteen of the piecewise schedule (bn_colstats x2, bn_finalize x2, bn_apply x2, matmul_f32 x3, bn_bwd_stats x2,
// bn_bwd_apply x2, bt_loss_grad x2).  Same arithmetic, fp32 throughout, products on the exact-fp32 MFMA v_mfma_f32_32x32x2_f32: the
// normalised values and the two gradient products are bit-identical to the piecewise kernels'; the cross-correlation adds its rows in
// four contiguous quarters and the BatchNorm backward sums in another (fixed) order, so c, the loss and dz agree to fp32 rounding
// (and are bit-reproducible run to run).  What it buys
// is launches on the critical path between the forward and the backward of a step: 16.8 MFLOP and 256 KiB at B = 128, D = 256 are latency.
#include "common.h"
#include "../../include/ssl_audio_hip.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int ST_RG = 8;
// ---- (mean, M2) per column of z1 and z2 (blockIdx.y = view): the two-pass column statistics of bn_colstats_kernel
__global__ __launch_bounds__(64 * ST_RG) void bt_stats2_kernel(const float* __restrict__ z1, const float* __restrict__ z2, int64_t ld, int B, int D,
                                                               float* __restrict__ stats) {
  __shared__ float red[ST_RG][64];
  const float* x = blockIdx.y ? z2 : z1;
  float* mean_out = stats + (int64_t)blockIdx.y * 2 * D;
  float* m2_out = mean_out + D;
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + col;
  const bool live = c < D;
  float s = 0.f;
  if (live)
    for (int b = rg; b < B; b += ST_RG) s += x[(int64_t)b * ld + c];
  red[rg][col] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < ST_RG; ++k) tot += red[k][col];
  const float mean = tot / (float)B;
  __syncthreads();
  float q = 0.f;
  if (live)
    for (int b = rg; b < B; b += ST_RG) {
      const float d = x[(int64_t)b * ld + c] - mean;
      q += d * d;
    }
  red[rg][col] = q;
  __syncthreads();
  if (rg == 0 && live) {
    float m2 = 0.f;
#pragma unroll
    for (int k = 0; k < ST_RG; ++k) m2 += red[k][col];
    mean_out[c] = mean;
    m2_out[c] = m2;
  }
}

// Chan combine of W ranks' (mean, M2) of one column (equal row counts), as bn_finalize_kernel does it
__device__ __forceinline__ void combine(const float* __restrict__ st, int64_t rank_stride, int W, int rows_per_rank, int D, int c, float eps, float& mean,
                                        float& rstd, float& var_unbiased) {
  float m = 0.f;
  for (int r = 0; r < W; ++r) m += st[r * rank_stride + c];
  m /= (float)W;
  float m2 = 0.f;
  for (int r = 0; r < W; ++r) {
    const float d = st[r * rank_stride + c] - m;
    m2 += st[r * rank_stride + D + c] + (float)rows_per_rank * d * d;
  }
  const float n = (float)W * (float)rows_per_rank;
  mean = m;
  rstd = rsqrtf(m2 / n + eps);
  var_unbiased = m2 / fmaxf(n - 1.f, 1.f);
}

// ---- one workgroup (four waves) per 32 x 32 tile (ti, tj) of c: statistics of its 32 view-1 columns and 32 view-2 columns combined over
// the ranks, the two views normalised in registers (zn = (z - mean) * rstd, the expression of bn_apply_kernel) and fed to the MFMA as
// A(i, b) = z1n[b][i], B(b, j) = z2n[b][j].  The four waves split the rows b (the reduction) into contiguous quarters and their partial
// tiles meet in LDS in wave order: at 16.8 MFLOP the product is a chain of load latencies, so it is cut four ways and every wave has the
// next 16 rows' loads in flight while it multiplies the current 16.  Tiles of the first tile row / column also write the normalised
// views and mean / rstd (saved for the backward); the DIAGONAL tiles update the running buffers, view 1 then view 2, as bn(z1), bn(z2)
// do (utils/loss.py:17).  allst: [W][2][2][D] (the all-gathered sa_bt_stats2 output).
__global__ __launch_bounds__(256) void bt_corr_kernel(const float* __restrict__ z1, const float* __restrict__ z2, int64_t ld, int B, int D,
                                                      const float* __restrict__ allst, int W, float eps, float momentum, float inv_n,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out, float* __restrict__ running_mean,
                                                      float* __restrict__ running_var, float* __restrict__ z1n, float* __restrict__ z2n,
                                                      float* __restrict__ c) {
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ti = blockIdx.y, tj = blockIdx.x;
  const int i = lane & 31, kk = lane >> 5;
  const int ci = ti * 32 + i, cj = tj * 32 + i;
  const bool iv = ci < D, jv = cj < D;
  float mu1 = 0.f, r1 = 0.f, vu1 = 0.f, mu2 = 0.f, r2 = 0.f, vu2 = 0.f;
  const int64_t rs = (int64_t)4 * D;                       // one rank's [2][2][D] block
  if (iv) combine(allst, rs, W, B, D, ci, eps, mu1, r1, vu1);
  if (jv) combine(allst + 2 * D, rs, W, B, D, cj, eps, mu2, r2, vu2);
  if (wave == 0 && kk == 0) {
    if (tj == 0 && iv) { mean_out[ci] = mu1; rstd_out[ci] = r1; }
    if (ti == 0 && jv) { mean_out[D + cj] = mu2; rstd_out[D + cj] = r2; }
    if (ti == tj && iv) {                                   // (ci == cj here)
      if (running_mean) {
        float rm = running_mean[ci];
        rm = (1.f - momentum) * rm + momentum * mu1;
        running_mean[ci] = (1.f - momentum) * rm + momentum * mu2;
      }
      if (running_var) {
        float rv = running_var[ci];
        rv = (1.f - momentum) * rv + momentum * vu1;
        running_var[ci] = (1.f - momentum) * rv + momentum * vu2;
      }
    }
  }
  const int per = ((B + 3) / 4 + 1) & ~1;                  // rows per wave (even: a lane pair covers rows k, k + 1)
  const int k_lo = wave * per, k_hi = min(B, k_lo + per);
  const float* ap = z1 + (iv ? ci : 0);
  const float* bp = z2 + (jv ? cj : 0);
  // rows past k_hi read row 0 (always valid) and are zeroed; a chunk = 16 rows = 8 MFMAs
  auto load_chunk = [&](int k, float (&a)[8], float (&b)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kq = k + 2 * u + kk;
      const int64_t row = kq < k_hi ? kq : 0;
      a[u] = ap[row * ld];
      b[u] = bp[row * ld];
    }
  };
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float a0[8], b0[8], a1[8], b1[8];
  if (k_lo < k_hi) load_chunk(k_lo, a0, b0);
  for (int k = k_lo; k < k_hi; k += 16) {
    const bool more = k + 16 < k_hi;
    if (more) load_chunk(k + 16, a1, b1);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kq = k + 2 * u + kk;
      const bool live = kq < k_hi;
      const float av = (iv && live) ? (a0[u] - mu1) * r1 : 0.f;
      const float bv = (jv && live) ? (b0[u] - mu2) * r2 : 0.f;
      if (live) {
        if (tj == 0 && iv) z1n[(int64_t)kq * D + ci] = av;
        if (ti == 0 && jv) z2n[(int64_t)kq * D + cj] = bv;
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { a0[u] = a1[u]; b0[u] = b1[u]; }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  if (wave == 0) {
    // C/D layout 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int n = tj * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const float t = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
      if (m < D && n < D) c[(int64_t)m * D + n] = inv_n * t;
    }
  }
}

// ---- backward products with their BatchNorm column sums.  blockIdx.y = view, blockIdx.x = 32-column tile; EIGHT waves, wave w takes the
// 32-row tiles w, w + 8, ...:  view 0: dz1n[b][i] = inv_n * sum_j z2n[b][j] G[i][j]; view 1: dz2n[b][j] = inv_n * sum_i z1n[b][i] G[i][j]
// (the operand order and k-order of the two matmul_f32 launches they replace; the next 16 k's loads travel while the current 16 are
// multiplied), then s1 = sum_b dzn, s2 = sum_b dzn * zn per column: a lane holds 16 rows of one column, the two half-waves and the
// eight waves meet in LDS in a fixed order.
constexpr int BP_WAVES = 8;
__global__ __launch_bounds__(64 * BP_WAVES) void bt_bwd_products_kernel(const float* __restrict__ z1n, const float* __restrict__ z2n, int B, int D,
                                                                        const float* __restrict__ G, float inv_n, float* __restrict__ dzn,
                                                                        float* __restrict__ s) {
  __shared__ float red[2][2 * BP_WAVES][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int view = blockIdx.y;
  const float* A = view == 0 ? z2n : z1n;                  // A(m, k) = A[m * D + k]
  const float* own = view == 0 ? z1n : z2n;                // the view whose gradient this is (zn of the s2 sum)
  float* out = dzn + (int64_t)view * B * D;
  const int n0 = blockIdx.x * 32;
  const int i = lane & 31, kk = lane >> 5;
  const bool nv = (n0 + i) < D;
  // B(k, n): view 0: G[n][k] (stride D over n, 1 over k); view 1: G[k][n]
  const int64_t sbk = view == 0 ? 1 : D, sbn = view == 0 ? D : 1;
  const float* bp = G + (int64_t)(nv ? n0 + i : 0) * sbn;
  float s1 = 0.f, s2 = 0.f;
  for (int m0 = wave * 32; m0 < B; m0 += BP_WAVES * 32) {
    const bool mv = (m0 + i) < B;
    const float* ap = A + (int64_t)(mv ? m0 + i : 0) * D;
    auto load_chunk = [&](int k, float (&a)[8], float (&b)[8]) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kq = k + 2 * u + kk;
        const int kc = kq < D ? kq : 0;
        a[u] = ap[kc];
        b[u] = bp[(int64_t)kc * sbk];
      }
    };
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float a0[8], b0[8], a1[8], b1[8];
    load_chunk(0, a0, b0);
    for (int k = 0; k < D; k += 16) {
      const bool more = k + 16 < D;
      if (more) load_chunk(k + 16, a1, b1);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool live = k + 2 * u + kk < D;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32((mv && live) ? a0[u] : 0.f, (nv && live) ? b0[u] : 0.f, acc, 0, 0, 0);
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { a0[u] = a1[u]; b0[u] = b1[u]; }
      }
    }
    const int n = n0 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m < B && n < D) {
        const float d = inv_n * acc[r];
        out[(int64_t)m * D + n] = d;
        s1 += d;
        s2 += d * own[(int64_t)m * D + n];
      }
    }
  }
  red[0][wave * 2 + kk][i] = s1;
  red[1][wave * 2 + kk][i] = s2;
  __syncthreads();
  if (threadIdx.x < 32 && n0 + (int)threadIdx.x < D) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int q = 0; q < 2 * BP_WAVES; ++q) { a += red[0][q][threadIdx.x]; b += red[1][q][threadIdx.x]; }
    float* sv = s + (int64_t)view * 2 * D;
    sv[n0 + threadIdx.x] = a;
    sv[D + n0 + threadIdx.x] = b;
  }
}

// ---- dz = scale * rstd * (dzn - s1 / N - zn * s2 / N) for both views (bn_bwd_apply_kernel without affine / ReLU, xhat = the saved zn)
__global__ void bt_bwd_apply_kernel(const float* __restrict__ z1n, const float* __restrict__ z2n, int B, int D, const float* __restrict__ rstd,
                                    const float* __restrict__ dzn, const float* __restrict__ s, float inv_n, const float* __restrict__ out_scale,
                                    float* __restrict__ dz1, float* __restrict__ dz2) {
  const int64_t n = (int64_t)B * D;
  const float osc = out_scale ? *out_scale : 1.f;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < 2 * n; t += (int64_t)gridDim.x * blockDim.x) {
    const int view = t >= n;
    const int64_t e = view ? t - n : t;
    const int c = (int)(e % D);
    const float xh = (view ? z2n : z1n)[e];
    const float r = rstd[view * D + c];
    const float d = dzn[t];
    const float* sv = s + (int64_t)view * 2 * D;
    const float o = osc * 1.f * r * (d - sv[c] * inv_n - xh * sv[D + c] * inv_n);
    (view ? dz2 : dz1)[e] = o;
  }
}

}  // namespace

extern "C" int sa_bt_stats2(const float* z1, const float* z2, int64_t ld, int32_t B, int32_t D, float* stats, void* stream) {
  SA_CHECK_ARG(z1 && z2 && stats && B > 0 && D > 0 && ld >= D, "sa_bt_stats2: bad args");
  hipLaunchKernelGGL(bt_stats2_kernel, dim3((D + 63) / 64, 2), dim3(64 * ST_RG), 0, (hipStream_t)stream, z1, z2, ld, B, D, stats);
  SA_LAUNCH_CHECK("sa_bt_stats2");
  return 0;
}

extern "C" int sa_bt_corr(const float* z1, const float* z2, int64_t ld, int32_t B, int32_t D, const float* all_stats, int32_t W, float eps,
                          float momentum, float inv_n, float* mean, float* rstd, float* running_mean, float* running_var, float* z1n, float* z2n,
                          float* c, void* stream) {
  SA_CHECK_ARG(z1 && z2 && all_stats && mean && rstd && z1n && z2n && c && B > 0 && D > 0 && W > 0 && ld >= D, "sa_bt_corr: bad args");
  const int t = (D + 31) / 32;
  hipLaunchKernelGGL(bt_corr_kernel, dim3(t, t), dim3(256), 0, (hipStream_t)stream, z1, z2, ld, B, D, all_stats, W, eps, momentum, inv_n, mean, rstd,
                     running_mean, running_var, z1n, z2n, c);
  SA_LAUNCH_CHECK("sa_bt_corr");
  return 0;
}

extern "C" int sa_bt_bwd_products(const float* z1n, const float* z2n, int32_t B, int32_t D, const float* G, float inv_n, float* dzn, float* sums,
                                  void* stream) {
  SA_CHECK_ARG(z1n && z2n && G && dzn && sums && B > 0 && D > 0, "sa_bt_bwd_products: bad args");
  hipLaunchKernelGGL(bt_bwd_products_kernel, dim3((D + 31) / 32, 2), dim3(64 * BP_WAVES), 0, (hipStream_t)stream, z1n, z2n, B, D, G, inv_n, dzn, sums);
  SA_LAUNCH_CHECK("sa_bt_bwd_products");
  return 0;
}

extern "C" int sa_bt_bwd_apply(const float* z1n, const float* z2n, int32_t B, int32_t D, const float* rstd, const float* dzn, const float* sums,
                               float inv_n, const float* out_scale, float* dz1, float* dz2, void* stream) {
  SA_CHECK_ARG(z1n && z2n && rstd && dzn && sums && dz1 && dz2 && B > 0 && D > 0, "sa_bt_bwd_apply: bad args");
  const int64_t n = 2 * (int64_t)B * D;
  int64_t want = (n + 255) / 256;
  hipLaunchKernelGGL(bt_bwd_apply_kernel, dim3((int)(want < 2048 ? want : 2048)), dim3(256), 0, (hipStream_t)stream, z1n, z2n, B, D, rstd, dzn, sums,
                     inv_n, out_scale, dz1, dz2);
  SA_LAUNCH_CHECK("sa_bt_bwd_apply");
  return 0;
}
