// BatchNorm1d (train mode) pieces + the Barlow Twins loss, fp32 throughout.
//
// The projector / predictor BN (model.py:20,42) and the loss's affine-less BN (utils/loss.py:13,17) are
// split into  column statistics -> [cross-rank combine by the host side] -> apply, so that the same
// kernels serve one GPU and the global-batch-exact data-parallel path.
// The cross-correlation c = z1n^T z2n / B (utils/loss.py:17-19) and the two backward products run on the
// exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): 16.8 MFLOP at B=128, D=256 -- latency, not throughput.
#include "common.h"
#include "../../include/ssl_audio_hip.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

// ---- per-column mean and M2 = sum (x - mean)^2 over the local rows (two-pass).  A block is 64 columns x BN_RG row groups: thread
// (col, rg) walks rows rg, rg + BN_RG, ... and the groups' partial sums meet in LDS (one thread per column was 44 us on the
// projector's [128, 8192] hidden: 128 dependent row visits per pass on 32 CUs' worth of waves)
constexpr int BN_RG = 8;
__global__ __launch_bounds__(64 * BN_RG) void bn_colstats_kernel(const float* __restrict__ x, int64_t ld, int B, int C, float* __restrict__ mean_out,
                                                                 float* __restrict__ m2_out) {
  __shared__ float red[BN_RG][64];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + col;
  const bool live = c < C;
  float s = 0.f;
  if (live)
    for (int b = rg; b < B; b += BN_RG) s += x[(int64_t)b * ld + c];
  red[rg][col] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < BN_RG; ++k) tot += red[k][col];
  const float mean = tot / (float)B;
  __syncthreads();
  float q = 0.f;
  if (live)
    for (int b = rg; b < B; b += BN_RG) {
      const float d = x[(int64_t)b * ld + c] - mean;
      q += d * d;
    }
  red[rg][col] = q;
  __syncthreads();
  if (rg == 0 && live) {
    float m2 = 0.f;
#pragma unroll
    for (int k = 0; k < BN_RG; ++k) m2 += red[k][col];
    mean_out[c] = mean;
    m2_out[c] = m2;
  }
}

// ---- combine per-rank (mean, M2) (Chan et al., equal row counts), produce mean / rstd and update the running buffers
// stats: [W][2][C] (rank-major; [r][0] = mean_r, [r][1] = M2_r), consecutive ranks rank_stride elements apart (several
// tensors' statistics packed into ONE all-gather).  running_var uses the unbiased variance (PyTorch).
__global__ void bn_finalize_kernel(const float* __restrict__ stats, int64_t rank_stride, int W, int rows_per_rank, int C, float eps, float momentum,
                                   float* __restrict__ mean_out, float* __restrict__ rstd_out, float* __restrict__ running_mean,
                                   float* __restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean = 0.f;
  for (int r = 0; r < W; ++r) mean += stats[r * rank_stride + c];
  mean /= (float)W;
  float m2 = 0.f;
  for (int r = 0; r < W; ++r) {
    const float d = stats[r * rank_stride + c] - mean;
    m2 += stats[r * rank_stride + C + c] + (float)rows_per_rank * d * d;
  }
  const float n = (float)W * (float)rows_per_rank;
  mean_out[c] = mean;
  rstd_out[c] = rsqrtf(m2 / n + eps);
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
  if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2 / fmaxf(n - 1.f, 1.f));
}

// ---- y = (x - mean) * rstd [* gamma + beta] [relu]; rstd = rsqrt(var + eps) is given
__global__ void bn_apply_kernel(const float* __restrict__ x, int64_t ld, int B, int C, const float* __restrict__ mean,
                                const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                int relu, float* __restrict__ y_f32, bf16_t* __restrict__ y_bf16, int64_t ldy) {
  const int64_t n = (int64_t)B * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / C), c = (int)(i % C);
    float v = (x[(int64_t)b * ld + c] - mean[c]) * rstd[c];
    if (gamma) v = v * gamma[c] + beta[c];
    if (relu) v = relu_f(v);
    if (y_f32) y_f32[(int64_t)b * ldy + c] = v;
    if (y_bf16) y_bf16[(int64_t)b * ldy + c] = f2bf(v);
  }
}

// ---- backward column sums: s1 = sum_b g, s2 = sum_b g * xhat, with g = dy * [relu mask] (dy w.r.t. BN output)
// dgamma = s2, dbeta = s1 (local contributions).
template <typename DY>
__global__ __launch_bounds__(64 * BN_RG) void bn_bwd_stats_kernel(const DY* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ld, int B, int C,
                                    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, int relu, float* __restrict__ s1_out, float* __restrict__ s2_out) {
  __shared__ float red[2][BN_RG][64];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + col;
  const bool live = c < C;
  float s1 = 0.f, s2 = 0.f;
  if (live) {
    const float mu = mean[c], r = rstd[c];
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    for (int b = rg; b < B; b += BN_RG) {
      const float xh = (x[(int64_t)b * ld + c] - mu) * r;
      float d = (float)dy[(int64_t)b * lddy + c];
      if (relu && !(xh * g + bt > 0.f)) d = 0.f;
      s1 += d;
      s2 += d * xh;
    }
  }
  red[0][rg][col] = s1;
  red[1][rg][col] = s2;
  __syncthreads();
  if (rg == 0 && live) {
    float a = 0.f, b2 = 0.f;
#pragma unroll
    for (int k = 0; k < BN_RG; ++k) { a += red[0][k][col]; b2 += red[1][k][col]; }
    s1_out[c] = a;
    s2_out[c] = b2;
  }
}

// ---- dx = gamma * rstd * (g - s1/N - xhat * s2/N), N = global row count, s1/s2 already summed over ranks
template <typename DY>
__global__ void bn_bwd_apply_kernel(const DY* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ld, int B, int C,
                                    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, int relu, const float* __restrict__ s1, const float* __restrict__ s2,
                                    float inv_n, const float* __restrict__ out_scale, float* __restrict__ dx_f32,
                                    bf16_t* __restrict__ dx_bf16, int64_t lddx) {
  const int64_t n = (int64_t)B * C;
  const float osc = out_scale ? *out_scale : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / C), c = (int)(i % C);
    const float r = rstd[c];
    const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float xh = (x[(int64_t)b * ld + c] - mean[c]) * r;
    float d = (float)dy[(int64_t)b * lddy + c];
    if (relu && !(xh * g + bt > 0.f)) d = 0.f;
    const float o = osc * g * r * (d - s1[c] * inv_n - xh * s2[c] * inv_n);
    if (dx_f32) dx_f32[(int64_t)b * lddx + c] = o;
    if (dx_bf16) dx_bf16[(int64_t)b * lddx + c] = f2bf(o);
  }
}

// ---- small exact-fp32 matmul on v_mfma_f32_32x32x2_f32: C[m][n] = alpha * sum_k A(m,k) * B(k,n)
// A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]  (general strides cover every transpose the loss needs).
// One wave per 32x32 output tile.  Lane l feeds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31].
__global__ __launch_bounds__(64) void matmul_f32_kernel(const float* __restrict__ A, int64_t sam, int64_t sak, const float* __restrict__ B,
                                                        int64_t sbk, int64_t sbn, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                        float alpha) {
  const int lane = threadIdx.x;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int i = lane & 31, kk = lane >> 5;
  const bool mv = (m0 + i) < M, nv = (n0 + i) < N;
  const float* ap = A + (int64_t)(m0 + i) * sam;
  const float* bp = B + (int64_t)(n0 + i) * sbn;
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= K; k += 8) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kq = k + 2 * u + kk;
      a[u] = mv ? ap[(int64_t)kq * sak] : 0.f;
      b[u] = nv ? bp[(int64_t)kq * sbk] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
  }
  for (; k < K; k += 2) {
    const int kq = k + kk;
    const float a = (mv && kq < K) ? ap[(int64_t)kq * sak] : 0.f;
    const float b = (nv && kq < K) ? bp[(int64_t)kq * sbk] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  // C/D layout 32x32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int n = n0 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m < M && n < N) C[(int64_t)m * ldc + n] = alpha * acc[r];
  }
}

// ---- loss = alpha * sum_i (c_ii - 1)^2 + lambda * sum_{i != j} (c_ij [+1])^2 ; G = dL/dc   (utils/loss.py:23-30)
// The loss scalar is summed in a fixed order (no float atomics: bit-reproducible): every block leaves its partial in bt_loss_partials and a
// one-wave launch adds them in block order.  The partials live in the caller's workspace (256 floats): launches on different streams or
// threads bring their own (a device global here was silently shared by all of them).
constexpr int BT_LOSS_BLOCKS = 256;

__global__ __launch_bounds__(256) void bt_loss_grad_kernel(const float* __restrict__ c, int D, float alpha, float lambda, int hsic,
                                                           float* __restrict__ G, float* __restrict__ bt_loss_partials) {
  __shared__ float red[4];
  float s = 0.f;
  const int n = D * D;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
    const int i = idx / D, j = idx % D;
    const float v = c[idx];
    float g, t;
    if (i == j) {
      t = v - 1.f;
      s += alpha * t * t;
      g = 2.f * alpha * t;
    } else {
      t = hsic ? v + 1.f : v;
      s += lambda * t * t;
      g = 2.f * lambda * t;
    }
    if (G) G[idx] = g;
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) bt_loss_partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(64) void bt_loss_finish_kernel(int nblocks, float* __restrict__ loss, const float* __restrict__ bt_loss_partials) {
  // lane l adds the partials l, l + 64, ... in order; the 64 lane sums are then added in lane order by lane 0
  __shared__ float lanes[64];
  float a = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += 64) a += bt_loss_partials[b];
  lanes[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int l = 0; l < 64; ++l) t += lanes[l];
    *loss = t;
  }
}

// ---- flat elementwise: AdamW (torch.optim.AdamW semantics, decoupled decay), EMA, scaled add
// hyper (optional, device): {lr, 1 / (1 - b1^t), 1 / sqrt(1 - b2^t)} replace the scalar arguments -- what changes from step to step then
// lives in memory, so a captured launch (HIP graph) replays with the current values; skip (optional, device): a non-zero word (the
// trainer's non-finite-loss counter) turns the launch into a no-op, so a NaN loss never reaches the weights or the moments.
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                             float lr, float b1, float b2, float eps, float wd, float inv_c1, float inv_sqrt_c2, float grad_scale,
                             bf16_t* __restrict__ p_bf16, const float* __restrict__ hyper, const int32_t* __restrict__ skip) {
  if (skip && *skip) return;
  if (hyper) { lr = hyper[0]; inv_c1 = hyper[1]; inv_sqrt_c2 = hyper[2]; }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * grad_scale;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) * inv_sqrt_c2 + eps;
    pi -= lr * inv_c1 * mi / denom;
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (p_bf16) p_bf16[i] = f2bf(pi);
  }
}

// 16-byte version for buffers whose pointers are 16-byte aligned and whose length is a multiple of 4 (the flat state is)
__global__ __launch_bounds__(256) void adamw_vec4_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                         float4* __restrict__ v, int64_t n4, float lr, float b1, float b2, float eps, float wd,
                                                         float inv_c1, float inv_sqrt_c2, float grad_scale, bf16x4* __restrict__ p_bf16,
                                                         const float* __restrict__ hyper, const int32_t* __restrict__ skip) {
  if (skip && *skip) return;
  if (hyper) { lr = hyper[0]; inv_c1 = hyper[1]; inv_sqrt_c2 = hyper[2]; }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 gi = g[i], pi = p[i], mi = m[i], vi = v[i];
    float gg[4] = {gi.x * grad_scale, gi.y * grad_scale, gi.z * grad_scale, gi.w * grad_scale};
    float pp[4] = {pi.x, pi.y, pi.z, pi.w}, mm[4] = {mi.x, mi.y, mi.z, mi.w}, vv[4] = {vi.x, vi.y, vi.z, vi.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float q = pp[k] * (1.f - lr * wd);
      mm[k] = b1 * mm[k] + (1.f - b1) * gg[k];
      vv[k] = b2 * vv[k] + (1.f - b2) * gg[k] * gg[k];
      const float denom = sqrtf(vv[k]) * inv_sqrt_c2 + eps;
      q -= lr * inv_c1 * mm[k] / denom;
      pp[k] = q;
    }
    p[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    m[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    v[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (p_bf16) p_bf16[i] = bf16x4{f2bf(pp[0]), f2bf(pp[1]), f2bf(pp[2]), f2bf(pp[3])};
  }
}

__global__ void ema_kernel(float* __restrict__ tgt, const float* __restrict__ src, int64_t n, float beta, const int32_t* __restrict__ skip) {
  if (skip && *skip) return;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    tgt[i] = tgt[i] * beta + (1.f - beta) * src[i];
}

__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n, float a) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] += a * x[i];
}

// ---- LARS (utils/utils.py:150-189): dp = g (+ wd * p); q = eta * |p| / |dp| (1 if either norm is 0); mu = momentum * mu + q * dp;
// p -= lr * mu.  Pass 1: the two squared norms into scratch[0..1]; pass 2: the update (also refreshes the bf16 weight copy).
__global__ __launch_bounds__(256) void lars_norms_kernel(const float* __restrict__ p, const float* __restrict__ g, int64_t n, float wd,
                                                         float* __restrict__ scratch) {
  __shared__ float red[8];
  float sp = 0.f, su = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float pi = p[i], di = g[i] + wd * pi;
    sp += pi * pi;
    su += di * di;
  }
  sp = block_sum_256(sp, red);
  su = block_sum_256(su, red + 4);
  if (threadIdx.x == 0) {
    atomicAdd(scratch + 0, sp);
    atomicAdd(scratch + 1, su);
  }
}

__global__ void lars_update_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mu, int64_t n, float lr, float wd,
                                   float momentum, float eta, int adapt, const float* __restrict__ scratch, bf16_t* __restrict__ p_bf16) {
  float q = 1.f;
  if (adapt) {
    const float pn = sqrtf(scratch[0]), un = sqrtf(scratch[1]);
    q = (pn > 0.f && un > 0.f) ? eta * pn / un : 1.f;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float pi = p[i];
    const float di = (g[i] + wd * pi) * q;
    const float mi = mu[i] * momentum + di;
    const float po = pi - lr * mi;
    mu[i] = mi;
    p[i] = po;
    if (p_bf16) p_bf16[i] = f2bf(po);
  }
}

inline int flat_grid(int64_t n, int per_thread = 1) {
  const int64_t want = (n + 256LL * per_thread - 1) / (256LL * per_thread);
  return (int)(want < 4096 ? (want < 1 ? 1 : want) : 4096);
}

}  // namespace

extern "C" int sa_bn_colstats(const float* x, int64_t ld, int32_t B, int32_t C, float* mean, float* m2, void* stream) {
  SA_CHECK_ARG(x && mean && m2 && B > 0 && C > 0, "sa_bn_colstats: bad args");
  hipLaunchKernelGGL(bn_colstats_kernel, dim3((C + 63) / 64), dim3(64 * BN_RG), 0, (hipStream_t)stream, x, ld, B, C, mean, m2);
  SA_LAUNCH_CHECK("sa_bn_colstats");
  return 0;
}

extern "C" int sa_bn_finalize(const float* stats, int64_t rank_stride, int32_t W, int32_t rows_per_rank, int32_t C, float eps, float momentum,
                              float* mean, float* rstd, float* running_mean, float* running_var, void* stream) {
  if (rank_stride == 0) rank_stride = 2 * (int64_t)C;
  SA_CHECK_ARG(stats && mean && rstd && W > 0 && rows_per_rank > 0 && C > 0 && rank_stride >= 2 * (int64_t)C, "sa_bn_finalize: bad args");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, rank_stride, W, rows_per_rank, C, eps, momentum,
                     mean, rstd, running_mean, running_var);
  SA_LAUNCH_CHECK("sa_bn_finalize");
  return 0;
}

extern "C" int sa_bn_apply(const float* x, int64_t ld, int32_t B, int32_t C, const float* mean, const float* rstd, const float* gamma,
                           const float* beta, int32_t relu, float* y_f32, void* y_bf16, int64_t ldy, void* stream) {
  SA_CHECK_ARG(x && mean && rstd && (y_f32 || y_bf16) && B > 0 && C > 0, "sa_bn_apply: bad args");
  SA_CHECK_ARG((gamma == nullptr) == (beta == nullptr), "sa_bn_apply: gamma and beta must be given together");
  hipLaunchKernelGGL(bn_apply_kernel, dim3(flat_grid((int64_t)B * C)), dim3(256), 0, (hipStream_t)stream, x, ld, B, C, mean, rstd, gamma, beta,
                     relu, y_f32, (bf16_t*)y_bf16, ldy);
  SA_LAUNCH_CHECK("sa_bn_apply");
  return 0;
}

extern "C" int sa_bn_bwd_stats(const void* dy, int32_t dy_is_bf16, int64_t lddy, const float* x, int64_t ld, int32_t B, int32_t C,
                               const float* mean, const float* rstd, const float* gamma, const float* beta, int32_t relu, float* s1,
                               float* s2, void* stream) {
  SA_CHECK_ARG(dy && x && mean && rstd && s1 && s2 && B > 0 && C > 0, "sa_bn_bwd_stats: bad args");
  if (dy_is_bf16)
    hipLaunchKernelGGL((bn_bwd_stats_kernel<bf16_t>), dim3((C + 63) / 64), dim3(64 * BN_RG), 0, (hipStream_t)stream, (const bf16_t*)dy, lddy, x, ld, B,
                       C, mean, rstd, gamma, beta, relu, s1, s2);
  else
    hipLaunchKernelGGL((bn_bwd_stats_kernel<float>), dim3((C + 63) / 64), dim3(64 * BN_RG), 0, (hipStream_t)stream, (const float*)dy, lddy, x, ld, B, C,
                       mean, rstd, gamma, beta, relu, s1, s2);
  SA_LAUNCH_CHECK("sa_bn_bwd_stats");
  return 0;
}

extern "C" int sa_bn_bwd_apply(const void* dy, int32_t dy_is_bf16, int64_t lddy, const float* x, int64_t ld, int32_t B, int32_t C,
                               const float* mean, const float* rstd, const float* gamma, const float* beta, int32_t relu, const float* s1,
                               const float* s2, float inv_n, const float* out_scale, float* dx_f32, void* dx_bf16, int64_t lddx,
                               void* stream) {
  SA_CHECK_ARG(dy && x && mean && rstd && s1 && s2 && (dx_f32 || dx_bf16) && B > 0 && C > 0, "sa_bn_bwd_apply: bad args");
  const int grid = flat_grid((int64_t)B * C);
  if (dy_is_bf16)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, lddy, x, ld, B, C, mean,
                       rstd, gamma, beta, relu, s1, s2, inv_n, out_scale, dx_f32, (bf16_t*)dx_bf16, lddx);
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)dy, lddy, x, ld, B, C, mean,
                       rstd, gamma, beta, relu, s1, s2, inv_n, out_scale, dx_f32, (bf16_t*)dx_bf16, lddx);
  SA_LAUNCH_CHECK("sa_bn_bwd_apply");
  return 0;
}

extern "C" int sa_matmul_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C, int64_t ldc,
                             int32_t M, int32_t N, int32_t K, float alpha, void* stream) {
  SA_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "sa_matmul_f32: bad args");
  hipLaunchKernelGGL(matmul_f32_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(64), 0, (hipStream_t)stream, A, sam, sak, B, sbk, sbn, C, ldc,
                     M, N, K, alpha);
  SA_LAUNCH_CHECK("sa_matmul_f32");
  return 0;
}

extern "C" int64_t sa_bt_loss_workspace_bytes(void) { return (int64_t)BT_LOSS_BLOCKS * sizeof(float); }

extern "C" int sa_bt_loss_grad(const float* c, int32_t D, float alpha, float lambda, int32_t hsic, float* loss, float* G, float* ws, void* stream) {
  SA_CHECK_ARG(c && loss && D > 0, "sa_bt_loss_grad: bad args");
  SA_CHECK_ARG(ws, "sa_bt_loss_grad: needs a workspace (sa_bt_loss_workspace_bytes)");
  int grid = (int)(((int64_t)D * D + 255) / 256 < BT_LOSS_BLOCKS ? ((int64_t)D * D + 255) / 256 : BT_LOSS_BLOCKS);
  hipLaunchKernelGGL(bt_loss_grad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, c, D, alpha, lambda, hsic, G, ws);
  hipLaunchKernelGGL(bt_loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, grid, loss, ws);
  SA_LAUNCH_CHECK("sa_bt_loss_grad");
  return 0;
}

namespace {
int adamw_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                 float inv_c1, float inv_sqrt_c2, float grad_scale, void* p_bf16, const float* hyper, const int32_t* skip, void* stream) {
  const bool vec = (n % 4 == 0) && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && (((uintptr_t)p_bf16 & 7) == 0);
  if (vec)
    hipLaunchKernelGGL(adamw_vec4_kernel, dim3(flat_grid(n / 4, 2)), dim3(256), 0, (hipStream_t)stream, (float4*)p, (const float4*)g, (float4*)m,
                       (float4*)v, n / 4, lr, beta1, beta2, eps, weight_decay, inv_c1, inv_sqrt_c2, grad_scale, (bf16x4*)p_bf16, hyper, skip);
  else
    hipLaunchKernelGGL(adamw_kernel, dim3(flat_grid(n, 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, inv_c1, inv_sqrt_c2, grad_scale, (bf16_t*)p_bf16, hyper, skip);
  SA_LAUNCH_CHECK("sa_adamw_step");
  return 0;
}
}  // namespace

extern "C" int sa_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int32_t step, float grad_scale, void* p_bf16, void* stream) {
  SA_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1, "sa_adamw_step: bad args");
  if (n == 0) return 0;
  const double c1 = 1.0 - pow((double)beta1, (double)step), c2 = 1.0 - pow((double)beta2, (double)step);
  return adamw_launch(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, (float)(1.0 / c1), (float)(1.0 / sqrt(c2)), grad_scale, p_bf16, nullptr,
                      nullptr, stream);
}

extern "C" int sa_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper3, float beta1, float beta2, float eps,
                                 float weight_decay, float grad_scale, void* p_bf16, const int32_t* skip_flag, void* stream) {
  SA_CHECK_ARG(p && g && m && v && n >= 0 && hyper3, "sa_adamw_step_dev: bad args (hyper3 = device {lr, 1/(1-b1^t), 1/sqrt(1-b2^t)})");
  if (n == 0) return 0;
  return adamw_launch(p, g, m, v, n, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, grad_scale, p_bf16, hyper3, skip_flag, stream);
}

namespace {
// bf16 -> fp32 over a flat buffer: the way back from the bf16 gradient buckets of the data-parallel exchange (train.GradSync, grad_dtype bf16)
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + i);
      *reinterpret_cast<float4*>(dst + i) = make_float4(bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3]));
      *reinterpret_cast<float4*>(dst + i + 4) = make_float4(bf2f(v[4]), bf2f(v[5]), bf2f(v[6]), bf2f(v[7]));
    } else {
      for (int64_t k = i; k < n; ++k) dst[k] = bf2f(src[k]);
    }
  }
}
}  // namespace

extern "C" int sa_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
  SA_CHECK_ARG(n >= 0 && (n == 0 || (src && dst)), "sa_cast_bf16_to_f32: bad args");
  if (n == 0) return 0;
  SA_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "sa_cast_bf16_to_f32: pointers must be 16-byte aligned");
  const int64_t want = (n + 8 * 256 - 1) / (8 * 256);
  const int grid = (int)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, n);
  SA_LAUNCH_CHECK("sa_cast_bf16_to_f32");
  return 0;
}

extern "C" int sa_ema_update(float* target, const float* online, int64_t n, float beta, void* stream) {
  SA_CHECK_ARG(target && online && n >= 0, "sa_ema_update: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(ema_kernel, dim3(flat_grid(n, 4)), dim3(256), 0, (hipStream_t)stream, target, online, n, beta, (const int32_t*)nullptr);
  SA_LAUNCH_CHECK("sa_ema_update");
  return 0;
}

extern "C" int sa_ema_update_gated(float* target, const float* online, int64_t n, float beta, const int32_t* skip_flag, void* stream) {
  SA_CHECK_ARG(target && online && n >= 0, "sa_ema_update_gated: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(ema_kernel, dim3(flat_grid(n, 4)), dim3(256), 0, (hipStream_t)stream, target, online, n, beta, skip_flag);
  SA_LAUNCH_CHECK("sa_ema_update_gated");
  return 0;
}

// flag[0] += number of non-finite values among x[0..n) (a handful of loss scalars; one wave)
__global__ void count_nonfinite_kernel(const float* __restrict__ x, int64_t n, int32_t* __restrict__ flag) {
  int bad = 0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) bad += !isfinite(x[i]);
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o);
  if (threadIdx.x == 0 && bad) atomicAdd(flag, bad);
}

extern "C" int sa_count_nonfinite(const float* x, int64_t n, int32_t* flag, void* stream) {
  SA_CHECK_ARG(x && flag && n >= 0, "sa_count_nonfinite: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(count_nonfinite_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, x, n, flag);
  SA_LAUNCH_CHECK("sa_count_nonfinite");
  return 0;
}

extern "C" int sa_axpy_f32(float* y, const float* x, int64_t n, float a, void* stream) {
  SA_CHECK_ARG(y && x && n >= 0, "sa_axpy_f32: bad args");
  if (n == 0) return 0;
  hipLaunchKernelGGL(axpy_kernel, dim3(flat_grid(n, 4)), dim3(256), 0, (hipStream_t)stream, y, x, n, a);
  SA_LAUNCH_CHECK("sa_axpy_f32");
  return 0;
}

extern "C" int sa_lars_step(float* p, const float* g, float* mu, int64_t n, float lr, float weight_decay, float momentum, float eta,
                            int32_t lars_adaptation, float* scratch2, void* p_bf16, void* stream) {
  SA_CHECK_ARG(p && g && mu && n >= 0 && (!lars_adaptation || scratch2), "sa_lars_step: bad args (the trust ratio needs a 2-float scratch)");
  if (n == 0) return 0;
  if (lars_adaptation) {
    if (hipMemsetAsync(scratch2, 0, 2 * sizeof(float), (hipStream_t)stream) != hipSuccess) {
      sa_set_error("sa_lars_step: memset failed");
      return 2;
    }
    hipLaunchKernelGGL(lars_norms_kernel, dim3(flat_grid(n, 8)), dim3(256), 0, (hipStream_t)stream, p, g, n, weight_decay, scratch2);
  }
  hipLaunchKernelGGL(lars_update_kernel, dim3(flat_grid(n, 4)), dim3(256), 0, (hipStream_t)stream, p, g, mu, n, lr, weight_decay, momentum, eta,
                     lars_adaptation, scratch2, (bf16_t*)p_bf16);
  SA_LAUNCH_CHECK("sa_lars_step");
  return 0;
}
