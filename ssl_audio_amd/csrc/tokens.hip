// Token-sequence bookkeeping around the encoder (models/mae.py:349-365, 309-347, 411-420, 460-463):
// CLS row fill and its gradient, row gather / scatter for random masking, mean pooling.  All HBM-trivial.
#include "common.h"
#include "../../include/ssl_audio_hip.h"

namespace {

// x[s][0][:] = cls + pos0   (cls_token + pos_embed[:, :1], models/mae.py:361-362)
__global__ void fill_cls_kernel(float* __restrict__ x, int S, int64_t seq_stride, int d, const float* __restrict__ cls,
                                const float* __restrict__ pos0) {
  const int64_t n = (int64_t)S * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int s = (int)(i / d), j = (int)(i % d);
    x[(int64_t)s * seq_stride + j] = cls[j] + (pos0 ? pos0[j] : 0.f);
  }
}

// dcls[j] += sum_s dx[s][0][j].  One 1024-thread block per 64 columns: wave w adds the rows s = w, w + 16, ... and the sixteen wave
// sums meet in LDS in wave order -- a fixed summation order and a single writer per column (no float atomics: bit-reproducible).
__global__ __launch_bounds__(1024) void cls_grad_kernel(const float* __restrict__ dx, int S, int64_t seq_stride, int d, float* __restrict__ dcls) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  float a = 0.f;
  if (j < d)
    for (int s = wave; s < S; s += 16) a += dx[(int64_t)s * seq_stride + j];
  red[wave][lane] = a;
  __syncthreads();
  if (wave == 0 && j < d) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][lane];
    dcls[j] += t;
  }
}

// dst[s][dst_row0 + k][:] = src[s][src_row0 + idx[s][k]][:]  (k < n_idx); fp32 rows of width d (multiple of 4)
__global__ void gather_rows_kernel(const float* __restrict__ src, int64_t src_seq_stride, int src_row0, const int* __restrict__ idx, int n_idx,
                                   float* __restrict__ dst, int64_t dst_seq_stride, int dst_row0, int S, int d) {
  const int nv = d >> 2;
  const int64_t n = (int64_t)S * n_idx * nv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % nv);
    const int64_t rk = i / nv;
    const int k = (int)(rk % n_idx), s = (int)(rk / n_idx);
    const int r = idx[(int64_t)s * n_idx + k];
    reinterpret_cast<float4*>(dst + (int64_t)s * dst_seq_stride + (int64_t)(dst_row0 + k) * d)[c] =
        reinterpret_cast<const float4*>(src + (int64_t)s * src_seq_stride + (int64_t)(src_row0 + r) * d)[c];
  }
}

// dst[s][dst_row0 + idx[s][k]][:] += src[s][src_row0 + k][:]   (indices within a sequence are distinct)
__global__ void scatter_add_rows_kernel(const float* __restrict__ src, int64_t src_seq_stride, int src_row0, const int* __restrict__ idx,
                                        int n_idx, float* __restrict__ dst, int64_t dst_seq_stride, int dst_row0, int S, int d) {
  const int nv = d >> 2;
  const int64_t n = (int64_t)S * n_idx * nv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % nv);
    const int64_t rk = i / nv;
    const int k = (int)(rk % n_idx), s = (int)(rk / n_idx);
    const int r = idx[(int64_t)s * n_idx + k];
    float4* o = reinterpret_cast<float4*>(dst + (int64_t)s * dst_seq_stride + (int64_t)(dst_row0 + r) * d) + c;
    const float4 v = reinterpret_cast<const float4*>(src + (int64_t)s * src_seq_stride + (int64_t)(src_row0 + k) * d)[c];
    float4 t = *o;
    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    *o = t;
  }
}


// ---- mean pooling over the patch tokens (rows 1..N-1) of a [S][N][d] sequence (models/mae.py:461-462) and its adjoint
__global__ void mean_tokens_kernel(const float* __restrict__ y, int S, int N, int d, float* __restrict__ out) {
  const int nv = d >> 2;
  const int64_t n = (int64_t)S * nv;
  const float inv = 1.f / (float)(N - 1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % nv), s = (int)(i / nv);
    const float4* p = reinterpret_cast<const float4*>(y + ((int64_t)s * N + 1) * d) + c;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < N - 1; ++r) {
      const float4 v = p[(int64_t)r * nv];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    reinterpret_cast<float4*>(out + (int64_t)s * d)[c] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
  }
}

__global__ void mean_tokens_bwd_kernel(const float* __restrict__ dout, int S, int N, int d, float* __restrict__ dy) {
  const int nv = d >> 2;
  const int64_t n = (int64_t)S * N * nv;
  const float inv = 1.f / (float)(N - 1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % nv);
    const int64_t sr = i / nv;
    const int r = (int)(sr % N), s = (int)(sr / N);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r > 0) {
      const float4 g = reinterpret_cast<const float4*>(dout + (int64_t)s * d)[c];
      v = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
    }
    reinterpret_cast<float4*>(dy)[i] = v;
  }
}

// ---- MAE decoder input (models/mae.py:413-420): out[b][0] = x[b][0] + pos[0];
// out[b][1 + l] = (ids_restore[b][l] < keep ? x[b][1 + ids_restore[b][l]] : mask_token) + pos[1 + l]
__global__ void mae_unshuffle_kernel(const float* __restrict__ x, int keep, const float* __restrict__ mask_token, const float* __restrict__ pos,
                                     const int* __restrict__ ids_restore, int B, int L, int d, float* __restrict__ out) {
  const int nv = d >> 2;
  const int64_t n = (int64_t)B * (L + 1) * nv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % nv);
    const int64_t br = i / nv;
    const int row = (int)(br % (L + 1)), b = (int)(br / (L + 1));
    float4 v;
    if (row == 0) {
      v = reinterpret_cast<const float4*>(x + (int64_t)b * (keep + 1) * d)[c];
    } else {
      const int r = ids_restore[(int64_t)b * L + row - 1];
      v = r < keep ? reinterpret_cast<const float4*>(x + ((int64_t)b * (keep + 1) + 1 + r) * d)[c] : reinterpret_cast<const float4*>(mask_token)[c];
    }
    const float4 pz = reinterpret_cast<const float4*>(pos + (int64_t)row * d)[c];
    reinterpret_cast<float4*>(out)[i] = make_float4(v.x + pz.x, v.y + pz.y, v.z + pz.z, v.w + pz.w);
  }
}

constexpr int UNSH_MAXD = 2048, UNSH_GROUPS = 128;      // the caller's workspace: per row group the mask-token gradient partial, [UNSH_GROUPS][UNSH_MAXD] floats

// adjoint: dx rows are a permutation of the kept rows of dout (written, not accumulated); dmask_token += sum over masked rows.
// One block column per 64 float4 columns, blockIdx.y strides over (b, row); the mask-token partial stays in registers.
__global__ __launch_bounds__(256) void mae_unshuffle_bwd_kernel(const float* __restrict__ dout, int keep, const int* __restrict__ ids_restore, int B, int L,
                                                                int d, float* __restrict__ dx, float* __restrict__ dmask,
                                                                float* __restrict__ unshuffle_partials) {
  __shared__ float4 red[4][64];
  const int nv = d >> 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t rows = (int64_t)B * (L + 1);
  for (int64_t br = (int64_t)blockIdx.y * 4 + wave; br < rows; br += (int64_t)gridDim.y * 4) {
    if (c >= nv) break;
    const int row = (int)(br % (L + 1)), b = (int)(br / (L + 1));
    const float4 g = reinterpret_cast<const float4*>(dout + br * d)[c];
    if (row == 0) {
      reinterpret_cast<float4*>(dx + (int64_t)b * (keep + 1) * d)[c] = g;
    } else {
      const int r = ids_restore[(int64_t)b * L + row - 1];
      if (r < keep) reinterpret_cast<float4*>(dx + ((int64_t)b * (keep + 1) + 1 + r) * d)[c] = g;
      else { acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w; }
    }
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && c < nv && dmask) {
    float4 t = red[0][lane];
    for (int w = 1; w < 4; ++w) { t.x += red[w][lane].x; t.y += red[w][lane].y; t.z += red[w][lane].z; t.w += red[w][lane].w; }
    // the row groups' partials are added in group order by mae_unshuffle_sum_kernel (no float atomics: bit-reproducible)
    reinterpret_cast<float4*>(unshuffle_partials + (int64_t)blockIdx.y * UNSH_MAXD)[c] = t;
  }
}

__global__ void mae_unshuffle_sum_kernel(int ngroups, int d, float* __restrict__ dmask, const float* __restrict__ unshuffle_partials) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  float t = 0.f;
  for (int y = 0; y < ngroups; ++y) t += unshuffle_partials[(int64_t)y * UNSH_MAXD + c];
  dmask[c] += t;
}

// ---- MAE reconstruction loss (models/mae.py:437-453, patchify :282-293; one input channel):
// target[b][l][py*pw + px] = img[b][gy*ph + py][gx*pw + px], l = gy*gw + gx;  loss = sum_l mask * mean_p (pred - target)^2 / sum mask.
// norm_pix (models/mae.py:443-446): target <- (target - mean_p target) / sqrt(var_p target + 1e-6), var unbiased (torch.var's default).
// Pass 1: one wave per (b, l) row; every block leaves {sum mask * mean_p (.)^2, sum mask} in the caller's workspace ([2048][2] floats) and a
// one-wave launch adds the blocks in block order (no float atomics: bit-reproducible) -> acc[0], acc[1].   Pass 2 (finalize): loss = acc[0] / acc[1].
constexpr int MAE_LOSS_BLOCKS = 2048;

// the patch's pixels of this lane (p = lane, lane + 64, ...) and, with norm_pix, its mean and 1 / sqrt(var + 1e-6) over the whole patch
template <int MAXP>
__device__ __forceinline__ void patch_pixels(const float* ip, int T, int pw, int P, int lane, int norm_pix, float (&t)[MAXP], float& mu, float& rs) {
#pragma unroll
  for (int q = 0; q < MAXP; ++q) {
    const int p = lane + 64 * q;
    t[q] = p < P ? ip[(int64_t)(p / pw) * T + (p % pw)] : 0.f;
  }
  mu = 0.f; rs = 1.f;
  if (norm_pix) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < MAXP; ++q) s += t[q];
    mu = wave_sum(s) / (float)P;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < MAXP; ++q) { const float d = (lane + 64 * q < P) ? t[q] - mu : 0.f; v += d * d; }
    rs = rsqrtf(wave_sum(v) / (float)(P - 1) + 1.e-6f);
  }
}
constexpr int MAE_MAXP = 8;      // patches of up to 512 pixels (16 x 16 = 256, 64 x 2 = 128, 16 x 8 = 128, 8 x 8 = 64)

__global__ __launch_bounds__(256) void mae_loss_rows_kernel(const float* __restrict__ pred, int64_t pred_seq_stride, int pred_row0,
                                                            const float* __restrict__ img, const float* __restrict__ mask,
                                                            int B, int F, int T, int ph, int pw, int norm_pix,
                                                            float* __restrict__ mae_loss_partials) {
  __shared__ float red[8];
  const int gh = F / ph, gw = T / pw, L = gh * gw, P = ph * pw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float num = 0.f, den = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < (int64_t)B * L; row += (int64_t)gridDim.x * 4) {
    const float m = mask[row];
    if (lane == 0) den += m;
    if (m == 0.f) continue;                               // visible patches do not enter the loss
    const int l = (int)(row % L), b = (int)(row / L);
    const int gy = l / gw, gx = l % gw;
    const float* ip = img + ((int64_t)b * F + (int64_t)gy * ph) * T + (int64_t)gx * pw;
    const float* pr = pred + (int64_t)b * pred_seq_stride + (int64_t)(pred_row0 + l) * P;
    float t[MAE_MAXP], mu, rs;
    patch_pixels<MAE_MAXP>(ip, T, pw, P, lane, norm_pix, t, mu, rs);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < MAE_MAXP; ++q) {
      const int p = lane + 64 * q;
      if (p < P) {
        const float e = pr[p] - (t[q] - mu) * rs;
        s += e * e;
      }
    }
    s = wave_sum(s);
    if (lane == 0) num += m * s / (float)P;
  }
  num = wave_sum(num); den = wave_sum(den);
  if (lane == 0) { red[wave] = num; red[4 + wave] = den; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mae_loss_partials[2 * blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    mae_loss_partials[2 * blockIdx.x + 1] = red[4] + red[5] + red[6] + red[7];
  }
}

__global__ __launch_bounds__(64) void mae_loss_sum_kernel(int nblocks, float* __restrict__ acc, float* __restrict__ loss,
                                                          const float* __restrict__ mae_loss_partials) {
  __shared__ float lanes[2][64];
  float a = 0.f, b = 0.f;
  for (int k = threadIdx.x; k < nblocks; k += 64) { a += mae_loss_partials[2 * k]; b += mae_loss_partials[2 * k + 1]; }
  lanes[0][threadIdx.x] = a; lanes[1][threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    float ta = 0.f, tb = 0.f;
    for (int l = 0; l < 64; ++l) { ta += lanes[0][l]; tb += lanes[1][l]; }
    acc[0] = ta; acc[1] = tb;
    if (loss) loss[0] = ta / tb;
  }
}

__global__ void mae_loss_finalize_kernel(const float* __restrict__ acc, float* __restrict__ loss) { loss[0] = acc[0] / acc[1]; }

// dpred[b][row0 + l][p] = gscale * 2 * mask * (pred - target) / (P * sum mask), rows below row0 (the CLS prediction) get 0;
// pred / dpred share the [B][row0 + L][P] layout (seq_stride elements per clip).  gscale = upstream gradient (device scalar).
// One wave per (b, r) row (the per-patch statistics of norm_pix are a wave reduction).
__global__ __launch_bounds__(256) void mae_loss_bwd_kernel(const float* __restrict__ pred, int64_t seq_stride, int row0, const float* __restrict__ img,
                                                           const float* __restrict__ mask, const float* __restrict__ acc,
                                                           const float* __restrict__ gscale, int B, int F, int T, int ph, int pw, int norm_pix,
                                                           float* __restrict__ dpred) {
  const int gh = F / ph, gw = T / pw, L = gh * gw, P = ph * pw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float k = gscale[0] * 2.f / ((float)P * acc[1]);
  for (int64_t br = (int64_t)blockIdx.x * 4 + wave; br < (int64_t)B * (row0 + L); br += (int64_t)gridDim.x * 4) {
    const int r = (int)(br % (row0 + L)), b = (int)(br / (row0 + L));
    const int64_t off = (int64_t)b * seq_stride + (int64_t)r * P;
    const int l = r - row0;
    const float m = r >= row0 ? mask[(int64_t)b * L + l] : 0.f;        // wave-uniform
    if (m == 0.f) {
      for (int p = lane; p < P; p += 64) dpred[off + p] = 0.f;
      continue;
    }
    const int gy = l / gw, gx = l % gw;
    const float* ip = img + ((int64_t)b * F + (int64_t)gy * ph) * T + (int64_t)gx * pw;
    float t[MAE_MAXP], mu, rs;
    patch_pixels<MAE_MAXP>(ip, T, pw, P, lane, norm_pix, t, mu, rs);
#pragma unroll
    for (int q = 0; q < MAE_MAXP; ++q) {
      const int p = lane + 64 * q;
      if (p < P) dpred[off + p] = k * m * (pred[off + p] - (t[q] - mu) * rs);
    }
  }
}

// ---- grouped token-row sums for the chunked embedding of the eval path (encode_vit, utils/utils.py:278-314):
// out[s][g][:] (+)= scale * sum_{t < count} y[s][row0 + g * group_stride + t][:]
__global__ void token_group_sum_kernel(const float* __restrict__ y, int S, int N, int d, int row0, int G, int group_stride, int count, float scale,
                                       int accumulate, float* __restrict__ out) {
  const int nv = d >> 2;
  const int64_t n = (int64_t)S * G * nv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % nv);
    const int64_t sg = i / nv;
    const int g = (int)(sg % G), s = (int)(sg / G);
    const float4* p = reinterpret_cast<const float4*>(y + ((int64_t)s * N + row0 + (int64_t)g * group_stride) * d) + c;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < count; ++t) {
      const float4 v = p[(int64_t)t * nv];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    float4* o = reinterpret_cast<float4*>(out) + i;
    float4 r = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
    if (accumulate) { const float4 old = *o; r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w; }
    *o = r;
  }
}

inline int grid_for(int64_t n) {
  const int64_t want = (n + 255) / 256;
  return (int)(want < 4096 ? (want < 1 ? 1 : want) : 4096);
}

}  // namespace

extern "C" int sa_fill_cls(float* x, int32_t S, int64_t seq_stride, int32_t d, const float* cls, const float* pos0, void* stream) {
  SA_CHECK_ARG(x && cls && S > 0 && d > 0, "sa_fill_cls: bad args");
  hipLaunchKernelGGL(fill_cls_kernel, dim3(grid_for((int64_t)S * d)), dim3(256), 0, (hipStream_t)stream, x, S, seq_stride, d, cls, pos0);
  SA_LAUNCH_CHECK("sa_fill_cls");
  return 0;
}

extern "C" int sa_cls_grad(const float* dx, int32_t S, int64_t seq_stride, int32_t d, float* dcls, void* stream) {
  SA_CHECK_ARG(dx && dcls && S > 0 && d > 0, "sa_cls_grad: bad args");
  hipLaunchKernelGGL(cls_grad_kernel, dim3((d + 63) / 64), dim3(1024), 0, (hipStream_t)stream, dx, S, seq_stride, d, dcls);
  SA_LAUNCH_CHECK("sa_cls_grad");
  return 0;
}

extern "C" int sa_gather_rows(const float* src, int64_t src_seq_stride, int32_t src_row0, const int32_t* idx, int32_t n_idx, float* dst,
                              int64_t dst_seq_stride, int32_t dst_row0, int32_t S, int32_t d, void* stream) {
  SA_CHECK_ARG(src && idx && dst && S > 0 && n_idx > 0 && d > 0 && d % 4 == 0, "sa_gather_rows: bad args");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((int64_t)S * n_idx * (d / 4))), dim3(256), 0, (hipStream_t)stream, src, src_seq_stride,
                     src_row0, idx, n_idx, dst, dst_seq_stride, dst_row0, S, d);
  SA_LAUNCH_CHECK("sa_gather_rows");
  return 0;
}

extern "C" int sa_scatter_add_rows(const float* src, int64_t src_seq_stride, int32_t src_row0, const int32_t* idx, int32_t n_idx, float* dst,
                                   int64_t dst_seq_stride, int32_t dst_row0, int32_t S, int32_t d, void* stream) {
  SA_CHECK_ARG(src && idx && dst && S > 0 && n_idx > 0 && d > 0 && d % 4 == 0, "sa_scatter_add_rows: bad args");
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for((int64_t)S * n_idx * (d / 4))), dim3(256), 0, (hipStream_t)stream, src,
                     src_seq_stride, src_row0, idx, n_idx, dst, dst_seq_stride, dst_row0, S, d);
  SA_LAUNCH_CHECK("sa_scatter_add_rows");
  return 0;
}

extern "C" int sa_mean_tokens_fwd(const float* y, int32_t S, int32_t N, int32_t d, float* out, void* stream) {
  SA_CHECK_ARG(y && out && S > 0 && N > 1 && d > 0 && d % 4 == 0, "sa_mean_tokens_fwd: bad args (need N > 1, d %% 4 == 0)");
  hipLaunchKernelGGL(mean_tokens_kernel, dim3(grid_for((int64_t)S * (d / 4))), dim3(256), 0, (hipStream_t)stream, y, S, N, d, out);
  SA_LAUNCH_CHECK("sa_mean_tokens_fwd");
  return 0;
}

extern "C" int sa_mean_tokens_bwd(const float* dout, int32_t S, int32_t N, int32_t d, float* dy, void* stream) {
  SA_CHECK_ARG(dout && dy && S > 0 && N > 1 && d > 0 && d % 4 == 0, "sa_mean_tokens_bwd: bad args (need N > 1, d %% 4 == 0)");
  hipLaunchKernelGGL(mean_tokens_bwd_kernel, dim3(grid_for((int64_t)S * N * (d / 4))), dim3(256), 0, (hipStream_t)stream, dout, S, N, d, dy);
  SA_LAUNCH_CHECK("sa_mean_tokens_bwd");
  return 0;
}

extern "C" int sa_mae_unshuffle_fwd(const float* x, int32_t keep, const float* mask_token, const float* pos, const int32_t* ids_restore, int32_t B,
                                    int32_t L, int32_t d, float* out, void* stream) {
  SA_CHECK_ARG(x && mask_token && pos && ids_restore && out && B > 0 && L > 0 && keep >= 0 && keep <= L && d > 0 && d % 4 == 0,
               "sa_mae_unshuffle_fwd: bad args");
  hipLaunchKernelGGL(mae_unshuffle_kernel, dim3(grid_for((int64_t)B * (L + 1) * (d / 4))), dim3(256), 0, (hipStream_t)stream, x, keep, mask_token,
                     pos, ids_restore, B, L, d, out);
  SA_LAUNCH_CHECK("sa_mae_unshuffle_fwd");
  return 0;
}

extern "C" int64_t sa_mae_unshuffle_bwd_workspace_bytes(void) { return (int64_t)UNSH_GROUPS * UNSH_MAXD * sizeof(float); }
extern "C" int64_t sa_mae_recon_loss_workspace_bytes(void) { return (int64_t)2 * MAE_LOSS_BLOCKS * sizeof(float); }

extern "C" int sa_mae_unshuffle_bwd(const float* dout, int32_t keep, const int32_t* ids_restore, int32_t B, int32_t L, int32_t d, float* dx,
                                    float* dmask_token, float* ws, void* stream) {
  SA_CHECK_ARG(dout && ids_restore && dx && B > 0 && L > 0 && keep >= 0 && keep <= L && d > 0 && d % 4 == 0, "sa_mae_unshuffle_bwd: bad args");
  SA_CHECK_ARG(!dmask_token || (ws && ((uintptr_t)ws & 15) == 0), "sa_mae_unshuffle_bwd: the mask-token gradient needs a 16-byte aligned workspace (sa_mae_unshuffle_bwd_workspace_bytes)");
  const int64_t rows = (int64_t)B * (L + 1);
  int gy = (int)((rows + 3) / 4);
  if (gy > UNSH_GROUPS) gy = UNSH_GROUPS;
  SA_CHECK_ARG(d <= UNSH_MAXD, "sa_mae_unshuffle_bwd: d = %d exceeds %d", d, UNSH_MAXD);
  hipLaunchKernelGGL(mae_unshuffle_bwd_kernel, dim3((d / 4 + 63) / 64, gy), dim3(256), 0, (hipStream_t)stream, dout, keep, ids_restore, B, L, d, dx,
                     dmask_token, ws);
  if (dmask_token) hipLaunchKernelGGL(mae_unshuffle_sum_kernel, dim3((d + 255) / 256), dim3(256), 0, (hipStream_t)stream, gy, d, dmask_token, ws);
  SA_LAUNCH_CHECK("sa_mae_unshuffle_bwd");
  return 0;
}

extern "C" int sa_mae_recon_loss_fwd(const float* pred, int64_t pred_seq_stride, int32_t pred_row0, const float* img, const float* mask, int32_t B, int32_t F, int32_t T, int32_t ph, int32_t pw,
                                     int32_t norm_pix, float* acc2, float* loss, float* ws, void* stream) {
  SA_CHECK_ARG(ws, "sa_mae_recon_loss_fwd: needs a workspace (sa_mae_recon_loss_workspace_bytes)");
  SA_CHECK_ARG(pred && img && mask && acc2 && loss && B > 0 && ph > 0 && pw > 0 && F >= ph && T >= pw && F % ph == 0 && T % pw == 0 &&
                   pred_row0 >= 0 && pred_seq_stride >= (int64_t)(pred_row0 + (F / ph) * (T / pw)) * ph * pw,
               "sa_mae_recon_loss_fwd: bad args (F, T must be multiples of the patch size; pred rows must fit the clip stride)");
  SA_CHECK_ARG(ph * pw <= 64 * MAE_MAXP && (!norm_pix || ph * pw > 1), "sa_mae_recon_loss_fwd: patches of at most %d pixels", 64 * MAE_MAXP);
  const int64_t rows = (int64_t)B * (F / ph) * (T / pw);
  int grid = (int)((rows + 3) / 4);
  if (grid > MAE_LOSS_BLOCKS) grid = MAE_LOSS_BLOCKS;
  hipLaunchKernelGGL(mae_loss_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pred, pred_seq_stride, pred_row0, img, mask, B, F, T, ph, pw,
                     norm_pix, ws);
  hipLaunchKernelGGL(mae_loss_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, grid, acc2, loss, ws);
  SA_LAUNCH_CHECK("sa_mae_recon_loss_fwd");
  return 0;
}

extern "C" int sa_mae_recon_loss_finalize(const float* acc2, float* loss, void* stream) {
  SA_CHECK_ARG(acc2 && loss, "sa_mae_recon_loss_finalize: bad args");
  hipLaunchKernelGGL(mae_loss_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, acc2, loss);
  SA_LAUNCH_CHECK("sa_mae_recon_loss_finalize");
  return 0;
}

extern "C" int sa_mae_recon_loss_bwd(const float* pred, int64_t pred_seq_stride, int32_t pred_row0, const float* img, const float* mask, const float* acc2, const float* gscale, int32_t B,
                                     int32_t F, int32_t T, int32_t ph, int32_t pw, int32_t norm_pix, float* dpred, void* stream) {
  SA_CHECK_ARG(pred && img && mask && acc2 && gscale && dpred && B > 0 && ph > 0 && pw > 0 && F % ph == 0 && T % pw == 0 && pred_row0 >= 0 &&
                   pred_seq_stride >= (int64_t)(pred_row0 + (F / ph) * (T / pw)) * ph * pw && ph * pw <= 64 * MAE_MAXP,
               "sa_mae_recon_loss_bwd: bad args");
  const int64_t rows = (int64_t)B * (pred_row0 + (F / ph) * (T / pw));
  int grid = (int)((rows + 3) / 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(mae_loss_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, pred, pred_seq_stride, pred_row0, img, mask, acc2,
                     gscale, B, F, T, ph, pw, norm_pix, dpred);
  SA_LAUNCH_CHECK("sa_mae_recon_loss_bwd");
  return 0;
}

extern "C" int sa_token_group_sum(const float* y, int32_t S, int32_t N, int32_t d, int32_t row0, int32_t G, int32_t group_stride, int32_t count,
                                  float scale, int32_t accumulate, float* out, void* stream) {
  SA_CHECK_ARG(y && out && S > 0 && N > 0 && d > 0 && d % 4 == 0 && G > 0 && row0 >= 0 && count >= 0 && group_stride >= 0 &&
                   (count == 0 || row0 + (int64_t)(G - 1) * group_stride + count <= N),
               "sa_token_group_sum: bad args (groups must lie inside the N token rows, d %% 4 == 0)");
  hipLaunchKernelGGL(token_group_sum_kernel, dim3(grid_for((int64_t)S * G * (d / 4))), dim3(256), 0, (hipStream_t)stream, y, S, N, d, row0, G,
                     group_stride, count, scale, accumulate, out);
  SA_LAUNCH_CHECK("sa_token_group_sum");
  return 0;
}
