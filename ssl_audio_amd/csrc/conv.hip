// Convolutional pieces of the ConvStem (vitc_*) and AudioNTT encoders for gfx950: everything channel-last (NHWC), so that a
// feature map [B][H][W][C] IS the row-major [M = B*H*W][C] matrix the bf16 MFMA GEMM and the BatchNorm kernels work on.
//
//   3x3 convolution, pad 1, stride (sh, sw)    models/mae.py:82-88 (ConvStem), model.py:138-152 (AudioNTT)
//     C_in = 1 (first layer): direct kernels -- forward (x fp32 [B][H][W] -> y fp32 [M][C_out]) and weight gradient
//     C_in >= 8            : im2col (bf16 [M][9*C_in], k = (ky*3 + kx)*C_in + ci, zero-padded to a multiple of 64) feeding
//                            sa_gemm_bf16; backward = the GEMM's dgrad + col2im (a gather: no atomics) and its wgrad
//   BatchNorm2d statistics over millions of rows: chunked two-level Chan combination (the projector's one-thread-per-column
//     kernels in bn_loss.hip are for a few hundred rows); sums for the backward likewise
//   MaxPool2d(2, 2) forward (with argmax) / backward                                   model.py:141,149
// All of them are HBM-bound streaming kernels: 16-byte accesses along the channel axis.
#include "common.h"
#include "../../include/ssl_audio_hip.h"

namespace {

inline int grid_for(int64_t n, int per_block = 256, int cap = 8192) {
  const int64_t want = (n + per_block - 1) / per_block;
  return (int)(want < cap ? (want < 1 ? 1 : want) : cap);
}

// ---------------------------------------------------------------------------------------------------- C_in = 1
// y[m][c] = bias[c] + sum_{ky,kx} w[c][ky*3+kx] * x[b][oy*sh-1+ky][ox*sw-1+kx]   (zero outside the image)
__global__ __launch_bounds__(256) void conv3x3_c1_fwd_kernel(const float* __restrict__ x, int B, int H, int W, int sh, int sw, int Ho, int Wo,
                                                             const float* __restrict__ w, const float* __restrict__ bias, int Cout,
                                                             float* __restrict__ y) {
  const int cg_n = Cout >> 2;
  const int64_t total = (int64_t)B * Ho * Wo * cg_n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cg_n);
    const int64_t m = i / cg_n;
    const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((int64_t)Wo * Ho));
    float acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = bias ? bias[cg * 4 + r] : 0.f;
    const float* xb = x + (int64_t)b * H * W;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * sh - 1 + ky;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * sw - 1 + kx;
        const float v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? xb[(int64_t)iy * W + ix] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fmaf(v, w[(cg * 4 + r) * 9 + ky * 3 + kx], acc[r]);
      }
    }
    *reinterpret_cast<float4*>(y + m * Cout + cg * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

// dw[c][tap] += sum_m dy[m][c] * x(m, tap);  dbias[c] += sum_m dy[m][c].  Block = PL pixel lanes x (Cout/4) channel groups.
__global__ __launch_bounds__(256) void conv3x3_c1_wgrad_kernel(const float* __restrict__ x, int B, int H, int W, int sh, int sw, int Ho, int Wo,
                                                               const bf16_t* __restrict__ dy, int Cout, float* __restrict__ dw,
                                                               float* __restrict__ dbias) {
  extern __shared__ float red[];                       // [PL][cg_n][40]
  const int cg_n = Cout >> 2, PL = blockDim.x / cg_n;
  const int cg = threadIdx.x % cg_n, pl = threadIdx.x / cg_n;
  float acc[4][10];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[r][t] = 0.f;
  const int64_t M = (int64_t)B * Ho * Wo;
  if (pl < PL) {
    for (int64_t m = (int64_t)blockIdx.x * PL + pl; m < M; m += (int64_t)gridDim.x * PL) {
      const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((int64_t)Wo * Ho));
      const bf16x4 d4 = *reinterpret_cast<const bf16x4*>(dy + m * Cout + cg * 4);
      const float d[4] = {bf2f(d4[0]), bf2f(d4[1]), bf2f(d4[2]), bf2f(d4[3])};
      const float* xb = x + (int64_t)b * H * W;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * sh - 1 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = ox * sw - 1 + kx;
          const float v = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? xb[(int64_t)iy * W + ix] : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r][ky * 3 + kx] = fmaf(d[r], v, acc[r][ky * 3 + kx]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r][9] += d[r];
    }
    float* mine = red + ((size_t)pl * cg_n + cg) * 40;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < 10; ++t) mine[r * 10 + t] = acc[r][t];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < cg_n * 40; e += blockDim.x) {       // sum over the pixel lanes, one atomic per (channel, tap) per block
    float s = 0.f;
    for (int q = 0; q < PL; ++q) s += red[(size_t)q * cg_n * 40 + e];
    const int c = (e / 40) * 4 + (e % 40) / 10, t = e % 10;
    if (t < 9) atomicAdd(dw + c * 9 + t, s);
    else if (dbias) atomicAdd(dbias + c, s);
  }
}

// ---------------------------------------------------------------------------------------------------- im2col / col2im (NHWC bf16)
// out[m][(ky*3+kx)*C + c] = x[b][oy*sh-1+ky][ox*sw-1+kx][c] (0 outside), columns 9*C .. Kpad-1 = 0.  One thread = 8 channels.
__global__ __launch_bounds__(256) void im2col3x3_kernel(const bf16_t* __restrict__ x, int B, int H, int W, int C, int sh, int sw, int Ho, int Wo,
                                                        bf16_t* __restrict__ out, int Kpad) {
  const int chunks = Kpad >> 3, c8 = C >> 3;
  const int64_t total = (int64_t)B * Ho * Wo * chunks;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % chunks);
    const int64_t m = i / chunks;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (ch < 9 * c8) {
      const int tap = ch / c8, cc = ch - tap * c8;
      const int ox = (int)(m % Wo), oy = (int)((m / Wo) % Ho), b = (int)(m / ((int64_t)Wo * Ho));
      const int iy = oy * sh - 1 + tap / 3, ix = ox * sw - 1 + tap % 3;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const uint4*>(x + (((int64_t)b * H + iy) * W + ix) * C + cc * 8);
    }
    *reinterpret_cast<uint4*>(out + m * Kpad + ch * 8) = v;
  }
}

// dx[b][iy][ix][c] = sum over the taps (ky,kx) and outputs (oy,ox) with oy*sh-1+ky == iy, ox*sw-1+kx == ix of dP[m(b,oy,ox)][(ky*3+kx)*C + c]
__global__ __launch_bounds__(256) void col2im3x3_kernel(const bf16_t* __restrict__ dP, int B, int H, int W, int C, int sh, int sw, int Ho, int Wo,
                                                        int Kpad, float* __restrict__ dx) {
  const int c8 = C >> 3;
  const int64_t total = (int64_t)B * H * W * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const int64_t pix = i / c8;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ty = iy + 1 - ky;
      if (ty < 0 || ty % sh != 0) continue;
      const int oy = ty / sh;
      if (oy >= Ho) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tx = ix + 1 - kx;
        if (tx < 0 || tx % sw != 0) continue;
        const int ox = tx / sw;
        if (ox >= Wo) continue;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(dP + (((int64_t)b * Ho + oy) * Wo + ox) * Kpad + (ky * 3 + kx) * C + cc * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += bf2f(v[k]);
      }
    }
    float* o = dx + pix * C + cc * 8;
    *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

// ---------------------------------------------------------------------------------------------------- tall-matrix BatchNorm statistics
constexpr int BN_CHUNK = 2048;      // rows per partial

// partial (mean, M2) of one chunk of rows for 64 columns: 4 row lanes x 64 columns per block, two passes over the chunk (L2-resident)
__global__ __launch_bounds__(256) void bn_tall_partial_kernel(const float* __restrict__ x, int64_t ld, int64_t M, int C, float* __restrict__ ws) {
  __shared__ float sh[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * BN_CHUNK, r1 = min(M, r0 + BN_CHUNK);
  const float n = (float)(r1 - r0);
  float s = 0.f;
  if (col < C)
    for (int64_t r = r0 + rl; r < r1; r += 4) s += x[r * ld + col];
  sh[rl][threadIdx.x & 63] = s;
  __syncthreads();
  const float mean = (sh[0][threadIdx.x & 63] + sh[1][threadIdx.x & 63] + sh[2][threadIdx.x & 63] + sh[3][threadIdx.x & 63]) / n;
  __syncthreads();
  float q = 0.f;
  if (col < C)
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      const float d = x[r * ld + col] - mean;
      q += d * d;
    }
  sh[rl][threadIdx.x & 63] = q;
  __syncthreads();
  if (rl == 0 && col < C) {
    float* o = ws + (int64_t)blockIdx.y * 2 * C;
    o[col] = mean;
    o[C + col] = sh[0][threadIdx.x & 63] + sh[1][threadIdx.x & 63] + sh[2][threadIdx.x & 63] + sh[3][threadIdx.x & 63];
  }
}

// merge the chunk partials (Chan et al., unequal counts: only the last chunk is short) -> local mean, M2
__global__ void bn_tall_merge_kernel(const float* __restrict__ ws, int64_t M, int C, int nchunks, float* __restrict__ mean_out,
                                     float* __restrict__ m2_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double na = 0.0, mean = 0.0, m2 = 0.0;
  for (int k = 0; k < nchunks; ++k) {
    const double nb = (double)min((int64_t)BN_CHUNK, M - (int64_t)k * BN_CHUNK);
    const double mb = ws[(int64_t)k * 2 * C + c], qb = ws[(int64_t)k * 2 * C + C + c];
    const double d = mb - mean, nt = na + nb;
    mean += d * nb / nt;
    m2 += qb + d * d * na * nb / nt;
    na = nt;
  }
  mean_out[c] = (float)mean;
  m2_out[c] = (float)m2;
}

// partial backward sums of one chunk: s1 = sum g, s2 = sum g * xhat, g = dy * [relu mask]
template <typename DY>
__global__ __launch_bounds__(256) void bn_tall_bwd_partial_kernel(const DY* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ld, int64_t M,
                                                                  int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                                  float* __restrict__ ws) {
  __shared__ float sh[2][4][64];
  const int lc = threadIdx.x & 63, col = blockIdx.x * 64 + lc, rl = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.y * BN_CHUNK, r1 = min(M, r0 + BN_CHUNK);
  float s1 = 0.f, s2 = 0.f;
  if (col < C) {
    const float mu = mean[col], rs = rstd[col], g = gamma ? gamma[col] : 1.f, bt = beta ? beta[col] : 0.f;
    for (int64_t r = r0 + rl; r < r1; r += 4) {
      const float xh = (x[r * ld + col] - mu) * rs;
      float d = (float)dy[r * lddy + col];
      if (relu && !(xh * g + bt > 0.f)) d = 0.f;
      s1 += d;
      s2 += d * xh;
    }
  }
  sh[0][rl][lc] = s1;
  sh[1][rl][lc] = s2;
  __syncthreads();
  if (rl == 0 && col < C) {
    float* o = ws + (int64_t)blockIdx.y * 2 * C;
    o[col] = sh[0][0][lc] + sh[0][1][lc] + sh[0][2][lc] + sh[0][3][lc];
    o[C + col] = sh[1][0][lc] + sh[1][1][lc] + sh[1][2][lc] + sh[1][3][lc];
  }
}

__global__ void bn_tall_bwd_merge_kernel(const float* __restrict__ ws, int C, int nchunks, float* __restrict__ s1, float* __restrict__ s2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int k = 0; k < nchunks; ++k) {
    a += ws[(int64_t)k * 2 * C + c];
    b += ws[(int64_t)k * 2 * C + C + c];
  }
  s1[c] = (float)a;
  s2[c] = (float)b;
}

// ---------------------------------------------------------------------------------------------------- MaxPool2d(2, 2), NHWC bf16
// y[b][oy][ox][c] = max over the 2x2 window (floor mode: odd trailing rows / columns are dropped); idx = which of the four (first max wins)
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const bf16_t* __restrict__ x, int B, int H, int W, int C, bf16_t* __restrict__ y,
                                                           uint8_t* __restrict__ idx) {
  const int Ho = H >> 1, Wo = W >> 1, c8 = C >> 3;
  const int64_t total = (int64_t)B * Ho * Wo * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const int64_t o = i / c8;
    const int ox = (int)(o % Wo), oy = (int)((o / Wo) % Ho), b = (int)(o / ((int64_t)Wo * Ho));
    bf16x8 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      v[q] = *reinterpret_cast<const bf16x8*>(x + (((int64_t)b * H + 2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * C + cc * 8);
    bf16x8 best = v[0];
    uint8_t which[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 1; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (bf2f(v[q][k]) > bf2f(best[k])) { best[k] = v[q][k]; which[k] = (uint8_t)q; }
    *reinterpret_cast<bf16x8*>(y + o * C + cc * 8) = best;
    uint64_t packed = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) packed |= (uint64_t)which[k] << (8 * k);
    *reinterpret_cast<uint64_t*>(idx + o * C + cc * 8) = packed;
  }
}

// dx (fp32, every element written): the window's gradient goes to the recorded position, 0 elsewhere; dropped rows / columns get 0
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, int B, int H, int W, int C,
                                                           float* __restrict__ dx) {
  const int Ho = H >> 1, Wo = W >> 1, c4 = C >> 2;
  const int64_t total = (int64_t)B * H * W * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4);
    const int64_t pix = i / c4;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    const int oy = iy >> 1, ox = ix >> 1;
    if (oy < Ho && ox < Wo) {
      const int64_t o = ((int64_t)b * Ho + oy) * Wo + ox;
      const float4 g = *reinterpret_cast<const float4*>(dy + o * C + cc * 4);
      const uint32_t w4 = *reinterpret_cast<const uint32_t*>(idx + o * C + cc * 4);
      const uint32_t me = (uint32_t)((iy & 1) * 2 + (ix & 1));
      out.x = ((w4 & 0xFF) == me) ? g.x : 0.f;
      out.y = (((w4 >> 8) & 0xFF) == me) ? g.y : 0.f;
      out.z = (((w4 >> 16) & 0xFF) == me) ? g.z : 0.f;
      out.w = (((w4 >> 24) & 0xFF) == me) ? g.w : 0.f;
    }
    *reinterpret_cast<float4*>(dx + pix * C + cc * 4) = out;
  }
}

// ---------------------------------------------------------------------------------------------------- AudioNTT glue (model.py:153-191)
// y = relu(x) [* keep * scale]  (nn.ReLU, then nn.Dropout in train mode with the caller's keep mask), x fp32 [M][C] with row stride ldx
__global__ __launch_bounds__(256) void relu_mask_fwd_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, int C, const uint8_t* __restrict__ keep,
                                                            float scale, bf16_t* __restrict__ y16, int64_t ldy16, float* __restrict__ y32, int64_t ldy32) {
  const int c4 = C >> 2;
  const int64_t total = M * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4);
    const int64_t m = i / c4;
    const float4 v = *reinterpret_cast<const float4*>(x + m * ldx + cc * 4);
    float o[4] = {relu_f(v.x), relu_f(v.y), relu_f(v.z), relu_f(v.w)};
    if (keep) {
      const uint32_t k4 = *reinterpret_cast<const uint32_t*>(keep + m * C + cc * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = ((k4 >> (8 * r)) & 0xFF) ? o[r] * scale : 0.f;
    }
    if (y16) *reinterpret_cast<bf16x4*>(y16 + m * ldy16 + cc * 4) = bf16x4{f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
    if (y32) *reinterpret_cast<float4*>(y32 + m * ldy32 + cc * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// dx = dy * [x_pre > 0] [* keep * scale] -> bf16 (the operand of the layer's dgrad / wgrad GEMMs)
__global__ __launch_bounds__(256) void relu_mask_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ xpre, int64_t ldx, int64_t M,
                                                            int C, const uint8_t* __restrict__ keep, float scale, bf16_t* __restrict__ dx16, int64_t lddx) {
  const int c4 = C >> 2;
  const int64_t total = M * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4);
    const int64_t m = i / c4;
    const float4 g = *reinterpret_cast<const float4*>(dy + m * lddy + cc * 4);
    const float4 v = *reinterpret_cast<const float4*>(xpre + m * ldx + cc * 4);
    float o[4] = {v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f};
    if (keep) {
      const uint32_t k4 = *reinterpret_cast<const uint32_t*>(keep + m * C + cc * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = ((k4 >> (8 * r)) & 0xFF) ? o[r] * scale : 0.f;
    }
    *reinterpret_cast<bf16x4*>(dx16 + m * lddx + cc * 4) = bf16x4{f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
  }
}

// x bf16 NHWC [B][H][W][C] -> frames[(b, w)][h * C + c]  (x.permute(0, 3, 2, 1).reshape(B, T, C * D), model.py:161-163): bf16 and / or fp32
__global__ __launch_bounds__(256) void nhwc_to_frames_kernel(const bf16_t* __restrict__ x, int B, int H, int W, int C, bf16_t* __restrict__ f16,
                                                             float* __restrict__ f32, int64_t ld32) {
  const int c8 = C >> 3;
  const int64_t total = (int64_t)B * H * W * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const int64_t r = i / c8;
    const int h = (int)(r % H), w = (int)((r / H) % W), b = (int)(r / ((int64_t)H * W));      // output-major: consecutive threads write one frame row
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (((int64_t)b * H + h) * W + w) * C + cc * 8);
    const int64_t row = (int64_t)b * W + w;
    if (f16) *reinterpret_cast<bf16x8*>(f16 + row * ((int64_t)H * C) + h * C + cc * 8) = v;
    if (f32) {
      float* o = f32 + row * ld32 + h * C + cc * 8;
      *reinterpret_cast<float4*>(o) = make_float4(bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3]));
      *reinterpret_cast<float4*>(o + 4) = make_float4(bf2f(v[4]), bf2f(v[5]), bf2f(v[6]), bf2f(v[7]));
    }
  }
}

// dx[b][h][w][c] = da[(b, w)][h * C + c] + db[(b, w)][h * C + c]   (either source may be null)
__global__ __launch_bounds__(256) void frames_to_nhwc_kernel(const float* __restrict__ da, int64_t lda, const float* __restrict__ db, int64_t ldb, int B,
                                                             int H, int W, int C, float* __restrict__ dx) {
  const int c4 = C >> 2;
  const int64_t total = (int64_t)B * H * W * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4);
    const int64_t pix = i / c4;
    const int w = (int)(pix % W), h = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    const int64_t row = (int64_t)b * W + w;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (da) v = *reinterpret_cast<const float4*>(da + row * lda + h * C + cc * 4);
    if (db) {
      const float4 u = *reinterpret_cast<const float4*>(db + row * ldb + h * C + cc * 4);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    *reinterpret_cast<float4*>(dx + pix * C + cc * 4) = v;
  }
}

// mean_max_pooling (model.py:185-190): out[b][f] = max_t x[b][t][f] + mean_t x[b][t][f]; arg = first t of the maximum
__global__ __launch_bounds__(256) void meanmax_fwd_kernel(const float* __restrict__ x, int B, int T, int D, float* __restrict__ out, int32_t* __restrict__ arg) {
  const int64_t total = (int64_t)B * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % D), b = (int)(i / D);
    const float* p = x + (int64_t)b * T * D + f;
    float mx = p[0], sum = 0.f;
    int am = 0;
    for (int t = 0; t < T; ++t) {
      const float v = p[(int64_t)t * D];
      sum += v;
      if (v > mx) { mx = v; am = t; }
    }
    out[i] = mx + sum / (float)T;
    arg[i] = am;
  }
}

__global__ __launch_bounds__(256) void meanmax_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ arg, int B, int T, int D,
                                                          float* __restrict__ dx) {
  const int64_t total = (int64_t)B * T * D;
  const float invT = 1.f / (float)T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % D), t = (int)((i / D) % T), b = (int)(i / ((int64_t)D * T));
    const float g = dout[(int64_t)b * D + f];
    dx[i] = g * (invT + (arg[(int64_t)b * D + f] == t ? 1.f : 0.f));
  }
}

}  // namespace

static inline int conv_out(int n, int s) { return (n + 2 - 3) / s + 1; }

extern "C" int sa_conv3x3_c1_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t sh, int32_t sw, const float* w, const float* bias,
                                 int32_t Cout, float* y, void* stream) {
  SA_CHECK_ARG(x && w && y && B > 0 && H > 0 && W > 0 && sh > 0 && sw > 0 && Cout > 0 && Cout % 4 == 0 && ((uintptr_t)y & 15) == 0,
               "sa_conv3x3_c1_fwd: bad args (C_out must be a multiple of 4, y 16-byte aligned)");
  const int Ho = conv_out(H, sh), Wo = conv_out(W, sw);
  hipLaunchKernelGGL(conv3x3_c1_fwd_kernel, dim3(grid_for((int64_t)B * Ho * Wo * (Cout / 4))), dim3(256), 0, (hipStream_t)stream, x, B, H, W, sh, sw, Ho,
                     Wo, w, bias, Cout, y);
  SA_LAUNCH_CHECK("sa_conv3x3_c1_fwd");
  return 0;
}

extern "C" int sa_conv3x3_c1_wgrad(const float* x, int32_t B, int32_t H, int32_t W, int32_t sh, int32_t sw, const void* dy_bf16, int32_t Cout,
                                   float* dw, float* dbias, void* stream) {
  SA_CHECK_ARG(x && dy_bf16 && dw && B > 0 && H > 0 && W > 0 && sh > 0 && sw > 0 && Cout >= 4 && Cout % 4 == 0 && Cout <= 1024,
               "sa_conv3x3_c1_wgrad: bad args");
  const int Ho = conv_out(H, sh), Wo = conv_out(W, sw);
  const int cg_n = Cout / 4, PL = 256 / cg_n;
  SA_CHECK_ARG(PL >= 1, "sa_conv3x3_c1_wgrad: C_out too wide");
  const size_t lds = (size_t)PL * cg_n * 40 * sizeof(float);
  const int64_t M = (int64_t)B * Ho * Wo;
  int grid = (int)((M + (int64_t)PL * 64 - 1) / ((int64_t)PL * 64));     // ~64 pixels per pixel lane
  grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
  hipLaunchKernelGGL(conv3x3_c1_wgrad_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, x, B, H, W, sh, sw, Ho, Wo, (const bf16_t*)dy_bf16, Cout,
                     dw, dbias);
  SA_LAUNCH_CHECK("sa_conv3x3_c1_wgrad");
  return 0;
}

extern "C" int sa_im2col3x3_bf16(const void* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t sh, int32_t sw, void* out, int32_t Kpad,
                                 void* stream) {
  SA_CHECK_ARG(x && out && B > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0 && sh > 0 && sw > 0 && Kpad >= 9 * C && Kpad % 8 == 0 &&
                   (((uintptr_t)x | (uintptr_t)out) & 15) == 0,
               "sa_im2col3x3_bf16: bad args (C %% 8 == 0, Kpad >= 9 C and %% 8 == 0, 16-byte aligned buffers)");
  const int Ho = conv_out(H, sh), Wo = conv_out(W, sw);
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for((int64_t)B * Ho * Wo * (Kpad / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, B, H, W, C,
                     sh, sw, Ho, Wo, (bf16_t*)out, Kpad);
  SA_LAUNCH_CHECK("sa_im2col3x3_bf16");
  return 0;
}

extern "C" int sa_col2im3x3_f32(const void* dP_bf16, int32_t B, int32_t H, int32_t W, int32_t C, int32_t sh, int32_t sw, int32_t Kpad, float* dx,
                                void* stream) {
  SA_CHECK_ARG(dP_bf16 && dx && B > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0 && sh > 0 && sw > 0 && Kpad >= 9 * C && Kpad % 8 == 0 &&
                   (((uintptr_t)dP_bf16 | (uintptr_t)dx) & 15) == 0,
               "sa_col2im3x3_f32: bad args");
  const int Ho = conv_out(H, sh), Wo = conv_out(W, sw);
  hipLaunchKernelGGL(col2im3x3_kernel, dim3(grid_for((int64_t)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dP_bf16, B, H, W, C,
                     sh, sw, Ho, Wo, Kpad, dx);
  SA_LAUNCH_CHECK("sa_col2im3x3_f32");
  return 0;
}

extern "C" int64_t sa_bn_tall_workspace_bytes(int64_t M, int32_t C) { return ((M + BN_CHUNK - 1) / BN_CHUNK) * 2 * (int64_t)C * (int64_t)sizeof(float); }

extern "C" int sa_bn_colstats_tall(const float* x, int64_t ld, int64_t M, int32_t C, float* ws, float* mean, float* m2, void* stream) {
  SA_CHECK_ARG(x && ws && mean && m2 && M > 0 && C > 0, "sa_bn_colstats_tall: bad args");
  const int nchunks = (int)((M + BN_CHUNK - 1) / BN_CHUNK);
  SA_CHECK_ARG(nchunks <= 65535, "sa_bn_colstats_tall: more than 65535 row chunks");
  hipLaunchKernelGGL(bn_tall_partial_kernel, dim3((C + 63) / 64, nchunks), dim3(256), 0, (hipStream_t)stream, x, ld, M, C, ws);
  hipLaunchKernelGGL(bn_tall_merge_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, ws, M, C, nchunks, mean, m2);
  SA_LAUNCH_CHECK("sa_bn_colstats_tall");
  return 0;
}

extern "C" int sa_bn_bwd_stats_tall(const void* dy, int32_t dy_is_bf16, int64_t lddy, const float* x, int64_t ld, int64_t M, int32_t C,
                                    const float* mean, const float* rstd, const float* gamma, const float* beta, int32_t relu, float* ws, float* s1,
                                    float* s2, void* stream) {
  SA_CHECK_ARG(dy && x && mean && rstd && ws && s1 && s2 && M > 0 && C > 0, "sa_bn_bwd_stats_tall: bad args");
  const int nchunks = (int)((M + BN_CHUNK - 1) / BN_CHUNK);
  SA_CHECK_ARG(nchunks <= 65535, "sa_bn_bwd_stats_tall: more than 65535 row chunks");
  if (dy_is_bf16)
    hipLaunchKernelGGL((bn_tall_bwd_partial_kernel<bf16_t>), dim3((C + 63) / 64, nchunks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, lddy, x,
                       ld, M, C, mean, rstd, gamma, beta, relu, ws);
  else
    hipLaunchKernelGGL((bn_tall_bwd_partial_kernel<float>), dim3((C + 63) / 64, nchunks), dim3(256), 0, (hipStream_t)stream, (const float*)dy, lddy, x, ld,
                       M, C, mean, rstd, gamma, beta, relu, ws);
  hipLaunchKernelGGL(bn_tall_bwd_merge_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, ws, C, nchunks, s1, s2);
  SA_LAUNCH_CHECK("sa_bn_bwd_stats_tall");
  return 0;
}

extern "C" int sa_maxpool2_fwd(const void* x_bf16, int32_t B, int32_t H, int32_t W, int32_t C, void* y_bf16, uint8_t* idx, void* stream) {
  SA_CHECK_ARG(x_bf16 && y_bf16 && idx && B > 0 && H >= 2 && W >= 2 && C >= 8 && C % 8 == 0, "sa_maxpool2_fwd: bad args");
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for((int64_t)B * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16,
                     B, H, W, C, (bf16_t*)y_bf16, idx);
  SA_LAUNCH_CHECK("sa_maxpool2_fwd");
  return 0;
}

extern "C" int sa_maxpool2_bwd(const float* dy, const uint8_t* idx, int32_t B, int32_t H, int32_t W, int32_t C, float* dx, void* stream) {
  SA_CHECK_ARG(dy && idx && dx && B > 0 && H >= 2 && W >= 2 && C >= 8 && C % 8 == 0, "sa_maxpool2_bwd: bad args");
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, idx, B, H, W, C, dx);
  SA_LAUNCH_CHECK("sa_maxpool2_bwd");
  return 0;
}

extern "C" int sa_relu_mask_fwd(const float* x, int64_t ldx, int64_t M, int32_t C, const uint8_t* keep, float scale, void* y_bf16, int64_t ldy16,
                                float* y_f32, int64_t ldy32, void* stream) {
  SA_CHECK_ARG(x && (y_bf16 || y_f32) && M > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && (!y_bf16 || ldy16 % 4 == 0) && (!y_f32 || ldy32 % 4 == 0),
               "sa_relu_mask_fwd: bad args (C and the leading dimensions must be multiples of 4)");
  hipLaunchKernelGGL(relu_mask_fwd_kernel, dim3(grid_for(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ldx, M, C, keep, scale, (bf16_t*)y_bf16, ldy16,
                     y_f32, ldy32);
  SA_LAUNCH_CHECK("sa_relu_mask_fwd");
  return 0;
}

extern "C" int sa_relu_mask_bwd(const float* dy, int64_t lddy, const float* x_pre, int64_t ldx, int64_t M, int32_t C, const uint8_t* keep, float scale,
                                void* dx_bf16, int64_t lddx, void* stream) {
  SA_CHECK_ARG(dy && x_pre && dx_bf16 && M > 0 && C > 0 && C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0, "sa_relu_mask_bwd: bad args");
  hipLaunchKernelGGL(relu_mask_bwd_kernel, dim3(grid_for(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, lddy, x_pre, ldx, M, C, keep, scale,
                     (bf16_t*)dx_bf16, lddx);
  SA_LAUNCH_CHECK("sa_relu_mask_bwd");
  return 0;
}

extern "C" int sa_nhwc_to_frames(const void* x_bf16, int32_t B, int32_t H, int32_t W, int32_t C, void* frames_bf16, float* frames_f32, int64_t ld32,
                                 void* stream) {
  SA_CHECK_ARG(x_bf16 && (frames_bf16 || frames_f32) && B > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0 && (!frames_f32 || ld32 % 4 == 0),
               "sa_nhwc_to_frames: bad args");
  hipLaunchKernelGGL(nhwc_to_frames_kernel, dim3(grid_for((int64_t)B * H * W * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, B, H, W,
                     C, (bf16_t*)frames_bf16, frames_f32, ld32);
  SA_LAUNCH_CHECK("sa_nhwc_to_frames");
  return 0;
}

extern "C" int sa_frames_to_nhwc(const float* da, int64_t lda, const float* db, int64_t ldb, int32_t B, int32_t H, int32_t W, int32_t C, float* dx,
                                 void* stream) {
  SA_CHECK_ARG((da || db) && dx && B > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0 && (!da || lda % 4 == 0) && (!db || ldb % 4 == 0),
               "sa_frames_to_nhwc: bad args");
  hipLaunchKernelGGL(frames_to_nhwc_kernel, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, da, lda, db, ldb, B, H, W, C,
                     dx);
  SA_LAUNCH_CHECK("sa_frames_to_nhwc");
  return 0;
}

extern "C" int sa_meanmax_time_fwd(const float* x, int32_t B, int32_t T, int32_t D, float* out, int32_t* arg, void* stream) {
  SA_CHECK_ARG(x && out && arg && B > 0 && T > 0 && D > 0, "sa_meanmax_time_fwd: bad args");
  hipLaunchKernelGGL(meanmax_fwd_kernel, dim3(grid_for((int64_t)B * D)), dim3(256), 0, (hipStream_t)stream, x, B, T, D, out, arg);
  SA_LAUNCH_CHECK("sa_meanmax_time_fwd");
  return 0;
}

extern "C" int sa_meanmax_time_bwd(const float* dout, const int32_t* arg, int32_t B, int32_t T, int32_t D, float* dx, void* stream) {
  SA_CHECK_ARG(dout && arg && dx && B > 0 && T > 0 && D > 0, "sa_meanmax_time_bwd: bad args");
  hipLaunchKernelGGL(meanmax_bwd_kernel, dim3(grid_for((int64_t)B * T * D)), dim3(256), 0, (hipStream_t)stream, dout, arg, B, T, D, dx);
  SA_LAUNCH_CHECK("sa_meanmax_time_bwd");
  return 0;
}
