// The pieces of the ResNet-18 encoders (models/resnet.py: `resnet18`, `resnet18_ReGP_NRF`; BASELINE config 1) that the conv stems in
// conv.hip do not already provide.  Same conventions: channel-last feature maps [B][H][W][C] = row-major [M][C] matrices, 16-byte
// accesses along the channel axis, HBM-bound streaming kernels, no atomics (backward passes are gathers).
//
//   MaxPool2d(kernel 3, stride 2, padding 1)                     models/resnet.py:191, 258   (windows overlap: the backward sums)
//   stride-(sh, sw) row subsampling = the data movement of a strided 1x1 convolution (downsample path, :223-226); backward adds
//   out = relu(bn(h) + identity)  (BasicBlock.forward :75-78) as `add + relu` after sa_bn_apply, and the fp32 ReLU mask of its backward
#include "common.h"
#include "../../include/ssl_audio_hip.h"

namespace {

inline int grid_for(int64_t n, int per_block = 256, int cap = 8192) {
  const int64_t want = (n + per_block - 1) / per_block;
  return (int)(want < cap ? (want < 1 ? 1 : want) : cap);
}

// y[b][oy][ox][c] = max over the 3x3 window at (2*oy - 1, 2*ox - 1), padding taps skipped (= -inf); idx = ky*3 + kx of the FIRST
// maximum in scan order (strictly-greater update, as ATen's max_pool2d does)
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const bf16_t* __restrict__ x, int B, int H, int W, int C, int Ho, int Wo, bf16_t* __restrict__ y,
                                                             float* __restrict__ y32, uint8_t* __restrict__ idx) {
  const int c8 = C >> 3;
  const int64_t total = (int64_t)B * Ho * Wo * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const int64_t o = i / c8;
    const int ox = (int)(o % Wo), oy = (int)((o / Wo) % Ho), b = (int)(o / ((int64_t)Wo * Ho));
    float best[8];
    uint8_t which[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; which[k] = 0; }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (((int64_t)b * H + iy) * W + ix) * C + cc * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float f = bf2f(v[k]);
          if (f > best[k]) { best[k] = f; which[k] = (uint8_t)(ky * 3 + kx); }
        }
      }
    }
    bf16x8 out;
#pragma unroll
    for (int k = 0; k < 8; ++k) out[k] = f2bf(best[k]);
    *reinterpret_cast<bf16x8*>(y + o * C + cc * 8) = out;
    if (y32) {
      *reinterpret_cast<float4*>(y32 + o * C + cc * 8) = make_float4(best[0], best[1], best[2], best[3]);
      *reinterpret_cast<float4*>(y32 + o * C + cc * 8 + 4) = make_float4(best[4], best[5], best[6], best[7]);
    }
    uint64_t packed = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) packed |= (uint64_t)which[k] << (8 * k);
    *reinterpret_cast<uint64_t*>(idx + o * C + cc * 8) = packed;
  }
}

// dx[b][iy][ix][c] = sum over the (<= 4) windows that contain the pixel of dy where the window's recorded maximum is this pixel
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, int B, int H, int W, int C, int Ho,
                                                             int Wo, float* __restrict__ dx) {
  const int c4 = C >> 2;
  const int64_t total = (int64_t)B * H * W * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4);
    const int64_t pix = i / c4;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((int64_t)W * H));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // window oy covers rows 2*oy - 1 .. 2*oy + 1  ->  oy in [ceil((iy - 1) / 2), floor((iy + 1) / 2)]
    for (int oy = iy >> 1; oy <= (iy + 1) >> 1; ++oy) {
      if (oy >= Ho) continue;
      const int ky = iy - (2 * oy - 1);
      for (int ox = ix >> 1; ox <= (ix + 1) >> 1; ++ox) {
        if (ox >= Wo) continue;
        const int kx = ix - (2 * ox - 1);
        const int64_t o = ((int64_t)b * Ho + oy) * Wo + ox;
        const uint32_t w4 = *reinterpret_cast<const uint32_t*>(idx + o * C + cc * 4);
        const float4 g = *reinterpret_cast<const float4*>(dy + o * C + cc * 4);
        const uint32_t me = (uint32_t)(ky * 3 + kx);
        if ((w4 & 0xffu) == me) acc[0] += g.x;
        if (((w4 >> 8) & 0xffu) == me) acc[1] += g.y;
        if (((w4 >> 16) & 0xffu) == me) acc[2] += g.z;
        if ((w4 >> 24) == me) acc[3] += g.w;
      }
    }
    *reinterpret_cast<float4*>(dx + pix * C + cc * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

// y[b][oy][ox][:] = x[b][oy*sh][ox*sw][:]
__global__ __launch_bounds__(256) void subsample_fwd_kernel(const bf16_t* __restrict__ x, int B, int H, int W, int C, int sh, int sw, int Ho, int Wo,
                                                            bf16_t* __restrict__ y) {
  const int c8 = C >> 3;
  const int64_t total = (int64_t)B * Ho * Wo * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const int64_t o = i / c8;
    const int ox = (int)(o % Wo), oy = (int)((o / Wo) % Ho), b = (int)(o / ((int64_t)Wo * Ho));
    *reinterpret_cast<bf16x8*>(y + o * C + cc * 8) = *reinterpret_cast<const bf16x8*>(x + (((int64_t)b * H + oy * sh) * W + ox * sw) * C + cc * 8);
  }
}

// dx[b][oy*sh][ox*sw][:] += dy[b][oy][ox][:]   (dy bf16: the dgrad GEMM's output; one thread per sampled pixel: no two threads share an address)
__global__ __launch_bounds__(256) void subsample_bwd_add_kernel(const bf16_t* __restrict__ dy, int64_t lddy, int B, int H, int W, int C, int sh, int sw, int Ho,
                                                                int Wo, float* __restrict__ dx) {
  const int c4 = C >> 2;
  const int64_t total = (int64_t)B * Ho * Wo * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4);
    const int64_t o = i / c4;
    const int ox = (int)(o % Wo), oy = (int)((o / Wo) % Ho), b = (int)(o / ((int64_t)Wo * Ho));
    const bf16x4 g = *reinterpret_cast<const bf16x4*>(dy + o * lddy + cc * 4);
    float4* p = reinterpret_cast<float4*>(dx + (((int64_t)b * H + oy * sh) * W + ox * sw) * C + cc * 4);
    float4 t = *p;
    t.x += bf2f(g[0]); t.y += bf2f(g[1]); t.z += bf2f(g[2]); t.w += bf2f(g[3]);
    *p = t;
  }
}

// y = max(z + identity, 0) -> fp32 (next block's identity) and bf16 (next convolution's operand)
__global__ __launch_bounds__(256) void add_relu_fwd_kernel(const float* __restrict__ z, const float* __restrict__ idn, int64_t n4, float* __restrict__ y32,
                                                           bf16_t* __restrict__ y16) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(z)[i], b = reinterpret_cast<const float4*>(idn)[i];
    const float4 r = make_float4(relu_f(a.x + b.x), relu_f(a.y + b.y), relu_f(a.z + b.z), relu_f(a.w + b.w));
    reinterpret_cast<float4*>(y32)[i] = r;
    if (y16) reinterpret_cast<bf16x4*>(y16)[i] = bf16x4{f2bf(r.x), f2bf(r.y), f2bf(r.z), f2bf(r.w)};
  }
}

// ds = (y > 0) ? dy (+ dy2) : 0     (y = the block's post-ReLU output; dy2: a second consumer's gradient, or null)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ dy2, const float* __restrict__ y, int64_t n4,
                                                       float* __restrict__ ds) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 g = reinterpret_cast<const float4*>(dy)[i];
    if (dy2) {
      const float4 h = reinterpret_cast<const float4*>(dy2)[i];
      g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w;
    }
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    reinterpret_cast<float4*>(ds)[i] = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
  }
}

// AdaptiveAvgPool2d((1, 1)) over a channel-last map: out[b][c] = mean over the L = H*W rows of x[b]; one thread per (b, 4 channels)
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, int B, int L, int C, float* __restrict__ out) {
  const int c4 = C >> 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * c4) return;
  const int b = i / c4, cc = i - b * c4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int l = 0; l < L; ++l) {
    const float4 v = *reinterpret_cast<const float4*>(x + ((int64_t)b * L + l) * C + cc * 4);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const float inv = 1.f / (float)L;
  *reinterpret_cast<float4*>(out + (int64_t)b * C + cc * 4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dout, int B, int L, int C, float* __restrict__ dx) {
  const int c4 = C >> 2;
  const int64_t total = (int64_t)B * L * c4;
  const float inv = 1.f / (float)L;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4);
    const int b = (int)(i / ((int64_t)L * c4));
    const float4 g = *reinterpret_cast<const float4*>(dout + (int64_t)b * C + cc * 4);
    reinterpret_cast<float4*>(dx)[i] = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
  }
}

}  // namespace

extern "C" int sa_maxpool3s2_fwd(const void* x_bf16, int32_t B, int32_t H, int32_t W, int32_t C, void* y_bf16, float* y_f32, uint8_t* idx, void* stream) {
  SA_CHECK_ARG(x_bf16 && y_bf16 && idx && B > 0 && H >= 1 && W >= 1 && C >= 8 && C % 8 == 0, "sa_maxpool3s2_fwd: bad args");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(grid_for((int64_t)B * Ho * Wo * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, B, H, W, C,
                     Ho, Wo, (bf16_t*)y_bf16, y_f32, idx);
  SA_LAUNCH_CHECK("sa_maxpool3s2_fwd");
  return 0;
}

extern "C" int sa_maxpool3s2_bwd(const float* dy, const uint8_t* idx, int32_t B, int32_t H, int32_t W, int32_t C, float* dx, void* stream) {
  SA_CHECK_ARG(dy && idx && dx && B > 0 && H >= 1 && W >= 1 && C >= 8 && C % 8 == 0, "sa_maxpool3s2_bwd: bad args");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(grid_for((int64_t)B * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, idx, B, H, W, C, Ho, Wo, dx);
  SA_LAUNCH_CHECK("sa_maxpool3s2_bwd");
  return 0;
}

extern "C" int sa_subsample_fwd(const void* x_bf16, int32_t B, int32_t H, int32_t W, int32_t C, int32_t sh, int32_t sw, void* y_bf16, void* stream) {
  SA_CHECK_ARG(x_bf16 && y_bf16 && B > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0 && sh >= 1 && sw >= 1, "sa_subsample_fwd: bad args");
  const int Ho = (H - 1) / sh + 1, Wo = (W - 1) / sw + 1;
  hipLaunchKernelGGL(subsample_fwd_kernel, dim3(grid_for((int64_t)B * Ho * Wo * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, B, H, W, C,
                     sh, sw, Ho, Wo, (bf16_t*)y_bf16);
  SA_LAUNCH_CHECK("sa_subsample_fwd");
  return 0;
}

extern "C" int sa_subsample_bwd_add(const void* dy_bf16, int64_t lddy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t sh, int32_t sw, float* dx,
                                    void* stream) {
  SA_CHECK_ARG(dy_bf16 && dx && B > 0 && H > 0 && W > 0 && C >= 4 && C % 4 == 0 && lddy >= C && lddy % 4 == 0 && sh >= 1 && sw >= 1,
               "sa_subsample_bwd_add: bad args");
  const int Ho = (H - 1) / sh + 1, Wo = (W - 1) / sw + 1;
  hipLaunchKernelGGL(subsample_bwd_add_kernel, dim3(grid_for((int64_t)B * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy_bf16, lddy, B,
                     H, W, C, sh, sw, Ho, Wo, dx);
  SA_LAUNCH_CHECK("sa_subsample_bwd_add");
  return 0;
}

extern "C" int sa_add_relu_fwd(const float* z, const float* identity, int64_t n, float* y_f32, void* y_bf16, void* stream) {
  SA_CHECK_ARG(z && identity && y_f32 && n > 0 && n % 4 == 0, "sa_add_relu_fwd: bad args (n must be a multiple of 4)");
  SA_CHECK_ARG((((uintptr_t)z | (uintptr_t)identity | (uintptr_t)y_f32) & 15) == 0 && ((uintptr_t)y_bf16 & 7) == 0, "sa_add_relu_fwd: misaligned pointers");
  hipLaunchKernelGGL(add_relu_fwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, z, identity, n / 4, y_f32, (bf16_t*)y_bf16);
  SA_LAUNCH_CHECK("sa_add_relu_fwd");
  return 0;
}

extern "C" int sa_relu_bwd(const float* dy, const float* dy2, const float* y, int64_t n, float* ds, void* stream) {
  SA_CHECK_ARG(dy && y && ds && n > 0 && n % 4 == 0, "sa_relu_bwd: bad args (n must be a multiple of 4)");
  SA_CHECK_ARG((((uintptr_t)dy | (uintptr_t)dy2 | (uintptr_t)y | (uintptr_t)ds) & 15) == 0, "sa_relu_bwd: misaligned pointers");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, dy, dy2, y, n / 4, ds);
  SA_LAUNCH_CHECK("sa_relu_bwd");
  return 0;
}

extern "C" int sa_avgpool_fwd(const float* x, int32_t B, int32_t L, int32_t C, float* out, void* stream) {
  SA_CHECK_ARG(x && out && B > 0 && L > 0 && C >= 4 && C % 4 == 0, "sa_avgpool_fwd: bad args");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for((int64_t)B * (C / 4), 256, 1 << 20)), dim3(256), 0, (hipStream_t)stream, x, B, L, C, out);
  SA_LAUNCH_CHECK("sa_avgpool_fwd");
  return 0;
}

extern "C" int sa_avgpool_bwd(const float* dout, int32_t B, int32_t L, int32_t C, float* dx, void* stream) {
  SA_CHECK_ARG(dout && dx && B > 0 && L > 0 && C >= 4 && C % 4 == 0, "sa_avgpool_bwd: bad args");
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((int64_t)B * L * (C / 4))), dim3(256), 0, (hipStream_t)stream, dout, B, L, C, dx);
  SA_LAUNCH_CHECK("sa_avgpool_bwd");
  return 0;
}
