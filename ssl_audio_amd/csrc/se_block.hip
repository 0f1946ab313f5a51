// Squeeze-and-excitation gate of AudioNTT2022 (`--squeeze_excitation`, model.py:141-142,150-151,196-213) on channel-last maps.
//   s[b][c] = mean over the L = H * W pixels of x[b][.][c]            (nn.AdaptiveAvgPool2d(1))
//   e[b][.] = sigmoid(W2 relu(W1 s[b][.]))                           (Linear(C, C/r, bias=False) -> ReLU -> Linear(C/r, C, bias=False) -> Sigmoid)
//   y[b][l][c] = x[b][l][c] * e[b][c]
// HBM-bound: x is read twice (squeeze, scale) and y written once; the excitation is a few hundred FLOPs per sample and runs inside the
// squeeze kernel (one workgroup per sample).  Maps are bf16 (they are GEMM operands on either side), statistics and gates fp32.
#include "common.h"
#include "../../include/ssl_audio_hip.h"

namespace {

constexpr int SE_MAXC = 256, SE_MAXR = 32;

// one workgroup (256 threads) per sample: column means through LDS partials, then the two tiny Linears
__global__ __launch_bounds__(256) void se_squeeze_excite_kernel(const bf16_t* __restrict__ x, int L, int C, const float* __restrict__ w1,
                                                                const float* __restrict__ w2, int R, float* __restrict__ s_out,
                                                                float* __restrict__ h_out, float* __restrict__ e_out) {
  __shared__ float part[256], s[SE_MAXC], h[SE_MAXR];
  const int b = blockIdx.x;
  const bf16_t* xb = x + (int64_t)b * L * C;
  const int c = threadIdx.x % C, lane_rows = 256 / C;          // C divides 256 (checked by the host): 256 / C rows in flight
  float acc = 0.f;
  for (int l = threadIdx.x / C; l < L; l += lane_rows) acc += bf2f(xb[(int64_t)l * C + c]);
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int k = 0; k < lane_rows; ++k) t += part[k * C + threadIdx.x];
    t /= (float)L;
    s[threadIdx.x] = t;
    s_out[(int64_t)b * C + threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x < R) {
    float t = 0.f;
    for (int k = 0; k < C; ++k) t += w1[threadIdx.x * C + k] * s[k];
    h_out[(int64_t)b * R + threadIdx.x] = t;                   // pre-activation (the backward needs its sign)
    h[threadIdx.x] = relu_f(t);
  }
  __syncthreads();
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int k = 0; k < R; ++k) t += w2[threadIdx.x * R + k] * h[k];
    e_out[(int64_t)b * C + threadIdx.x] = 1.f / (1.f + __expf(-t));
  }
}

// y = x * e[b][c]  (8 bf16 per thread, 16-byte accesses; C % 8 == 0)
__global__ __launch_bounds__(256) void se_scale_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ e, int64_t total8, int C, int64_t per_sample,
                                                           bf16_t* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t el = i * 8;
    const int c = (int)(el % C);
    const int b = (int)(el / per_sample);
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + el);
    const float* eb = e + (int64_t)b * C + c;
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = f2bf(bf2f(v[k]) * eb[k]);
    *reinterpret_cast<bf16x8*>(y + el) = o;
  }
}

// de[b][c] = sum_l dy[b][l][c] * x[b][l][c]; one workgroup per (sample, slab of rows), partial sums added atomically (few slabs)
__global__ __launch_bounds__(256) void se_gate_grad_kernel(const float* __restrict__ dy, const bf16_t* __restrict__ x, int L, int C, int rows_per_block,
                                                           float* __restrict__ de) {
  __shared__ float part[256];
  const int b = blockIdx.y;
  const int c = threadIdx.x % C, lane_rows = 256 / C;
  const int l0 = blockIdx.x * rows_per_block, l1 = min(L, l0 + rows_per_block);
  const float* dyb = dy + (int64_t)b * L * C;
  const bf16_t* xb = x + (int64_t)b * L * C;
  float acc = 0.f;
  for (int l = l0 + threadIdx.x / C; l < l1; l += lane_rows) acc += dyb[(int64_t)l * C + c] * bf2f(xb[(int64_t)l * C + c]);
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int k = 0; k < lane_rows; ++k) t += part[k * C + threadIdx.x];
    atomicAdd(de + (int64_t)b * C + threadIdx.x, t);
  }
}

// per sample: dz = de * e (1 - e); dW2 += dz (x) relu(h); dh = W2^T dz masked by h > 0; dW1 += dh (x) s; ds = W1^T dh
__global__ __launch_bounds__(256) void se_excite_bwd_kernel(const float* __restrict__ de, const float* __restrict__ s, const float* __restrict__ hpre,
                                                            const float* __restrict__ e, const float* __restrict__ w1, const float* __restrict__ w2, int C,
                                                            int R, float* __restrict__ dw1, float* __restrict__ dw2, float* __restrict__ ds) {
  __shared__ float dz[SE_MAXC], dh[SE_MAXR], sh[SE_MAXC], hh[SE_MAXR];
  const int b = blockIdx.x;
  if (threadIdx.x < C) {
    const float ev = e[(int64_t)b * C + threadIdx.x];
    dz[threadIdx.x] = de[(int64_t)b * C + threadIdx.x] * ev * (1.f - ev);
    sh[threadIdx.x] = s[(int64_t)b * C + threadIdx.x];
  }
  if (threadIdx.x < R) hh[threadIdx.x] = hpre[(int64_t)b * R + threadIdx.x];
  __syncthreads();
  if (threadIdx.x < R) {
    float t = 0.f;
    for (int k = 0; k < C; ++k) t += w2[k * R + threadIdx.x] * dz[k];
    dh[threadIdx.x] = hh[threadIdx.x] > 0.f ? t : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * R; i += 256) {
    const int cc = i / R, rr = i - cc * R;
    atomicAdd(dw2 + i, dz[cc] * relu_f(hh[rr]));               // W2 [C][R]
    atomicAdd(dw1 + rr * C + cc, dh[rr] * sh[cc]);             // W1 [R][C]
  }
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int k = 0; k < R; ++k) t += w1[k * C + threadIdx.x] * dh[k];
    ds[(int64_t)b * C + threadIdx.x] = t;
  }
}

// dx = dy * e + ds / L   (gradient through the scale and through the squeeze)
__global__ __launch_bounds__(256) void se_scale_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ e, const float* __restrict__ ds,
                                                           int64_t total4, int C, int64_t per_sample, float inv_L, float* __restrict__ dx) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t el = i * 4;
    const int c = (int)(el % C);
    const int b = (int)(el / per_sample);
    const float4 g = *reinterpret_cast<const float4*>(dy + el);
    const float4 ev = *reinterpret_cast<const float4*>(e + (int64_t)b * C + c);
    const float4 sv = *reinterpret_cast<const float4*>(ds + (int64_t)b * C + c);
    *reinterpret_cast<float4*>(dx + el) = make_float4(g.x * ev.x + sv.x * inv_L, g.y * ev.y + sv.y * inv_L, g.z * ev.z + sv.z * inv_L, g.w * ev.w + sv.w * inv_L);
  }
}

inline int se_grid(int64_t n) {
  const int64_t want = (n + 255) / 256;
  return (int)(want < 4096 ? (want < 1 ? 1 : want) : 4096);
}

int se_check(const char* who, int B, int L, int C, int R) {
  SA_CHECK_ARG(B > 0 && L > 0 && C >= 8 && C <= SE_MAXC && 256 % C == 0 && R >= 1 && R <= SE_MAXR, "%s: C must divide 256 (8..256) and 1 <= R <= %d (C=%d, R=%d)", who, SE_MAXR,
               C, R);
  return 0;
}

}  // namespace

extern "C" int sa_se_fwd(const void* x_bf16, int32_t B, int32_t L, int32_t C, const float* w1, const float* w2, int32_t R, float* s, float* h, float* e,
                         void* y_bf16, void* stream) {
  SA_CHECK_ARG(x_bf16 && w1 && w2 && s && h && e && y_bf16, "sa_se_fwd: null pointer");
  if (se_check("sa_se_fwd", B, L, C, R)) return 1;
  SA_CHECK_ARG((((uintptr_t)x_bf16 | (uintptr_t)y_bf16) & 15) == 0, "sa_se_fwd: maps must be 16-byte aligned");
  hipLaunchKernelGGL(se_squeeze_excite_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, L, C, w1, w2, R, s, h, e);
  const int64_t total8 = (int64_t)B * L * C / 8;
  hipLaunchKernelGGL(se_scale_fwd_kernel, dim3(se_grid(total8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x_bf16, e, total8, C, (int64_t)L * C,
                     (bf16_t*)y_bf16);
  SA_LAUNCH_CHECK("sa_se_fwd");
  return 0;
}

extern "C" int sa_se_bwd(const float* dy, const void* x_bf16, int32_t B, int32_t L, int32_t C, const float* w1, const float* w2, int32_t R, const float* s,
                         const float* h, const float* e, float* dx, float* dw1, float* dw2, float* scratch, void* stream) {
  SA_CHECK_ARG(dy && x_bf16 && w1 && w2 && s && h && e && dx && dw1 && dw2 && scratch, "sa_se_bwd: null pointer (scratch: 2 * B * C floats)");
  if (se_check("sa_se_bwd", B, L, C, R)) return 1;
  SA_CHECK_ARG((((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)e | (uintptr_t)scratch) & 15) == 0 && C % 4 == 0, "sa_se_bwd: 16-byte aligned fp32 buffers");
  float* de = scratch;
  float* ds = scratch + (int64_t)B * C;
  if (hipMemsetAsync(de, 0, sizeof(float) * (size_t)B * C, (hipStream_t)stream) != hipSuccess) {
    sa_set_error("sa_se_bwd: memset failed");
    return 2;
  }
  const int rows_per_block = 2048;
  hipLaunchKernelGGL(se_gate_grad_kernel, dim3((L + rows_per_block - 1) / rows_per_block, B), dim3(256), 0, (hipStream_t)stream, dy, (const bf16_t*)x_bf16, L, C,
                     rows_per_block, de);
  hipLaunchKernelGGL(se_excite_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, de, s, h, e, w1, w2, C, R, dw1, dw2, ds);
  const int64_t total4 = (int64_t)B * L * C / 4;
  hipLaunchKernelGGL(se_scale_bwd_kernel, dim3(se_grid(total4)), dim3(256), 0, (hipStream_t)stream, dy, e, ds, total4, C, (int64_t)L * C, 1.f / (float)L, dx);
  SA_LAUNCH_CHECK("sa_se_bwd");
  return 0;
}
