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

// dcls[j] += sum_s dx[s][0][j]
__global__ void cls_grad_kernel(const float* __restrict__ dx, int S, int64_t seq_stride, int d, float* __restrict__ dcls) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= d) return;
  float a = 0.f;
  for (int s = blockIdx.y; s < S; s += gridDim.y) a += dx[(int64_t)s * seq_stride + j];
  atomicAdd(dcls + j, a);
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
  int gy = S < 64 ? S : 64;
  hipLaunchKernelGGL(cls_grad_kernel, dim3((d + 255) / 256, gy), dim3(256), 0, (hipStream_t)stream, dx, S, seq_stride, d, dcls);
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
