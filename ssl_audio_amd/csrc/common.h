// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of ssl_audio_amd.
// wave = 64 lanes everywhere; no other architecture is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SA_WAVE 64

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---- error plumbing shared by all C-ABI entry points (capi.cpp owns the storage)
extern "C" void sa_set_error(const char* fmt, ...);
#define SA_CHECK_ARG(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      sa_set_error(__VA_ARGS__);                \
      return 1;                                 \
    }                                           \
  } while (0)
#define SA_LAUNCH_CHECK(name)                                                  \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      sa_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
      return 2;                                                                \
    }                                                                          \
  } while (0)

// ---- wave-level reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// ReLU that keeps a NaN a NaN, as torch's relu does (fmaxf / v_max_f32 return the non-NaN operand, which would turn a diverged
// activation into a clean-looking zero and hide the non-finite loss the step's device-side gate and the host check look for)
__device__ __forceinline__ float relu_f(float v) { return v < 0.f ? 0.f : v; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); `red` is >= 4 floats of LDS. All threads get the result.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---- bf16 helpers (plain casts: hipcc emits v_cvt_pk_bf16_f32, round-to-nearest-even, NaN preserving)
__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }

// erf-GELU (timm Mlp act_layer=nn.GELU, models/mae.py:155) and its derivative.  erf is evaluated branch-free with
// Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 + one fast-exp ulp): libm's erff is ~40 instructions with divergent
// branches, and 64 unrolled copies of it in a GEMM epilogue cost 150 VGPRs and spills; the result is rounded to bf16
// (8 significant bits) right afterwards, so the two are indistinguishable downstream.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float r = 1.0f - poly * t * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// GELU(x) and GELU'(x) from one exponential and one reciprocal (forward fc1 with act = 3 stores the derivative for the backward)
__device__ __forceinline__ void gelu_pair(float x, float& y, float& dy) {
  // t = 1 / (1 + p |x| / sqrt 2) straight from x (one fma with an |.| source modifier), exp(-x^2 / 2) from x * x: the two products
  // pack across neighbouring elements (v_pk_mul_f32), which the |x| / sqrt 2 intermediate did not
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x), 1.0f));
  const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);     // exp(-x^2 / 2)
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float cdf = 0.5f * (1.0f + copysignf(1.0f - poly * t * e, x));
  y = x * cdf;
  dy = fmaf(x, 0.39894228040143267794f * e, cdf);
}

// GELU'(x) = Phi(x) + x * phi(x).  erf(x / sqrt 2) needs exp(-x^2 / 2), which is also phi(x) up to a constant: one v_exp and one
// v_rcp per element instead of three transcendentals (the fc2-dgrad epilogue evaluates this 128 times per lane per tile).
__device__ __forceinline__ float dgelu_f(float x) {
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, fabsf(x), 1.0f));
  const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);     // exp(-x^2 / 2)
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  const float erf_abs = 1.0f - poly * t * e;                                       // erf(|x| / sqrt 2)
  const float cdf = 0.5f * (1.0f + copysignf(erf_abs, x));
  return fmaf(x, 0.39894228040143267794f * e, cdf);
}

// buffer resource over [base, base+bytes): out-of-range lanes of a buffer load return 0 (used for ragged tiles)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
