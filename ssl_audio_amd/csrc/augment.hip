// Spectrogram augmentation stack on the GPU, one fused kernel per view:
//   MixupBYOLA / log_mixup_exp (augmentations.py:81-85,103-117)
//   -> RandomResizeCrop on a *virtual* zero canvas, bicubic align_corners=True (augmentations.py:40-55)
//   -> RandomLinearFader (augmentations.py:69-74)
// plus NormalizeBatch (augmentations.py:229-232) and the bf16 patch layout the patch-embed GEMM consumes.
//
// HBM-bound.  A workgroup owns an output slab [F_out x TT] of one view: it first materialises the source
// columns that slab needs -- log-mixed once per source pixel, zero outside the pasted input -- as an LDS
// tile (coalesced row reads of x and of the bank entry z), then every thread runs the 4x4 cubic taps out of
// LDS and adds the fade slope.  The canvas is never built in memory.
// Sampling is NOT done here: (alpha, bank slot, i, j, h, w, head, tail) arrive as explicit parameters so the
// CPU oracle and this kernel consume identical draws.
#include <stdlib.h>
#include "common.h"
#include "../../include/ssl_audio_hip.h"

namespace {

constexpr int LDS_H = 64;           // source rows per tile (crop height <= canvas height)
constexpr float kEps32 = 1.1920929e-07f;
constexpr float kA = -0.75f;        // PyTorch bicubic coefficient

struct ViewParams {  // mirrored by ssl_audio_amd/augmentations.py (8 x float32 per view)
  float alpha;       // mixup weight of the bank entry (0.2 * U[0,1)); ignored when mix_slot < 0
  float i, j, h, w;  // crop rectangle on the virtual canvas (integers stored as float)
  float head, tail;  // fader end points
  float lambd;       // MixGaussianNoise weight (ratio * U[0,1)); used only when the launch carries a noise tensor
};

__device__ __forceinline__ void cubic_w(float t, float w[4]) {
  const float x0 = t + 1.f, x3 = 2.f - t, x2 = 1.f - t;
  w[0] = ((kA * x0 - 5.f * kA) * x0 + 8.f * kA) * x0 - 4.f * kA;
  w[1] = ((kA + 2.f) * t - (kA + 3.f)) * t * t + 1.f;
  w[2] = ((kA + 2.f) * x2 - (kA + 3.f)) * x2 * x2 + 1.f;
  w[3] = ((kA * x3 - 5.f * kA) * x3 + 8.f * kA) * x3 - 4.f * kA;
}

// LDS_W = source columns per tile (incl. the cubic halo), TXW = output columns per slab.  <132, 64>: 34 KiB of LDS, four workgroups per
// CU; <72, 32>: 18 KiB, eight per CU -- a workgroup's load phase and its interpolation phase are serial, so more, smaller workgroups
// keep the memory pipe busier (at 8 % more halo columns fetched twice).
template <int LDS_W, int TXW>
__global__ __launch_bounds__(256) void augment_kernel(const float* __restrict__ lms, int64_t clip_stride, const int* __restrict__ src_slot,
                                                      const int* __restrict__ mix_slot, const ViewParams* __restrict__ params,
                                                      float* __restrict__ out, int F_in, int T_in, int canvas_h, int canvas_w, int F_out,
                                                      int T_out, int TT, int do_fade, const float* __restrict__ noise) {
  __shared__ float tile[LDS_H * LDS_W];
  const int view = blockIdx.y;
  const int t0 = blockIdx.x * TT;
  const int nt = min(TT, T_out - t0);
  const ViewParams pr = params[view];
  const int ci = (int)pr.i, cj = (int)pr.j, ch = (int)pr.h, cw = (int)pr.w;
  const float* x = lms + (int64_t)src_slot[view] * clip_stride;
  const int ms = mix_slot ? mix_slot[view] : -1;
  const float* z = ms >= 0 ? lms + (int64_t)ms * clip_stride : nullptr;
  const float wa = 1.f - pr.alpha, wb = pr.alpha;  // log_mixup_exp(x, z, 1 - alpha): weight of x is 1 - alpha
  // MixGaussianNoise between mixup and the crop (utils/transforms.py:21-22, augmentations.py:132-141): standard-normal draws of this
  // view's [F_in, T_in] grid, a function of the SOURCE pixel, so the halo columns two slabs share get the same noise
  const float* nz = noise ? noise + (int64_t)view * F_in * T_in : nullptr;
  // paste offsets of the input on the canvas (augmentations.py:47)
  const int px = (canvas_w - T_in) / 2, py = (canvas_h - F_in) / 2;

  // source-column window of this slab (crop coordinates), align_corners=True
  const float sx = T_out > 1 ? __fdiv_rn((float)(cw - 1), (float)(T_out - 1)) : 0.f;   // IEEE division, as the CPU reference
  const float sy = F_out > 1 ? __fdiv_rn((float)(ch - 1), (float)(F_out - 1)) : 0.f;
  const int xlo = max(0, (int)floorf(__fmul_rn(sx, (float)t0)) - 1);
  const int xhi = min(cw - 1, (int)floorf(__fmul_rn(sx, (float)(t0 + nt - 1))) + 2);
  const int wt = xhi - xlo + 1;  // host guarantees wt <= LDS_W via TT

  // ---- stage 1: mixed source tile [ch x wt].  Thread -> (column, row parity): a wave reads 64 consecutive source columns of one row
  // (256 contiguous bytes of x, of the bank entry z and of the noise grid).  EIGHT rows' loads are issued before the first value is
  // used: with one row per iteration every iteration paid a full HBM round trip (the loop was latency-bound: 32 dependent trips per
  // block), now there are four batches of up to 24 independent loads.
  constexpr int CT = TXW * 2, RP = 256 / CT;          // column threads per row, row phases (CT = 128: 2 phases; 64: 4)
  for (int xx = threadIdx.x % CT; xx < wt; xx += CT) {
    const int cc = cj + xlo + xx - px;  // input coordinates
    const bool col_in = cc >= 0 && cc < T_in;
    for (int yy0 = threadIdx.x / CT; yy0 < ch; yy0 += 8 * RP) {
      float xv[8], zv[8], nv[8];
      bool ok[8];
      // UNCONDITIONAL loads from an always-valid (clamped) address, selected afterwards: a predicated load compiles to a branch whose
      // join waits for it (`s_waitcnt vmcnt(0)` after every row: 32 serial HBM round trips per block -- what this loop used to cost)
      const float* zs = z ? z : x;
      const float* ns = nz ? nz : x;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int yy = yy0 + RP * u, r = ci + yy - py;
        ok[u] = col_in && yy < ch && r >= 0 && r < F_in;
        const int64_t off = (int64_t)min(max(r, 0), F_in - 1) * T_in + min(max(cc, 0), T_in - 1);
        xv[u] = x[off];
        zv[u] = zs[off];
        nv[u] = ns[off];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = ok[u] ? xv[u] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int yy = yy0 + RP * u;
        float v = xv[u];
        if (ok[u]) {
          // v_exp_f32 / v_log_f32 (1 ulp each) instead of libm's expf / logf: those were ~1 500 of the 2 200 vector instructions a
          // wave executed (PMC, profiles/r04_frontend_pmc.txt); the results move by <= 1e-6 (the golden tests hold 1e-5)
          if (z) v = __logf(wa * __expf(v) + wb * __expf(zv[u]) + kEps32);
          if (nz) v = __logf((1.f - pr.lambd) * __expf(v) + __expf(pr.lambd * nv[u]) + kEps32);
        }
        if (yy < ch) tile[yy * LDS_W + xx] = v;
      }
    }
  }
  __syncthreads();

  // ---- stage 2: bicubic + fade.  Thread -> output column tx (fixed) x a run of CONSECUTIVE output rows: the horizontal taps and
  // weights are computed once per thread, and the horizontally interpolated value h(r) of a source row is computed once and kept in a
  // four-deep register window that slides down the tile (consecutive output rows share three of their four source rows at scale <= 1):
  // 4-8 LDS reads per output instead of 16.  The vertical taps are wave-uniform; a wave stores 64 consecutive frames of one mel row.
  const float fade_step = T_out > 1 ? (pr.tail - pr.head) / (float)(T_out - 1) : 0.f;
  float* o = out + (int64_t)view * F_out * T_out;
  constexpr int NG = 256 / TXW;                       // output row groups (4 or 8)
  const int tx = threadIdx.x % TXW, rg = threadIdx.x / TXW;
  if (tx < nt) {
    const int t = t0 + tx;
    // rounded products (no FMA contraction into the fraction): PyTorch's CPU kernel rounds scale*index to fp32 first
    float rx = sx * (float)t;
    asm volatile("" : "+v"(rx));             // keep the ROUNDED product: hipcc would otherwise contract x*y - floor into an fma
    const float fx = floorf(rx);
    float wx[4];
    cubic_w(rx - fx, wx);
    const int ix = (int)fx;
    int cx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cx[k] = min(max(ix - 1 + k, 0), cw - 1) - xlo;  // taps clamp to the CROP bounds (PyTorch border rule)
    float fade = 0.f;
    if (do_fade) {
      // torch.linspace: first half counts up from head, second half counts down from tail
      const int half = T_out / 2;
      fade = (t < half) ? pr.head + fade_step * (float)t : pr.tail - fade_step * (float)(T_out - 1 - t);
    }
    auto hrow = [&](int r) {                 // source row r (clamped to the crop), interpolated at this thread's column
      const float* row = tile + min(max(r, 0), ch - 1) * LDS_W;
      return row[cx[0]] * wx[0] + row[cx[1]] * wx[1] + row[cx[2]] * wx[2] + row[cx[3]] * wx[3];
    };
    const int rows_per = (F_out + NG - 1) / NG;
    const int fy_end = min(F_out, (rg + 1) * rows_per);
    int rwin = 0;
    bool have = false;
    float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f;   // h(rwin) .. h(rwin + 3)
    for (int fy = rg * rows_per; fy < fy_end; ++fy) {
      float ry = sy * (float)fy;
      asm volatile("" : "+v"(ry));
      const float fyf = floorf(ry);
      float wy[4];
      cubic_w(ry - fyf, wy);
      const int want = (int)fyf - 1;         // wave-uniform: the window moves in uniform control flow
      if (!have || want - rwin >= 4) {
        h0 = hrow(want); h1 = hrow(want + 1); h2 = hrow(want + 2); h3 = hrow(want + 3);
        rwin = want;
        have = true;
      } else {
        while (rwin < want) {
          h0 = h1; h1 = h2; h2 = h3;
          h3 = hrow(rwin + 4);
          ++rwin;
        }
      }
      float acc = 0.f;
      acc += h0 * wy[0];
      acc += h1 * wy[1];
      acc += h2 * wy[2];
      acc += h3 * wy[3];
      if (do_fade) acc += fade;
      o[(int64_t)fy * T_out + t] = acc;
    }
  }
}

// ---- NormalizeBatch: (X - mean) / clamp(std_unbiased, eps) over the whole [B,1,F,T] tensor
// (block partials in a device global, added in block order by a one-wave launch: a fixed summation order, no atomics)
__device__ double sumsq_partials[2 * 1024];

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, float shift) {
  __shared__ float red[4];
  float s = 0.f, q = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = x[i] - shift;
    s += d;
    q += d * d;
  }
  s = block_sum_256(s, red);
  q = block_sum_256(q, red);
  if (threadIdx.x == 0) {
    sumsq_partials[2 * blockIdx.x] = (double)s;
    sumsq_partials[2 * blockIdx.x + 1] = (double)q;
  }
}

__global__ __launch_bounds__(64) void sumsq_finish_kernel(int nblocks, double* __restrict__ acc) {
  __shared__ double lanes[2][64];
  double s = 0.0, q = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) {
    s += sumsq_partials[2 * b];
    q += sumsq_partials[2 * b + 1];
  }
  lanes[0][threadIdx.x] = s;
  lanes[1][threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tq = 0.0;
    for (int l = 0; l < 64; ++l) {
      ts += lanes[0][l];
      tq += lanes[1][l];
    }
    acc[0] = ts;
    acc[1] = tq;
  }
}

__global__ void normalize_apply_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float shift, const double* __restrict__ acc,
                                       float eps, float stat_div) {
  const double s = acc[0], q = acc[1];
  const double mean_d = s / (double)n;
  const double var = (q - s * mean_d) / (double)(n - 1);
  const float mean = ((float)mean_d + shift) / stat_div;
  const float stdv = fmaxf((float)sqrt(var > 0.0 ? var : 0.0) / stat_div, eps);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = (x[i] - mean) / stdv;
}

// ---- patch layout for the patch-embed GEMM: [S,1,F,T] fp32 -> bf16 [S * gh * gw, ph * pw], k = i * pw + j
// (conv weight [d,1,ph,pw] flattened; patch order h-major as conv output .flatten(2), models/mae.py:42)
__global__ void patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int S, int F, int T, int ph, int pw, int gh, int gw) {
  const int64_t n = (int64_t)S * gh * gw * ph * pw;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx;
    const int j = (int)(r % pw); r /= pw;
    const int i = (int)(r % ph); r /= ph;
    const int gx = (int)(r % gw); r /= gw;
    const int gy = (int)(r % gh); r /= gh;
    const int s = (int)r;
    out[idx] = f2bf(img[((int64_t)s * F + gy * ph + i) * T + gx * pw + j]);
  }
}

}  // namespace

extern "C" int sa_augment_views(const float* lms, int64_t clip_stride, const int32_t* src_slot, const int32_t* mix_slot, const float* params,
                                float* out, int32_t n_views, int32_t F_in, int32_t T_in, int32_t canvas_h, int32_t canvas_w, int32_t F_out,
                                int32_t T_out, float max_w_ratio, int32_t do_fade, const float* noise, void* stream) {
  SA_CHECK_ARG(lms && src_slot && params && out && n_views > 0, "sa_augment_views: bad args");
  SA_CHECK_ARG(canvas_h <= LDS_H, "sa_augment_views: canvas height %d exceeds the %d-row LDS tile", canvas_h, LDS_H);
  SA_CHECK_ARG(F_out > 0 && T_out > 0 && F_in > 0 && T_in > 0 && max_w_ratio > 0.f, "sa_augment_views: bad sizes");
  // slab width so that ceil(TT * ratio) + 4 source columns fit the LDS tile; ratio = max crop width / T_out
  static const char* wide_env = getenv("SA_AUG_WIDE");      // experiment knob: 1 = the 132-column tile / 64-column slabs only
  const bool wide = wide_env && wide_env[0] == '1';
  int TT = (int)floorf((float)((wide ? 132 : 72) - 5) / max_w_ratio);
  const int cap = wide ? 64 : 32;
  if (TT > cap) TT = cap;
  if (!wide && TT >= 8) {
    dim3 grid((T_out + TT - 1) / TT, n_views);
    hipLaunchKernelGGL((augment_kernel<72, 32>), grid, dim3(256), 0, (hipStream_t)stream, lms, clip_stride, src_slot, mix_slot,
                       reinterpret_cast<const ViewParams*>(params), out, F_in, T_in, canvas_h, canvas_w, F_out, T_out, TT, do_fade, noise);
  } else {                                                  // steep down-sampling (local crops of a long clip): the wide source tile
    TT = (int)floorf((float)(132 - 5) / max_w_ratio);
    if (TT > 64) TT = 64;
    SA_CHECK_ARG(TT >= 1, "sa_augment_views: crop/out width ratio %f too large", max_w_ratio);
    dim3 grid((T_out + TT - 1) / TT, n_views);
    hipLaunchKernelGGL((augment_kernel<132, 64>), grid, dim3(256), 0, (hipStream_t)stream, lms, clip_stride, src_slot, mix_slot,
                       reinterpret_cast<const ViewParams*>(params), out, F_in, T_in, canvas_h, canvas_w, F_out, T_out, TT, do_fade, noise);
  }
  SA_LAUNCH_CHECK("sa_augment_views");
  return 0;
}

namespace {
// MixGaussianNoise (augmentations.py:125-141): log((1 - lambda) * exp(x) + exp(lambda * n) + eps), n ~ N(0, 1) supplied by the caller
__global__ void mix_gaussian_noise_kernel(const float* __restrict__ x, const float* __restrict__ nrm, int64_t n, float lambd, float eps,
                                          float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = logf((1.f - lambd) * expf(x[i]) + expf(lambd * nrm[i]) + eps);
}

// RunningNorm (augmentations.py:144-214), one workgroup per channel.  state = {mu, s2} per channel; n_mean / n_var = how many samples the
// running mean / the running second moment have seen (the reference divides the increment by that count, not count + 1: kept).
__global__ __launch_bounds__(1024) void running_norm_kernel(const float* __restrict__ x, int64_t per_channel, float* __restrict__ state,
                                                            int n_seen, int update, float eps, float* __restrict__ out) {
  __shared__ double red[16];
  __shared__ float bc[2];
  const float* xc = x + (int64_t)blockIdx.x * per_channel;
  float* oc = out + (int64_t)blockIdx.x * per_channel;
  float* st = state + 2 * blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  auto block_sum = [&](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    return t;
  };
  if (update) {
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < per_channel; i += 1024) s += (double)xc[i];
    const float m = (float)(block_sum(s) / (double)per_channel);
    if (threadIdx.x == 0) {
      const float mu = n_seen == 0 ? m : st[0] + (m - st[0]) / (float)n_seen;
      st[0] = mu;
      bc[0] = mu;
    }
    __syncthreads();
    const float mu = bc[0];
    double q = 0.0;
    for (int64_t i = threadIdx.x; i < per_channel; i += 1024) {
      const float dlt = xc[i] - mu;
      q += (double)(dlt * dlt);
    }
    const float v = (float)(block_sum(q) / (double)per_channel);
    if (threadIdx.x == 0) st[1] = n_seen == 0 ? v : st[1] + (v - st[1]) / (float)n_seen;
    __syncthreads();
  }
  const float mu = st[0];
  const float sd = fmaxf(sqrtf(st[1]), eps);
  for (int64_t i = threadIdx.x; i < per_channel; i += 1024) oc[i] = (xc[i] - mu) / sd;
}
}  // namespace

extern "C" int sa_mix_gaussian_noise(const float* x, const float* normal, int64_t n, float lambd, float eps, float* out, void* stream) {
  SA_CHECK_ARG(x && normal && out && n > 0, "sa_mix_gaussian_noise: bad args");
  const int64_t want = (n + 255) / 256;
  hipLaunchKernelGGL(mix_gaussian_noise_kernel, dim3((int)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream, x, normal, n, lambd, eps, out);
  SA_LAUNCH_CHECK("sa_mix_gaussian_noise");
  return 0;
}

extern "C" int sa_running_norm(const float* x, int32_t channels, int64_t per_channel, float* state, int32_t n_seen, int32_t update, float eps,
                               float* out, void* stream) {
  SA_CHECK_ARG(x && state && out && channels > 0 && per_channel > 0 && n_seen >= 0, "sa_running_norm: bad args");
  hipLaunchKernelGGL(running_norm_kernel, dim3(channels), dim3(1024), 0, (hipStream_t)stream, x, per_channel, state, n_seen, update, eps, out);
  SA_LAUNCH_CHECK("sa_running_norm");
  return 0;
}

extern "C" int sa_normalize_batch(const float* x, float* y, int64_t n, float shift, double* workspace2, float eps, float stat_div, void* stream) {
  SA_CHECK_ARG(x && y && workspace2 && n > 1 && stat_div > 0.f, "sa_normalize_batch: bad args");
  int64_t want = (n + 256 * 16 - 1) / (256 * 16);
  const int grid = (int)(want < 1024 ? (want < 1 ? 1 : want) : 1024);
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, shift);
  hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, grid, workspace2);
  hipLaunchKernelGGL(normalize_apply_kernel, dim3(grid * 4), dim3(256), 0, (hipStream_t)stream, x, y, n, shift, workspace2, eps, stat_div);
  SA_LAUNCH_CHECK("sa_normalize_batch");
  return 0;
}

extern "C" int sa_patchify_bf16(const float* img, void* out, int32_t S, int32_t F, int32_t T, int32_t ph, int32_t pw, void* stream) {
  SA_CHECK_ARG(img && out && S > 0 && F >= ph && T >= pw && ph > 0 && pw > 0, "sa_patchify_bf16: bad args");
  const int gh = F / ph, gw = T / pw;
  const int64_t n = (int64_t)S * gh * gw * ph * pw;
  int64_t want = (n + 255) / 256;
  const int grid = (int)(want < 8192 ? want : 8192);
  hipLaunchKernelGGL(patchify_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, img, (bf16_t*)out, S, F, T, ph, pw, gh, gw);
  SA_LAUNCH_CHECK("sa_patchify_bf16");
  return 0;
}
