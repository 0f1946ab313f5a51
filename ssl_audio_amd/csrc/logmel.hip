// Log-mel frontend: waveform -> reflect-padded frames -> Hann -> 1024-point real FFT -> |X|^2 -> 64 HTK mel
// bands -> log(x + eps) -> dataset crop / right-pad -> (x - mean) / std.
// Restates torchaudio.transforms.MelSpectrogram as called at datasets.py:39-48 followed by datasets.py:115
// and the crop/pad/normalise of datasets.py:342-354 (frontend parity is unpinned by the reference: DESIGN.md §3).
//
// One wave per frame.  The real FFT runs as a 512-point complex FFT of z[n] = x[2n] + i x[2n+1]:
// three radix-8 passes (512 = 8*8*8) with the 8 points of each butterfly in one lane's registers and two
// LDS exchanges (padded, conflict-free) in between, then the real-input split, the power spectrum into LDS
// and the triangular mel filters with lane <-> mel band (64 bands = 64 lanes).  A workgroup (4 waves)
// produces 16 consecutive frames and writes them as 64-byte row segments.  fp32 throughout; twiddles,
// window and filter weights are host-computed fp64 tables rounded to fp32 (ssl_audio_amd/frontend.py).
#include "common.h"
#include "../../include/ssl_audio_hip.h"

namespace {

constexpr int NFFT = 1024, NC = 512, NBINS = 513, NMEL = 64;
constexpr int FR_PER_WAVE = 4, FR_PER_BLOCK = 16;
constexpr int XS = 72;  // LDS row stride (floats) of the exchange buffers
constexpr int MAXLEN_CAP = 48;  // longest mel band (in bins) whose weight table is staged in LDS (64 HTK bands over 513 bins: 42)

struct cpx { float re, im; };
__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cpx mul_mi(cpx a) { return {a.im, -a.re}; }  // * (-i)

// in-place 8-point DFT, forward (e^{-2 pi i nk/8}), natural order in and out
__device__ __forceinline__ void dft8(cpx v[8]) {
  const float h = 0.70710678118654752440f;
  // stage 1 (radix-2 DIF)
  cpx a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
  cpx a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
  cpx a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
  cpx a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
  // twiddles W8^k on the odd branch: 1, (1-i)/sqrt2, -i, (-1-i)/sqrt2
  a5 = {h * (a5.re + a5.im), h * (a5.im - a5.re)};
  a6 = mul_mi(a6);
  a7 = {h * (a7.im - a7.re), -h * (a7.re + a7.im)};
  // stage 2
  cpx b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = mul_mi(csub(a1, a3));
  cpx b4 = cadd(a4, a6), b6 = csub(a4, a6), b5 = cadd(a5, a7), b7 = mul_mi(csub(a5, a7));
  // stage 3 + bit-reversal to natural order
  v[0] = cadd(b0, b1); v[4] = csub(b0, b1);
  v[2] = cadd(b2, b3); v[6] = csub(b2, b3);
  v[1] = cadd(b4, b5); v[5] = csub(b4, b5);
  v[3] = cadd(b6, b7); v[7] = csub(b6, b7);
}

__device__ __forceinline__ int reflect_idx(int i, int L) {
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  return i;
}

// tw: [1024] (cos, -sin) of 2 pi k / 1024, i.e. W_1024^k.  melw: [maxlen][64] weights, mel_lo/mel_len: per band bin range.
__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ wave, int64_t wave_stride, int L, const float* __restrict__ window,
                                                     const float2* __restrict__ tw, const float* __restrict__ melw,
                                                     const int* __restrict__ mel_lo, const int* __restrict__ mel_len,
                                                     float* __restrict__ out, int64_t out_stride, int n_frames, int T_out, int start,
                                                     float mean, float inv_std, float pad_value, int hop) {
  __shared__ float ex_re[4][8 * XS], ex_im[4][8 * XS];   // per-wave exchange / spectrum buffers
  __shared__ float pw[4][NBINS + MAXLEN_CAP + 3];      // power spectrum + a zero tail: band `lo + q` never needs a clamp
  __shared__ float stage[NMEL][FR_PER_BLOCK + 1];
  __shared__ float melw_s[MAXLEN_CAP * NMEL];            // the filter table, staged once per workgroup
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int clip = blockIdx.y;
  const int tb = blockIdx.x * FR_PER_BLOCK;  // first OUTPUT frame of this block
  const float* wv = wave + (int64_t)clip * wave_stride;
  float* xr = ex_re[w];
  float* xi = ex_im[w];
  const int lo = mel_lo[lane], len = mel_len[lane];
  // longest band of the table (wave-uniform): the filter loop below runs this many iterations on every lane -- the weight table is
  // zero-padded to it -- so that it has a uniform trip count and its loads pipeline (a per-lane `len` bound made every iteration a
  // dependent global load -> LDS read -> fma round trip: ~8 k cycles of latency per frame)
  int maxlen = len;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o, 64));
  maxlen = __builtin_amdgcn_readfirstlane(maxlen);
  // Everything a frame needs besides its samples is the same for every frame: the window and the three twiddle sets live in registers
  // for the wave's four frames, the filter weights in LDS.  (They used to be re-read from global memory in every frame: ~30 loads and
  // four dependent L2 round trips per frame, with one frame in flight per wave.)
  const bool mel_lds = maxlen <= MAXLEN_CAP;
  if (mel_lds)
    for (int i = threadIdx.x; i < maxlen * NMEL; i += 256) melw_s[i] = melw[i];
  for (int i = lane; i < MAXLEN_CAP + 2; i += 64) pw[w][NBINS + i] = 0.f;
  float2 twa[8], twb[8], tws[8];
#pragma unroll
  for (int k1 = 1; k1 < 8; ++k1) twa[k1] = tw[(2 * lane * k1) & 1023];
#pragma unroll
  for (int c = 1; c < 8; ++c) twb[c] = tw[(16 * (lane & 7) * c) & 1023];
#pragma unroll
  for (int r = 0; r < 8; ++r) tws[r] = tw[lane + 64 * r];
  __syncthreads();
  // samples of one frame: lane n2 holds x[2 (64 n1 + n2)], x[2 (64 n1 + n2) + 1], n1 = 0..7 (reflect padding at the clip's ends)
  auto load_frame = [&](int t, float2 (&sm)[8]) {
    const int base = t * hop - NFFT / 2;
    const bool interior = (base >= 0) && (base + NFFT <= L);
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
      const int n = 64 * n1 + lane;
      if (interior) sm[n1] = *reinterpret_cast<const float2*>(wv + base + 2 * n);
      else sm[n1] = make_float2(wv[reflect_idx(base + 2 * n, L)], wv[reflect_idx(base + 2 * n + 1, L)]);
    }
  };
  auto frame_live = [&](int f) { const int to = tb + w * FR_PER_WAVE + f; return (f < FR_PER_WAVE) && (to < T_out) && (to + start < n_frames); };
  float2 smp[8], smp_next[8];
  if (frame_live(0)) load_frame(tb + w * FR_PER_WAVE + start, smp);

  for (int f = 0; f < FR_PER_WAVE; ++f) {
    const int to = tb + w * FR_PER_WAVE + f;   // output frame index
    const int t = to + start;                  // source frame index (dataset crop offset)
    float result = pad_value;
    const bool live = (to < T_out) && (t < n_frames);  // wave-uniform
    if (frame_live(f + 1)) load_frame(t + 1, smp_next);       // the next frame's samples travel while this one is transformed
    if (live) {
      // ---- window: lane n2 holds z[64*n1 + n2], n1 = 0..7
      // (the window is re-read per frame: 4 KiB that stay in the CU's L1; holding it in 16 registers cost a wave per SIMD)
      cpx v[8];
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) {
        const float2 wn = *reinterpret_cast<const float2*>(window + 2 * (64 * n1 + lane));
        v[n1] = {smp[n1].x * wn.x, smp[n1].y * wn.y};
      }
      // ---- pass A: DFT8 over n1, twiddle W_512^{n2*k1} = W_1024^{2*n2*k1}
      dft8(v);
#pragma unroll
      for (int k1 = 1; k1 < 8; ++k1) v[k1] = cmul(v[k1], {twa[k1].x, twa[k1].y});
#pragma unroll
      for (int k1 = 0; k1 < 8; ++k1) { xr[k1 * XS + lane] = v[k1].re; xi[k1 * XS + lane] = v[k1].im; }
      __builtin_amdgcn_wave_barrier();
      // ---- pass B1: lane = (k1, b); DFT8 over a of y[k1][8a + b]; twiddle W_64^{b*c} = W_1024^{16*b*c}
      const int k1 = lane >> 3, b = lane & 7;
#pragma unroll
      for (int a = 0; a < 8; ++a) v[a] = {xr[k1 * XS + 8 * a + b], xi[k1 * XS + 8 * a + b]};
      dft8(v);
#pragma unroll
      for (int c = 1; c < 8; ++c) v[c] = cmul(v[c], {twb[c].x, twb[c].y});
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < 8; ++c) { xr[k1 * XS + 9 * c + b] = v[c].re; xi[k1 * XS + 9 * c + b] = v[c].im; }
      __builtin_amdgcn_wave_barrier();
      // ---- pass B2: lane = (k1, c) [lane = 8*k1 + c]; DFT8 over b of u[k1][c][b] -> Z[k1 + 8c + 64d].  (With lane = 8*c + k1 the
      // reads of rows k1 and k1 + 4 met in one bank: the row stride of 72 floats is 8 mod 32.)
      const int k2 = lane >> 3, c2 = lane & 7;
#pragma unroll
      for (int bb = 0; bb < 8; ++bb) v[bb] = {xr[k2 * XS + 9 * c2 + bb], xi[k2 * XS + 9 * c2 + bb]};
      dft8(v);
      __builtin_amdgcn_wave_barrier();
      // natural-order spectrum Z[k], k = k2 + 8*c2 + 64*d, kept at position k + 4 * (k >> 5) of the exchange buffers (576 floats:
      // the pad makes this store, whose lanes step by 8, and the loads below conflict-free)
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const int k = k2 + 8 * c2 + 64 * d;
        xr[k + 4 * (k >> 5)] = v[d].re; xi[k + 4 * (k >> 5)] = v[d].im;
      }
      __builtin_amdgcn_wave_barrier();
      // ---- real-input split + power: X[k] = E[k] + W_1024^k O[k]
      float* P = pw[w];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int k = lane + 64 * r;
        const int kc = (NC - k) & (NC - 1);
        const int pk = k + 4 * (k >> 5), pc = kc + 4 * (kc >> 5);
        const cpx zk = {xr[pk], xi[pk]}, zc = {xr[pc], -xi[pc]};          // Z[k], conj(Z[N-k])
        const cpx e = {0.5f * (zk.re + zc.re), 0.5f * (zk.im + zc.im)};
        const cpx dd = csub(zk, zc);
        const cpx o = {0.5f * dd.im, -0.5f * dd.re};                     // (-i/2) (Z[k] - conj(Z[N-k]))
        const cpx xk = cadd(e, cmul(o, {tws[r].x, tws[r].y}));
        P[k] = xk.re * xk.re + xk.im * xk.im;
      }
      if (lane == 0) {  // Nyquist bin: X[512] = E[0] - O[0] = Re Z[0] - Im Z[0]
        const float nyq = xr[0] - xi[0];
        P[NC] = nyq * nyq;
      }
      __builtin_amdgcn_wave_barrier();
      // ---- mel band `lane`: sum over its bin range
      float m = 0.f;
      if (mel_lds) {
        // (weights past a band's end are zero and P has a zero tail past bin 512: no clamp, one address per lane, immediate offsets)
        const float* Pl = P + lo;
        const float* Wl = melw_s + lane;
#pragma unroll 8
        for (int q = 0; q < maxlen; ++q) m += Wl[q * NMEL] * Pl[q];                                   // same order of additions as before
      } else {
#pragma unroll 8
        for (int q = 0; q < maxlen; ++q) m += melw[q * NMEL + lane] * P[min(lo + q, NBINS - 1)];
      }
      result = (__logf(m + 1.1920929e-07f) - mean) * inv_std;     // v_log_f32: 1 ulp
      __builtin_amdgcn_wave_barrier();
    }
    stage[lane][w * FR_PER_WAVE + f] = result;
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) smp[n1] = smp_next[n1];
  }
  __syncthreads();
  // ---- write [64 mel][16 frames]: thread -> (mel = tid / 4, 4 consecutive frames)
  const int mel = threadIdx.x >> 2, f0 = (threadIdx.x & 3) * 4;
  float* orow = out + (int64_t)clip * out_stride + (int64_t)mel * T_out;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int to = tb + f0 + q;
    if (to < T_out) orow[to] = stage[mel][f0 + q];
  }
}

}  // namespace

extern "C" int sa_logmel_fwd(const float* wave, int64_t wave_stride, int32_t n_clips, int32_t n_samples, const float* window,
                             const float* twiddle, const float* mel_weights, const int32_t* mel_lo, const int32_t* mel_len, float* out,
                             int64_t out_stride, int32_t T_out, int32_t start, float mean, float stdv, int32_t hop, void* stream) {
  SA_CHECK_ARG(wave && window && twiddle && mel_weights && mel_lo && mel_len && out, "sa_logmel_fwd: null pointer");
  SA_CHECK_ARG(n_clips > 0 && n_samples > NFFT / 2 && T_out > 0 && hop > 0 && hop % 2 == 0 && start >= 0, "sa_logmel_fwd: bad sizes");
  SA_CHECK_ARG(stdv != 0.f, "sa_logmel_fwd: std must be non-zero");
  SA_CHECK_ARG(((uintptr_t)wave & 7) == 0 && wave_stride % 2 == 0, "sa_logmel_fwd: waveform rows must be 8-byte aligned");
  const int n_frames = 1 + n_samples / hop;
  const float pad_value = (0.f - mean) / stdv;  // right zero-pad happens BEFORE normalisation (datasets.py:346-354)
  dim3 grid((T_out + FR_PER_BLOCK - 1) / FR_PER_BLOCK, n_clips);
  hipLaunchKernelGGL(logmel_kernel, grid, dim3(256), 0, (hipStream_t)stream, wave, wave_stride, n_samples, window,
                     reinterpret_cast<const float2*>(twiddle), mel_weights, mel_lo, mel_len, out, out_stride, n_frames, T_out, start, mean,
                     1.0f / stdv, pad_value, hop);
  SA_LAUNCH_CHECK("sa_logmel_fwd");
  return 0;
}
