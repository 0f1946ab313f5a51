// Log-mel frontend: waveform -> reflect-padded frames -> Hann -> 1024-point real FFT -> |X|^2 -> 64 HTK mel
// bands -> log(x + eps) -> dataset crop / right-pad -> (x - mean) / std.
// Restates torchaudio.transforms.MelSpectrogram as called at datasets.py:39-48 followed by datasets.py:115
// and the crop/pad/normalise of datasets.py:342-354 (frontend parity is unpinned by the reference: DESIGN.md §3).
//
// A workgroup (4 waves) produces 16 consecutive frames of one clip in two phases, and is persistent: it walks the list of
// (16-frame group, clip) pairs with the stride of the grid (3 workgroups per CU), so the tables are fetched once per workgroup.
//   1. One wave per frame, four frames per wave.  The real FFT runs as a 512-point complex FFT of z[n] = x[2n] + i x[2n+1]:
//      three radix-8 passes (512 = 8*8*8) with the 8 points of each butterfly in one lane's registers as (re, im) pairs -- all
//      complex arithmetic is packed fp32 (v_pk_add/mul/fma_f32, the rotations by -i folded into op_sel / neg modifiers) -- and two
//      LDS exchanges of 8-byte complex values (ds_write_b64 / ds_read_b64, padded conflict-free).  The real-input split pairs
//      Z[k] with Z[512 - k], which sits in another lane's registers: a ds_bpermute instead of a third exchange.  The power
//      spectrum of the frame goes to its row of P[16][514] in LDS.
//   2. The 513 -> 64 mel projection of the 16 frames is a [64 x K] x [K x 16] product on the fp32 MFMA
//      (v_mfma_f32_16x16x4_f32) over the bins each group of 16 bands covers, shared equally by the four waves (their weights
//      stay in registers for the life of the workgroup), partial sums combined through LDS in a fixed order.
//      Then log, normalise, and 64-byte row segments out.
// Compile with -DSA_LOGMEL_DBG=1 for cycle stamps of the phases (scripts/diag/logmel_stamps.py).
// fp32 throughout; twiddles, window and filter weights are host-computed fp64 tables rounded to fp32 (ssl_audio_amd/frontend.py).
#include "common.h"
#include "../../include/ssl_audio_hip.h"

#ifndef SA_LOGMEL_DBG
#define SA_LOGMEL_DBG 0
#endif
namespace {

constexpr int NFFT = 1024, NC = 512, NBINS = 513, NMEL = 64;
constexpr int FR_PER_WAVE = 4, FR_PER_BLOCK = 16;
constexpr int XS = 72;    // row stride (complex values) of the exchange buffers: 72 = 8 mod 32 keeps every b64 access below conflict-free
constexpr int PS = 514;   // row stride (floats) of the power spectra: 514 = 2 mod 32, so the MFMA B reads (16 frames x 2 bins per half-wave) hit 32 banks

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// x + (-i) y = (x.re + y.im, x.im - y.re)
__device__ __forceinline__ v2f radd(v2f x, v2f y) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
// x - (-i) y = (x.re - y.im, x.im + y.re)
__device__ __forceinline__ v2f rsub(v2f x, v2f y) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
// complex product v * t in two packed instructions
__device__ __forceinline__ v2f cmul(v2f v, v2f t) {
  v2f r, o;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(v), "v"(t));   // (-v.im t.im, v.im t.re)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(o) : "v"(v), "v"(t), "v"(r));                // + (v.re t.re, v.re t.im)
  return o;
}

// in-place 8-point DFT, forward (e^{-2 pi i nk/8}), natural order in and out; 28 packed instructions
__device__ __forceinline__ void dft8(v2f v[8]) {
  const float h = 0.70710678118654752440f;
  // stage 1 (radix-2 DIF)
  const v2f a0 = v[0] + v[4], a4 = v[0] - v[4];
  const v2f a1 = v[1] + v[5], a2 = v[2] + v[6], a6 = v[2] - v[6], a3 = v[3] + v[7];
  v2f a5 = v[1] - v[5], a7 = v[3] - v[7];
  // twiddles W8^k on the odd branch: 1, (1-i)/sqrt2, -i, (-1-i)/sqrt2   (the -i on a6 rides on the adds of stage 2)
  a5 = radd(a5, a5) * h;
  a7 = rsub(a7, a7) * (-h);
  // stage 2 (b3 and b7 still owe a factor -i: it rides on the adds of stage 3)
  const v2f b0 = a0 + a2, b2 = a0 - a2, b1 = a1 + a3, b3 = a1 - a3;
  const v2f b4 = radd(a4, a6), b6 = rsub(a4, a6), b5 = a5 + a7, b7 = a5 - a7;
  // stage 3 + bit-reversal to natural order
  v[0] = b0 + b1; v[4] = b0 - b1;
  v[2] = radd(b2, b3); v[6] = rsub(b2, b3);
  v[1] = b4 + b5; v[5] = b4 - b5;
  v[3] = radd(b6, b7); v[7] = rsub(b6, b7);
}

__device__ __forceinline__ int reflect_idx(int i, int L) {
  if (i < 0) i = -i;
  if (i >= L) i = 2 * (L - 1) - i;
  return i;
}

// LDS accesses of the exchanges are volatile so that they stay single b64 operations: two ds_read_b64 cost 2 LDS cycles each, the
// ds_read2_b64 the compiler would merge them into costs 8
typedef __attribute__((address_space(3))) v2f lds_v2f;
__device__ __forceinline__ v2f lds_ld(const v2f* p) { return *(const volatile lds_v2f*)(p); }
__device__ __forceinline__ void lds_st(v2f* p, v2f v) { *(volatile lds_v2f*)(p) = v; }

// tw: [1024] (cos, -sin) of 2 pi k / 1024, i.e. W_1024^k.  melw: [maxlen][64] weights, mel_lo/mel_len: per band bin range.
// Persistent: a workgroup takes the 16-frame groups blockIdx.x, blockIdx.x + gridDim.x, ... of the (clip, group) list, so the tables
// below are fetched once per workgroup, not once per 16 frames.
template <bool PER_CLIP>
__global__ __launch_bounds__(256, 3) void logmel_kernel(const float* __restrict__ wave, int64_t wave_stride, int L, const float* __restrict__ window,
                                                        const v2f* __restrict__ tw, const float* __restrict__ melw,
                                                        const int* __restrict__ mel_lo, const int* __restrict__ mel_len,
                                                        float* __restrict__ out, int64_t out_stride, int T_out, int start_all,
                                                        const int* __restrict__ starts, const int* __restrict__ lengths,
                                                        const int* __restrict__ offsets, float mean, float inv_std, float pad_value, int hop,
                                                        int n_clips, int n_groups) {
  __shared__ v2f ex[4][8 * XS];                         // per-wave exchange buffers; the mel partial sums [wave][group][lane] after a group's frames
  __shared__ float pw[FR_PER_BLOCK * PS + 32];          // power spectra [frame][bin], one pad bin per row (+ slack past the last row), zeroed
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  v2f* x = ex[w];
#if SA_LOGMEL_DBG
  long long st[9];
  st[0] = __builtin_readcyclecounter();
#endif
  if (threadIdx.x < FR_PER_BLOCK) pw[threadIdx.x * PS + NBINS] = 0.f;
  if (threadIdx.x < 32) pw[FR_PER_BLOCK * PS + threadIdx.x] = 0.f;
  // Everything a group of frames needs besides its samples is the same for every group: the window, the three twiddle sets and this wave's
  // slice of the mel weights live in registers for the life of the workgroup (LDS, not registers, is what caps the CU at three workgroups).
  const int k2 = lane >> 3, c2 = lane & 7;       // pass B: lane = (row of the 8 x 64 exchange, column group)
  v2f twa[8], twb[8], tws[8], win[8];
#pragma unroll
  for (int n1 = 0; n1 < 8; ++n1) win[n1] = *reinterpret_cast<const v2f*>(window + 2 * (64 * n1 + lane));
#pragma unroll
  for (int k1 = 1; k1 < 8; ++k1) twa[k1] = tw[(2 * lane * k1) & 1023];
#pragma unroll
  for (int c = 1; c < 8; ++c) twb[c] = tw[(16 * c2 * c) & 1023];
#pragma unroll
  for (int d = 0; d < 8; ++d) tws[d] = tw[k2 + 8 * c2 + 64 * d];
  // the lane that holds Z[512 - k] for this lane's Z[k], k = k2 + 8 c2 + 64 d: (8 - k2, 7 - c2) with register 7 - d; row k2 = 0 pairs
  // with (0, 8 - c2), and lane 0 (k = 64 d) with itself, register (8 - d) & 7
  const int partner = (k2 != 0) ? 8 * (8 - k2) + (7 - c2) : ((8 - c2) & 7);
  const int partner_addr = partner * 4;

  // ---- the mel projection's work list (the same for every group of frames).  D[band][frame] = sum_k W[band][k] P[frame][k] on the fp32 MFMA:
  // A (16 bands x 4 bins): lane holds W[16 g + (lane & 15)][k0 + (lane >> 4)]; B (4 bins x 16 frames): P[lane & 15][k0 + (lane >> 4)];
  // D: lane holds bands 16 g + 4 (lane >> 4) + r, r = 0..3, of frame lane & 15.
  // Band group g only meets the bins its 16 triangles cover (44 / 77 / 141 / 255 bins from bin kb[g] on for the 64 HTK bands of the
  // reference).  Its products are taken eight at a time (32 bins: "octets", 2 / 3 / 5 / 8 of them); the octets of all groups form one list,
  // cut into four equal slices -- one per wave, so that the four SIMDs share the MFMA work -- and a wave's partial sums per group meet
  // in LDS.  Bins an octet reaches past its group's last carry weight zero (q >= len) and read finite values (the next row, the slack).
  const int i16 = lane & 15, kk = lane >> 4;
  int kb[4], cum[5];
  int lo4[4], len4[4];
  {
    const int lo_l = mel_lo[lane], len_l = mel_len[lane];
    int kmin = len_l > 0 ? lo_l : NBINS, kmax = len_l > 0 ? lo_l + len_l : 0;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      kmin = min(kmin, __shfl_xor(kmin, o, 64));
      kmax = max(kmax, __shfl_xor(kmax, o, 64));
    }
    cum[0] = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int b0 = __builtin_amdgcn_readlane(kmin, 16 * g) & ~3, b1 = min(__builtin_amdgcn_readlane(kmax, 16 * g), NBINS);
      kb[g] = b1 > b0 ? b0 : 0;
      cum[g + 1] = cum[g] + (b1 > b0 ? (b1 - b0 + 31) >> 5 : 0);
      lo4[g] = __shfl(lo_l, 16 * g + i16, 64);
      len4[g] = __shfl(len_l, 16 * g + i16, 64);
    }
  }
  constexpr int NO = 5;                        // octets of a slice whose weights are held in registers (18 octets -> 5 per wave for the reference's bands)
  const int total = cum[4], per = (total + 3) >> 2;
  const int o0 = w * per, o1 = min(total, o0 + per);
  // octet -> (group, first bin); octets past the slice repeat its last one with weight zero
  auto decode = [&](int oct, int& g, int& k0) {
    oct = max(min(oct, o1 - 1), 0);
    g = (oct >= cum[1]) + (oct >= cum[2]) + (oct >= cum[3]);
    const int first = g == 0 ? 0 : g == 1 ? cum[1] : g == 2 ? cum[2] : cum[3];
    const int kbg = g == 0 ? kb[0] : g == 1 ? kb[1] : g == 2 ? kb[2] : kb[3];
    k0 = kbg + 32 * (oct - first);
  };
  auto load_w = [&](int base, float (&a)[NO * 8]) {
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      int g, k0;
      decode(base + o, g, k0);
      const int lo = g == 0 ? lo4[0] : g == 1 ? lo4[1] : g == 2 ? lo4[2] : lo4[3];
      const int len = (base + o < o1) ? (g == 0 ? len4[0] : g == 1 ? len4[1] : g == 2 ? len4[2] : len4[3]) : 0;
      const float* Wg = melw + 16 * g + i16;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int q = k0 + 4 * j + kk - lo;
        const bool in = (unsigned)q < (unsigned)len;
        const float t = Wg[(in ? q : 0) * NMEL];
        a[o * 8 + j] = in ? t : 0.f;
      }
    }
  };

  // A group's geometry: its clip and first output frame, and the clip's own numbers (datasets.py:342-351 per sample) -- its first sample inside
  // the row (offsets), its length in samples (lengths), the source frame its output frame 0 is (starts).  Wave-uniform scalar loads, made
  // once per group (for the NEXT group at the top of a group's work, so that they have landed when the cross-group prefetch needs them);
  // NULL arrays = the launch-wide values.  Values that would reach outside the row are clamped; a clip too short to reflect-pad
  // (<= n_fft / 2 samples) has no frames: its output is all padding.  The list is ordered group-major (all clips' group 0, then all
  // clips' group 1, ...): the groups that lie in the padding of short clips then sit together at the end of the list instead of
  // striking some workgroups' walks more than others.
  struct Geo { int clip, tb, off, L, start, nfr; };
  auto geometry = [&](int grp, Geo& g) {
    if (grp >= n_groups) { g.clip = 0; g.tb = 0; g.off = 0; g.L = 0; g.start = 0; g.nfr = 0; return; }
    const int gi = grp / n_clips;
    g.clip = grp - gi * n_clips;
    g.tb = gi * FR_PER_BLOCK;
    if constexpr (PER_CLIP) {
      g.off = offsets ? min(max(offsets[g.clip], 0), L) : 0;
      g.L = lengths ? min(max(lengths[g.clip], 0), L - g.off) : L - g.off;
      g.start = starts ? max(starts[g.clip], 0) : start_all;
    } else {
      g.off = 0; g.L = L; g.start = start_all;
    }
    g.nfr = g.L > NFFT / 2 ? 1 + g.L / hop : 0;
  };
  // samples of one frame: lane n2 holds x[2 (64 n1 + n2)], x[2 (64 n1 + n2) + 1], n1 = 0..7 (reflect padding at the clip's ends).
  // (g, f): frame f of this wave in the group g describes
  auto fetch = [&](const Geo& g, int f, v2f (&sm)[8]) {
    const int to = g.tb + w * FR_PER_WAVE + f;
    if (to >= T_out || to + g.start >= g.nfr) return;
    const float* wv = wave + (int64_t)g.clip * wave_stride + g.off;
    const int Lc = g.L;
    const int base = (to + g.start) * hop - NFFT / 2;
    const bool interior = (base >= 0) && (base + NFFT <= Lc) && !(g.off & 1);   // (8-byte loads need an even first sample)
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
      const int n = 64 * n1 + lane;
      if (interior) sm[n1] = *reinterpret_cast<const v2f*>(wv + base + 2 * n);
      else sm[n1] = v2f{wv[reflect_idx(base + 2 * n, Lc)], wv[reflect_idx(base + 2 * n + 1, Lc)]};
    }
  };
  float a[NO * 8];
  load_w(o0, a);
  v2f smp[8] = {};
  Geo cur, nxt;
  geometry(blockIdx.x, cur);
  fetch(cur, 0, smp);
#if SA_LOGMEL_DBG
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  st[1] = __builtin_readcyclecounter();
#endif

  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x, cur = nxt) {
    geometry(grp + gridDim.x, nxt);
    const int clip = cur.clip;
    const int tb = cur.tb;                       // first OUTPUT frame of this group
    const int start = cur.start, n_frames = cur.nfr;
    if (tb + start >= n_frames) {
      // every frame of the group lies past the clip's end (a clip shorter than the crop): nothing to transform, the rows are the
      // normalised zero pad.  No LDS traffic, so no barrier; the next group's first frame is requested as after a full group.
      const int to = tb + i16;
      if (to < T_out) {
        float* orow = out + (int64_t)clip * out_stride + (int64_t)(16 * w + 4 * kk) * T_out + to;
#pragma unroll
        for (int r = 0; r < 4; ++r) orow[(int64_t)r * T_out] = pad_value;
      }
      fetch(nxt, 0, smp);
      continue;
    }
#pragma unroll 1
    for (int f = 0; f < FR_PER_WAVE; ++f) {
      const int to = tb + w * FR_PER_WAVE + f;   // output frame index
      const int t = to + start;                  // source frame index (dataset crop offset)
      const bool live = (to < T_out) && (t < n_frames);  // wave-uniform
      float* P = pw + (w * FR_PER_WAVE + f) * PS;
      // ---- window: lane n2 holds z[64*n1 + n2], n1 = 0..7
      v2f v[8];
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) v[n1] = smp[n1] * win[n1];
      if (f + 1 < FR_PER_WAVE) fetch(cur, f + 1, smp);       // the next frame's samples travel, into the registers just read, while this one is transformed
      if (live) {
        // ---- pass A: DFT8 over n1, twiddle W_512^{n2*k1} = W_1024^{2*n2*k1}
        dft8(v);
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1) v[k1] = cmul(v[k1], twa[k1]);
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) lds_st(x + k1 * XS + lane, v[k1]);
        __builtin_amdgcn_wave_barrier();
        // ---- pass B1: lane = (k1, b); DFT8 over a of y[k1][8a + b]; twiddle W_64^{b*c} = W_1024^{16*b*c}
#pragma unroll
        for (int a = 0; a < 8; ++a) v[a] = lds_ld(x + k2 * XS + 8 * a + c2);
        dft8(v);
#pragma unroll
        for (int c = 1; c < 8; ++c) v[c] = cmul(v[c], twb[c]);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 8; ++c) lds_st(x + k2 * XS + 9 * c + c2, v[c]);
        __builtin_amdgcn_wave_barrier();
        // ---- pass B2: lane = (k1, c); DFT8 over b of u[k1][c][b] -> Z[k1 + 8c + 64d] in register d
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) v[bb] = lds_ld(x + k2 * XS + 9 * c2 + bb);
        dft8(v);
        __builtin_amdgcn_wave_barrier();
        // ---- real-input split + power: X[k] = E[k] + W_1024^k O[k], E = (Z[k] + conj Z[512-k]) / 2, O = -i (Z[k] - conj Z[512-k]) / 2.
        // 2 X = s + (-i dd) W with s = Z[k] + conj(Zc), dd = Z[k] - conj(Zc); the factor 1/4 of |X|^2 is applied after the mel sum
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          const v2f own0 = v[(8 - d) & 7], own1 = v[7 - d];          // (loaded before the select: a select of the two array slots compiles to a 16-way register pick)
          const v2f own = {lane == 0 ? own0.x : own1.x, lane == 0 ? own0.y : own1.y};
          v2f zc;
          zc.x = __int_as_float(__builtin_amdgcn_ds_bpermute(partner_addr, __float_as_int(own.x)));
          zc.y = __int_as_float(__builtin_amdgcn_ds_bpermute(partner_addr, __float_as_int(own.y)));
          v2f s, dd, r0, x2;
          asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(s) : "v"(v[d]), "v"(zc));
          asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(dd) : "v"(v[d]), "v"(zc));
          asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]" : "=v"(r0) : "v"(dd), "v"(tws[d]), "v"(s));
          asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(x2) : "v"(dd), "v"(tws[d]), "v"(r0));
          P[k2 + 8 * c2 + 64 * d] = x2.x * x2.x + x2.y * x2.y;
          if (d == 0 && lane == 0) {  // Nyquist bin: X[512] = E[0] - O[0] = Re Z[0] - Im Z[0]; stored x 4 like the others
            const float nyq = 2.f * (v[0].x - v[0].y);
            P[NC] = nyq * nyq;
          }
        }
      } else {
        // (a row past the clip's end or the output's: defined, finite contents for the product below; its outputs are the pad value)
#pragma unroll
        for (int r = 0; r < 8; ++r) P[lane + 64 * r] = 0.f;
        if (lane == 0) P[NC] = 0.f;
      }
    }
#if SA_LOGMEL_DBG
    if (grp == blockIdx.x) st[2] = __builtin_readcyclecounter();
#endif
#if SA_LOGMEL_DBG
    if (grp == blockIdx.x) st[3] = __builtin_readcyclecounter();
#endif
    __syncthreads();
#if SA_LOGMEL_DBG
    if (grp == blockIdx.x) st[4] = __builtin_readcyclecounter();
#endif
    // A wave's octets are ordered by group: one running pair of accumulators, written to the group's partial-sum slot
    // [wave][group][lane] (over the exchange buffers, free since the barrier above) when the group changes; the slots of the groups the
    // slice does not meet get zeros.  (Four accumulators picked by a run-time group index compile to indexed register moves.)
    f32x4_t* part = reinterpret_cast<f32x4_t*>(&ex[0][0]) + w * 4 * 64 + lane;
    const float* Pb = pw + i16 * PS + kk;
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    int g_first, g_last, k0;
    decode(o0, g_first, k0);
    if (per <= NO) {
      // (B one octet ahead of its products: eight LDS reads in flight behind eight MFMAs, not forty registers of them)
      float b_cur[8], b_nxt[8];
      int g_cur = g_first, g_nxt;
      g_last = g_first;
#pragma unroll
      for (int j = 0; j < 8; ++j) b_cur[j] = Pb[k0 + 4 * j];
      f32x4_t c = zero4, c1 = zero4;                 // (two chains: a product does not wait for the one before it)
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        if (o + 1 < NO) {
          decode(o0 + o + 1, g_nxt, k0);
#pragma unroll
          for (int j = 0; j < 8; ++j) b_nxt[j] = Pb[k0 + 4 * j];
        }
        if (g_cur != g_last) {
          part[g_last * 64] = c + c1;
          c = zero4; c1 = zero4;
          g_last = g_cur;
        }
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[o * 8 + j], b_cur[j], c, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[o * 8 + j + 1], b_cur[j + 1], c1, 0, 0, 0);
        }
        if (o + 1 < NO) {
          g_cur = g_nxt;
#pragma unroll
          for (int j = 0; j < 8; ++j) b_cur[j] = b_nxt[j];
        }
      }
      part[g_last * 64] = c + c1;
    } else {
      // a filter bank whose slices do not fit the registers (wider groups than the reference's): one product at a time, weights from memory
      g_last = g_first;
      f32x4_t c = zero4;
#pragma unroll 1
      for (int oct = o0; oct < o1; ++oct) {
        int g;
        decode(oct, g, k0);
        if (g != g_last) {
          part[g_last * 64] = c;
          c = zero4;
          g_last = g;
        }
        const int lo = g == 0 ? lo4[0] : g == 1 ? lo4[1] : g == 2 ? lo4[2] : lo4[3];
        const int len = g == 0 ? len4[0] : g == 1 ? len4[1] : g == 2 ? len4[2] : len4[3];
#pragma unroll 1
        for (int j = 0; j < 8; ++j) {
          const int q = k0 + 4 * j + kk - lo;
          const bool in = (unsigned)q < (unsigned)len;
          const float t = melw[(in ? q : 0) * NMEL + 16 * g + i16];
          c = __builtin_amdgcn_mfma_f32_16x16x4f32(in ? t : 0.f, Pb[k0 + 4 * j], c, 0, 0, 0);
        }
      }
      part[g_last * 64] = c;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (g < g_first || g > g_last) part[g * 64] = zero4;
#if SA_LOGMEL_DBG
    if (grp == blockIdx.x) st[5] = __builtin_readcyclecounter();
#endif
    fetch(nxt, 0, smp);                          // the first frame of the workgroup's next group travels behind the two barriers below
    // group w's four partials, in wave order
    part = reinterpret_cast<f32x4_t*>(&ex[0][0]);
    __syncthreads();
    f32x4_t acc = part[w * 64 + lane];
#pragma unroll
    for (int ww = 1; ww < 4; ++ww) {
      const f32x4_t t = part[(ww * 4 + w) * 64 + lane];
      acc += t;
    }
    __syncthreads();                             // (the next group's frames write the exchange buffers)
#if SA_LOGMEL_DBG
    if (grp == blockIdx.x) st[6] = __builtin_readcyclecounter();
#endif
    const int to = tb + i16;
    if (to < T_out) {
      const bool live = (to + start) < n_frames;
      float* orow = out + (int64_t)clip * out_stride + (int64_t)(16 * w + 4 * kk) * T_out + to;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        orow[(int64_t)r * T_out] = live ? (__logf(0.25f * acc[r] + 1.1920929e-07f) - mean) * inv_std : pad_value;     // v_log_f32: 1 ulp
    }
#if SA_LOGMEL_DBG
    if (grp == blockIdx.x) st[7] = __builtin_readcyclecounter();
#endif
  }
#if SA_LOGMEL_DBG
  // (diagnostic build, scripts/diag/logmel_stamps.py: the phase durations of the workgroup's first group and its whole life overwrite that group's outputs)
  st[8] = __builtin_readcyclecounter();
  {
    const int gi0 = blockIdx.x / n_clips, clip = blockIdx.x - gi0 * n_clips, tb = gi0 * FR_PER_BLOCK;
    if (lane == 0 && tb + 8 <= T_out) {
      float* o = out + (int64_t)clip * out_stride + (int64_t)(16 * w) * T_out + tb;
      for (int i = 0; i < 7; ++i) o[i] = (float)(st[i + 1] - st[i]);
      o[7] = (float)(st[8] - st[0]);
    }
  }
#endif
}

}  // namespace

extern "C" int sa_logmel_fwd(const float* wave, int64_t wave_stride, int32_t n_clips, int32_t n_samples, const float* window,
                             const float* twiddle, const float* mel_weights, const int32_t* mel_lo, const int32_t* mel_len, float* out,
                             int64_t out_stride, int32_t T_out, int32_t start, const int32_t* starts, const int32_t* lengths,
                             const int32_t* offsets, float mean, float stdv, int32_t hop, void* stream) {
  SA_CHECK_ARG(wave && window && twiddle && mel_weights && mel_lo && mel_len && out, "sa_logmel_fwd: null pointer");
  SA_CHECK_ARG(n_clips > 0 && n_samples > NFFT / 2 && T_out > 0 && hop > 0 && hop % 2 == 0 && start >= 0, "sa_logmel_fwd: bad sizes");
  SA_CHECK_ARG(stdv != 0.f, "sa_logmel_fwd: std must be non-zero");
  SA_CHECK_ARG(((uintptr_t)wave & 7) == 0 && wave_stride % 2 == 0, "sa_logmel_fwd: waveform rows must be 8-byte aligned");
  const float pad_value = (0.f - mean) / stdv;  // right zero-pad happens BEFORE normalisation (datasets.py:346-354)
  const int groups_per_clip = (T_out + FR_PER_BLOCK - 1) / FR_PER_BLOCK;
  const int64_t n_groups = (int64_t)groups_per_clip * n_clips;
  SA_CHECK_ARG(n_groups < (1ll << 31), "sa_logmel_fwd: too many frames");
  static int slots = 0;                  // three workgroups per CU (LDS)
  if (!slots) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      sa_set_error("sa_logmel_fwd: cannot query the device");
      return 2;
    }
    slots = 3 * prop.multiProcessorCount;
  }
  dim3 grid((unsigned)(n_groups < slots ? n_groups : slots));
  auto kernel = (starts || lengths || offsets) ? logmel_kernel<true> : logmel_kernel<false>;
  hipLaunchKernelGGL(kernel, grid, dim3(256), 0, (hipStream_t)stream, wave, wave_stride, n_samples, window,
                     reinterpret_cast<const v2f*>(twiddle), mel_weights, mel_lo, mel_len, out, out_stride, T_out, start, starts, lengths,
                     offsets, mean, 1.0f / stdv, pad_value, hop, n_clips, (int)n_groups);
  SA_LAUNCH_CHECK("sa_logmel_fwd");
  return 0;
}
