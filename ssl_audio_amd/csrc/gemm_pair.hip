// 128 x 256 bf16 MFMA GEMM, TWO workgroups per CU ("mode P"; forward layout: both operands k-major).  Own translation unit.
#include "gemm_common.h"

namespace {

// =====================================================================================================
// Why a second geometry (VERDICT r4 #4c): the 256 x 256 kernels run ONE 8-wave workgroup per CU, so a tile's epilogue -- 128 exact-erf
// GELU pairs per lane (fc1 forward), or 256 KiB of fp32 residual in and 256 KiB out per CU (proj / fc2 forward) -- runs with the matrix
// pipe idle: fc1 forward 392 us against 257 us for the plain product, proj forward 138 against 67 (profiles/r05_bench_gemm_block.txt).
// Here a workgroup is FOUR waves on a 128 x 256 tile (wave w: all 128 rows x columns [64 w, 64 w + 64) = the 8 x 4 MFMA tiles and
// 128 accumulator registers of the 256^2 kernels' waves, so their compact epilogues apply unchanged), 72 KiB of LDS, one tile per
// workgroup: two workgroups share a CU, and the hardware starts a new one whenever one retires, so their phases drift apart and one
// workgroup's epilogue (VALU / HBM) runs under the other's main loop (MFMA).
// Main loop: the streaming kernel's (gemm_stream.hip) -- a three-stage ring of 32-deep K-steps, 24 KiB per stage (A 128 x 32, B 256 x 32),
// two stages always in flight, `vmcnt(6)` (six 1 KiB pieces per wave and stage), one barrier per K-step.
// Image of a k-major operand at 32-deep steps: 64-byte rows (four 16-byte chunks), chunk XOR (-(row >> 2)) & 3: the four 16-lane groups a
// ds_read_b128 is served in ({0-3, 12-15, 20-27}, ... MI355X_MICROARCH.md "LDS") then touch sixteen different bank quads each.
// The price: 48 KiB of operands per 4.2 MFLOP against 64 KiB per 8.4 (the feed the 256^2 tile halves), a barrier per 32 MFMAs.
constexpr int PR_A = 128 * 32 * 2;                  // 8 KiB
constexpr int PR_B = 256 * 32 * 2;                  // 16 KiB
constexpr int PR_STAGE = PR_A + PR_B;               // 24 KiB
constexpr int PR_NSTAGE = 3;
constexpr int PR_LDS = PR_NSTAGE * PR_STAGE;        // 72 KiB (the epilogue's 4 x 8 KiB scratch reuses it)

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_pair_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qq = nwg >> 3, rr = nwg & 7;
  const int lid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
  const int GM = p.gm;                                       // row panels per group: a group's tiles share its A panels and every B panel
  const int group_sz = GM * p.tiles_n;
  const int grp = lid / group_sz, within = lid - grp * group_sz;
  const int gm = min(GM, p.tiles_m - grp * GM);
  const int tm = grp * GM + within % gm, tn = within / gm;
  const int m0 = tm * 128, n0 = tn * 256;
  const int nk = p.K >> 5;                                   // (K is a multiple of 64 for k-major operands)

  // ---- requests: wave w sends A pieces 2 w, 2 w + 1 (rows 16 piece + [0, 16)) and B pieces 4 w .. 4 w + 3; lane = (row, chunk)
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.B, p.b_bytes);
  const int lrow = lane >> 2;
  const int lchunk = (lane & 3) ^ ((-(lrow >> 2)) & 3);
  uint32_t rq[6];
#pragma unroll
  for (int e = 0; e < 2; ++e) rq[e] = (uint32_t)(m0 + 16 * (2 * wave + e) + lrow) * (uint32_t)p.lda * 2u + (uint32_t)lchunk * 16u;
#pragma unroll
  for (int e = 0; e < 4; ++e) rq[2 + e] = (uint32_t)(n0 + 16 * (4 * wave + e) + lrow) * (uint32_t)p.ldb * 2u + (uint32_t)lchunk * 16u;
  auto request = [&](int slot, int kt, bool valid) {
    char* base = smem + slot * PR_STAGE;
    const uint32_t k_off = (uint32_t)kt * 64u;               // 32 k x 2 B
#pragma unroll
    for (int e = 0; e < 2; ++e) lds_dma16<true>(ra, base + (2 * wave + e) * 1024, valid ? rq[e] + k_off : 0xFFFFFFF0u);
#pragma unroll
    for (int e = 0; e < 4; ++e) lds_dma16<true>(rb, base + PR_A + (4 * wave + e) * 1024, valid ? rq[2 + e] + k_off : 0xFFFFFFF0u);
  };

  // ---- fragment address: row (lane & 15) of a 16-row block, chunk (lane >> 4) of its four
  const int frow = lane & 15;
  const int fbase = frow * 64 + (((lane >> 4) ^ ((-(frow >> 2)) & 3)) << 4);
  const int fb_base = PR_A + wave * 4096 + fbase;            // this wave's 64 B rows (columns of the output)

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  request(0, 0, true);
  request(1, 1, 1 < nk);
  int slot = 0, fill = 2;
  for (int t = 0; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __syncthreads();
    request(fill, t + 2, t + 2 < nk);
    const char* st = smem + slot * PR_STAGE;
    bf16x8 fa[8], fb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(st + fb_base + j * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(st + fbase + i * 1024);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    slot = slot == PR_NSTAGE - 1 ? 0 : slot + 1;
    fill = fill == PR_NSTAGE - 1 ? 0 : fill + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                           // every wave is past its reads: the ring becomes the epilogue scratch

  char* wl = smem + wave * 8192;
  const int nb = n0 + wave * 64;
  if constexpr (EPI == 4 || EPI == 5) {
    uint4 auxv[16];
    {
      typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
      const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.aux_in, nb < p.N ? (uint32_t)((((int64_t)p.M - 1) * p.ldaux + p.N) * 2) : 0u);
      const uint32_t ld2 = (uint32_t)p.ldaux * 2u;
      const uint32_t voff = (uint32_t)(lane >> 3) * ld2 + (uint32_t)(lane & 7) * 16u;
      const uint32_t soff = (uint32_t)m0 * ld2 + (uint32_t)nb * 2u;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, voff, soff + (uint32_t)q * 8u * ld2, 0);
        auxv[q] = __builtin_bit_cast(uint4, v);
      }
    }
    wave_epilogue_compact<true, EPI>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[0]), m0, nb, wl, lane, nullptr, nullptr, &auxv[0]);
    wave_epilogue_compact<true, EPI>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[4]), m0 + 64, nb, wl, lane, nullptr, nullptr, &auxv[8]);
  } else {
    float4 bias4[4];
    const bool has_bias = p.bias != nullptr && nb < p.N;
#pragma unroll
    for (int j = 0; j < 4; ++j) bias4[j] = has_bias ? *reinterpret_cast<const float4*>(p.bias + nb + j * 16 + 4 * (lane >> 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
    wave_epilogue_compact<true, EPI>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[0]), m0, nb, wl, lane, nullptr, bias4);
    wave_epilogue_compact<true, EPI>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[4]), m0 + 64, nb, wl, lane, nullptr, bias4);
  }
}

template <int EPI>
int launch_pair_one(GemmParams p, hipStream_t stream) {
  static bool cfg = false;
  if (!cfg) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pair_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, PR_LDS) != hipSuccess) {
      sa_set_error("sa_gemm_bf16: 72 KiB of LDS per workgroup refused");
      return 2;
    }
    cfg = true;
  }
  p.tiles_m = (p.M + 127) / 128;
  p.tiles_n = (p.N + 255) / 256;
  hipLaunchKernelGGL(gemm_pair_kernel<EPI>, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(256), PR_LDS, stream, p);
  SA_LAUNCH_CHECK("sa_gemm_bf16(128 x 256 pair)");
  return 0;
}

}  // namespace

// both operands k-major, split_k == 1, one of the compact epilogues; -1 when the kind is not covered
int sagemm::launch_pair(GemmParams p, hipStream_t stream) {
  switch (p.epi_kind) {
    case 1: return launch_pair_one<1>(p, stream);
    case 3: return launch_pair_one<3>(p, stream);
    case 5: return launch_pair_one<5>(p, stream);
    case 6: return launch_pair_one<6>(p, stream);
    case 8: return launch_pair_one<8>(p, stream);
    default: return -1;
  }
}
