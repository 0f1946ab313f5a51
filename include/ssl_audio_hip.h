/* ssl_audio_hip.h -- C ABI of libssl_audio_hip.so (gfx950 / MI355X only).
 *
 * The drop-in boundary of the Audio Barlow Twins pre-training hot path.  The reference
 * (jonahanton/SSL_audio) has no FFI: its boundary is the Python import surface consumed by
 * main_bt_byol.py:20-25.  Every entry point below therefore cites the reference
 * *op site* it replaces (file:line of the PyTorch call that does the work there).  The Python host side
 * in ssl_audio_amd/ mirrors the reference's classes and calls these through ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); all work is enqueued, never synchronised;
 *   - row-major everywhere; `ld*` are leading dimensions in ELEMENTS;
 *   - bf16 tensors are raw 16-bit storage (`void*`), fp32 tensors are `float*`;
 *   - return 0 on success; non-zero on failure with a message in sa_last_error().
 *   - no entry point allocates, frees or synchronises (safe to capture into a hipGraph).
 */
#ifndef SSL_AUDIO_HIP_H
#define SSL_AUDIO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* sa_last_error(void);
int sa_abi_version(void);
/* fills name (e.g. "gfx950:sramecc+:xnack-") and CU count of the current device; host call */
int sa_device_info(char* name_host, int name_len, int* cu_count_host);

/* ------------------------------------------------------------------ GEMM (bf16 MFMA, fp32 accumulate)
 * D[M,N] = epilogue( alpha * A . B ),  A: M x K, B: K x N in the maths; storage selected per operand:
 *   a_kmajor=1: A stored [M][K] (k contiguous)       a_kmajor=0: A stored [K][M]
 *   b_kmajor=1: B stored [N][K] (nn.Linear weight)   b_kmajor=0: B stored [K][N]
 * Replaces F.linear / nn.Linear fwd+bwd at models/mae.py:128,138 (qkv, proj), timm Mlp fc1/fc2
 * (models/mae.py:155,162), PatchEmbed conv-as-GEMM (models/mae.py:42), projector/predictor Linear
 * (model.py:19,23,41,44), MAE decoder_embed / decoder_pred (models/mae.py:413,430).
 * Epilogue order: v = alpha*acc; v += bias[n]; act; v += residual[row_r][n]; store.
 */
typedef struct SaGemmArgs {
  const void* A; int64_t lda; int32_t a_kmajor;
  const void* B; int64_t ldb; int32_t b_kmajor;
  int32_t M, N, K;
  float alpha;
  const float* bias;            /* [N] fp32 or NULL */
  int32_t act;                  /* 0 none | 1 GELU(erf); pre-activation copied to aux_out if non-NULL
                                   | 2 multiply by GELU'(aux_in[m][n]) (backward through fc1's activation, aux = pre-activation)
                                   | 3 GELU(erf); aux_out receives GELU'(pre-activation) -- the exponential is shared, so the
                                       derivative is free in the forward and the backward needs no transcendental
                                   | 4 multiply by aux_in[m][n] (backward partner of 3) */
  const void* aux_in;           /* bf16 [M][ldaux] */
  void* aux_out;                /* bf16 [M][ldaux] */
  int64_t ldaux;
  const float* residual;        /* fp32 or NULL; row = res_mod > 0 ? m % res_mod : m */
  int64_t ldr; int32_t res_mod;
  float* out_f32; int64_t ldo_f32;   /* either or both outputs may be given */
  void* out_bf16; int64_t ldo_bf16;
  int32_t row_group;            /* > 0: output row = m + m / row_group + 1 (patch tokens skip each clip's CLS row) */
  int32_t split_k;              /* >= 1.  > 1: K is split over workgroups and partial tiles are atomically
                                   ADDED into out_f32 (caller zeroes / owns accumulation); only alpha is applied */
  int32_t accumulate;           /* split_k == 1 only: out_f32 += v instead of = v */
  int32_t tile256;              /* split_k > 1 only: 1 = the 256 x 256 tile (one workgroup per CU) instead of 128 x 128; 2 = the 192 x 192
                                   streaming kernel (both operands k-strided; three K-steps in LDS, for outputs a few hundred wide) */
  float* colsum_out;            /* optional [N]: += column sums of the fp32 epilogue result (the bias gradient of the Linear that
                                   produced this GEMM's A operand's gradient, e.g. fc1's bias from the fc2 dgrad: utils of
                                   models/mae.py:155 backward).  Needs split_k == 1, N % 64 == 0 and colsum_ws. */
  float* colsum_ws;             /* scratch of sa_gemm_colsum_workspace_bytes(M, N) bytes */
  float* splitk_ws;             /* optional, split_k > 1 only: scratch of sa_gemm_splitk_workspace_bytes(M, N, split_k) bytes.  When
                                   given, every K slice STORES its partial tile there and a second launch adds the slices into
                                   out_f32 in slice order: the result is bit-reproducible from run to run (the atomic path is
                                   exact only up to the fp32 rounding of an arbitrary summation order).  Served by the two default
                                   split-K kernels (128 x 128 and tile256) */
  float* asum_out;              /* optional, the streaming split-K kernels only (both operands k-strided, split_k > 1, tile256 = 2 or 1; also per
                                   product of sa_gemm_wgrad_group): asum_out[m] += sum_k A[k][m] for m < M outside [asum_skip_lo, asum_skip_hi)
                                   -- the BIAS gradient of the Linear whose weight gradient this product is (A = dY), taken from the dY tiles the
                                   kernel has in LDS anyway instead of a second pass over dY (sa_colsum_bf16).  The skipped rows: the k third
                                   of a packed qkv gradient, whose bias is fixed at zero (models/mae.py:125-128).  Other kernels refuse it. */
  float* asum_ws;               /* scratch of split_k * M floats; required with splitk_ws (slice sums added in slice order), unused otherwise */
  int32_t asum_skip_lo, asum_skip_hi;
} SaGemmArgs;
int sa_gemm_bf16(const SaGemmArgs* args_host, void* stream);
int64_t sa_gemm_splitk_workspace_bytes(int32_t M, int32_t N, int32_t split_k);
int64_t sa_gemm_colsum_workspace_bytes(int32_t M, int32_t N);
/* n <= 8 weight gradients over the SAME reduction in one pair of launches: out_f32_i[M_i, N_i] += alpha * A_i^T B_i for i < n (both
 * operands k-strided, the same K = number of rows, the same alpha and split_k > 1; args[0].tile256 picks the streaming kernel for
 * the whole group: 2 (or 0) = 192 x 192 tiles for narrow outputs, 1 = 256 x 256 tiles for wide ones).  The four weight gradients of a transformer block (models/mae.py:106-129,149-163: qkv, proj, fc1, fc2 at backward) are
 * such a group.  Why one launch: split-K costs one fp32 partial tile per WORKGROUP whatever the split, and a launch needs a
 * workgroup per CU to stream at the HBM rate -- four launches write and re-read four chips' worth of partials, a group one.
 * Every args[i] as for sa_gemm_bf16 with split_k > 1 (only A, B, lda, ldb, M, N, K, alpha, out_f32, ldo_f32, split_k, splitk_ws are
 * read); splitk_ws set in all of them (deterministic: slices added in slice order by a second launch) or in none (fp32 atomics). */
int sa_gemm_wgrad_group(const SaGemmArgs* args_host, int32_t n, void* stream);
/* Host policy knob (no reference counterpart): CUs the persistent GEMM grids may occupy, 0 = all.  Data-parallel runs reserve
 * the CUs RCCL's collective kernels hold during an all-reduce, so an overlapped GEMM does not spill into a second wave. */
int sa_set_cu_budget(int32_t cus);
/* Host policy knob (no reference counterpart; DDP / NCCL hide the problem it solves): dynamic tile hand-out in the persistent GEMM
 * kernels.  on != 0: a workgroup starts on tile blockIdx and draws every later tile from a per-launch ticket counter (one memset node
 * + the kernel, still capturable), so a workgroup that got its CU late -- RCCL's all-reduce kernels hold CUs while they run -- simply
 * draws fewer tiles instead of finishing a whole static share late.  Results do not depend on the order tiles are drawn in.
 * Costs 1.5-2 % of a launch on an undisturbed chip (one returning atomic per tile on the critical wave), so the default is off
 * (SA_GEMM_DYNAMIC=1 in the environment turns it on) and the data-parallel trainer switches it on when collectives are active. */
int sa_set_dynamic_tiles(int32_t on);

/* fp32 -> bf16 cast of a flat buffer (weights once per step, activations where needed) */
int sa_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* bf16 -> fp32 of a flat buffer (ABI v6): the way back from the bf16 gradient buckets of the data-parallel exchange -- a gradient range is cast
 * to bf16, SUM all-reduced over the ranks at half the bytes, and widened back into the fp32 gradient buffer (train.GradSync, grad_dtype
 * "bf16"; the reference's DDP all-reduces fp32 or, under autocast, fp16 gradient buckets: utils/utils.py:410-417). */
int sa_cast_bf16_to_f32(const void* src_bf16, float* dst, int64_t n, void* stream);
/* dst[c][r] = src[r][c], bf16 [R][C] -> [C][R] (R, C multiples of 8).  The engine keeps a transposed bf16 copy of every Linear weight so that
 * the data gradients dX = dY W run in the forward (k-contiguous x k-contiguous) operand layout, measured 6-16 % faster than reading W
 * k-strided; refreshed once per optimiser step. */
int sa_transpose_bf16(const void* src_bf16, int32_t R, int32_t C, void* dst_bf16, void* stream);
/* the same for n matrices in ONE launch (all Linear weights after an optimiser step; ABI v5).  desc_dev: device array [n][4] of int64
 * {src pointer, dst pointer, R | (int64)C << 32, index of the matrix' first 64 x 64 tile among all tiles}, matrices in tile order;
 * n_tiles = sum over the matrices of ceil(R / 64) * ceil(C / 64). */
int sa_transpose_bf16_batch(const int64_t* desc_dev, int32_t n_matrices, int32_t n_tiles, void* stream);
/* column sums of a bf16 [M][N] matrix into fp32 out[N] (bias gradients of nn.Linear, models/mae.py:125-131,155 backward);
 * accumulate != 0 adds.  ws: scratch of n_ranges x sa_colsum_workspace_bytes(M, N) bytes -- row slabs store their partial sums there
 * and a second launch adds them in slab order (bit-reproducible; ABI v5).  ws == NULL: one float atomic per column per block instead
 * (exact only up to the fp32 rounding of an arbitrary summation order).  n_ranges > 1: the same for the column ranges
 * [z * range_stride, z * range_stride + N) of x, z < n_ranges, into out + z * range_stride, in the same launches (q and v bias
 * gradients out of the packed dqkv: models/mae.py:125-128). */
int sa_colsum_bf16(const void* x, int64_t ld, int32_t M, int32_t N, float* out, int32_t accumulate, float* ws, int32_t n_ranges,
                   int64_t range_stride, void* stream);
int64_t sa_colsum_workspace_bytes(int32_t M, int32_t N);


/* ------------------------------------------------------------------ LayerNorm (fp32 stream -> bf16/fp32)
 * Replaces nn.LayerNorm(eps=1e-6) at models/mae.py:149,153,208,227 (fwd + bwd).  One wave per row, D <= 2048.
 * bwd: dx = dres + LN'(dy); dgamma/dbeta/dxsum are ADDED into (caller zeroes or accumulates) by a deterministic
 * two-stage reduction through `workspace` (sa_layernorm_bwd_workspace_bytes(M, D) bytes, required iff any of the three
 * is non-NULL; contents are scratch). */
int sa_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, void* y_bf16, float* y_f32,
                     int64_t ldy, float* mean, float* rstd, int32_t M, int32_t D, float eps, void* stream);
int sa_layernorm_bwd(const void* dy, int32_t dy_is_bf16, int64_t lddy, const float* x, int64_t ldx, const float* gamma,
                     const float* mean, const float* rstd, const float* dres, int64_t lddres, float* dx_f32, void* dx_bf16,
                     int64_t lddx, float* dgamma, float* dbeta, float* dxsum /* += column sums of dx, or NULL */, float* workspace,
                     int32_t M, int32_t D, void* stream);
int64_t sa_layernorm_bwd_workspace_bytes(int32_t M, int32_t D);

/* ------------------------------------------------------------------ fused attention (head_dim 64, N <= 512)
 * Replaces models/mae.py:130-138 (reshape/permute, q k^T * scale, softmax, attn v, transpose/reshape).
 * qkv: bf16 [rows][ld], per row [q(C) | k(C) | v(C)], C = H*64, rows = sequences * N.  out: bf16 [rows][ldo].
 * lse: fp32 [sequences*H][N] log-sum-exp of the scaled scores (saved for backward; may be NULL in fwd).
 * bwd writes dqkv in the same packed layout as qkv (every element of the [rows][3C] block is written).
 * n_query (0 = all N): only the first n_query tokens of each sequence act as queries (forward writes only those rows
 * of `out`; backward treats dout of the other rows as zero) -- the last encoder block needs the CLS row only, because
 * MaskedAutoencoderViT.forward returns x[:, 0] (models/mae.py:463). */
int sa_attention_fwd(const void* qkv, int64_t rows, int64_t ld, int32_t C, int32_t H, int32_t N, int32_t n_query, float scale,
                     void* out, int64_t ldo, float* lse, void* stream);
int sa_attention_bwd(const void* qkv, int64_t rows, int64_t ld, int32_t C, int32_t H, int32_t N, int32_t n_query, float scale,
                     const void* out, const void* dout, int64_t ldo, const float* lse, void* dqkv, void* stream);

/* ------------------------------------------------------------------ BatchNorm1d pieces (train mode, fp32)
 * Replaces nn.BatchNorm1d at model.py:20,42 (projector / predictor, affine + ReLU) and utils/loss.py:13,17
 * (loss, affine=False).  Split into statistics -> [host-side cross-rank combine] -> apply so one code path
 * serves 1 GPU and the global-batch-exact data-parallel mode.
 *   colstats : mean[c], m2[c] = sum_b (x - mean)^2 over the LOCAL rows
 *   apply    : y = (x - mean) * rstd [* gamma + beta] [ReLU] -> fp32 and/or bf16
 *   bwd_stats: s1[c] = sum_b g, s2[c] = sum_b g*xhat, g = dy * [ReLU mask]  (dbeta = s1, dgamma = s2)
 *   bwd_apply: dx = gamma * rstd * (g - s1*inv_n - xhat * s2*inv_n), s1/s2 summed over ranks, inv_n = 1/global rows */
int sa_bn_colstats(const float* x, int64_t ld, int32_t B, int32_t C, float* mean, float* m2, void* stream);
/* finalize: combine W ranks' (mean, m2) rows ([W][2][C], equal rows_per_rank; rank r's block starts at stats + r * rank_stride,
 * rank_stride = 0 means 2*C -- a larger stride reads one tensor's statistics out of a packed all-gather of several) -> global
 * mean, rstd = rsqrt(var+eps); optionally updates running_mean / running_var (momentum, unbiased variance) as nn.BatchNorm1d
 * does in train mode */
int sa_bn_finalize(const float* stats, int64_t rank_stride, int32_t W, int32_t rows_per_rank, int32_t C, float eps, float momentum,
                   float* mean, float* rstd, float* running_mean, float* running_var, void* stream);
int sa_bn_apply(const float* x, int64_t ld, int32_t B, int32_t C, const float* mean, const float* rstd, const float* gamma,
                const float* beta, int32_t relu, float* y_f32, void* y_bf16, int64_t ldy, void* stream);
int sa_bn_bwd_stats(const void* dy, int32_t dy_is_bf16, int64_t lddy, const float* x, int64_t ld, int32_t B, int32_t C,
                    const float* mean, const float* rstd, const float* gamma, const float* beta, int32_t relu, float* s1,
                    float* s2, void* stream);
int sa_bn_bwd_apply(const void* dy, int32_t dy_is_bf16, int64_t lddy, const float* x, int64_t ld, int32_t B, int32_t C,
                    const float* mean, const float* rstd, const float* gamma, const float* beta, int32_t relu, const float* s1,
                    const float* s2, float inv_n, const float* out_scale /* device scalar or NULL */, float* dx_f32, void* dx_bf16,
                    int64_t lddx, void* stream);

/* ------------------------------------------------------------------ Barlow Twins loss pieces (fp32, exact-fp32 MFMA)
 * Replaces utils/loss.py:15-30.  sa_matmul_f32: C[m][n] = alpha * sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn]
 * (c = z1n^T z2n / B, dz1n = z2n G^T / B, dz2n = z1n G / B).  sa_bt_loss_grad: loss (1 float, overwritten) and
 * G = dL/dc from the (already all-reduced) cross-correlation matrix c [D][D]; G may be NULL.  ws: sa_bt_loss_workspace_bytes() of
 * scratch for the per-block partial sums, added in block order by a second launch (no float atomics).  ABI v6: the scratch is the
 * CALLER's (until v5 a device global of the library, shared by every stream and thread of a process): callers that launch from several
 * streams or threads hand in one buffer per stream. */
int sa_matmul_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C, int64_t ldc,
                  int32_t M, int32_t N, int32_t K, float alpha, void* stream);
int64_t sa_bt_loss_workspace_bytes(void);
int sa_bt_loss_grad(const float* c, int32_t D, float alpha, float lambda, int32_t hsic, float* loss, float* G, float* ws, void* stream);
/* The loss term fused around its three collectives (ABI v5; SURVEY.md section 2.3 "K7", utils/loss.py:15-30): five launches instead of
 * the fifteen of the piecewise schedule above, same arithmetic.
 *   sa_bt_stats2        stats [2][2][D] = per view (mean, M2) of the local rows of z1, z2 [B][D] (row stride ld)       -> all-gather
 *   sa_bt_corr          all_stats [W][2][2][D]: Chan combine over the W ranks (B rows each) -> mean / rstd [2][D], running buffers
 *                       updated with view 1 then view 2 (bn(z1), bn(z2): utils/loss.py:17), z1n / z2n [B][D] (saved for the backward)
 *                       and c [D][D] = inv_n * z1n^T z2n (the local partial of utils/loss.py:19)                     -> all-reduce
 *   sa_bt_loss_grad     (above)
 *   sa_bt_bwd_products  dzn [2][B][D] = inv_n * (z2n G^T | z1n G) and sums [2][2][D] = per view (sum_b dzn, sum_b dzn * zn) -> all-reduce
 *   sa_bt_bwd_apply     dz_v = scale * rstd_v * (dzn_v - s1_v * inv_n - zn_v * s2_v * inv_n); out_scale: device scalar (dloss) or NULL */
int sa_bt_stats2(const float* z1, const float* z2, int64_t ld, int32_t B, int32_t D, float* stats, void* stream);
int sa_bt_corr(const float* z1, const float* z2, int64_t ld, int32_t B, int32_t D, const float* all_stats, int32_t W, float eps,
               float momentum, float inv_n, float* mean, float* rstd, float* running_mean, float* running_var, float* z1n, float* z2n,
               float* c, void* stream);
int sa_bt_bwd_products(const float* z1n, const float* z2n, int32_t B, int32_t D, const float* G, float inv_n, float* dzn, float* sums,
                       void* stream);
int sa_bt_bwd_apply(const float* z1n, const float* z2n, int32_t B, int32_t D, const float* rstd, const float* dzn, const float* sums,
                    float inv_n, const float* out_scale, float* dz1, float* dz2, void* stream);

/* ------------------------------------------------------------------ optimiser / EMA on flat fp32 buffers
 * AdamW: torch.optim.AdamW semantics (main_bt_byol.py:312-313); p_bf16 (optional) receives the bf16 copy the
 * next step's GEMMs read.  EMA: utils/utils.py:317-331 (target = beta*target + (1-beta)*online). */
int sa_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t step, float grad_scale, void* p_bf16, void* stream);
int sa_ema_update(float* target, const float* online, int64_t n, float beta, void* stream);
/* The same two updates with everything that changes from step to step read from DEVICE memory, so that a launch captured in a HIP
 * graph replays correctly: hyper3 = {lr (the value utils.adjust_learning_rate set, utils/utils.py:47-65), 1 / (1 - beta1^t),
 * 1 / sqrt(1 - beta2^t)}; skip_flag (optional): when *skip_flag != 0 the launch changes nothing -- the counterpart of
 * main_bt_byol.py:116-118 (a non-finite loss stops training BEFORE optimizer.step()) for a host that reads the flag lazily. */
int sa_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper3, float beta1, float beta2, float eps,
                      float weight_decay, float grad_scale, void* p_bf16, const int32_t* skip_flag, void* stream);
int sa_ema_update_gated(float* target, const float* online, int64_t n, float beta, const int32_t* skip_flag, void* stream);
/* LARS on one parameter tensor (utils/utils.py:150-189, selected by `--optimizer LARS`, main_bt_byol.py:326-345):
 * dp = g + weight_decay * p (pass 0 where the reference's weight_decay_filter excludes the tensor);
 * lars_adaptation != 0: dp *= eta * |p| / |dp| (1 when either norm is 0); mu = momentum * mu + dp; p -= lr * mu.
 * scratch2: 2 floats (squared norms), required with lars_adaptation; p_bf16 (optional): refreshed bf16 copy. */
int sa_lars_step(float* p, const float* g, float* mu, int64_t n, float lr, float weight_decay, float momentum, float eta,
                 int32_t lars_adaptation, float* scratch2, void* p_bf16, void* stream);
/* y += a * x on flat fp32 buffers (gradient accumulation of small vectors) */
int sa_axpy_f32(float* y, const float* x, int64_t n, float a, void* stream);
/* flag[0] += number of non-finite values in x[0..n): the device-side half of the reference's per-step `math.isfinite(loss.item())`
 * guard (main_bt_byol.py:116-118).  The host reads the flag every k steps instead of synchronising every step. */
int sa_count_nonfinite(const float* x, int64_t n, int32_t* flag, void* stream);

/* ------------------------------------------------------------------ log-mel frontend
 * Replaces torchaudio MelSpectrogram + log at datasets.py:39-48,115 and crop/pad/normalise at datasets.py:342-354.
 * wave [n_clips][wave_stride] fp32 -> out [n_clips][64][T_out] fp32 (clip stride out_stride).  Output frame t is
 * source frame t + start; frames past the clip are the normalised zero pad.
 * Per-clip geometry (ABI v6; each array optional, device memory, [n_clips] int32; NULL = the launch-wide value), so that ONE launch does
 * what the reference's Dataset.__getitem__ does per sample (a random crop start and a right pad of its own for every clip):
 *   starts[b]   source frame of clip b's output frame 0 -- `start = np.random.randint(l - crop_frames)` of datasets.py:342-345 (replaces `start`);
 *   lengths[b]  clip b's length in SAMPLES (replaces n_samples, which stays the row bound): it has 1 + lengths[b] / hop frames, the
 *               reflect padding of the last frames mirrors at ITS end, and output frames past its last frame are the normalised zero pad
 *               (datasets.py:346-351).  A clip of <= 512 samples cannot be reflect-padded: all of its output is the pad;
 *   offsets[b]  first sample of clip b inside its row (default 0): the waveform-level crop `wav[start:start + unit_length]` of
 *               datasets.py:108-112 without a copy (the frames are those of the cropped waveform, reflecting at the crop's ends).
 *   The kernel clamps offsets[b] to [0, n_samples] and lengths[b] to [0, n_samples - offsets[b]] (no read leaves the row); 16-frame groups that
 *   lie wholly in a clip's padding are written without any transform work.
 * Tables (host-built, uploaded once):
 * window [1024]; twiddle [1024][2] = (cos, -sin)(2 pi k / 1024); mel_lo/mel_len [64] = first power bin and number of bins of each
 * band (mel_lo + mel_len <= 513); mel_weights [maxlen][64] with maxlen = max(mel_len), entry [q][m] = weight of bin mel_lo[m] + q in band m (entries with
 * q >= mel_len[m] are not read).  The 513 -> 64 projection runs on the fp32 MFMA per group of 16 bands over the bins the group's bands cover,
 * so a non-finite power bin of a frame reaches every band of the groups whose range contains it (0 x NaN), not only the bands that weight it. */
int sa_logmel_fwd(const float* wave, int64_t wave_stride, int32_t n_clips, int32_t n_samples, const float* window,
                  const float* twiddle, const float* mel_weights, const int32_t* mel_lo, const int32_t* mel_len, float* out,
                  int64_t out_stride, int32_t T_out, int32_t start, const int32_t* starts, const int32_t* lengths, const int32_t* offsets,
                  float mean, float stdv, int32_t hop, void* stream);

/* ------------------------------------------------------------------ augmentation stack (one fused launch per batch of views)
 * Replaces MixupBYOLA.forward / log_mixup_exp (augmentations.py:81-85,103-117), RandomResizeCrop.forward
 * (augmentations.py:40-55) and RandomLinearFader.forward (augmentations.py:69-74) with explicit parameters.
 * lms: base of the clip store (ring of normalised log-mels, clip k at lms + k*clip_stride, [F_in][T_in]);
 * src_slot[v]: clip of view v; mix_slot[v]: bank clip to mix in, or -1 (NULL = no mixing);
 * params [n_views][8] fp32 = (alpha, i, j, h, w, head, tail, lambda); out [n_views][F_out][T_out].
 * max_w_ratio >= max crop width / (T_out - 1) bounds the LDS source tile.
 * noise (ABI v5, optional): standard-normal draws [n_views][F_in][T_in]; when given, MixGaussianNoise (augmentations.py:132-141,
 * `--Gnoise`, utils/transforms.py:21-22) is applied to the mixed source pixel before the crop:
 * log((1 - lambda) exp(v) + exp(lambda n) + eps), lambda = params[view][7]. */
int sa_augment_views(const float* lms, int64_t clip_stride, const int32_t* src_slot, const int32_t* mix_slot, const float* params,
                     float* out, int32_t n_views, int32_t F_in, int32_t T_in, int32_t canvas_h, int32_t canvas_w, int32_t F_out,
                     int32_t T_out, float max_w_ratio, int32_t do_fade, const float* noise, void* stream);
/* NormalizeBatch (augmentations.py:229-232) over n contiguous floats; workspace2 = 2 doubles; shift ~ mean guess.
 * stat_div = 1 is NormalizeBatch and the HEAR scene normalisation (hear/sample/vit.py:97-100); stat_div = number of frames
 * reproduces hear/utils.py:36-53, which divides BOTH statistics by len(melspec) before (x - mean) / std (hear/sample/vit.py:203-205) */
int sa_normalize_batch(const float* x, float* y, int64_t n, float shift, double* workspace2, float eps, float stat_div, void* stream);
/* MixGaussianNoise (augmentations.py:125-141): out = log((1 - lambd) * exp(x) + exp(lambd * normal) + eps); `normal` holds the
 * caller's N(0,1) draws (torch.normal(0, lambd) == lambd * N(0,1)). */
int sa_mix_gaussian_noise(const float* x, const float* normal, int64_t n, float lambd, float eps, float* out, void* stream);
/* RunningNorm (augmentations.py:144-214, axis [1, 2]): state = {running mean, running second moment} per channel, updated while
 * update != 0 with the reference's increment (divided by the number of samples seen so far, n_seen; first sample initialises),
 * then out = (x - mean) / max(sqrt(second moment), eps). */
int sa_running_norm(const float* x, int32_t channels, int64_t per_channel, float* state, int32_t n_seen, int32_t update, float eps,
                    float* out, void* stream);
/* [S][1][F][T] fp32 -> bf16 patch rows [S*(F/ph)*(T/pw)][ph*pw] for the patch-embed GEMM (models/mae.py:42) */
int sa_patchify_bf16(const float* img, void* out, int32_t S, int32_t F, int32_t T, int32_t ph, int32_t pw, void* stream);

/* ------------------------------------------------------------------ token bookkeeping (fp32 residual stream [S][N][d])
 * fill_cls: x[s][0][:] = cls + pos0 (models/mae.py:361-363); cls_grad: dcls += sum_s dx[s][0][:];
 * gather_rows / scatter_add_rows: random-masking keep-gather and its adjoint, decoder un-shuffle
 * (models/mae.py:335-337, 416-418); idx is [S][n_idx] int32, rows are offset by *_row0 inside each sequence. */
int sa_fill_cls(float* x, int32_t S, int64_t seq_stride, int32_t d, const float* cls, const float* pos0, void* stream);
int sa_cls_grad(const float* dx, int32_t S, int64_t seq_stride, int32_t d, float* dcls, void* stream);
int sa_gather_rows(const float* src, int64_t src_seq_stride, int32_t src_row0, const int32_t* idx, int32_t n_idx, float* dst,
                   int64_t dst_seq_stride, int32_t dst_row0, int32_t S, int32_t d, void* stream);
int sa_scatter_add_rows(const float* src, int64_t src_seq_stride, int32_t src_row0, const int32_t* idx, int32_t n_idx, float* dst,
                        int64_t dst_seq_stride, int32_t dst_row0, int32_t S, int32_t d, void* stream);

/* mean pooling of the patch tokens (`x[:, 1:].mean(dim=1)`, models/mae.py:461-462) and its adjoint (row 0 gets zero) */
int sa_mean_tokens_fwd(const float* y, int32_t S, int32_t N, int32_t d, float* out, void* stream);
int sa_mean_tokens_bwd(const float* dout, int32_t S, int32_t N, int32_t d, float* dy, void* stream);
/* eval path (encode_vit, utils/utils.py:278-314): out[s][g][:] (+)= scale * sum_{t<count} y[s][row0 + g*group_stride + t][:]
 * (mean of stacked CLS embeddings: G=1; mean over time of the patch tokens per frequency row: G = grid rows, stride = grid cols) */
int sa_token_group_sum(const float* y, int32_t S, int32_t N, int32_t d, int32_t row0, int32_t G, int32_t group_stride, int32_t count,
                       float scale, int32_t accumulate, float* out, void* stream);

/* MAE decoder input (forward_decoder, models/mae.py:413-420): mask tokens appended, un-shuffled by ids_restore, decoder
 * positional table added.  x [B][1+keep][d], out [B][1+L][d]; bwd WRITES dx (kept rows are a permutation) and ADDS the
 * masked rows into dmask_token (d <= 2048) through ws: sa_mae_unshuffle_bwd_workspace_bytes() of caller-owned, 16-byte aligned scratch for
 * the row groups' partial sums, added in group order (ABI v6; required when dmask_token is given; one buffer per stream). */
int64_t sa_mae_unshuffle_bwd_workspace_bytes(void);
int sa_mae_unshuffle_fwd(const float* x, int32_t keep, const float* mask_token, const float* pos, const int32_t* ids_restore, int32_t B,
                         int32_t L, int32_t d, float* out, void* stream);
int sa_mae_unshuffle_bwd(const float* dout, int32_t keep, const int32_t* ids_restore, int32_t B, int32_t L, int32_t d, float* dx,
                         float* dmask_token, float* ws, void* stream);

/* MAE reconstruction loss (forward_loss + patchify, models/mae.py:437-453, :282-293; one input channel):
 * loss = sum_l mask * mean_p (pred - target)^2 / sum mask, target = patchify(img), with norm_pix != 0 normalised per patch
 * ((t - mean) / sqrt(var + 1e-6), unbiased variance: models/mae.py:443-446; ABI v5).  acc2 = {numerator, sum mask} (kept for the backward);
 * bwd: dpred = gscale[0] * 2 * mask * (pred - target) / (P * sum mask).  pred / dpred are [B][pred_row0 + L][P] with
 * pred_seq_stride elements per clip: pred_row0 = 1 reads decoder_pred's output in place (its CLS row gets gradient 0).
 * ws (ABI v6): sa_mae_recon_loss_workspace_bytes() of caller-owned scratch for the per-block partial sums (one buffer per stream). */
int64_t sa_mae_recon_loss_workspace_bytes(void);
int sa_mae_recon_loss_fwd(const float* pred, int64_t pred_seq_stride, int32_t pred_row0, const float* img, const float* mask, int32_t B, int32_t F, int32_t T, int32_t ph, int32_t pw,
                          int32_t norm_pix, float* acc2, float* loss, float* ws, void* stream);
int sa_mae_recon_loss_bwd(const float* pred, int64_t pred_seq_stride, int32_t pred_row0, const float* img, const float* mask, const float* acc2, const float* gscale, int32_t B,
                          int32_t F, int32_t T, int32_t ph, int32_t pw, int32_t norm_pix, float* dpred, void* stream);
/* loss[0] = acc2[0] / acc2[1] again, after acc2 was summed over data-parallel ranks (global masked mean: what one process computes
 * on the global batch, models/mae.py:451-452). */
int sa_mae_recon_loss_finalize(const float* acc2, float* loss, void* stream);

/* ------------------------------------------------------------------ convolutional stems, channel-last
 * ConvStem (models/mae.py:46-99: 3x3 conv stride 2 / (2,1), pad 1, no bias -> BatchNorm2d -> ReLU, x4 (x6), then a 1x1 conv) and the
 * AudioNTT encoder (model.py:130-191: 3x3 conv stride 1 + bias -> BatchNorm2d -> ReLU -> MaxPool2d(2), x2, then an MLP per frame).
 * A feature map lives as NHWC = row-major [M = B*H*W][C]: it is directly the GEMM / BatchNorm matrix, and the token order of
 * `x.flatten(2).transpose(1, 2)` (models/mae.py:96).  H_out = (H - 1) / sh + 1 (kernel 3, pad 1), likewise W_out.
 *   c1_fwd   : first layer (C_in = 1): x fp32 [B][H][W], w fp32 [C_out][9], bias or NULL -> y fp32 [M][C_out]
 *   c1_wgrad : dw[C_out][9] += dy^T im2col(x), dbias[C_out] += column sums of dy (NULL: skipped); dy bf16 [M][C_out]
 *   im2col   : x bf16 NHWC -> out bf16 [M][Kpad], column (ky*3 + kx)*C + c, zero padding taps and columns 9C..Kpad-1 (the conv is
 *              then sa_gemm_bf16 against the weight laid out [C_out][ky][kx][C_in], zero-padded to Kpad)
 *   col2im   : dP bf16 [M][Kpad] (the GEMM's dgrad) -> dx fp32 NHWC, every element written (gather form, no atomics) */
int sa_conv3x3_c1_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t sh, int32_t sw, const float* w, const float* bias, int32_t Cout,
                      float* y, void* stream);
int sa_conv3x3_c1_wgrad(const float* x, int32_t B, int32_t H, int32_t W, int32_t sh, int32_t sw, const void* dy_bf16, int32_t Cout, float* dw,
                        float* dbias, void* stream);
int sa_im2col3x3_bf16(const void* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t sh, int32_t sw, void* out, int32_t Kpad, void* stream);
int sa_col2im3x3_f32(const void* dP_bf16, int32_t B, int32_t H, int32_t W, int32_t C, int32_t sh, int32_t sw, int32_t Kpad, float* dx, void* stream);
/* BatchNorm2d statistics of a tall [M][C] map (M = B*H*W in the millions): chunked (mean, M2) partials merged with Chan's formula;
 * results feed sa_bn_finalize / sa_bn_apply / sa_bn_bwd_apply exactly like sa_bn_colstats / sa_bn_bwd_stats do for the projector.
 * ws: sa_bn_tall_workspace_bytes(M, C) bytes of scratch. */
int64_t sa_bn_tall_workspace_bytes(int64_t M, int32_t C);
int sa_bn_colstats_tall(const float* x, int64_t ld, int64_t M, int32_t C, float* ws, float* mean, float* m2, void* stream);
int sa_bn_bwd_stats_tall(const void* dy, int32_t dy_is_bf16, int64_t lddy, const float* x, int64_t ld, int64_t M, int32_t C, const float* mean,
                         const float* rstd, const float* gamma, const float* beta, int32_t relu, float* ws, float* s1, float* s2, void* stream);
/* MaxPool2d(2, 2) on a bf16 NHWC map (model.py:141,149): y [B][H/2][W/2][C], idx = position of the maximum inside its window (0..3);
 * backward routes dy (fp32) to that position and writes every element of dx [B][H][W][C] */
int sa_maxpool2_fwd(const void* x_bf16, int32_t B, int32_t H, int32_t W, int32_t C, void* y_bf16, uint8_t* idx, void* stream);
int sa_maxpool2_bwd(const float* dy, const uint8_t* idx, int32_t B, int32_t H, int32_t W, int32_t C, float* dx, void* stream);
/* AudioNTT glue (model.py:153-191).  relu_mask: y = relu(x) [* keep * scale] (nn.ReLU, and nn.Dropout in train mode with the caller's
 * keep mask: uint8 [M][C], scale = 1 / (1 - p)); backward dx = dy * [x_pre > 0] [* keep * scale] as bf16.  nhwc_to_frames:
 * x.permute(0, 3, 2, 1).reshape(B, T, C * D) of the pooled map, frames[(b, w)][h * C + c], as bf16 (the MLP's GEMM operand) and / or
 * fp32 (into the stacked output, row stride ld32); frames_to_nhwc is its adjoint, summing two gradient sources.  meanmax_time:
 * mean_max_pooling over the time axis of [B][T][D], arg = position of the maximum. */
int sa_relu_mask_fwd(const float* x, int64_t ldx, int64_t M, int32_t C, const uint8_t* keep, float scale, void* y_bf16, int64_t ldy16, float* y_f32,
                     int64_t ldy32, void* stream);
int sa_relu_mask_bwd(const float* dy, int64_t lddy, const float* x_pre, int64_t ldx, int64_t M, int32_t C, const uint8_t* keep, float scale, void* dx_bf16,
                     int64_t lddx, void* stream);
int sa_nhwc_to_frames(const void* x_bf16, int32_t B, int32_t H, int32_t W, int32_t C, void* frames_bf16, float* frames_f32, int64_t ld32, void* stream);
int sa_frames_to_nhwc(const float* da, int64_t lda, const float* db, int64_t ldb, int32_t B, int32_t H, int32_t W, int32_t C, float* dx, void* stream);
int sa_meanmax_time_fwd(const float* x, int32_t B, int32_t T, int32_t D, float* out, int32_t* arg, void* stream);
int sa_meanmax_time_bwd(const float* dout, const int32_t* arg, int32_t B, int32_t T, int32_t D, float* dx, void* stream);
/* Squeeze-and-excitation gate of AudioNTT2022 (`squeeze_excitation=True`: SE_Block, model.py:196-213, placed after every MaxPool2d,
 * model.py:141-142,150-151) on a channel-last bf16 map x [B][L = H*W][C]:  s = mean_l x;  e = sigmoid(W2 relu(W1 s))  (W1 [R][C], W2 [C][R],
 * no biases, R = C / 16);  y = x * e.  fwd writes s [B][C], h [B][R] (the pre-ReLU hidden vector) and e [B][C] for the backward and y (bf16).
 * bwd: dy fp32 [B][L][C] at the gate's output -> dx fp32 (through the scale and through the squeeze), dW1 / dW2 ACCUMULATED (atomics);
 * scratch: 2 * B * C floats.  C must divide 256 (8..256), R <= 32. */
int sa_se_fwd(const void* x_bf16, int32_t B, int32_t L, int32_t C, const float* w1, const float* w2, int32_t R, float* s, float* h, float* e,
              void* y_bf16, void* stream);
int sa_se_bwd(const float* dy, const void* x_bf16, int32_t B, int32_t L, int32_t C, const float* w1, const float* w2, int32_t R, const float* s,
              const float* h, const float* e, float* dx, float* dw1, float* dw2, float* scratch, void* stream);

/* ------------------------------------------------------------------ ResNet-18 encoders (models/resnet.py: `resnet18`, `resnet18_ReGP_NRF`;
 * BASELINE config 1), channel-last like the stems above.  The 3x3 convolutions are sa_im2col3x3_bf16 + sa_gemm_bf16, the 1x1
 * downsample convolutions sa_subsample_fwd + sa_gemm_bf16, BatchNorm2d the sa_bn_*_tall kernels; these are the remaining pieces.
 *   maxpool3s2 : MaxPool2d(3, stride 2, padding 1) (models/resnet.py:191); H_out = (H - 1) / 2 + 1.  idx[m][c] = ky*3 + kx of the first
 *                maximum; the backward SUMS over the (overlapping) windows that recorded a pixel and writes every dx element
 *   subsample  : y[b][oy][ox] = x[b][oy*sh][ox*sw] (input rows of a 1x1 convolution with stride, :223-226); bwd_add: dx at the sampled
 *                pixels += dy (bf16, leading dimension lddy: the dgrad GEMM's padded output)
 *   add_relu   : y = max(z + identity, 0) -> fp32 and (optional) bf16    (BasicBlock.forward :75-78, z = bn2(conv2(.)))
 *   relu_bwd   : ds = y > 0 ? dy + dy2 : 0 (dy2 optional: the gradient arriving over the next block's identity path)
 *   avgpool    : AdaptiveAvgPool2d((1, 1)) + flatten (:266-268): out[b][c] = mean over the L = H*W rows; bwd writes dx = dout / L */
int sa_maxpool3s2_fwd(const void* x_bf16, int32_t B, int32_t H, int32_t W, int32_t C, void* y_bf16, float* y_f32, uint8_t* idx, void* stream);
int sa_maxpool3s2_bwd(const float* dy, const uint8_t* idx, int32_t B, int32_t H, int32_t W, int32_t C, float* dx, void* stream);
int sa_subsample_fwd(const void* x_bf16, int32_t B, int32_t H, int32_t W, int32_t C, int32_t sh, int32_t sw, void* y_bf16, void* stream);
int sa_subsample_bwd_add(const void* dy_bf16, int64_t lddy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t sh, int32_t sw, float* dx, void* stream);
int sa_add_relu_fwd(const float* z, const float* identity, int64_t n, float* y_f32, void* y_bf16, void* stream);
int sa_relu_bwd(const float* dy, const float* dy2, const float* y, int64_t n, float* ds, void* stream);
int sa_avgpool_fwd(const float* x, int32_t B, int32_t L, int32_t C, float* out, void* stream);
int sa_avgpool_bwd(const float* dout, int32_t B, int32_t L, int32_t C, float* dx, void* stream);

#ifdef __cplusplus
}
#endif
#endif
