"""`torch.ops.ssl_audio.*`: the C-ABI entry points of include/ssl_audio_hip.h as PyTorch custom operators (torch.library), the form
BASELINE.json's north_star names for driving the HIP kernels from Python.

Each operator is the out-variant the C ABI itself is (device pointers in, nothing allocated, nothing returned): outputs are arguments
and the schema marks them as written (`Tensor(a!)`), so the dispatcher, autograd's version counters and functionalisation see the
mutation.  The implementation registered for the CUDA (= ROCm) dispatch key is the ctypes call in `ops.py`; there is no CPU kernel
and no fallback -- calling an operator on CPU tensors raises NotImplementedError from the dispatcher.  The schema below is the single
source: argument names and order are those of the `ops` function of the same name (keyword-only arguments after `*`).

    import ssl_audio_amd.custom_ops                      # registers the namespace (idempotent)
    torch.ops.ssl_audio.layernorm_fwd(x, w, b, 1e-6, y_bf16=h)

The engine keeps calling `ops.*` directly (one Python frame less per launch on the eager path); both routes reach the same symbol.
"""
import torch

from . import ops

NAMESPACE = "ssl_audio"

# name -> (C-ABI symbol it reaches, schema).  T = Tensor, W = written tensor; `?` optional.
_T, _O = "Tensor", "Tensor?"
SCHEMAS = {
    "gemm": ("sa_gemm_bf16", "(Tensor A, Tensor B, *, bool a_kmajor=True, bool b_kmajor=True, float alpha=1.0, Tensor? bias=None, int act=0, "
             "Tensor? aux_in=None, Tensor(a!)? aux_out=None, Tensor? residual=None, int res_mod=0, Tensor(b!)? out_f32=None, "
             "Tensor(c!)? out_bf16=None, int row_group=0, int split_k=1, bool accumulate=False, bool tile256=False, "
             "Tensor(d!)? colsum_out=None, Tensor(e!)? asum_out=None, int asum_skip_lo=0, int asum_skip_hi=0) -> ()"),
    "gemm_wgrad_group": ("sa_gemm_wgrad_group", "(Tensor[] dY, Tensor[] X, Tensor(a!)[] out, int split_k, int tile=192, Tensor(b!)? asum_out=None, int asum_index=-1, "
                         "int asum_skip_lo=0, int asum_skip_hi=0) -> ()"),
    "transpose_bf16": ("sa_transpose_bf16", "(Tensor src, Tensor(a!) dst) -> ()"),
    # (the matrices written are named by pointers inside `desc`; the schema marks desc so that the operator counts as one with a side effect)
    "transpose_bf16_batch": ("sa_transpose_bf16_batch", "(Tensor(a!) desc, int n_tiles) -> ()"),
    "cast_bf16": ("sa_cast_f32_to_bf16", "(Tensor src, Tensor(a!) dst) -> ()"),
    "cast_f32_from_bf16": ("sa_cast_bf16_to_f32", "(Tensor src, Tensor(a!) dst) -> ()"),
    "colsum_bf16": ("sa_colsum_bf16", "(Tensor x, Tensor(a!) out, bool accumulate=False, int n_ranges=1, int range_stride=0) -> ()"),
    "layernorm_fwd": ("sa_layernorm_fwd", "(Tensor x, Tensor gamma, Tensor beta, float eps, *, Tensor(a!)? y_bf16=None, Tensor(b!)? y_f32=None, "
                      "Tensor(c!)? mean=None, Tensor(d!)? rstd=None) -> ()"),
    "layernorm_bwd": ("sa_layernorm_bwd", "(Tensor dy, Tensor x, Tensor gamma, Tensor mean, Tensor rstd, *, Tensor? dres=None, Tensor(a!)? dx_f32=None, "
                      "Tensor(b!)? dx_bf16=None, Tensor(c!)? dgamma=None, Tensor(d!)? dbeta=None, Tensor(e!)? dxsum=None) -> ()"),
    "attention_fwd": ("sa_attention_fwd", "(Tensor qkv, int H, int N, float scale, Tensor(a!) out, Tensor(b!)? lse=None, int n_query=0) -> ()"),
    "attention_bwd": ("sa_attention_bwd", "(Tensor qkv, int H, int N, float scale, Tensor out, Tensor dout, Tensor lse, Tensor(a!) dqkv, int n_query=0) -> ()"),
    "bn_colstats": ("sa_bn_colstats", "(Tensor x, Tensor(a!) mean, Tensor(b!) m2) -> ()"),
    "bn_colstats_tall": ("sa_bn_colstats_tall", "(Tensor x, Tensor(a!) mean, Tensor(b!) m2) -> ()"),
    "bn_finalize": ("sa_bn_finalize", "(Tensor stats, int rows_per_rank, float eps, float momentum, Tensor(a!) mean, Tensor(b!) rstd, "
                    "Tensor(c!)? running_mean=None, Tensor(d!)? running_var=None) -> ()"),
    "bn_apply": ("sa_bn_apply", "(Tensor x, Tensor mean, Tensor rstd, Tensor? gamma=None, Tensor? beta=None, bool relu=False, *, "
                 "Tensor(a!)? y_f32=None, Tensor(b!)? y_bf16=None) -> ()"),
    "bn_bwd_stats": ("sa_bn_bwd_stats", "(Tensor dy, Tensor x, Tensor mean, Tensor rstd, Tensor? gamma, Tensor? beta, bool relu, Tensor(a!) s1, Tensor(b!) s2) -> ()"),
    "bn_bwd_stats_tall": ("sa_bn_bwd_stats_tall", "(Tensor dy, Tensor x, Tensor mean, Tensor rstd, Tensor? gamma, Tensor? beta, bool relu, Tensor(a!) s1, Tensor(b!) s2) -> ()"),
    "bn_bwd_apply": ("sa_bn_bwd_apply", "(Tensor dy, Tensor x, Tensor mean, Tensor rstd, Tensor? gamma, Tensor? beta, bool relu, Tensor s1, Tensor s2, "
                     "float inv_n, *, Tensor? out_scale=None, Tensor(a!)? dx_f32=None, Tensor(b!)? dx_bf16=None) -> ()"),
    "matmul_f32": ("sa_matmul_f32", "(Tensor A, Tensor B, Tensor(a!) out, *, bool trans_a=False, bool trans_b=False, float alpha=1.0) -> ()"),
    "bt_loss_grad": ("sa_bt_loss_grad", "(Tensor c, float alpha, float lmbda, bool hsic, Tensor(a!) loss, Tensor(b!)? G=None) -> ()"),
    "bt_stats2": ("sa_bt_stats2", "(Tensor z1, Tensor z2, Tensor(a!) stats) -> ()"),
    "bt_corr": ("sa_bt_corr", "(Tensor z1, Tensor z2, Tensor all_stats, float eps, float momentum, float inv_n, Tensor(a!) mean, Tensor(b!) rstd, "
                "Tensor(c!)? running_mean, Tensor(d!)? running_var, Tensor(e!) z1n, Tensor(f!) z2n, Tensor(g!) c) -> ()"),
    "bt_bwd_products": ("sa_bt_bwd_products", "(Tensor z1n, Tensor z2n, Tensor G, float inv_n, Tensor(a!) dzn, Tensor(b!) sums) -> ()"),
    "bt_bwd_apply": ("sa_bt_bwd_apply", "(Tensor z1n, Tensor z2n, Tensor rstd, Tensor dzn, Tensor sums, float inv_n, Tensor? out_scale, "
                     "Tensor(a!) dz1, Tensor(b!) dz2) -> ()"),
    "adamw_step": ("sa_adamw_step", "(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, float lr, float beta1, float beta2, float eps, float wd, int step, "
                   "float grad_scale=1.0, Tensor(d!)? p_bf16=None) -> ()"),
    "lars_step": ("sa_lars_step", "(Tensor(a!) p, Tensor g, Tensor(b!) mu, float lr, float wd, float momentum, float eta, bool adapt, "
                  "Tensor(c!)? scratch2=None, Tensor(d!)? p_bf16=None) -> ()"),
    "adamw_step_dev": ("sa_adamw_step_dev", "(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, Tensor hyper3, float beta1, float beta2, float eps, float wd, "
                       "float grad_scale=1.0, Tensor(d!)? p_bf16=None, Tensor? skip_flag=None) -> ()"),
    "ema_update": ("sa_ema_update", "(Tensor(a!) target, Tensor online, float beta) -> ()"),
    "ema_update_gated": ("sa_ema_update_gated", "(Tensor(a!) target, Tensor online, float beta, Tensor? skip_flag) -> ()"),
    "axpy": ("sa_axpy_f32", "(Tensor(a!) y, Tensor x, float a=1.0) -> ()"),
    "count_nonfinite": ("sa_count_nonfinite", "(Tensor x, Tensor(a!) flag) -> ()"),
    "logmel_fwd": ("sa_logmel_fwd", "(Tensor wave, Tensor window, Tensor twiddle, Tensor mel_weights, Tensor mel_lo, Tensor mel_len, Tensor(a!) out, int T_out, "
                   "int start, float mean, float std, int hop, Tensor? starts=None, Tensor? lengths=None, Tensor? offsets=None) -> ()"),
    "augment_views": ("sa_augment_views", "(Tensor lms, int clip_stride, Tensor src_slot, Tensor? mix_slot, Tensor params, Tensor(a!) out, int F_in, int T_in, "
                      "int[] canvas, float max_w_ratio, bool do_fade, Tensor? noise=None) -> ()"),
    "normalize_batch": ("sa_normalize_batch", "(Tensor x, Tensor(a!) y, float shift, Tensor(b!) workspace, float eps, float stat_div=1.0) -> ()"),
    "patchify_bf16": ("sa_patchify_bf16", "(Tensor img, Tensor(a!) out, int ph, int pw) -> ()"),
    "fill_cls": ("sa_fill_cls", "(Tensor(a!) x, int S, int seq_stride, int d, Tensor cls, Tensor pos0) -> ()"),
    "cls_grad": ("sa_cls_grad", "(Tensor dx, int S, int seq_stride, int d, Tensor(a!) dcls) -> ()"),
    "gather_rows": ("sa_gather_rows", "(Tensor src, int src_seq_stride, int src_row0, Tensor idx, Tensor(a!) dst, int dst_seq_stride, int dst_row0, int S, int d) -> ()"),
    "scatter_add_rows": ("sa_scatter_add_rows", "(Tensor src, int src_seq_stride, int src_row0, Tensor idx, Tensor(a!) dst, int dst_seq_stride, int dst_row0, int S, int d) -> ()"),
    "mix_gaussian_noise": ("sa_mix_gaussian_noise", "(Tensor x, Tensor normal, float lambd, float eps, Tensor(a!) out) -> ()"),
    "running_norm": ("sa_running_norm", "(Tensor x, Tensor(a!) state, int n_seen, bool update, float eps, Tensor(b!) out) -> ()"),
    "token_group_sum": ("sa_token_group_sum", "(Tensor y, int row0, int G, int group_stride, int count, float scale, Tensor(a!) out, bool accumulate=False) -> ()"),
    "mean_tokens_fwd": ("sa_mean_tokens_fwd", "(Tensor y, Tensor(a!) out) -> ()"),
    "mean_tokens_bwd": ("sa_mean_tokens_bwd", "(Tensor dout, Tensor(a!) dy) -> ()"),
    "mae_unshuffle_fwd": ("sa_mae_unshuffle_fwd", "(Tensor x, Tensor mask_token, Tensor pos, Tensor ids_restore, Tensor(a!) out) -> ()"),
    "mae_unshuffle_bwd": ("sa_mae_unshuffle_bwd", "(Tensor dout, int keep, Tensor ids_restore, Tensor(a!) dx, Tensor(b!) dmask_token) -> ()"),
    "mae_recon_loss_fwd": ("sa_mae_recon_loss_fwd", "(Tensor pred, int row0, Tensor img, Tensor mask, int ph, int pw, Tensor(a!) acc2, Tensor(b!) loss, bool norm_pix=False) -> ()"),
    "mae_recon_loss_finalize": ("sa_mae_recon_loss_finalize", "(Tensor acc2, Tensor(a!) loss) -> ()"),
    "mae_recon_loss_bwd": ("sa_mae_recon_loss_bwd", "(Tensor pred, int row0, Tensor img, Tensor mask, int ph, int pw, Tensor acc2, Tensor gscale, Tensor(a!) dpred, bool norm_pix=False) -> ()"),
    "conv3x3_c1_fwd": ("sa_conv3x3_c1_fwd", "(Tensor x, Tensor w, Tensor? bias, int[] stride, Tensor(a!) y) -> ()"),
    "conv3x3_c1_wgrad": ("sa_conv3x3_c1_wgrad", "(Tensor x, Tensor dy16, int[] stride, Tensor(a!) dw, Tensor(b!)? dbias=None) -> ()"),
    "im2col3x3": ("sa_im2col3x3_bf16", "(Tensor x16, int B, int H, int W, int C, int[] stride, Tensor(a!) out) -> ()"),
    "col2im3x3": ("sa_col2im3x3_f32", "(Tensor dP16, int B, int H, int W, int C, int[] stride, Tensor(a!) dx) -> ()"),
    "maxpool2_fwd": ("sa_maxpool2_fwd", "(Tensor x16, int B, int H, int W, int C, Tensor(a!) y16, Tensor(b!) idx) -> ()"),
    "maxpool2_bwd": ("sa_maxpool2_bwd", "(Tensor dy, Tensor idx, int B, int H, int W, int C, Tensor(a!) dx) -> ()"),
    "relu_mask_fwd": ("sa_relu_mask_fwd", "(Tensor x, Tensor? keep, float scale, *, Tensor(a!)? y_bf16=None, Tensor(b!)? y_f32=None) -> ()"),
    "relu_mask_bwd": ("sa_relu_mask_bwd", "(Tensor dy, Tensor x_pre, Tensor? keep, float scale, Tensor(a!) dx_bf16) -> ()"),
    "nhwc_to_frames": ("sa_nhwc_to_frames", "(Tensor x16, int B, int H, int W, int C, Tensor(a!)? frames_bf16=None, Tensor(b!)? frames_f32=None) -> ()"),
    "frames_to_nhwc": ("sa_frames_to_nhwc", "(Tensor? da, Tensor? db, int B, int H, int W, int C, Tensor(a!) dx) -> ()"),
    "meanmax_time_fwd": ("sa_meanmax_time_fwd", "(Tensor x, Tensor(a!) out, Tensor(b!) arg) -> ()"),
    "se_fwd": ("sa_se_fwd", "(Tensor x16, int B, int L, int C, Tensor w1, Tensor w2, Tensor(a!) s, Tensor(b!) h, Tensor(c!) e, Tensor(d!) y16) -> ()"),
    "se_bwd": ("sa_se_bwd", "(Tensor dy, Tensor x16, int B, int L, int C, Tensor w1, Tensor w2, Tensor s, Tensor h, Tensor e, Tensor(a!) dx, "
               "Tensor(b!) dw1, Tensor(c!) dw2) -> ()"),
    "meanmax_time_bwd": ("sa_meanmax_time_bwd", "(Tensor dout, Tensor arg, Tensor(a!) dx) -> ()"),
    "maxpool3s2_fwd": ("sa_maxpool3s2_fwd", "(Tensor x16, int B, int H, int W, int C, Tensor(a!) y16, Tensor(b!)? y32, Tensor(c!) idx) -> ()"),
    "maxpool3s2_bwd": ("sa_maxpool3s2_bwd", "(Tensor dy, Tensor idx, int B, int H, int W, int C, Tensor(a!) dx) -> ()"),
    "subsample_fwd": ("sa_subsample_fwd", "(Tensor x16, int B, int H, int W, int C, int[] stride, Tensor(a!) y16) -> ()"),
    "subsample_bwd_add": ("sa_subsample_bwd_add", "(Tensor dy16, int B, int H, int W, int C, int[] stride, Tensor(a!) dx) -> ()"),
    "add_relu_fwd": ("sa_add_relu_fwd", "(Tensor z, Tensor identity, Tensor(a!) y_f32, Tensor(b!)? y_bf16=None) -> ()"),
    "relu_bwd": ("sa_relu_bwd", "(Tensor dy, Tensor? dy2, Tensor y, Tensor(a!) ds) -> ()"),
    "avgpool_fwd": ("sa_avgpool_fwd", "(Tensor x, int B, int L, int C, Tensor(a!) out) -> ()"),
    "avgpool_bwd": ("sa_avgpool_bwd", "(Tensor dout, int B, int L, int C, Tensor(a!) dx) -> ()"),
}

_LIB = None


def _logmel_fwd(wave, window, twiddle, mel_weights, mel_lo, mel_len, out, T_out, start, mean, std, hop, starts=None, lengths=None, offsets=None):
    """The dispatcher carries tensors, `ops.logmel_fwd` takes frontend.build_tables' dict: the five tables travel as five arguments
    (the registered schema had one `Tensor tables` and could not be called)."""
    ops.logmel_fwd(wave, {"window": window, "twiddle": twiddle, "mel_weights": mel_weights, "mel_lo": mel_lo, "mel_len": mel_len},
                   out, T_out, start, mean, std, hop, starts, lengths, offsets)


_ADAPTERS = {"logmel_fwd": _logmel_fwd}       # operators whose dispatcher signature differs from the `ops` function's


def _arg_names(schema):
    """Argument names of a schema string, with the keyword-only ones (after `*`) flagged."""
    inner = schema[schema.index("(") + 1:schema.rindex(") ->")]
    names, kwonly = [], False
    depth, cur = 0, ""
    for ch in inner + ",":
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            tok = cur.strip()
            cur = ""
            if tok == "*":
                kwonly = True
            elif tok:
                names.append((tok.split("=")[0].split()[-1], kwonly))
            continue
        cur += ch
    return names


def register():
    """Define the `ssl_audio` operator namespace (once per process) and bind every operator's CUDA implementation."""
    global _LIB
    if _LIB is not None:
        return _LIB
    lib = torch.library.Library(NAMESPACE, "DEF")
    for name, (_symbol, schema) in SCHEMAS.items():
        lib.define(name + schema)
        fn = _ADAPTERS.get(name) or getattr(ops, name)
        names = _arg_names(schema)

        def impl(*args, _fn=fn, _names=names, **kwargs):
            pos = [a for a, (_, kw) in zip(args, _names) if not kw]
            kw = {n: a for a, (n, k) in zip(args, _names) if k}
            kw.update(kwargs)
            _fn(*pos, **kw)

        lib.impl(name, impl, "CUDA")
        # what torch.library.custom_op(mutates_args=...) adds for an in-place operator: the written tensors' version counters advance
        # (autograd's saved-tensor checks, and engine._Bf16Cache's staleness test, then see a HIP kernel's write like any torch write)
        written = [i for i, a in enumerate(getattr(torch.ops, NAMESPACE).__getattr__(name).default._schema.arguments)
                   if a.alias_info is not None and a.alias_info.is_write]
        op = getattr(torch.ops, NAMESPACE).__getattr__(name).default

        def bump(keyset, *args, _op=op, _written=written, _names=[n for n, _ in names], **kwargs):
            for i in _written:
                t = args[i] if i < len(args) else kwargs.get(_names[i])
                for tt in (t if isinstance(t, (list, tuple)) else (t,)):
                    if tt is not None:
                        torch.autograd.graph.increment_version(tt)
            with torch._C._AutoDispatchBelowADInplaceOrView():
                return _op.redispatch(keyset & torch._C._after_ADInplaceOrView_keyset, *args, **kwargs)

        lib.impl(name, bump, "ADInplaceOrView", with_keyset=True)
    _LIB = lib
    return lib


register()
