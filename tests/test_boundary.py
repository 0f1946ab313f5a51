"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol the public header
declares, the product package never touches the oracle, and the host-side mirrors keep the reference's names."""
import ast
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ssl_audio_amd")


def test_library_exports_every_header_symbol():
    from ssl_audio_amd import _lib
    syms = _lib.header_symbols()
    assert len(syms) >= 25 and "sa_gemm_bf16" in syms and "sa_attention_bwd" in syms
    h = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(h, s)]
    assert not missing, missing
    h.sa_abi_version.restype = ctypes.c_int
    assert h.sa_abi_version() == 6
    # every declared function has ctypes argument types (a missing entry would silently pass ints as 32-bit)
    assert [s for s in syms if s not in _lib._SIGNATURES and s != "sa_last_error"] == []


def test_header_has_no_torch_types_and_cites_reference():
    text = open(os.path.join(ROOT, "include", "ssl_audio_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)          # declarations only (comments may cite torch semantics)
    assert "at::" not in code and "torch" not in code.lower() and "Tensor" not in code
    for cite in ["models/mae.py:", "utils/loss.py:", "augmentations.py:", "datasets.py:", "model.py:", "utils/utils.py:"]:
        assert cite in text, cite


def test_gemm_struct_layout_matches_header():
    """Field order of the ctypes mirror == field order in the C struct."""
    from ssl_audio_amd._lib import SaGemmArgs
    text = open(os.path.join(ROOT, "include", "ssl_audio_hip.h")).read()
    body = text[text.index("typedef struct SaGemmArgs {"):text.index("} SaGemmArgs;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split("{", 1)[1].split(";"):
        parts = [p.strip().split()[-1].lstrip("*") for p in decl.split(",") if p.strip()]
        names.extend(parts)
    assert names == [f[0] for f in SaGemmArgs._fields_]


def test_product_package_never_imports_the_oracle():
    """Only selfcheck.py (smoke's checker) may mention the oracle, and only inside functions (lazy import)."""
    for dirpath, _, files in os.walk(PKG):
        for fn in files:
            if not fn.endswith(".py"):
                continue
            src = open(os.path.join(dirpath, fn)).read()
            tree = ast.parse(src)
            for node in ast.walk(tree):
                mods = []
                if isinstance(node, ast.Import):
                    mods = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    mods = [node.module or ""]
                for m in mods:
                    if m == "oracle" or m.startswith("oracle."):
                        assert fn == "selfcheck.py", f"{fn} imports the oracle"
            if fn == "selfcheck.py":
                top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
                assert all(not ((getattr(n, "module", "") or "").startswith("oracle")) for n in top)
    for fn in os.listdir(os.path.join(PKG, "csrc")):
        if fn.endswith((".hip", ".h")):
            src = open(os.path.join(PKG, "csrc", fn)).read()
            assert "oracle/" not in src and not re.search(r"#include\s*[\"<][^\">]*oracle", src), fn


def test_missing_library_fails_loudly(monkeypatch):
    from ssl_audio_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libssl_audio_hip.so")
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.lib()


def test_cpu_tensors_are_rejected():
    """There is no CPU fallback: a CPU tensor is an error, not a slow path."""
    import torch
    from ssl_audio_amd import ops
    with pytest.raises(ValueError):
        ops.gemm(torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16), out_f32=torch.zeros(64, 64))


def test_reference_names_and_state_dict_keys():
    import torch
    from ssl_audio_amd import augmentations, hyperparameters as hp, loss, mae, model, transforms, utils
    for mod, names in [(model, ["ModelWrapper", "BarlowTwinsHead", "BarlowTwinsPredictor", "ViT"]),
                       (loss, ["BarlowTwinsLoss"]),
                       (utils, ["MultiCropWrapper", "EMA", "update_moving_average", "off_diagonal", "init_distributed_mode",
                                "model_setup_ddp", "get_param_groups", "is_main_process", "save_on_master"]),
                       (transforms, ["AudioPairTransform"]),
                       (augmentations, ["RandomResizeCrop", "RandomLinearFader", "MixupBYOLA", "NormalizeBatch", "log_mixup_exp"]),
                       (mae, ["MaskedAutoencoderViT", "PatchEmbed", "AttentionKBiasZero", "BlockKBiasZero", "get_mae_vit"])]:
        for n in names:
            assert hasattr(mod, n), (mod.__name__, n)
    cfg = hp.make_args(model_type="vit_tiny")
    net = utils.MultiCropWrapper(model.ModelWrapper(cfg), model.BarlowTwinsHead(cfg, 192))
    keys = set(net.state_dict())
    for k in ["backbone.encoder.encoder.cls_token", "backbone.encoder.encoder.blocks.0.attn.qkv.weight",
              "backbone.encoder.encoder.blocks.11.mlp.fc2.bias", "head.projector.0.weight", "head.projector.1.running_mean",
              "head.projector.3.weight"]:
        assert k in keys, k
    crit = loss.BarlowTwinsLoss(cfg, ncrops=2)
    assert set(crit.state_dict()) == {"bn.running_mean", "bn.running_var", "bn.num_batches_tracked"}
    groups = utils.get_param_groups(net)
    assert all(p.requires_grad for g in groups for p in g["params"]) and groups[1]["weight_decay"] == 0.
    frozen = [n for n, p in net.named_parameters() if not p.requires_grad]
    assert sorted(frozen) == sorted(["backbone.encoder.encoder.pos_embed", "backbone.encoder.encoder.patch_embed.proj.weight",
                                     "backbone.encoder.encoder.patch_embed.proj.bias"])


def test_hyperparameter_defaults_match_reference_contract():
    from ssl_audio_amd import hyperparameters as hp
    a = hp.get_std_parameters().parse_args(["--model_type", "vit_base", "--batch_size", "1024"])
    hp.setup_hyperparameters(a)
    assert (a.optimizer, a.wd) == ("AdamW", 0.06) and abs(a.lr - 1e-4 * 1024 / 128) < 1e-12
    assert (a.lmbda, a.alpha, a.projector_out_dim, a.projector_hidden_dim) == (0.005, 1, 256, 8192)
    assert (a.n_fft, a.win_length, a.hop_length, a.n_mels, a.f_min, a.f_max, a.crop_frames) == (1024, 1024, 160, 64, 60, 7800, 96)
    assert a.mixup and a.RRC and a.RLF and not a.Gnoise and a.load_lms
    b = hp.get_std_parameters().parse_args(["--no_mixup", "--load_wav", "--stop_gradient", "--predictor"])
    assert not b.mixup and not b.load_lms and b.stop_gradient and b.predictor


def test_host_side_pos_embed_and_sampling_match_oracle():
    """Host logic that runs without a GPU: positional tables and the augmentation sampler's RNG call order."""
    import random
    from oracle import augment as oaug, vit as ovit
    from ssl_audio_amd import augmentations as aug, pos_embed
    np.testing.assert_allclose(pos_embed.get_2d_sincos_pos_embed(192, (4, 6)), ovit.sincos_2d(192, (4, 6)), atol=1e-12)
    np.testing.assert_allclose(pos_embed.get_sinusoid_encoding_table(24, 384), ovit.sinusoid_table(24, 384), atol=1e-12)
    base = pos_embed.get_2d_sincos_pos_embed(768, (4, 6))[None]
    np.testing.assert_allclose(pos_embed.interpolate_pos_encoding(base, (4, 6), 64, 1001), ovit.interpolate_pos_embed(base, (4, 6), 64, 1001), atol=1e-12)
    assert pos_embed.interpolate_pos_encoding(base, (4, 6), 64, 1001).shape == (1, 249, 768)
    ba = aug.BatchedPairAugment("cpu", 64, 208, 208, seed=3, n_memory=6)
    ba.capacity = 12                      # host-side draw only (no device store needed)
    orc = oaug.PairTransformOracle(crop_frames=208, seed=3, n_memory=6)
    clips = np.zeros((8, 1, 64, 208))
    for it in range(2):
        src, mix, par, canvas = ba.draw(4)
        ba.clips += 4
        for b in range(4):
            orc(clips[4 * it + b])
    recs = orc.records[-8:]
    assert [r["rrc"] for r in ba.records] == [tuple(r["rrc"]) for r in recs]
    assert [r["bank_index"] for r in ba.records] == [r["bank_index"] for r in recs]
    np.testing.assert_allclose([r["alpha"] for r in ba.records], [r["alpha"] for r in recs])
    # ring-slot resolution of FIFO entry k of the second batch: event e = 2*clip + view has seen e appends, the FIFO
    # holds the last min(e, 6) of them, entry k is global append g = e - min(e, 6) + k and therefore clip g // 2
    for r in ba.records:
        e = 2 * r["clip"] + r["view"]
        g = e - min(e, 6) + r["bank_index"]
        slot = mix[r["view"] * 4 + (r["clip"] - 4)]
        assert slot == (g // 2) % 12 and src[r["view"] * 4 + (r["clip"] - 4)] == r["clip"] % 12


def test_sampler_draws_dataset_crop_before_the_views_and_matches_numpy_draw_for_draw():
    """(i) BatchedPairAugment.draw(B, src_frames): each clip's `np.random.randint(l - crop_frames)` (datasets.py:342-345; only when
    l > crop_frames) comes BEFORE that clip's view draws, as Dataset.__getitem__ crops and then transforms -- starts, crop boxes, mixup
    draws and fades equal the oracle's sequential `dataset_item`, local crops included.  (ii) The sampler's shortcuts are the numpy / random
    calls they replace, draw for draw: uniform(a, b) = a + (b - a) * random_sample(), rand(2) = two random_sample(), randint(0, n) =
    randrange(n + 1)."""
    import random
    from oracle import augment as oaug
    from ssl_audio_amd import augmentations as aug
    ba = aug.BatchedPairAugment("cpu", 64, 96, 96, seed=3, n_memory=6, local_crops_number=2)
    ba.capacity = 12
    orc = oaug.PairTransformOracle(crop_frames=96, seed=3, n_memory=6, local_crops_number=2)
    for it in range(6):
        fr = [1001, 50, 96, 300 + it]
        ba.draw(4, src_frames=fr)
        ba.clips += 4
        st = [orc.dataset_item(np.zeros((1, 64, l)))[1] for l in fr]
        assert st == ba.starts and st[1] == st[2] == 0 and 0 <= st[0] < 1001 - 96
        recs = orc.records[-16:]
        assert [tuple(r["rrc"]) for r in ba.records] == [tuple(r["rrc"]) for r in recs]
        glob = [r for r in recs if "alpha" in r]
        mine = [r for r in ba.records if "alpha" in r]
        assert [r["alpha"] for r in mine] == [r["alpha"] for r in glob] and [r["bank_index"] for r in mine] == [r["bank_index"] for r in glob]
        assert [r["head_tail"] for r in mine] == [tuple(r["head_tail"]) for r in glob]
    assert ba.draw(4) and ba.starts is None                       # without src_frames no crop draw is made (the pre-cropped path)
    a, b = np.random.RandomState(7), np.random.RandomState(7)
    for lo, hi in ((0.6, 1.5), (0.05, 0.6)):
        assert all(a.uniform(lo, hi) == lo + (hi - lo) * b.random_sample() for _ in range(50000))
    assert all(tuple(a.rand(2)) == (b.random_sample(), b.random_sample()) for _ in range(1000))
    p, q = random.Random(5), random.Random(5)
    assert all(p.randint(0, n) == q.randrange(n + 1) for n in list(range(1, 700)) * 5)


def test_c_abi_rejects_bad_arguments_before_touching_the_gpu():
    """Error behaviour of the boundary: argument checks run first, return non-zero and leave a message naming the entry point
    in sa_last_error -- no launch is attempted, so this runs on a CPU-only host."""
    import ctypes as C
    from ssl_audio_amd import _lib
    h = _lib.lib()
    P = C.c_void_p
    null = P(0)
    one = P(8)          # non-null, never dereferenced: every call below fails validation first
    cases = {
        "sa_layernorm_fwd": lambda: h.sa_layernorm_fwd(null, 0, null, null, null, null, 0, null, null, 4, 768, 1e-6, null),
        "sa_layernorm_fwd(D)": lambda: h.sa_layernorm_fwd(one, 770, one, one, one, null, 770, null, null, 4, 770, 1e-6, null),
        "sa_layernorm_bwd(ws)": lambda: h.sa_layernorm_bwd(one, 1, 768, one, 768, one, one, one, null, 0, one, null, 768, one, null, null, null, 4, 768, null),
        "sa_attention_fwd": lambda: h.sa_attention_fwd(one, 8, 2304, 768, 12, 300, 0, 0.125, one, 768, null, null),
        "sa_gather_rows": lambda: h.sa_gather_rows(one, 0, 0, one, 4, one, 0, 0, 2, 6, null),
        "sa_mae_unshuffle_fwd": lambda: h.sa_mae_unshuffle_fwd(one, 9, one, one, one, 2, 8, 64, one, null),
        "sa_mae_recon_loss_fwd": lambda: h.sa_mae_recon_loss_fwd(one, 10, 1, one, one, 2, 60, 96, 16, 16, 0, one, one, one, null),
        "sa_mean_tokens_fwd": lambda: h.sa_mean_tokens_fwd(one, 2, 1, 64, one, null),
    }
    for name, call in cases.items():
        rc = call()
        msg = h.sa_last_error().decode()
        assert rc != 0, name
        assert name.split("(")[0] in msg, (name, msg)
    g = _lib.SaGemmArgs()
    g.A, g.B, g.M, g.N, g.K, g.lda, g.ldb, g.split_k = 8, 8, 16, 16, 16, 12, 16, 1        # lda not a multiple of 8 elements
    g.out_f32, g.ldo_f32, g.a_kmajor, g.b_kmajor, g.alpha = 8, 16, 1, 1, 1.0
    assert h.sa_gemm_bf16(C.byref(g), null) != 0 and "sa_gemm_bf16" in h.sa_last_error().decode()


def test_driver_import_block_resolves_through_the_shims():
    """main_bt_byol.py:20-25 imports these names; with amd_shims/ first on sys.path every one must resolve to the MI355X package
    (`import datasets` is the reference's own data module and stays the reference's)."""
    import subprocess
    import sys
    code = ("from augmentations import RunningNorm, NormalizeBatch\n"
            "from utils.loss import BarlowTwinsLoss\n"
            "from utils import utils, transforms, hyperparameters\n"
            "from model import ModelWrapper, BarlowTwinsHead, BarlowTwinsPredictor\n"
            "mods = {RunningNorm, NormalizeBatch, BarlowTwinsLoss, ModelWrapper, BarlowTwinsHead, BarlowTwinsPredictor, utils.MultiCropWrapper,\n"
            "        utils.EMA, utils.LARS, transforms.AudioPairTransform}\n"
            "assert all(m.__module__.startswith('ssl_audio_amd') for m in mods), [m.__module__ for m in mods]\n"
            "assert callable(utils.update_moving_average) and callable(utils.adjust_learning_rate) and callable(hyperparameters.get_hyperparameters)\n"
            "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "amd_shims"), ROOT]))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_custom_ops_cover_every_compute_entry_point():
    """`torch.ops.ssl_audio.*` (ssl_audio_amd/custom_ops.py): one registered operator per compute entry point of the header -- only the
    queries (version, error string, device info, workspace sizes, CU budget) stay plain C calls -- with a parsed schema that marks the
    written tensors, and no CPU implementation (the dispatcher refuses CPU tensors: there is no fallback)."""
    import inspect
    import torch
    from ssl_audio_amd import _lib, custom_ops, ops
    covered = {sym for sym, _ in custom_ops.SCHEMAS.values()}
    plain = {s for s in _lib.header_symbols() if s not in covered}
    assert plain == {"sa_abi_version", "sa_last_error", "sa_device_info", "sa_set_cu_budget", "sa_set_dynamic_tiles", "sa_bn_tall_workspace_bytes",
                     "sa_gemm_colsum_workspace_bytes", "sa_colsum_workspace_bytes", "sa_gemm_splitk_workspace_bytes", "sa_layernorm_bwd_workspace_bytes",
                     "sa_bt_loss_workspace_bytes", "sa_mae_recon_loss_workspace_bytes", "sa_mae_unshuffle_bwd_workspace_bytes"}, plain
    assert covered <= set(_lib.header_symbols())
    for name, (sym, schema) in custom_ops.SCHEMAS.items():
        op = getattr(torch.ops.ssl_audio, name).default
        sch = op._schema
        assert any(a.alias_info is not None and a.alias_info.is_write for a in sch.arguments), name      # every operator writes an argument
        assert len(sch.returns) == 0, name
        impl = custom_ops._ADAPTERS.get(name) or getattr(ops, name)       # (an adapter re-packs arguments the dispatcher cannot carry)
        assert [a.name for a in sch.arguments] == list(inspect.signature(impl).parameters), name
        src = inspect.getsource(getattr(ops, name))
        assert f"lib().{sym}(" in src, (name, sym)                                                           # ... and reaches the symbol it names
    with pytest.raises(NotImplementedError):
        torch.ops.ssl_audio.axpy(torch.zeros(4), torch.ones(4), 2.0)


def test_qv_bias_gradient_fuses_only_inside_one_allocation():
    """engine.qv_row_sums: the q / v bias gradient rides along with the qkv weight-gradient launch only when both vectors are the [q | 0 | v]
    slices of ONE buffer (train.FlatState).  Two separately allocated gradients that merely sit 2 d apart -- the caching allocator hands the
    per-module path exactly that now and then -- must take the fallback: a [3 d] view of the first one's storage does not exist (a GPU
    run of tests/test_modules_gpu.py::test_mae_decoder_golden died on it, allocation-order dependent)."""
    import torch
    from ssl_audio_amd import engine
    d = 64
    base = torch.zeros(3 * d)
    r = engine.qv_row_sums(None, d, base[:d], base[2 * d:])
    assert r.out is not None and r.out.shape == (3 * d,) and (r.skip_lo, r.skip_hi) == (d, 2 * d)
    st = base.untyped_storage()
    gq = torch.empty(0).set_(st[0:4 * d], 0, (d,), (1,))
    gv = torch.empty(0).set_(st[8 * d:12 * d], 0, (d,), (1,))
    assert gv.data_ptr() == gq.data_ptr() + 8 * d                      # as far apart as the flat layout's slices ...
    assert engine.qv_row_sums(None, d, gq, gv).out is None             # ... but two storages: no fused form
    assert engine.qv_row_sums(None, d, base[:d], base[2 * d:].clone()).out is None


def test_weight_gradient_kernel_selection():
    """engine.stream_wgrad: which weight gradients (models/mae.py:106-129,149-163's Linear layers) take the 192 x 192 streaming split-K
    kernel -- ViT-T's four block shapes and the MAE decoder's wide ones; square d = 384, padded outputs, short reductions and d = 768 do
    not -- and the split ops.pick_split_k gives it fills one workgroup per CU without a nearly empty last round."""
    from ssl_audio_amd import engine, ops
    rows = 127488
    for shape in [(576, 192), (192, 192), (768, 192), (192, 768), (1152, 384), (1536, 384), (384, 1536), (384, 1024)]:
        assert engine.stream_wgrad(*shape, rows), shape
    for shape in [(384, 384), (192, 256), (256, 384), (2304, 768), (768, 768), (768, 3072), (8192, 192)]:
        assert not engine.stream_wgrad(*shape, rows), shape
    assert not engine.stream_wgrad(576, 192, 2048)
    for N, K in [(576, 192), (192, 192), (768, 192), (1152, 384)]:
        tiles = ((N + 191) // 192) * ((K + 191) // 192)
        split = ops.pick_split_k(N, K, rows, cu_count=256, tile=192)
        assert 128 < tiles * split <= 256 or (tiles * split) % 256 > 192, (N, K, split)
