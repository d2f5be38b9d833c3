"""Host-side logic that needs no GPU: the drop-in module's parameter contract, loud failures,
and the C ABI library (loads, exports every symbol include/mtadgat.h declares, validates configs)."""
import ctypes
import os
import re

import pytest
import torch

from helpers import SHIPPED_CASES, Case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_module_state_dict_contract():
    case = Case("msl")
    model = case.build_model()          # strict load of the shipped checkpoint + digest check
    sd = model.state_dict()
    expect = {
        "conv.conv.weight": (55, 55, 7), "conv.conv.bias": (55,),
        "feature_gat.lin.weight": (200, 200), "feature_gat.lin.bias": (200,), "feature_gat.a": (200, 1),
        "feature_gat.bias": (55, 55),
        "temporal_gat.lin.weight": (110, 110), "temporal_gat.lin.bias": (110,), "temporal_gat.a": (110, 1),
        "temporal_gat.bias": (100, 100),
        "gru.gru.weight_ih_l0": (450, 165), "gru.gru.weight_hh_l0": (450, 150),
        "gru.gru.bias_ih_l0": (450,), "gru.gru.bias_hh_l0": (450,),
        "recon_model.decoder.rnn.weight_ih_l0": (450, 150), "recon_model.decoder.rnn.weight_hh_l0": (450, 150),
        "recon_model.fc.weight": (1, 150), "recon_model.fc.bias": (1,),
    }
    for k, shp in expect.items():
        assert tuple(sd[k].shape) == shp, k
    assert len(sd) == 28 and sum(v.numel() for v in sd.values()) == 433777
    assert all(p.requires_grad for p in model.parameters())


@pytest.mark.parametrize("name", SHIPPED_CASES)
def test_shipped_checkpoints_load_strict(name):
    Case(name).build_model()


def test_seeded_init_equals_reference():
    # cases without a stored state_dict rebuild it from the seed; build_model asserts the digest
    for name in ("syn_v2_wide", "syn_v1_default"):
        Case(name).build_model()


def test_embed_dim_doubling_and_v1_shapes():
    from mtad_gat import MTAD_GAT
    m = MTAD_GAT(9, 16, 3, feat_gat_embed_dim=5, time_gat_embed_dim=3, use_gatv2=True)
    assert m.feature_gat.lin.weight.shape == (10, 32) and m.feature_gat.a.shape == (10, 1)
    assert m.temporal_gat.lin.weight.shape == (6, 18)
    m = MTAD_GAT(9, 16, 3, feat_gat_embed_dim=5, time_gat_embed_dim=3, use_gatv2=False)
    assert m.feature_gat.lin.weight.shape == (5, 16) and m.feature_gat.a.shape == (10, 1)
    with pytest.raises(ValueError):
        MTAD_GAT(9, 16, 3, kernel_size=4)


def test_cpu_tensors_take_the_packages_torch_path():
    """A model and input kept on the CPU (the reference's `--use_cuda False` branch, predict.py:122) are
    evaluated by _torchpath.py -- the package's own algebra, not the oracle -- and match the reference."""
    for name in ("syn_v2_embed", "syn_v1_small"):
        case = Case(name)
        model = case.build_model()
        with torch.no_grad():
            p, r = model(case.x)
            assert (p - case.preds).abs().max().item() <= 1e-5 and (r - case.recons).abs().max().item() <= 1e-5
            xc = model.conv(case.x)
            assert (xc - case.stages["xc"]).abs().max().item() <= 1e-5
            assert (model.feature_gat(xc) - case.stages["h_feat"]).abs().max().item() <= 1e-5
            assert (model.temporal_gat(xc) - case.stages["h_temp"]).abs().max().item() <= 1e-5
    # the GPU-side data-path entry points have no CPU form
    with pytest.raises(RuntimeError, match="GPU-side data path"):
        model.score_series(torch.rand(200, case.kwargs["n_features"]))


def test_gpu_tensors_never_fall_back(monkeypatch):
    """GPU tensors must reach the HIP library or raise: with the library 'missing' the engine refuses to
    exist (no torch-op fallback for device tensors)."""
    import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "_LIB_PATH", "/nonexistent/libmtadgat.so")
    with pytest.raises(RuntimeError, match="has not been built"):
        _native.Engine(Case("syn_v2_embed").build_model()._native_cfg, "cuda:0")


def test_deepcopy_and_pickle_rebind_the_stages():
    import copy
    import io
    case = Case("syn_v2_embed")
    m = case.build_model()
    m2 = copy.deepcopy(m)
    assert m2.conv._owner() is m2 and m.conv._owner() is m and m2._engine is None
    with torch.no_grad():
        m2.conv.conv.bias.add_(1.0)                     # the copy owns its parameters
        assert not torch.equal(m2.conv(case.x), m.conv(case.x))
    buf = io.BytesIO()
    torch.save(m, buf)                                  # whole-module pickle, as the reference supports
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert m3.gru._owner() is m3
    with torch.no_grad():
        assert torch.equal(m3(case.x)[0], m(case.x)[0])


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "mtad-gat-pytorch_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), fn


# ---- C ABI ---------------------------------------------------------------------------------
def _header_functions():
    hdr = open(os.path.join(ROOT, "include", "mtadgat.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mtadgat_[a-z_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import _native
    lib = _native.load_library()
    names = _header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mtadgat.h but not exported"
    assert lib.mtadgat_abi_version() == 1


def _cfg(**over):
    import _native
    base = dict(n_features=55, window_size=100, out_dim=1, kernel_size=7, use_gatv2=1, feat_embed=200, time_embed=110,
                gru_n_layers=1, gru_hid_dim=150, forecast_n_linear=4, forecast_hid_dim=150, recon_n_layers=1,
                recon_hid_dim=150, alpha=0.2)
    base.update(over)
    return _native.Config(**base)


def test_create_validates_and_plans_without_a_gpu():
    import _native
    lib = _native.load_library()
    h = ctypes.c_void_p()
    assert lib.mtadgat_create(ctypes.byref(_cfg()), ctypes.byref(h)) == 0
    ws1 = lib.mtadgat_workspace_bytes(h, 1)
    ws256 = lib.mtadgat_workspace_bytes(h, 256)
    assert 0 < ws1 < ws256
    # fused front at (W=100, F=55): h_cat (shared with the stage entry points' xc / xc^T copies) + h_end ~ 69 KB / window;
    # batches the hidden-tile-split GRU serves (<= 16 384 windows) add the pre-projected GRU input, W * 3 * Hp floats, and
    # batches of the 16-window-group recurrences (<= 4096 windows) the decoder's state sequence, W * Hp floats
    assert 60_000 + 192_000 + 64_000 < ws256 / 256 < 80_000 + 192_000 + 64_000
    assert 60_000 + 192_000 < lib.mtadgat_workspace_bytes(h, 8192) / 8192 < 80_000 + 192_000
    assert 60_000 < lib.mtadgat_workspace_bytes(h, 65536) / 65536 < 80_000
    chunk = lib.mtadgat_chunk_windows(h)
    assert lib.mtadgat_workspace_bytes(h, 10 * chunk) == lib.mtadgat_workspace_bytes(h, chunk)
    # the piece schedule of mode 2 (two lanes, each with its own part of the workspace): calls of 8 193 .. 16 384 windows are two
    # halves, calls beyond 32 768 with a partial last round are rounds of 32 768 + the rest; "lanes" = 1 keeps one plan per chunk
    assert lib.mtadgat_set_precision(h, 2) == 0
    per = lambda n: lib.mtadgat_workspace_bytes(h, n) / n
    assert per(10240) == pytest.approx(per(5120), rel=0.02)                       # 2 x 5 120 windows, hoisted GRU input and all
    assert lib.mtadgat_workspace_bytes(h, 36864) >= lib.mtadgat_workspace_bytes(h, 32768) + lib.mtadgat_workspace_bytes(h, 4096) - 4096
    assert per(65536) == pytest.approx(per(32768), rel=0.01)                      # a whole multiple of 32 768: one piece
    assert lib.mtadgat_set_option(h, b"lanes", 1) == 0
    assert lib.mtadgat_workspace_bytes(h, 36864) < lib.mtadgat_workspace_bytes(h, 32768) * 1.2
    assert lib.mtadgat_set_option(h, b"lanes", 0) == 0 and lib.mtadgat_set_option(h, b"lanes", 2) != 0
    for name, top in ((b"rowgemm_kernel", 2), (b"wgrad_kernel", 2), (b"conv_kernel", 2), (b"gat_kernel", 3)):
        assert lib.mtadgat_set_option(h, name, top) == 0 and lib.mtadgat_set_option(h, name, top + 1) != 0
        assert lib.mtadgat_set_option(h, name, 0) == 0
    assert lib.mtadgat_set_option(h, b"gat_kernel", 2) != 0                       # (round 4's column-sliced kernel is gone)
    assert lib.mtadgat_set_option(h, b"series_band", 0) != 0                      # (... and so is the shared-score band of rounds 4-5)
    assert lib.mtadgat_set_option(h, b"no_such_option", 0) != 0
    # arithmetic switch: 0 fp32 MFMA, 1 bf16 operands, 2 fp32 through split 16-bit operands; anything else is refused
    for mode, ok in ((0, True), (1, True), (2, True), (3, False), (-1, False)):
        assert (lib.mtadgat_set_precision(h, mode) == 0) == ok
    assert lib.mtadgat_set_precision(h, 0) == 0
    # the device-side re-pack needs an earlier load on a GPU: refused here with an error code, nothing is touched
    assert lib.mtadgat_update_weights_device(h, None, 0, None) != 0
    # forward before load_weights: an error code, not a crash, and nothing touches the device
    rc = lib.mtadgat_forward(h, None, 4, None, None, None, None, 0, None)
    assert rc == -4 and b"load_weights" in lib.mtadgat_last_error()
    assert lib.mtadgat_destroy(h) == 0
    for bad in (dict(kernel_size=4), dict(n_features=0), dict(window_size=600), dict(gru_hid_dim=300),
                dict(gru_n_layers=9)):
        h = ctypes.c_void_p()
        assert lib.mtadgat_create(ctypes.byref(_cfg(**bad)), ctypes.byref(h)) == -2, bad
        assert len(lib.mtadgat_last_error()) > 0
    assert lib.mtadgat_create(None, ctypes.byref(h)) == -1


_GATHER_CONFIGS = {
    "msl_shape": dict(n_features=55, window_size=100, out_dim=1, kernel_size=7, gru_hid_dim=150, forecast_n_layers=3,
                      forecast_hid_dim=150, recon_hid_dim=150),
    "gat_v1_stacked": dict(n_features=7, window_size=12, out_dim=7, kernel_size=3, use_gatv2=False, feat_gat_embed_dim=5,
                           time_gat_embed_dim=6, gru_n_layers=2, gru_hid_dim=20, forecast_n_layers=2, forecast_hid_dim=24,
                           recon_n_layers=2, recon_hid_dim=18),
    "wide_hidden": dict(n_features=6, window_size=12, out_dim=2, kernel_size=3, gru_hid_dim=200, recon_hid_dim=180,
                        forecast_n_layers=1, forecast_hid_dim=8),
    "many_nodes": dict(n_features=5, window_size=140, out_dim=5, kernel_size=3, gru_hid_dim=16, recon_hid_dim=16,
                       forecast_n_layers=1, forecast_hid_dim=8),
}


@pytest.mark.parametrize("name", list(_GATHER_CONFIGS))
def test_device_repack_gather_table_is_consistent_with_the_host_packer(name):
    """The device-side re-pack copies parameters into the tile image through an index table that is harvested from
    the host packer (run over parameters whose values are their own indices).  Host-only self check of that table:
    every position it covers must reproduce what the host packer put there, for real parameter values."""
    import _native
    import torch
    from mtad_gat import MTAD_GAT
    torch.manual_seed(3)
    model = MTAD_GAT(**_GATHER_CONFIGS[name])
    lib = _native.load_library()
    lib.mtadgat_selfcheck_gather_table.restype = ctypes.c_int64
    h = ctypes.c_void_p()
    assert lib.mtadgat_create(ctypes.byref(_native.Config(**model._native_cfg)), ctypes.byref(h)) == 0
    sd = {k: v.detach().float().contiguous() for k, v in model.state_dict().items()}
    cfg = model._native_cfg
    p = _native.Params()
    ptr = lambda key: ctypes.c_void_p(sd[key].data_ptr())      # noqa: E731
    p.conv_weight, p.conv_bias = ptr("conv.conv.weight"), ptr("conv.conv.bias")
    p.feat_lin_weight, p.feat_lin_bias = ptr("feature_gat.lin.weight"), ptr("feature_gat.lin.bias")
    p.feat_a, p.feat_bias = ptr("feature_gat.a"), ptr("feature_gat.bias")
    p.temp_lin_weight, p.temp_lin_bias = ptr("temporal_gat.lin.weight"), ptr("temporal_gat.lin.bias")
    p.temp_a, p.temp_bias = ptr("temporal_gat.a"), ptr("temporal_gat.bias")
    for l in range(cfg["gru_n_layers"]):
        p.gru_w_ih[l], p.gru_w_hh[l] = ptr(f"gru.gru.weight_ih_l{l}").value, ptr(f"gru.gru.weight_hh_l{l}").value
        p.gru_b_ih[l], p.gru_b_hh[l] = ptr(f"gru.gru.bias_ih_l{l}").value, ptr(f"gru.gru.bias_hh_l{l}").value
    for i in range(cfg["forecast_n_linear"]):
        p.fc_weight[i] = ptr(f"forecasting_model.layers.{i}.weight").value
        p.fc_bias[i] = ptr(f"forecasting_model.layers.{i}.bias").value
    for l in range(cfg["recon_n_layers"]):
        pre = "recon_model.decoder.rnn."
        p.rec_w_ih[l], p.rec_w_hh[l] = ptr(f"{pre}weight_ih_l{l}").value, ptr(f"{pre}weight_hh_l{l}").value
        p.rec_b_ih[l], p.rec_b_hh[l] = ptr(f"{pre}bias_ih_l{l}").value, ptr(f"{pre}bias_hh_l{l}").value
    p.rec_fc_weight, p.rec_fc_bias = ptr("recon_model.fc.weight"), ptr("recon_model.fc.bias")
    covered = ctypes.c_int64(0)
    bad = lib.mtadgat_selfcheck_gather_table(h, ctypes.byref(p), ctypes.byref(covered))
    assert bad == 0, (name, bad, lib.mtadgat_last_error())
    # the recurrent layers, the convolution and the heads are plain copies (the attention projections are computed
    # regions, filled by their own kernels)
    n_copied = sum(v.numel() for k, v in sd.items() if not k.startswith(("feature_gat", "temporal_gat")))
    assert covered.value >= n_copied
    assert lib.mtadgat_destroy(h) == 0
