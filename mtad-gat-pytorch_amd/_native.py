"""ctypes binding of the C ABI in include/mtadgat.h (libmtadgat.so, gfx950 HIP kernels).

PyTorch is used for device memory and the current HIP stream only.  There is no
fallback: if the library is missing or the tensors are not on a HIP device the
calls raise.
"""
import ctypes
import os

import torch  # must be imported before the library: both bind libamdhip64.so.7, torch's copy wins

# Debugging aid (tests/test_gpu_fuzz.py): every scratch buffer handed to the library (workspace, training tape) is
# filled with a large finite value before each call, so a kernel that USES scratch it did not write shows up in the outputs of a
# single call instead of depending on what the caching allocator left in the block (finite, inside the fp16 range: padding that
# is read and multiplied by zero weights is legitimate and stays harmless).
_POISON_SCRATCH = bool(os.environ.get("MTADGAT_POISON_SCRATCH"))
_POISON_VALUE = 7777.0


def _empty(*shape, **kw):
    """torch.empty, poisoned under MTADGAT_POISON_SCRATCH (outputs included: a kernel that accumulates into an output it was
    supposed to write, or leaves part of it unwritten, shows up the same way)."""
    t = torch.empty(*shape, **kw)
    if _POISON_SCRATCH and t.is_floating_point() and t.device.type == "cuda":
        t.fill_(_POISON_VALUE)
    return t


def _empty_like(x):
    return _empty(x.shape, dtype=x.dtype, device=x.device)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmtadgat.so")
MAX_LAYERS = 8
PROFILE_SLOTS = 6

_c_float_p = ctypes.POINTER(ctypes.c_float)


class Config(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "n_features", "window_size", "out_dim", "kernel_size", "use_gatv2", "feat_embed", "time_embed",
        "gru_n_layers", "gru_hid_dim", "forecast_n_linear", "forecast_hid_dim", "recon_n_layers",
        "recon_hid_dim")] + [("alpha", ctypes.c_float)]


class Params(ctypes.Structure):
    _fields_ = (
        [(n, ctypes.c_void_p) for n in (
            "conv_weight", "conv_bias", "feat_lin_weight", "feat_lin_bias", "feat_a", "feat_bias",
            "temp_lin_weight", "temp_lin_bias", "temp_a", "temp_bias")]
        + [(n, ctypes.c_void_p * MAX_LAYERS) for n in (
            "gru_w_ih", "gru_w_hh", "gru_b_ih", "gru_b_hh", "fc_weight", "fc_bias",
            "rec_w_ih", "rec_w_hh", "rec_b_ih", "rec_b_hh")]
        + [("rec_fc_weight", ctypes.c_void_p), ("rec_fc_bias", ctypes.c_void_p)])


_lib = None


def library_path():
    return _LIB_PATH


def load_library():
    """Load libmtadgat.so (raises with build instructions if it is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} is missing: the MI355X HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`python mtad-gat-pytorch_amd/build.py`). There is no non-HIP implementation.")
    lib = ctypes.CDLL(_LIB_PATH)
    vp, i64, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_size_t
    lib.mtadgat_abi_version.restype = ctypes.c_int
    lib.mtadgat_last_error.restype = ctypes.c_char_p
    lib.mtadgat_create.argtypes = [ctypes.POINTER(Config), ctypes.POINTER(vp)]
    lib.mtadgat_destroy.argtypes = [vp]
    lib.mtadgat_load_weights.argtypes = [vp, ctypes.POINTER(Params), vp]
    lib.mtadgat_update_weights_device.argtypes = [vp, vp, i64, vp]
    lib.mtadgat_params_fingerprint.argtypes = [vp, vp, ctypes.c_int, vp, vp]
    lib.mtadgat_packed_floats.argtypes = [vp]
    lib.mtadgat_packed_floats.restype = i64
    lib.mtadgat_read_packed.argtypes = [vp, vp, i64, vp]
    lib.mtadgat_derived_regions.argtypes = [vp, ctypes.POINTER(i64), ctypes.c_int]
    lib.mtadgat_workspace_bytes.argtypes = [vp, i64]
    lib.mtadgat_workspace_bytes.restype = sz
    lib.mtadgat_set_precision.argtypes = [vp, ctypes.c_int]
    lib.mtadgat_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_int]
    lib.mtadgat_last_conv_max.argtypes = [vp, vp, i64, ctypes.POINTER(ctypes.c_float), vp]
    lib.mtadgat_bf16_ready.argtypes = [vp]
    lib.mtadgat_chunk_windows.argtypes = [vp]
    lib.mtadgat_chunk_windows.restype = i64
    lib.mtadgat_set_chunk_windows.argtypes = [vp, i64]
    lib.mtadgat_forward.argtypes = [vp, vp, i64, vp, vp, vp, vp, sz, vp]
    lib.mtadgat_forward_xbf16.argtypes = [vp, vp, i64, vp, vp, vp, vp, sz, vp]
    lib.mtadgat_forward_series.argtypes = [vp, vp, i64, vp, i64, i64, i64, vp, vp, vp, vp, sz, vp]
    lib.mtadgat_conv.argtypes = [vp, vp, i64, vp, vp, sz, vp]
    lib.mtadgat_gat.argtypes = [vp, ctypes.c_int, vp, i64, vp, vp, sz, vp]
    lib.mtadgat_gru.argtypes = [vp, vp, i64, vp, vp, sz, vp]
    lib.mtadgat_heads.argtypes = [vp, vp, i64, vp, vp, vp, sz, vp]
    u64, f32 = ctypes.c_uint64, ctypes.c_float
    lib.mtadgat_backward_supported.argtypes = [vp]
    lib.mtadgat_tape_bytes.argtypes = [vp, i64]
    lib.mtadgat_tape_bytes.restype = sz
    lib.mtadgat_backward_workspace_bytes.argtypes = [vp, i64]
    lib.mtadgat_backward_input.argtypes = [vp, i64, vp, sz, vp, vp]
    lib.mtadgat_backward_input.restype = ctypes.c_int
    lib.mtadgat_backward_workspace_bytes.restype = sz
    lib.mtadgat_grad_floats.argtypes = [vp]
    lib.mtadgat_grad_floats.restype = i64
    lib.mtadgat_grad_offsets.argtypes = [vp, ctypes.POINTER(i64), ctypes.c_int]
    lib.mtadgat_train_layout.argtypes = [vp, i64, ctypes.POINTER(i64), ctypes.c_int]
    lib.mtadgat_forward_train.argtypes = [vp, vp, i64, i64, f32, u64, vp, vp, vp, sz, vp]
    lib.mtadgat_backward.argtypes = [vp, vp, i64, i64, f32, u64, vp, vp, vp, sz, vp, vp, sz, vp]
    lib.mtadgat_dropout_masks.argtypes = [vp, i64, i64, f32, u64, vp, vp, vp, vp]
    lib.mtadgat_dropout_masks_rnn.argtypes = [vp, i64, i64, f32, u64, vp, vp, vp]
    lib.mtadgat_profile_enable.argtypes = [vp, ctypes.c_int]
    lib.mtadgat_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64)]
    lib.mtadgat_profile_name.argtypes = [ctypes.c_int]
    lib.mtadgat_profile_name.restype = ctypes.c_char_p
    if lib.mtadgat_abi_version() != 1:
        raise RuntimeError("libmtadgat.so ABI version mismatch")
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = load_library().mtadgat_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"mtadgat {what} failed (status {rc}): {msg}")


def _dev_ptr(t, name, shape=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if t.device.type != "cuda":
        raise RuntimeError(
            f"{name} is on '{t.device}': the MI355X HIP path needs tensors on the GPU "
            "('cuda' = HIP on PyTorch-ROCm); there is no CPU implementation")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype})")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


class Engine:
    """One model instance on the native side: packed weights + launch plans."""

    def __init__(self, cfg: dict, device):
        self.lib = load_library()
        self.cfg = Config(**cfg)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"the MI355X HIP engine lives on a GPU, not on '{device}'")
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):      # the library sizes its chunks from this device's memory
            _check(self.lib.mtadgat_create(ctypes.byref(self.cfg), ctypes.byref(self.handle)), "create")
        self._ws = None
        self._keep = None

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.handle.value:
                self.lib.mtadgat_destroy(self.handle)
                self.handle = ctypes.c_void_p()
        except Exception:
            pass

    # -- training step ------------------------------------------------------------------------------
    def backward_supported(self):
        """True when the library has a HIP backward for this configuration (else why_not() says why)."""
        return bool(self.lib.mtadgat_backward_supported(self.handle))

    def why_not(self):
        self.lib.mtadgat_backward_supported(self.handle)
        return self.lib.mtadgat_last_error().decode("utf-8", "replace")

    def grad_layout(self):
        """[offsets] of the flat gradient buffer in the field order of mtadgat_params, total floats."""
        n = 10 + 4 * self.cfg.gru_n_layers + 2 * self.cfg.forecast_n_linear + 4 * self.cfg.recon_n_layers + 2
        offs = (ctypes.c_int64 * n)()
        got = self.lib.mtadgat_grad_offsets(self.handle, offs, n)
        if got != n:
            raise RuntimeError("mtadgat_grad_offsets: unexpected parameter count")
        return list(offs), int(self.lib.mtadgat_grad_floats(self.handle))

    TAPE_FIELDS = ("hcat", "xct", "att_f", "att_t", "hend", "gates_g", "seq_g", "gates_d", "seq_d", "xdec")
    WS_FIELDS = ("da", "dhcat", "dhdec", "dhend", "dz0", "dz1", "de_f", "de_t", "dv_f", "dv_t", "dlr_f", "dlr_t",
                 "dap_f", "dap_t", "dpre")

    def train_layout(self, batch):
        """Diagnostics: float offsets of the tape / backward-workspace regions for `batch` windows."""
        n = len(self.TAPE_FIELDS) + len(self.WS_FIELDS)
        offs = (ctypes.c_int64 * n)()
        if self.lib.mtadgat_train_layout(self.handle, batch, offs, n) != n:
            raise RuntimeError("mtadgat_train_layout failed")
        v = list(offs)
        return dict(zip(self.TAPE_FIELDS, v[:len(self.TAPE_FIELDS)])), dict(zip(self.WS_FIELDS, v[len(self.TAPE_FIELDS):]))

    def _buf(self, name, nbytes, device, fresh=True):
        cur = getattr(self, name, None)
        if cur is None or cur.device != device or cur.numel() * 4 < nbytes:
            setattr(self, name, None)
            cur = _empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
            setattr(self, name, cur)
        if _POISON_SCRATCH and fresh:
            cur.fill_(_POISON_VALUE)
        return cur

    def forward_train(self, x, p, seed, window0=0, tape=None):
        """Training forward of `x` (one chunk): (preds, recons, tape).  Dropout with probability p inside the kernels."""
        c = self.cfg
        b = x.shape[0]
        xp = _dev_ptr(x, "x", (b, c.window_size, c.n_features))
        preds = _empty((b, c.out_dim), dtype=torch.float32, device=x.device)
        recons = _empty((b, c.window_size, c.out_dim), dtype=torch.float32, device=x.device)
        if b == 0:
            return preds, recons, None
        need = self.lib.mtadgat_tape_bytes(self.handle, b)
        if tape is None or tape.numel() * 4 < need:
            tape = _empty((need + 3) // 4, dtype=torch.float32, device=x.device)
        if _POISON_SCRATCH:
            tape.fill_(_POISON_VALUE)
        self._call(self.lib.mtadgat_forward_train, "forward_train", x.device, xp, b, int(window0), float(p), int(seed),
                   _dev_ptr(preds, "preds"), _dev_ptr(recons, "recons"), _dev_ptr(tape, "tape"), need)
        return preds, recons, tape

    def backward(self, x, p, seed, d_preds, d_recons, tape, grads, window0=0):
        """Accumulates the parameter gradients of one chunk into the flat buffer `grads`."""
        c = self.cfg
        b = x.shape[0]
        if b == 0:
            return
        xp = _dev_ptr(x, "x", (b, c.window_size, c.n_features))
        need_t = self.lib.mtadgat_tape_bytes(self.handle, b)
        need_w = self.lib.mtadgat_backward_workspace_bytes(self.handle, b)
        ws = self._buf("_bws", need_w, x.device)
        self._call(self.lib.mtadgat_backward, "backward", x.device, xp, b, int(window0), float(p), int(seed),
                   _dev_ptr(d_preds, "d_preds", (b, c.out_dim)), _dev_ptr(d_recons, "d_recons", (b, c.window_size, c.out_dim)),
                   _dev_ptr(tape, "tape"), need_t, _dev_ptr(grads, "grads"), _dev_ptr(ws, "workspace"), need_w)

    def backward_input(self, x_like):
        """d loss / d x of the chunk mtadgat_backward just processed (same stream, same workspace): (b, W, F)."""
        c = self.cfg
        b = x_like.shape[0]
        dx = _empty((b, c.window_size, c.n_features), dtype=torch.float32, device=x_like.device)
        if b == 0:
            return dx
        need_w = self.lib.mtadgat_backward_workspace_bytes(self.handle, b)
        ws = self._buf("_bws", need_w, x_like.device, fresh=False)      # (mtadgat_backward's d pre-activations are read from it)
        self._call(self.lib.mtadgat_backward_input, "backward_input", x_like.device, b, _dev_ptr(ws, "workspace"), need_w, _dev_ptr(dx, "dx"))
        return dx

    def dropout_masks(self, batch, p, seed, device, window0=0):
        """The keep-masks the kernels apply: {"feat": (b,F,F), "temp": (b,W,W), "fc": [(b,hid)] * hidden layers}."""
        c = self.cfg
        mf = _empty((batch, c.n_features, c.n_features), dtype=torch.float32, device=device)
        mt = _empty((batch, c.window_size, c.window_size), dtype=torch.float32, device=device)
        nh = c.forecast_n_linear - 1
        mfc = _empty((max(nh, 1), batch, c.forecast_hid_dim), dtype=torch.float32, device=device)
        self._call(self.lib.mtadgat_dropout_masks, "dropout_masks", device, batch, int(window0), float(p), int(seed),
                   _dev_ptr(mf, "mask"), _dev_ptr(mt, "mask"), _dev_ptr(mfc, "mask"))
        out = {"feat": mf, "temp": mt, "fc": [mfc[i] for i in range(nh)]}
        lg, ld = c.gru_n_layers - 1, c.recon_n_layers - 1
        if lg > 0 or ld > 0:      # nn.GRU's dropout between stacked layers
            mg = _empty((max(lg, 1), batch, c.window_size, c.gru_hid_dim), dtype=torch.float32, device=device)
            mr = _empty((max(ld, 1), batch, c.window_size, c.recon_hid_dim), dtype=torch.float32, device=device)
            self._call(self.lib.mtadgat_dropout_masks_rnn, "dropout_masks_rnn", device, batch, int(window0), float(p), int(seed),
                       _dev_ptr(mg, "mask") if lg > 0 else None, _dev_ptr(mr, "mask") if ld > 0 else None)
            out["gru"] = [mg[i] for i in range(lg)]
            out["rec"] = [mr[i] for i in range(ld)]
        return out

    # -- weights ------------------------------------------------------------------------------
    def flat_keys(self):
        """state_dict keys in the field order of mtadgat_params (= the flat parameter / gradient buffer)."""
        names = ["conv.conv.weight", "conv.conv.bias"]
        for g in ("feature_gat", "temporal_gat"):
            names += [f"{g}.lin.weight", f"{g}.lin.bias", f"{g}.a", f"{g}.bias"]
        for l in range(self.cfg.gru_n_layers):
            names += [f"gru.gru.{k}_l{l}" for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        for i in range(self.cfg.forecast_n_linear):
            names += [f"forecasting_model.layers.{i}.weight", f"forecasting_model.layers.{i}.bias"]
        for l in range(self.cfg.recon_n_layers):
            names += [f"recon_model.decoder.rnn.{k}_l{l}" for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        return names + ["recon_model.fc.weight", "recon_model.fc.bias"]

    def update_weights_device(self, sd, device):
        """Re-pack from parameters that live on `device` without leaving it (after an optimizer step).  False when the
        library declines (no earlier host-side load, bf16 image, ...): the caller then uses load_weights."""
        if self._keep is None:
            return False
        flat = torch.cat([sd[k].detach().reshape(-1).to(torch.float32) for k in self.flat_keys()])
        if flat.device != device:
            return False
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream().cuda_stream
            rc = self.lib.mtadgat_update_weights_device(self.handle, ctypes.c_void_p(flat.data_ptr()), flat.numel(), ctypes.c_void_p(stream))
        if rc == -2:          # MTADGAT_ERR_UNSUPPORTED
            return False
        _check(rc, "update_weights_device")
        self._flat_dev = flat     # read by the kernels queued on the stream
        return True

    def fingerprint(self, params, device):
        """Bit-exact checksum of the parameter tensors (one kernel + one 8-byte read-back)."""
        n = len(params)
        cache = getattr(self, "_fp_cache", None)
        key = tuple(p.data_ptr() for p in params)
        if cache is None or cache[0] != key:
            ptrs = (ctypes.c_void_p * n)(*[p.data_ptr() for p in params])
            counts = (ctypes.c_int64 * n)(*[p.numel() for p in params])
            out = torch.zeros(1, dtype=torch.int64, device=device)
            cache = (key, ptrs, counts, out)
            self._fp_cache = cache
        _, ptrs, counts, out = cache
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream().cuda_stream
            _check(self.lib.mtadgat_params_fingerprint(ptrs, counts, n, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(stream)), "fingerprint")
        return int(out.item())

    def fingerprint_async(self, params, device):
        """The same checksum without a host wait: the kernel and an 8-byte copy into pinned host memory are enqueued on a side
        stream that starts where the current stream is now; returns a function that waits for THAT copy (an event, not a
        stream: the caller's work keeps running) and returns the value."""
        n = len(params)
        cache = getattr(self, "_fp_cache", None)
        key = tuple(p.data_ptr() for p in params)
        if cache is None or cache[0] != key:
            ptrs = (ctypes.c_void_p * n)(*[p.data_ptr() for p in params])
            counts = (ctypes.c_int64 * n)(*[p.numel() for p in params])
            out = torch.zeros(1, dtype=torch.int64, device=device)
            cache = (key, ptrs, counts, out)
            self._fp_cache = cache
        _, ptrs, counts, out = cache
        ring = getattr(self, "_fp_ring", None)
        if ring is None:
            ring = self._fp_ring = [[torch.zeros(1, dtype=torch.int64).pin_memory(), torch.cuda.Event(), torch.cuda.Event()] for _ in range(4)]
            self._fp_slot = 0
            self._fp_stream = torch.cuda.Stream(device)
        self._fp_slot = (self._fp_slot + 1) % len(ring)
        pin, ev, ev0 = ring[self._fp_slot]
        # on a side stream that starts where the caller's stream is now: the check runs beside the call's first kernels instead of
        # in front of them.  (The caller waits for `ev` before it returns -- _finish_weight_check -- so nothing the caller enqueues
        # after this call can overtake the read of the parameters.)
        with torch.cuda.device(device):
            cur = torch.cuda.current_stream()
            ev0.record(cur)
            side = self._fp_stream
            side.wait_event(ev0)
            with torch.cuda.stream(side):
                _check(self.lib.mtadgat_params_fingerprint(ptrs, counts, n, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(side.cuda_stream)), "fingerprint")
                pin.copy_(out, non_blocking=True)
                ev.record(side)

        def wait():
            ev.synchronize()
            return int(pin[0])
        return wait

    def read_packed(self, device):
        """Diagnostic: the packed weight image as a CPU tensor."""
        n = self.lib.mtadgat_packed_floats(self.handle)
        out = _empty(n, dtype=torch.float32)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream().cuda_stream
            _check(self.lib.mtadgat_read_packed(self.handle, ctypes.c_void_p(out.data_ptr()), n, ctypes.c_void_p(stream)), "read_packed")
        return out

    def derived_regions(self):
        """(offset, length) pairs in floats of the image regions derived on the device from other regions (split-bf16 packs)."""
        buf = (ctypes.c_int64 * 64)()
        n = self.lib.mtadgat_derived_regions(self.handle, buf, 32)
        return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(min(n, 32))]

    def load_weights(self, sd, device, allow_device_pack=True):
        """sd: reference-format state_dict (any device).  The first load packs on the host and uploads on the current
        stream; later ones (fp32 image, parameters on the GPU) re-pack on the device."""
        if allow_device_pack and self.update_weights_device(sd, device):
            return
        # one device-side concatenation + one transfer instead of a copy (and a sync) per tensor
        keys = list(sd.keys())
        flat = torch.cat([sd[k].detach().reshape(-1).to(torch.float32) for k in keys]).cpu()
        host, off = {}, 0
        for k in keys:
            n = sd[k].numel()
            host[k] = flat[off:off + n]
            off += n
        p = Params()

        def ptr(key):
            return ctypes.c_void_p(host[key].data_ptr())

        p.conv_weight, p.conv_bias = ptr("conv.conv.weight"), ptr("conv.conv.bias")
        p.feat_lin_weight, p.feat_lin_bias = ptr("feature_gat.lin.weight"), ptr("feature_gat.lin.bias")
        p.feat_a, p.feat_bias = ptr("feature_gat.a"), ptr("feature_gat.bias")
        p.temp_lin_weight, p.temp_lin_bias = ptr("temporal_gat.lin.weight"), ptr("temporal_gat.lin.bias")
        p.temp_a, p.temp_bias = ptr("temporal_gat.a"), ptr("temporal_gat.bias")
        for l in range(self.cfg.gru_n_layers):
            p.gru_w_ih[l], p.gru_w_hh[l] = ptr(f"gru.gru.weight_ih_l{l}").value, ptr(f"gru.gru.weight_hh_l{l}").value
            p.gru_b_ih[l], p.gru_b_hh[l] = ptr(f"gru.gru.bias_ih_l{l}").value, ptr(f"gru.gru.bias_hh_l{l}").value
        for i in range(self.cfg.forecast_n_linear):
            p.fc_weight[i] = ptr(f"forecasting_model.layers.{i}.weight").value
            p.fc_bias[i] = ptr(f"forecasting_model.layers.{i}.bias").value
        for l in range(self.cfg.recon_n_layers):
            pre = "recon_model.decoder.rnn."
            p.rec_w_ih[l], p.rec_w_hh[l] = ptr(f"{pre}weight_ih_l{l}").value, ptr(f"{pre}weight_hh_l{l}").value
            p.rec_b_ih[l], p.rec_b_hh[l] = ptr(f"{pre}bias_ih_l{l}").value, ptr(f"{pre}bias_hh_l{l}").value
        p.rec_fc_weight, p.rec_fc_bias = ptr("recon_model.fc.weight"), ptr("recon_model.fc.bias")
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream().cuda_stream
            _check(self.lib.mtadgat_load_weights(self.handle, ctypes.byref(p), ctypes.c_void_p(stream)), "load_weights")
        self._keep = flat

    # -- scratch ------------------------------------------------------------------------------
    def _workspace(self, batch, device):
        need = self.lib.mtadgat_workspace_bytes(self.handle, batch)
        ws = self._ws
        if ws is None or ws.device != device or ws.numel() * 4 < need:
            self._ws = None
            ws = _empty((need + 3) // 4, dtype=torch.float32, device=device)
            self._ws = ws
        if _POISON_SCRATCH:
            ws.fill_(_POISON_VALUE)
        return ws, need

    def set_precision(self, mode):
        """0 / False: fp32 MFMA operands (<= 1e-5 parity); 1 / True: bf16 MFMA operands, fp32 accumulation / state
        (<= 2e-2); 2: fp32-class results through split-bf16 operands on the large-batch kernels (<= 1e-5)."""
        _check(self.lib.mtadgat_set_precision(self.handle, int(mode)), "set_precision")

    def bf16_ready(self):
        return bool(self.lib.mtadgat_bf16_ready(self.handle))

    def set_option(self, name, value):
        """Testing / measurement hook (include/mtadgat.h): e.g. ("gru_kernel", 0 automatic | 1 tile-major | 2 chunk-major | 3 hidden-tile split on split operands)."""
        _check(self.lib.mtadgat_set_option(self.handle, name.encode(), int(value)), "set_option")

    def last_conv_max(self, batch, device):
        """Largest convolution output of the last forward of `batch` windows (range guard of the fp16 operand pieces);
        synchronises the current stream."""
        if self._ws is None:
            return 0.0
        out = ctypes.c_float(0.0)
        with torch.cuda.device(device):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _check(self.lib.mtadgat_last_conv_max(self.handle, ctypes.c_void_p(self._ws.data_ptr()), int(batch), ctypes.byref(out), stream),
                   "last_conv_max")
        return float(out.value)

    def chunk_windows(self):
        return int(self.lib.mtadgat_chunk_windows(self.handle))

    def set_chunk_windows(self, n):
        _check(self.lib.mtadgat_set_chunk_windows(self.handle, int(n)), "set_chunk_windows")
        self._ws = None

    def _call(self, fn, what, device, *args):
        with torch.cuda.device(device):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            _check(fn(self.handle, *args, stream), what)

    # -- forward ------------------------------------------------------------------------------
    def forward(self, x, want_hend=False):
        c = self.cfg
        b = x.shape[0]
        fn = self.lib.mtadgat_forward
        if x.dtype == torch.bfloat16:       # read directly by the convolution: no fp32 copy of the batch
            if x.device.type != "cuda" or not x.is_contiguous() or tuple(x.shape) != (b, c.window_size, c.n_features):
                raise RuntimeError("bfloat16 input must be a contiguous (b, W, F) tensor on the GPU")
            xp, fn = ctypes.c_void_p(x.data_ptr()), self.lib.mtadgat_forward_xbf16
        else:
            xp = _dev_ptr(x, "x", (b, c.window_size, c.n_features))
        preds = _empty((b, c.out_dim), dtype=torch.float32, device=x.device)
        recons = _empty((b, c.window_size, c.out_dim), dtype=torch.float32, device=x.device)
        hend = _empty((b, c.gru_hid_dim), dtype=torch.float32, device=x.device) if want_hend else None
        ws, need = self._workspace(b, x.device)
        self._call(fn, "forward", x.device, xp, b, _dev_ptr(preds, "preds"),
                   _dev_ptr(recons, "recons"), _dev_ptr(hend, "hend") if want_hend else None,
                   _dev_ptr(ws, "workspace"), need)
        return (preds, recons, hend) if want_hend else (preds, recons)

    def forward_series(self, series, starts=None, start0=0, stride=1, count=None, want_recons=True, want_last=False):
        """Windows gathered on the GPU from the device-resident series (n_rows, F); returns
        (preds, recons or None, recons[:, -1] or None)."""
        c = self.cfg
        if series.dim() != 2 or series.shape[1] != c.n_features:
            raise RuntimeError(f"series must have shape (n_rows, {c.n_features}), got {tuple(series.shape)}")
        sp = _dev_ptr(series, "series")
        n_rows = series.shape[0]
        if starts is not None:
            if starts.dtype != torch.int64 or starts.device != series.device or not starts.is_contiguous() or starts.dim() != 1:
                raise RuntimeError("starts must be a contiguous 1-D int64 tensor on the series' device")
            b = starts.shape[0]
            if b and (int(starts.min()) < 0 or int(starts.max()) + c.window_size > n_rows):
                raise RuntimeError("a window does not lie inside the series")
            stp = ctypes.c_void_p(starts.data_ptr())
        else:
            b = count if count is not None else max(0, (n_rows - c.window_size - start0) // max(stride, 1) + 1)
            stp = None
        dev = series.device
        preds = _empty((b, c.out_dim), dtype=torch.float32, device=dev)
        recons = _empty((b, c.window_size, c.out_dim), dtype=torch.float32, device=dev) if want_recons else None
        last = _empty((b, c.out_dim), dtype=torch.float32, device=dev) if want_last else None
        ws, need = self._workspace(b, dev)
        self._call(self.lib.mtadgat_forward_series, "forward_series", dev, sp, n_rows, stp, int(start0), int(stride), b,
                   _dev_ptr(preds, "preds"), _dev_ptr(recons, "recons") if want_recons else None,
                   _dev_ptr(last, "recons_last") if want_last else None, _dev_ptr(ws, "workspace") if b else None, need)
        return preds, recons, last

    def conv(self, x):
        c = self.cfg
        b = x.shape[0]
        xp = _dev_ptr(x, "x", (b, c.window_size, c.n_features))
        y = _empty_like(x)
        self._call(self.lib.mtadgat_conv, "conv", x.device, xp, b, _dev_ptr(y, "y"), None, 0)
        return y

    def gat(self, which, xc):
        c = self.cfg
        b = xc.shape[0]
        xp = _dev_ptr(xc, "x", (b, c.window_size, c.n_features))
        y = _empty_like(xc)
        ws, need = self._workspace(b, xc.device)
        self._call(self.lib.mtadgat_gat, "gat", xc.device, which, xp, b, _dev_ptr(y, "y"), _dev_ptr(ws, "workspace"), need)
        return y

    def gru(self, hcat):
        c = self.cfg
        b = hcat.shape[0]
        xp = _dev_ptr(hcat, "h_cat", (b, c.window_size, 3 * c.n_features))
        hend = _empty((b, c.gru_hid_dim), dtype=torch.float32, device=hcat.device)
        ws, need = self._workspace(b, hcat.device)
        self._call(self.lib.mtadgat_gru, "gru", hcat.device, xp, b, _dev_ptr(hend, "h_end"), _dev_ptr(ws, "workspace"), need)
        return hend

    def heads(self, hend, want_preds=True, want_recons=True):
        c = self.cfg
        b = hend.shape[0]
        hp = _dev_ptr(hend, "h_end", (b, c.gru_hid_dim))
        preds = _empty((b, c.out_dim), dtype=torch.float32, device=hend.device) if want_preds else None
        recons = _empty((b, c.window_size, c.out_dim), dtype=torch.float32, device=hend.device) if want_recons else None
        ws, need = self._workspace(b, hend.device)
        self._call(self.lib.mtadgat_heads, "heads", hend.device, hp, b,
                   _dev_ptr(preds, "preds") if want_preds else None,
                   _dev_ptr(recons, "recons") if want_recons else None, _dev_ptr(ws, "workspace"), need)
        return preds, recons

    # -- per-kernel timing (bench.py) ------------------------------------------------------------
    def profile_enable(self, on=True):
        _check(self.lib.mtadgat_profile_enable(self.handle, 1 if on else 0), "profile_enable")

    def profile_read(self):
        ms = (ctypes.c_double * PROFILE_SLOTS)()
        n = (ctypes.c_int64 * PROFILE_SLOTS)()
        _check(self.lib.mtadgat_profile_read(self.handle, ms, n), "profile_read")
        return {self.lib.mtadgat_profile_name(i).decode(): (ms[i], n[i]) for i in range(PROFILE_SLOTS)}
