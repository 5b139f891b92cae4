"""ForceUnet and the airfoil design objective on MI355X.

``ForceUnet`` keeps the reference's constructor and state-dict surface (model/diffusion_2d.py:411-486) so its
checkpoints load strictly; ``ForceObjective`` is the ``design_fn`` of inference/inverse_design_2d.py:208-214 -- the
gradient of (summed lift / drag surrogate forces + lambda * boundary overlap) with respect to the diffusion state --
evaluated by libcindm_hip.so (forward AND input-gradient kernels) instead of ``torch.autograd.grad``.  It plugs into
``GaussianDiffusion.sample(design_fn=..., design_guidance="standard-alpha")`` unchanged: the 2-D convention is that
``design_fn(x)`` returns the gradient tensor (model/diffusion_2d.py:813)."""
import ctypes as C
import math

import torch
from torch import nn

from . import _ffi
from .unet1d import _attach
from .unet2d import from_device_layout, to_device_layout


class ForceUnet(nn.Module):
    """Drop-in for ``ForceUnet(dim, dim_mults=(1, 2, 4, 8), channels=4)``.  ``forward(x[N, 4, H, W]) -> [N, 2]``;
    ``input_grad(x, lambda_force)`` returns d(sum lambda |out[:, 0]| + out[:, 1]) / dx as well."""

    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=3, self_condition=False,
                 resnet_block_groups=8, learned_variance=False, *, image_size=64):
        super().__init__()
        if self_condition or learned_variance or resnet_block_groups != 8 or (init_dim not in (None, dim)):
            raise NotImplementedError("ForceUnet: only the configuration of inference/inverse_design_2d.py:157-161 is built")
        self.channels, self.dim, self.dim_mults, self.image_size = channels, dim, tuple(dim_mults), int(image_size)
        L = _ffi.lib()
        d = _ffi.ForceUnetDesc()
        d.dim, d.n_mults, d.channels, d.image_size = dim, len(self.dim_mults), channels, self.image_size
        for i, m in enumerate(self.dim_mults):
            d.dim_mults[i] = m
        h = C.c_void_p()
        _ffi.check(L.cindm_forceunet_create(C.byref(d), C.byref(h)))
        self._h, self._sig, self._ws = h, None, None
        self._py_recovered = 0
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        manifest = []
        for i in range(L.cindm_forceunet_num_params(h)):
            _ffi.check(L.cindm_forceunet_param_info(h, i, name, 256, C.byref(shape), C.byref(nd)))
            manifest.append((name.value.decode(), tuple(int(shape[j]) for j in range(nd.value))))
        fan = {k[:-7]: int(torch.tensor(s[1:]).prod()) for k, s in manifest if k.endswith(".weight") and len(s) >= 2}
        for k, s in manifest:
            t = torch.empty(s)
            if k.endswith(".g") or k.endswith(".norm.weight"):
                t.fill_(1.0)
            elif k.endswith(".norm.bias"):
                t.zero_()
            else:
                t.uniform_(-1.0 / math.sqrt(fan[k.rsplit(".", 1)[0]]), 1.0 / math.sqrt(fan[k.rsplit(".", 1)[0]]))
            _attach(self, k, nn.Parameter(t))

    def __del__(self):
        h = self.__dict__.get("_h")
        if h is not None and h.value:
            try:
                _ffi.lib().cindm_forceunet_destroy(h)
            except Exception:
                pass
            self.__dict__["_h"] = None

    def sync_weights(self):
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if sig == self._sig:
            return
        L = _ffi.lib()
        dev = None
        for k, p in self.named_parameters():
            t = p.detach().to(torch.float32).contiguous()
            if t.is_cuda:
                dev = t.device
            _ffi.check(L.cindm_forceunet_set_param(self._h, k.encode(), _ffi.ptr(t), t.numel(), int(t.is_cuda)))
        if dev is None:
            raise _ffi.CindmError("ForceUnet parameters are on the CPU: move the module to a ROCm device; there is no CPU execution path")
        with torch.cuda.device(dev):
            _ffi.check(L.cindm_forceunet_finalize(self._h, _ffi.current_stream(dev)))
        self._sig = sig

    def set_option(self, key, value):
        """Selects a kernel path of this model (``cindm_forceunet_set_option``): ``h3`` / ``h3_bwd`` = 0 run the forward /
        input-gradient 3x3 convolutions on the exact fp32 MFMA kernel, ``auto_range`` = 0 skips the range rule's calibration
        forward.  Takes effect at the next call."""
        _ffi.check(_ffi.lib().cindm_forceunet_set_option(self._h, key.encode(), int(value)))
        if key not in ("no_exchange", "recover", "dbg", "stress"):       # (run-time options: the packed weights stay valid)
            self._sig = None
        return self

    @property
    def recovered(self):
        """How many gradient calls / guided chains of this model were re-run on the exchange-free GroupNorm derivative after an
        in-kernel exchange timed out (foreign load on the device); 0 in normal operation."""
        return int(_ffi.lib().cindm_forceunet_recovered(self._h)) + self._py_recovered

    # True (default): every gradient call reads the handle's exchange flag before its result is handed back -- a stream synchronise
    # and a 4-byte device-to-host copy per call, which serialises a Python-driven guided loop on the host.  False: asynchronous calls
    # (as TemporalUnet1D.forward(check=False)); the caller reads ``cindm_forceunet_status`` / ``poll_status()`` before trusting results.
    check_exchange = True

    def poll_status(self, device=None):
        """Reads and clears the exchange flag (synchronises the current stream): 0 = clean, 1 = an exchange timed out since the last
        read (the gradients computed since then contain NaN)."""
        dev = device if device is not None else next(self.parameters()).device
        with torch.cuda.device(dev):
            return int(_ffi.lib().cindm_forceunet_status(self._h, _ffi.current_stream(dev)))

    def _checked(self, call, device):
        """Runs ``call()`` (one library gradient call on the current stream) and hands its result back only after the handle's
        exchange flag has been read: a timed-out exchange (NaN gradients) is re-run once with ``no_exchange`` = 1 -- or raised,
        with ``recover`` = 0.  Skipped under stream capture (a capture cannot synchronise; the chain entry points check)."""
        out = call()
        if torch.cuda.is_current_stream_capturing() or not self.check_exchange:
            return out
        L = _ffi.lib()
        st = L.cindm_forceunet_status(self._h, _ffi.current_stream(device))
        if st < 0:
            _ffi.check(st)
        if st == 0:
            return out
        v = C.c_int32()
        _ffi.check(L.cindm_forceunet_get_option(self._h, b"recover", C.byref(v)))
        if not v.value:
            raise _ffi.CindmError("an in-kernel exchange of the surrogate's GroupNorm derivative timed out (foreign load on the device)")
        _ffi.check(L.cindm_forceunet_set_option(self._h, b"no_exchange", 1))
        try:
            out = call()
            st = L.cindm_forceunet_status(self._h, _ffi.current_stream(device))
        finally:
            _ffi.check(L.cindm_forceunet_set_option(self._h, b"no_exchange", 0))
        if st != 0:
            raise _ffi.CindmError("an exchange timed out during the exchange-free re-run (internal error)")
        self._py_recovered += 1
        return out

    def get_option(self, key):
        """Current option value; ``get_option("range_fallback")`` is 1 after the calibration forward left fp16's range."""
        self.sync_weights()
        v = C.c_int32()
        _ffi.check(_ffi.lib().cindm_forceunet_get_option(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def _workspace(self, nbytes, device):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._ws

    def _run(self, x, lambda_force=None, dout=None):
        if not x.is_cuda:
            raise _ffi.CindmError("ForceUnet needs a ROCm device tensor; there is no CPU execution path")
        n, c, hh, ww = x.shape
        if c != self.channels or hh != self.image_size or ww != self.image_size:
            raise ValueError(f"expected [N, {self.channels}, {self.image_size}, {self.image_size}], got {tuple(x.shape)}")
        self.sync_weights()
        L = _ffi.lib()
        xd = x.detach().float().permute(0, 2, 3, 1).reshape(n, hh * ww, c).contiguous()
        out = torch.empty((n, 2), dtype=torch.float32, device=x.device)
        with_grad = lambda_force is not None or dout is not None
        ws = self._workspace(L.cindm_forceunet_workspace_bytes(self._h, n, int(with_grad)), x.device)
        with torch.cuda.device(x.device):
            if not with_grad:
                _ffi.check(L.cindm_forceunet_forward(self._h, _ffi.ptr(xd), _ffi.ptr(out), n, _ffi.ptr(ws), ws.numel(),
                                                     _ffi.current_stream(x.device)))
                return out, None
            dx = torch.empty_like(xd)
            if dout is not None:
                do = dout.detach().to(device=x.device, dtype=torch.float32).reshape(n, 2).contiguous()
                self._checked(lambda: _ffi.check(L.cindm_forceunet_vjp(self._h, _ffi.ptr(xd), _ffi.ptr(do), _ffi.ptr(out), _ffi.ptr(dx), n,
                                                                       _ffi.ptr(ws), ws.numel(), _ffi.current_stream(x.device))), x.device)
            else:
                self._checked(lambda: _ffi.check(L.cindm_forceunet_grad(self._h, _ffi.ptr(xd), float(lambda_force), _ffi.ptr(out), _ffi.ptr(dx), n,
                                                                        _ffi.ptr(ws), ws.numel(), _ffi.current_stream(x.device))), x.device)
        return out, dx.reshape(n, hh, ww, c).permute(0, 3, 1, 2).contiguous()

    def forward(self, x, x_self_cond=None):
        """``[N, channels, H, W] -> [N, 2]`` (lift, drag).  Differentiable with respect to ``x`` under torch.autograd (the
        reference's ``force_fn`` takes ``autograd.grad`` of a function of this output, inverse_design_2d.py:113-117): the
        backward is the library's input-gradient pass (``cindm_forceunet_vjp``; the forward is recomputed there -- the
        kernels keep no state between calls).  Weights are frozen: no parameter gradients exist."""
        if x.requires_grad and torch.is_grad_enabled():
            return _ForceUnetFn.apply(x, self)
        with torch.no_grad():
            return self._run(x)[0]

    @torch.no_grad()
    def input_grad(self, x, lambda_force=1.0):
        """(out [N, 2], d(sum_n lambda_force * |out[n, 0]| + out[n, 1]) / dx [N, C, H, W])."""
        return self._run(x, lambda_force)


class _ForceUnetFn(torch.autograd.Function):
    """ForceUnet.forward under autograd: input gradients only (frozen weights), computed by the HIP backward pass."""

    @staticmethod
    def forward(ctx, x, model):
        ctx.model = model
        ctx.save_for_backward(x)
        return model._run(x)[0]

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        with torch.no_grad():
            _, dx = ctx.model._run(x, dout=dout)
        return dx.to(x.dtype), None


class ForceObjective:
    """``design_fn`` of the airfoil inverse design (inference/inverse_design_2d.py:208-214; ``sum_boundary`` as force_fn's):
    ``g = grad_force + lambda_overlap * grad_overlap`` for a state ``x [B * nb, 3 * frames + 3, 64, 64]``."""

    def __init__(self, force_model, batch_size, num_boundaries, frames, p_min, p_max, lambda_force=1.0, lambda_overlap=1.0,
                 downsampling_factor=4, frames_per_pass=None, sum_boundary=True):
        self.model, self.B, self.nb, self.frames = force_model, int(batch_size), int(num_boundaries), int(frames)
        # all frames of a design as ONE surrogate batch by default (frames * B * nb images per pass; workspace scales with it)
        self.frames_per_pass = int(frames_per_pass or frames)
        if self.frames % self.frames_per_pass:
            raise ValueError("frames_per_pass must divide frames")
        self.p_min, self.p_max = float(p_min), float(p_max)
        self.lambda_force, self.lambda_overlap, self.factor = float(lambda_force), float(lambda_overlap), int(downsampling_factor)
        self.sum_boundary = bool(sum_boundary)          # force_fn's branch (:98-132); the script's default is True
        self._ws = None

    @torch.no_grad()
    def __call__(self, x):
        if not x.is_cuda:
            raise _ffi.CindmError("ForceObjective needs a ROCm device tensor; there is no CPU execution path")
        n, c, hh, ww = x.shape
        if n != self.B * self.nb or c != 3 * self.frames + 3:
            raise ValueError(f"expected [{self.B * self.nb}, {3 * self.frames + 3}, H, W], got {tuple(x.shape)}")
        m = self.model
        m.sync_weights()
        L = _ffi.lib()
        cp = (c + 3) // 4 * 4
        xd = to_device_layout(x.detach().float(), cp)
        g = torch.empty_like(xd)
        nbytes = L.cindm_airfoil_design_workspace_bytes(m._h, self.B, self.nb, self.frames_per_pass)
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != x.device:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            m._checked(lambda: _ffi.check(L.cindm_airfoil_design_grad(m._h, _ffi.ptr(xd), self.B, self.nb, self.frames, cp, self.p_min, self.p_max,
                                                                      self.lambda_force, self.lambda_overlap, self.factor, int(self.sum_boundary),
                                                                      _ffi.ptr(g), _ffi.ptr(self._ws), self._ws.numel(),
                                                                      _ffi.current_stream(x.device))), x.device)
        return from_device_layout(g, c, hh, ww)
