"""TemporalUnet1D on MI355X: the reference's constructor / state_dict / forward surface
(model/diffusion_1d.py:517-646 of AI4Science-WestlakeU/cindm) over the HIP library.

The module owns ordinary ``nn.Parameter`` tensors under the reference's state-dict key names
(SURVEY.md Appendix A.3), so reference checkpoints load with ``load_state_dict(strict=True)``
and ``.to(device)`` / ``.eval()`` / ``.parameters()`` behave as callers expect.  ``forward``
hands raw device pointers to ``libcindm_hip.so``; there is no PyTorch compute path.
"""
import ctypes as C
import math

import torch
from torch import nn

from . import _ffi


class _Node(nn.Module):
    """Anonymous container used to rebuild the reference's dotted key hierarchy."""


def _attach(root, dotted, param):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], param)


def sinusoid_table(timesteps, dim):
    """Row t = SinusoidalPosEmb(dim)(t) (model/diffusion_1d.py:151-158): fp32 frequencies
    ``exp(arange(half) * -(ln 1e4 / (half-1)))``, fp32 product with the integer timestep, fp32
    sin/cos -- evaluated once on the host for t = 0..timesteps-1."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half) * -emb)
    arg = torch.arange(timesteps)[:, None] * freq[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1).contiguous()


class TemporalUnet1D(nn.Module):
    """Drop-in for ``TemporalUnet1D(horizon, transition_dim, cond_dim, dim=64,
    dim_mults=(1, 2, 4, 8), attention=False)`` (model/diffusion_1d.py:519-527).

    Extra keyword ``timesteps`` (default 1000) sizes the per-timestep bias table that replaces
    the time-embedding MLP at run time."""

    def __init__(self, horizon, transition_dim, cond_dim, dim=64, dim_mults=(1, 2, 4, 8), attention=False,
                 *, timesteps=1000):
        super().__init__()
        self.horizon = horizon
        self.transition_dim = transition_dim
        self.channels = transition_dim
        self.cond_dim = cond_dim
        self.dim = dim
        self.dim_mults = tuple(dim_mults)
        self.attention = bool(attention)
        self.timesteps = int(timesteps)
        L = _ffi.lib()
        d = _ffi.UnetDesc()
        d.horizon, d.transition_dim, d.dim, d.n_mults = horizon, transition_dim, dim, len(self.dim_mults)
        for i, m in enumerate(self.dim_mults):
            d.dim_mults[i] = m
        d.attention, d.timesteps = int(self.attention), self.timesteps
        h = C.c_void_p()
        _ffi.check(L.cindm_unet1d_create(C.byref(d), C.byref(h)))
        self._h = h
        self._sig = None
        self._ws = None
        self._ws_rows = 0
        # An in-kernel exchange between workgroups that timed out (foreign load on the device kept a partner workgroup from
        # becoming resident) is recovered by re-running the work once on the exchange-free kernels; False: raise CindmError
        self._recover = True
        # parameters under the reference's key names, PyTorch-default initialisation
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        manifest = []
        for i in range(L.cindm_unet1d_num_params(h)):
            _ffi.check(L.cindm_unet1d_param_info(h, i, name, 256, C.byref(shape), C.byref(nd)))
            manifest.append((name.value.decode(), tuple(int(shape[j]) for j in range(nd.value))))
        fan = {}
        for k, s in manifest:
            if k.endswith(".weight") and len(s) >= 2:
                f = s[1] * (s[2] if len(s) > 2 else 1)
                fan[k[:-7]] = f
        for k, s in manifest:
            base = k.rsplit(".", 1)[0]
            t = torch.empty(s)
            if k.endswith(".norm.g") or (".block.2." in k and k.endswith(".weight")):
                t.fill_(1.0)
            elif ".block.2." in k:
                t.zero_()
            else:
                bound = 1.0 / math.sqrt(fan[base])
                t.uniform_(-bound, bound)
            _attach(self, k, nn.Parameter(t))
        self._manifest = manifest

    def __del__(self):
        h = self.__dict__.get("_h")
        if h is not None and h.value:
            try:
                _ffi.lib().cindm_unet1d_destroy(h)
            except Exception:
                pass
            self.__dict__["_h"] = None

    # ------------------------------------------------------------------ weights -> library
    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def sync_weights(self, force=False):
        """Copies the current parameter values into the library handle and re-runs its
        finalisation (weight repack + per-timestep bias table) if anything changed."""
        sig = self._signature()
        if not force and sig == self._sig:
            return
        L = _ffi.lib()
        dev = None
        for k, p in self.named_parameters():
            if p.dtype != torch.float32:
                raise TypeError(f"{k}: fp32 parameters required, got {p.dtype}")
            t = p.detach().contiguous()
            if t.is_cuda:
                dev = t.device
            _ffi.check(L.cindm_unet1d_set_param(self._h, k.encode(), _ffi.ptr(t), t.numel(), int(t.is_cuda)))
        if dev is None:
            raise _ffi.CindmError("TemporalUnet1D parameters are on the CPU: move the module to a ROCm device "
                                  "(.to('cuda')); there is no CPU execution path")
        tab = sinusoid_table(self.timesteps, self.dim)
        _ffi.check(L.cindm_unet1d_set_sinusoid_table(self._h, _ffi.ptr(tab), tab.numel()))
        with torch.cuda.device(dev):
            _ffi.check(L.cindm_unet1d_finalize(self._h, _ffi.current_stream(dev)))
        self._sig = sig

    def set_option(self, key, value):
        """Selects a kernel path of this model (``cindm_unet1d_set_option``; keys in include/cindm_hip.h), e.g.
        ``set_option("mfma_f32", 1)``.  Every path computes the same function; takes effect at the next call."""
        _ffi.check(_ffi.lib().cindm_unet1d_set_option(self._h, key.encode(), int(value)))
        self._sig = None
        self._ws = None
        return self

    def get_option(self, key):
        """Current value of a kernel-path option; ``get_option("range_fallback")`` is 1 after the weights were found outside
        the split-fp16 window and the exact fp32 kernels were selected (evaluated when the weights are synchronised)."""
        self.sync_weights()
        v = C.c_int32()
        _ffi.check(_ffi.lib().cindm_unet1d_get_option(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def workspace(self, rows, device):
        L = _ffi.lib()
        if self._ws is None or self._ws_rows < rows or self._ws.device != device:
            nbytes = L.cindm_unet1d_workspace_bytes(self._h, rows)
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._ws_rows = rows
        return self._ws

    @property
    def launches_per_forward(self):
        return _ffi.lib().cindm_unet1d_launches_per_forward(self._h)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x, time, cond=None, *, check=None):
        """x [B, horizon, transition_dim] fp32 on a ROCm device, time [B] (all equal) -> eps [B, horizon, F]
        (model/diffusion_1d.py:610-646).  ``cond`` is ignored, as in the reference.

        ``check`` (build-only keyword): read the handle's exchange flag before returning -- a device-to-host copy and a
        stream synchronise -- and recover from a timed-out exchange (``recover_exchange_timeouts``).  Default: on, except
        while the current stream is being captured (a synchronise would invalidate the capture).  Callers that pass
        ``check=False`` (asynchronous pipelines, graph captures) must call ``check_status()`` / ``poll_status()`` themselves
        before they trust the results."""
        if not x.is_cuda:
            raise _ffi.CindmError("TemporalUnet1D.forward needs a ROCm device tensor; there is no CPU execution path")
        if x.dim() != 3 or x.shape[1] != self.horizon or x.shape[2] != self.transition_dim:
            raise ValueError(f"expected x of shape [B, {self.horizon}, {self.transition_dim}], got {tuple(x.shape)}")
        if torch.is_tensor(time):
            lo, hi = torch.aminmax(time)
            lo, hi = int(lo), int(hi)
            if lo != hi:
                raise NotImplementedError("per-row timesteps are not supported on the sampling path (all rows share t)")
            t = lo
        else:
            t = int(time)
        self.sync_weights()
        x = x.contiguous().float()
        out = torch.empty_like(x)
        ws = self.workspace(x.shape[0], x.device)
        if check is None:
            check = not torch.cuda.is_current_stream_capturing()

        def launch():
            with torch.cuda.device(x.device):
                _ffi.check(_ffi.lib().cindm_unet1d_forward(self._h, _ffi.ptr(x), t, None, _ffi.ptr(out), x.shape[0],
                                                           _ffi.ptr(ws), ws.numel(), _ffi.current_stream(x.device)))

        launch()
        if check and self.poll_status(x.device):
            self.rerun_exchange_free(launch, x.device)
        if check and self.range_guard_pending():
            # the range rule on the caller's own data: the FIRST checked forward after a weight synchronisation
            self.range_guard(lambda: out, launch, x.device)
        return out

    # ------------------------------------------------------------------ range rule on the caller's data (round 6)
    def range_guard_pending(self):
        """True until the first result after the last weight synchronisation has been checked -- and only where the check can
        change anything: the handle runs the split-fp16 kernels with ``auto_range`` on."""
        if self.__dict__.get("_range_checked_sig") == self._sig and self._sig is not None:
            return False
        return bool(self.get_option("auto_range")) and not self.get_option("mfma_f32") and not self.get_option("range_fallback")

    def range_guard(self, result, rerun, device):
        """``result()`` is what the split-fp16 kernels just produced for the caller's own inputs.  Finite: nothing to do (one
        reduction and one synchronise, once per weight synchronisation).  inf / nan: the un-normalised residual stream of this
        checkpoint left fp16's exponent range on real data although the synthetic calibration batch passed -- the handle is repacked
        for the exact fp32-MFMA kernels (``get_option("range_fallback")`` reads 3), ``rerun()`` repeats the work; if that is not
        finite either the cause was not the range (non-finite inputs / weights) and the split-fp16 pack is restored.
        Returns True when the handle was escalated."""
        self._range_checked_sig = self._sig
        if bool(torch.isfinite(result()).all()):
            return False
        import warnings
        with torch.cuda.device(device):
            _ffi.check(_ffi.lib().cindm_unet1d_range_escalate(self._h, 1, _ffi.current_stream(device)))
        self._ws = None; self._ws_rows = 0           # (the fp32 plan has its own workspace size)
        rerun()
        if bool(torch.isfinite(result()).all()):
            warnings.warn("cindm_amd: the first result after loading these weights was not finite on the split-fp16 kernels (an activation left "
                          "fp16's exponent range on the caller's data); the model now runs on the exact fp32-MFMA kernels, about 3x slower "
                          "(get_option('range_fallback') == 3)", RuntimeWarning)
            return True
        with torch.cuda.device(device):
            _ffi.check(_ffi.lib().cindm_unet1d_range_escalate(self._h, 0, _ffi.current_stream(device)))
        self._ws = None; self._ws_rows = 0
        return False

    TIMEOUT_TEXT = ("an in-kernel exchange between workgroups timed out (GroupNorm pair / attention head exchange): "
                    "the results of that forward are invalid")

    @property
    def recover_exchange_timeouts(self):
        return self._recover

    @recover_exchange_timeouts.setter
    def recover_exchange_timeouts(self, on):
        """Also tells the library (run-time option ``recover``): its chain entry points (sample / ddim_sample / the built-in guided
        loop) return an error instead of re-running a timed-out chain."""
        self._recover = bool(on)
        _ffi.check(_ffi.lib().cindm_unet1d_set_option(self._h, b"recover", int(self._recover)))

    def poll_raw(self, device):
        """True when an exchange of a forward issued so far timed out (the flag is cleared); synchronises; never raises for a
        time-out."""
        with torch.cuda.device(device):
            rc = _ffi.lib().cindm_unet1d_poll(self._h, _ffi.current_stream(device))
        if rc < 0:
            _ffi.check(rc)
        return rc == 1

    def check_status(self, device):
        """Raises CindmError when an in-kernel exchange between workgroups of a forward issued so far timed out (its
        results are invalid); synchronises the current stream."""
        with torch.cuda.device(device):
            _ffi.check(_ffi.lib().cindm_unet1d_status(self._h, _ffi.current_stream(device)))

    def poll_status(self, device):
        """True when an exchange of a forward issued so far timed out (the flag is cleared); synchronises the current
        stream.  With ``recover_exchange_timeouts = False`` a time-out raises CindmError here instead."""
        with torch.cuda.device(device):
            rc = _ffi.lib().cindm_unet1d_poll(self._h, _ffi.current_stream(device))
        if rc < 0:
            _ffi.check(rc)
        if rc == 1 and not self.recover_exchange_timeouts:
            raise _ffi.CindmError(self.TIMEOUT_TEXT)
        return rc == 1

    def exchange_free(self, on):
        """Run-time switch (``cindm_unet1d_set_option("no_exchange")``; does not touch the packed weights or the workspace):
        only kernels without an in-launch exchange between workgroups."""
        _ffi.check(_ffi.lib().cindm_unet1d_set_option(self._h, b"no_exchange", int(bool(on))))

    def rerun_exchange_free(self, fn, device):
        """``fn()`` once more with this model on the exchange-free kernels (after a time-out); a second time-out cannot
        happen there and raises."""
        self.exchange_free(True)
        try:
            fn()
            with torch.cuda.device(device):
                rc = _ffi.lib().cindm_unet1d_poll(self._h, _ffi.current_stream(device))
            if rc != 0:
                raise _ffi.CindmError(self.TIMEOUT_TEXT if rc == 1 else _ffi.lib().cindm_last_error().decode())
        finally:
            self.exchange_free(False)
        self._py_recovered = getattr(self, "_py_recovered", 0) + 1

    @property
    def recovered(self):
        """Forwards / chains of this model that were re-run on the exchange-free kernels after a time-out."""
        return _ffi.lib().cindm_unet1d_recovered(self._h) + getattr(self, "_py_recovered", 0)

    # kind 4 = the k=5 convolutions: conv_gemm_h3_kernel<5,48,*> (split-fp16 MFMA; default) or
    # conv_gemm_kernel<5,32,48,*> (fp32 MFMA; CINDM_MFMA=f32)
    KERNEL_KINDS = ("conv_gemm_kernel<0>", "conv_gemm_kernel<1>", "conv_gemm_kernel<3>", "conv_gemm_kernel<4>",
                    "conv5_gemm", "attention_and_level_kernels")

    @torch.no_grad()
    def profile(self, x, t):
        """One forward with every launch bracketed by HIP events on the current stream.
        Returns {kernel kind: (launches, total ms, total algorithmic FLOPs)}."""
        self.sync_weights()
        x = x.contiguous().float()
        out = torch.empty_like(x)
        ws = self.workspace(x.shape[0], x.device)
        cnt, ms, fl = (C.c_int32 * 6)(), (C.c_float * 6)(), (C.c_double * 6)()
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().cindm_unet1d_profile(self._h, _ffi.ptr(x), int(t), _ffi.ptr(out), x.shape[0], _ffi.ptr(ws),
                                                       ws.numel(), _ffi.current_stream(x.device), C.byref(cnt), C.byref(ms),
                                                       C.byref(fl)))
        return {k: (cnt[i], ms[i], fl[i]) for i, k in enumerate(self.KERNEL_KINDS)}

    @torch.no_grad()
    def profile_detail(self, x, t, cap=256):
        """Per-launch records of one instrumented forward: [(kind, ms, flops, grid_x, grid_y, stages)]."""
        self.sync_weights()
        x = x.contiguous().float()
        out = torch.empty_like(x)
        ws = self.workspace(x.shape[0], x.device)
        n = C.c_int32()
        kind, ms, fl, g = (C.c_int32 * cap)(), (C.c_float * cap)(), (C.c_double * cap)(), (C.c_int32 * (3 * cap))()
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().cindm_unet1d_profile_detail(self._h, _ffi.ptr(x), int(t), _ffi.ptr(out), x.shape[0], _ffi.ptr(ws),
                                                              ws.numel(), _ffi.current_stream(x.device), cap, C.byref(n), kind, ms, fl, g))
        return [(self.KERNEL_KINDS[kind[i]], ms[i], fl[i], g[3 * i], g[3 * i + 1], g[3 * i + 2]) for i in range(n.value)]

    def tap(self, name, rows):
        """Intermediate activation of the last forward as [rows, C, L] (the reference's layout)."""
        shape = (C.c_int64 * 3)()
        ws = self._ws
        dst = torch.empty(rows * 1536 * 8, dtype=torch.float32, device=ws.device)
        with torch.cuda.device(ws.device):
            _ffi.check(_ffi.lib().cindm_unet1d_tap(self._h, name.encode(), rows, _ffi.ptr(ws), _ffi.ptr(dst),
                                                   dst.numel(), C.byref(shape), _ffi.current_stream(ws.device)))
        n = shape[0] * shape[1] * shape[2]
        return dst[:n].view(shape[0], shape[1], shape[2]).transpose(1, 2).contiguous()
