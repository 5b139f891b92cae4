"""2-D Unet of the airfoil path on MI355X: the reference's constructor / state_dict / forward surface
(model/diffusion_2d.py:281-408 of AI4Science-WestlakeU/cindm) over the HIP library.

Parameters live under the reference's state-dict key names, so reference checkpoints load with
``load_state_dict(strict=True)``.  ``forward`` hands raw device pointers to ``libcindm_hip.so``; there is no
PyTorch compute path.  The library works on channel-last images ``[images, H*W, CP]`` (CP = channels rounded up
to a multiple of 4); ``to_device_layout`` / ``from_device_layout`` are the boundary conversions.
"""
import ctypes as C
import math

import torch
from torch import nn

from . import _ffi
from .unet1d import _attach, sinusoid_table


def to_device_layout(x, cp):
    """[N, C, H, W] -> [N, H*W, cp] channel-last, zero padding channels."""
    n, c, h, w = x.shape
    out = torch.zeros((n, h * w, cp), dtype=torch.float32, device=x.device)
    out[:, :, :c] = x.permute(0, 2, 3, 1).reshape(n, h * w, c)
    return out


def from_device_layout(y, c, h, w):
    """[N, H*W, cp] -> [N, c, H, W]."""
    n = y.shape[0]
    return y[:, :, :c].reshape(n, h, w, c).permute(0, 3, 1, 2).contiguous()


class Unet(nn.Module):
    """Drop-in for ``Unet(dim, dim_mults=(1, 2), channels=21)`` (model/diffusion_2d.py:282-367) on the sampling
    path.  Options the airfoil checkpoints never use (self-conditioning, learned variance, learned / random
    sinusoidal embeddings, init_dim / out_dim overrides, GroupNorm groups != 8) are rejected.

    Extra keywords: ``image_size`` (default 64) and ``timesteps`` (default 1000) size the launch plan and the
    per-timestep scale/shift table that replaces the time-embedding MLPs at run time."""

    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=3, self_condition=False,
                 resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, learned_sinusoidal_dim=16, *, image_size=64, timesteps=1000):
        super().__init__()
        if self_condition or learned_variance or learned_sinusoidal_cond or random_fourier_features:
            raise NotImplementedError("self_condition / learned_variance / learned or random sinusoidal embeddings are "
                                      "outside the sampling path this build covers")
        if (init_dim not in (None, dim)) or (out_dim not in (None, channels)) or resnet_block_groups != 8:
            raise NotImplementedError("init_dim / out_dim overrides and resnet_block_groups != 8 are not supported")
        self.channels = channels
        self.self_condition = False
        self.out_dim = channels
        self.random_or_learned_sinusoidal_cond = False
        self.dim = dim
        self.dim_mults = tuple(dim_mults)
        self.image_size = int(image_size)
        self.timesteps = int(timesteps)
        L = _ffi.lib()
        d = _ffi.Unet2dDesc()
        d.dim, d.n_mults = dim, len(self.dim_mults)
        for i, m in enumerate(self.dim_mults):
            d.dim_mults[i] = m
        d.channels, d.image_size, d.timesteps = channels, self.image_size, self.timesteps
        h = C.c_void_p()
        _ffi.check(L.cindm_unet2d_create(C.byref(d), C.byref(h)))
        self._h = h
        self._sig = None
        self._ws = None
        self._ws_images = 0
        self.padded_channels = L.cindm_unet2d_padded_channels(h)
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        manifest = []
        for i in range(L.cindm_unet2d_num_params(h)):
            _ffi.check(L.cindm_unet2d_param_info(h, i, name, 256, C.byref(shape), C.byref(nd)))
            manifest.append((name.value.decode(), tuple(int(shape[j]) for j in range(nd.value))))
        fan = {k[:-7]: int(torch.tensor(s[1:]).prod()) for k, s in manifest if k.endswith(".weight") and len(s) >= 2}
        for k, s in manifest:
            t = torch.empty(s)
            if k.endswith(".g") or k.endswith(".norm.weight"):
                t.fill_(1.0)
            elif k.endswith(".norm.bias"):
                t.zero_()
            else:
                bound = 1.0 / math.sqrt(fan[k.rsplit(".", 1)[0]])
                t.uniform_(-bound, bound)
            _attach(self, k, nn.Parameter(t))
        self._manifest = manifest

    def __del__(self):
        h = self.__dict__.get("_h")
        if h is not None and h.value:
            try:
                _ffi.lib().cindm_unet2d_destroy(h)
            except Exception:
                pass
            self.__dict__["_h"] = None

    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def sync_weights(self, force=False):
        """Copies the current parameter values into the library handle and re-runs its finalisation (weight
        standardisation + repack + per-timestep scale/shift table) if anything changed."""
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
            _ffi.check(L.cindm_unet2d_set_param(self._h, k.encode(), _ffi.ptr(t), t.numel(), int(t.is_cuda)))
        if dev is None:
            raise _ffi.CindmError("Unet parameters are on the CPU: move the module to a ROCm device (.to('cuda')); "
                                  "there is no CPU execution path")
        tab = sinusoid_table(self.timesteps, self.dim)
        _ffi.check(L.cindm_unet2d_set_sinusoid_table(self._h, _ffi.ptr(tab), tab.numel()))
        with torch.cuda.device(dev):
            _ffi.check(L.cindm_unet2d_finalize(self._h, _ffi.current_stream(dev)))
        self._sig = sig

    def set_option(self, key, value):
        """Selects a kernel path of this model (``cindm_unet2d_set_option``; keys in include/cindm_hip.h)."""
        _ffi.check(_ffi.lib().cindm_unet2d_set_option(self._h, key.encode(), int(value)))
        self._sig = None
        self._ws = None
        return self

    def get_option(self, key):
        """Current value of a kernel-path option; ``get_option("range_fallback")`` is 1 after the weights were found outside
        the split-fp16 window and the exact fp32 kernels were selected (evaluated when the weights are synchronised)."""
        self.sync_weights()
        v = C.c_int32()
        _ffi.check(_ffi.lib().cindm_unet2d_get_option(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def workspace(self, images, device):
        if self._ws is None or self._ws_images < images or self._ws.device != device:
            nbytes = _ffi.lib().cindm_unet2d_workspace_bytes(self._h, images)
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self._ws_images = images
        return self._ws

    @property
    def launches_per_forward(self):
        return _ffi.lib().cindm_unet2d_launches_per_forward(self._h)

    @staticmethod
    def _t_int(time):
        if torch.is_tensor(time):
            lo, hi = torch.aminmax(time)
            lo, hi = int(lo), int(hi)
            if lo != hi:
                raise NotImplementedError("per-image timesteps are not supported on the sampling path (all images share t)")
            return lo
        return int(time)

    @torch.no_grad()
    def forward_device_layout(self, x, t):
        """x [images, H*W, CP] (library layout) -> eps, same layout."""
        self.sync_weights()
        out = torch.zeros_like(x)
        ws = self.workspace(x.shape[0], x.device)
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().cindm_unet2d_forward(self._h, _ffi.ptr(x), int(t), None, _ffi.ptr(out), x.shape[0],
                                                       _ffi.ptr(ws), ws.numel(), _ffi.current_stream(x.device)))
        return out

    @torch.no_grad()
    def forward(self, x, time, x_self_cond=None):
        """x [images, channels, H, W] fp32 on a ROCm device, time [images] (all equal) -> eps, same shape
        (model/diffusion_2d.py:369-408)."""
        if not x.is_cuda:
            raise _ffi.CindmError("Unet.forward needs a ROCm device tensor; there is no CPU execution path")
        S = self.image_size
        if x.dim() != 4 or x.shape[1] != self.channels or x.shape[2] != S or x.shape[3] != S:
            raise ValueError(f"expected x of shape [N, {self.channels}, {S}, {S}], got {tuple(x.shape)}")
        xd = to_device_layout(x.float(), self.padded_channels)
        out = self.forward_device_layout(xd, self._t_int(time))
        return from_device_layout(out, self.channels, S, S)

    KERNEL_KINDS = ("conv3x3", "conv1x1", "stem7x7", "qkv1x1", "linear_attention", "full_attention", "stats_ln")

    @torch.no_grad()
    def profile(self, x, t):
        """One forward (x in the library layout [images, H*W, CP]) with every launch bracketed by HIP events on the
        current stream.  Returns {kernel kind: (launches, total ms, total algorithmic FLOPs)}."""
        self.sync_weights()
        out = torch.zeros_like(x)
        ws = self.workspace(x.shape[0], x.device)
        cnt, ms, fl = (C.c_int32 * 7)(), (C.c_float * 7)(), (C.c_double * 7)()
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().cindm_unet2d_profile(self._h, _ffi.ptr(x), int(t), _ffi.ptr(out), x.shape[0], _ffi.ptr(ws),
                                                       ws.numel(), _ffi.current_stream(x.device), C.byref(cnt), C.byref(ms),
                                                       C.byref(fl)))
        return {k: (cnt[i], ms[i], fl[i]) for i, k in enumerate(self.KERNEL_KINDS)}

    def tap(self, name, images):
        """Intermediate activation of the last forward as [images, C, H, W] (the reference's layout)."""
        shape = (C.c_int64 * 3)()
        ws = self._ws
        S = self.image_size
        dst = torch.empty(images * S * S * 384, dtype=torch.float32, device=ws.device)
        with torch.cuda.device(ws.device):
            _ffi.check(_ffi.lib().cindm_unet2d_tap(self._h, name.encode(), images, _ffi.ptr(ws), _ffi.ptr(dst),
                                                   dst.numel(), C.byref(shape), _ffi.current_stream(ws.device)))
        n, hw, c = shape[0], shape[1], shape[2]
        s = int(round(math.sqrt(hw)))
        return dst[:n * hw * c].view(n, s, s, c).permute(0, 3, 1, 2).contiguous()
