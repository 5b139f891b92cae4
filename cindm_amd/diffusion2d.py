"""GaussianDiffusion (2-D airfoil design, sampling half) on MI355X: the reference's constructor / buffers /
``sample`` surface (model/diffusion_2d.py:551-963 of AI4Science-WestlakeU/cindm) over the HIP library.

What runs where
  * the Unet on all ``batch * num_boundaries`` images, the boundary sharing of the predicted noise, x0, clamp,
    posterior mean and the (boundary-shared) noise add run in ``libcindm_hip.so``;
  * with ``design_fn=None`` the whole reverse loop is ONE call (``cindm_ddpm2d_sample``: one captured HIP graph
    step replayed per timestep), the state staying in the library's channel-last layout for the whole chain;
  * ``design_fn`` is a user Python callable returning a gradient tensor (:813); it is evaluated between library
    calls, exactly where the reference evaluates it.
Training, DDIM and self-conditioning are outside this build's scope and raise NotImplementedError.
"""
import ctypes as C
from collections import namedtuple

import torch
from torch import nn

from . import _ffi
from .schedule import make_schedule
from .unet2d import from_device_layout, to_device_layout


ModelPrediction = namedtuple("ModelPrediction", ["pred_noise", "pred_x_start"])      # model/diffusion_2d.py:43


class NoiseTape2D:
    """Explicit noise for parity runs, replacing the reference's ``sample_noise`` draws (:775-785):
    ``init`` = (state [B,1,C-3,H,W], boundary [B,nb,3,H,W]) for x_T (:895);
    ``step_state`` [T,B,1,C-3,H,W] / ``step_boundary`` [T,B,nb,3,H,W] indexed by timestep (:807)."""

    def __init__(self, init, step_state, step_boundary):
        self.init, self.step_state, self.step_boundary = init, step_state, step_boundary


def _state_cl(s):
    """[..., B, 1, Cs, H, W] -> channel-last [..., B, H*W, Cs]"""
    *lead, b, one, cs, h, w = s.shape
    return s.reshape(*lead, b, cs, h * w).transpose(-1, -2).contiguous()


def _boundary_cl(s):
    """[..., B, nb, 3, H, W] -> channel-last [..., B*nb, H*W, 3]"""
    *lead, b, nb, c, h, w = s.shape
    return s.reshape(*lead, b * nb, c, h * w).transpose(-1, -2).contiguous()


class GaussianDiffusion(nn.Module):
    """Drop-in for the reference's 2-D ``GaussianDiffusion`` (constructor :552-676)."""

    def __init__(self, model, *, image_size, frames=6, cond_frames=4, timesteps=1000, sampling_timesteps=None,
                 loss_type="l1", objective="pred_noise", beta_schedule="sigmoid", schedule_fn_kwargs=dict(),
                 ddim_sampling_eta=0., auto_normalize=True, min_snr_loss_weight=False, min_snr_gamma=5,
                 diffuse_cond=True, backward_steps=5, backward_lr=0.01, standard_fixed_ratio=0.01,
                 forward_fixed_ratio=0.01, coeff_ratio=0.1, share_noise=True, use_average_share=True):
        super().__init__()
        assert model.channels == model.out_dim
        assert not model.random_or_learned_sinusoidal_cond
        if objective not in _ffi.OBJECTIVES:
            raise ValueError("objective must be either pred_noise (predict noise) or pred_x0 (predict image start) or pred_v (predict v)")
        if schedule_fn_kwargs or min_snr_loss_weight:
            raise NotImplementedError("schedule_fn_kwargs / min_snr_loss_weight only matter for training")
        self.model = model
        self.channels = model.channels
        self.self_condition = False
        self.frames, self.cond_frames = frames, cond_frames
        self.image_size = image_size
        self.objective = objective
        self.diffuse_cond = diffuse_cond
        self.backward_steps, self.backward_lr = backward_steps, backward_lr
        self.standard_fixed_ratio, self.forward_fixed_ratio = standard_fixed_ratio, forward_fixed_ratio
        self.coeff_ratio = coeff_ratio
        self.share_noise, self.use_average_share = share_noise, use_average_share
        assert self.channels == frames * 3 + 3, "channels must be frames * 3 + 3 (states + boundary mask/offsets)"
        assert image_size == model.image_size, "image_size must match the Unet's launch plan"
        tables = make_schedule(beta_schedule, timesteps, objective)
        self.num_timesteps = int(timesteps)
        # the model's time path is a table with model.timesteps rows (the reference evaluates its time MLP per call)
        mt = getattr(model, "timesteps", None)
        if mt is not None and int(mt) < int(timesteps):
            raise ValueError(f"model was built with timesteps={mt} < diffusion timesteps={timesteps}: pass timesteps={timesteps} to the model")
        self.loss_type = loss_type
        self.sampling_timesteps = sampling_timesteps if sampling_timesteps is not None else timesteps
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        for name in _ffi.SCHED_NAMES:               # same buffer names as the reference (:626-674)
            self.register_buffer(name, tables[name])
        self._h = None
        self._tab_sig = None
        self._ws = None

    def __del__(self):
        h = self.__dict__.get("_h")
        if h is not None and h.value:
            try:
                _ffi.lib().cindm_ddpm1d_destroy(h)
            except Exception:
                pass
            self.__dict__["_h"] = None

    def _share_mode(self):
        """The library's ``use_average_share`` word: bit 0 = mean (1) / sum (0) over the boundary copies of a design, bit 1 =
        share_noise False -- the clamped x_start and the posterior mean are shared instead of the prediction (:757-773); bits 4-5 =
        the objective (pred_noise / pred_x0 / pred_v, :743-753)."""
        return int(bool(self.use_average_share)) | (0 if self.share_noise else 2) | (_ffi.OBJECTIVES[self.objective] << 4)

    # ------------------------------------------------------------------ library handle
    def _handle(self):
        sig = tuple((getattr(self, n).data_ptr(), getattr(self, n)._version) for n in _ffi.SCHED_NAMES)
        if self._h is not None and sig == self._tab_sig:
            return self._h
        L = _ffi.lib()
        if self._h is not None:
            L.cindm_ddpm1d_destroy(self._h)
        dev = self.betas.device
        if dev.type != "cuda":
            raise _ffi.CindmError("GaussianDiffusion is on the CPU: move it to a ROCm device (.to('cuda')); "
                                  "there is no CPU execution path")
        d = _ffi.SchedDesc()
        d.timesteps = self.num_timesteps
        keep = []
        for n in _ffi.SCHED_NAMES:
            t = getattr(self, n).detach().to("cpu", torch.float32).contiguous()
            keep.append(t)
            setattr(d, n, t.data_ptr())
        h = C.c_void_p()
        with torch.cuda.device(dev):
            _ffi.check(L.cindm_ddpm1d_create(C.byref(d), C.byref(h)))
        self._h, self._tab_sig = h, sig
        return h

    def _prepare(self, images, device):
        self.model.sync_weights()
        h = self._handle()
        nbytes = _ffi.lib().cindm_ddpm2d_workspace_bytes(self.model._h, images)
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return h, self._ws

    # ------------------------------------------------------------------ reference-named helpers (plumbing)
    def predict_start_from_noise(self, x_t, t, noise):
        return self.sqrt_recip_alphas_cumprod[t].view(-1, 1, 1, 1) * x_t - self.sqrt_recipm1_alphas_cumprod[t].view(-1, 1, 1, 1) * noise

    def q_posterior(self, x_start, x_t, t):
        mean = self.posterior_mean_coef1[t].view(-1, 1, 1, 1) * x_start + self.posterior_mean_coef2[t].view(-1, 1, 1, 1) * x_t
        return mean, self.posterior_variance[t].view(-1, 1, 1, 1), self.posterior_log_variance_clipped[t].view(-1, 1, 1, 1)

    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        return self.sqrt_alphas_cumprod[t].view(-1, 1, 1, 1) * x_start + self.sqrt_one_minus_alphas_cumprod[t].view(-1, 1, 1, 1) * noise

    def share_states_over_boundaries(self, shape, x, use_average_share=True):
        """:712-725 (torch; the library does this inside its update kernel -- this is for callers' own tensors)."""
        s = x[:, :-3].reshape(shape[0], shape[1], shape[2] - 3, shape[3], shape[4])
        s = s.mean(dim=1, keepdim=True) if use_average_share else s.sum(dim=1, keepdim=True)
        x[:, :-3] = s.expand(-1, shape[1], -1, -1, -1).reshape(shape[0] * shape[1], shape[2] - 3, shape[3], shape[4])
        return x

    def sample_noise(self, shape, device):
        """:775-785."""
        state = torch.randn((shape[0], 1, shape[2] - 3, shape[3], shape[4]), device=device)
        boundary = torch.randn((shape[0], shape[1], 3, shape[3], shape[4]), device=device)
        return torch.cat([state.expand(-1, shape[1], -1, -1, -1), boundary], dim=2)

    @staticmethod
    def _t_int(t):
        return int(t.reshape(-1)[0]) if torch.is_tensor(t) else int(t)

    # ------------------------------------------------------------------ one reverse step
    @torch.no_grad()
    def _step(self, shape, x, t, clip_denoised, noise, add_noise=True):
        """(x_{t-1} without guidance, x_start, model_mean), all [B*nb, C, H, W]."""
        if not x.is_cuda:
            raise _ffi.CindmError("sampling needs ROCm device tensors; there is no CPU execution path")
        B, nb, Cc, H, W = shape
        cp = self.model.padded_channels
        xd = to_device_layout(x.float(), cp)
        x0 = torch.zeros_like(xd)
        mean = torch.zeros_like(xd)
        h, ws = self._prepare(B * nb, x.device)
        ns = nbnd = None
        if noise is not None and t > 0:
            nz = noise.reshape(B, nb, Cc, H, W).to(x.device, torch.float32)
            ns = _state_cl(nz[:, :1, :-3])
            nbnd = _boundary_cl(nz[:, :, -3:])
        elif t > 0 and add_noise:
            nz = self.sample_noise(shape, x.device)
            ns = _state_cl(nz[:, :1, :-3])
            nbnd = _boundary_cl(nz[:, :, -3:])
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().cindm_ddpm2d_step(h, self.model._h, _ffi.ptr(xd), B, nb, self._share_mode(),
                                                    int(clip_denoised), _ffi.ptr(ns), _ffi.ptr(nbnd), 0, 0, int(t), None,
                                                    _ffi.ptr(x0), _ffi.ptr(mean), _ffi.ptr(ws), ws.numel(),
                                                    _ffi.current_stream(x.device)))
        f = lambda y: from_device_layout(y, Cc, H, W)
        return f(xd), f(x0), f(mean)

    @torch.no_grad()
    def model_predictions(self, shape, x, t, x_self_cond=None, clip_x_start=False, rederive_pred_noise=False, share_noise=True):
        """:727-754 (all three objectives).  x [B*nb, C, H, W] -> ModelPrediction(pred_noise, pred_x_start): the Unet's output
        with its state channels shared over the boundary copies of a design (``share_noise``; mean or sum by
        ``use_average_share``), x_start = predict_start_from_noise (clamped when ``clip_x_start``), and with
        ``clip_x_start and rederive_pred_noise`` the noise re-derived from the clamped x_start.  One library call
        (``cindm_ddpm2d_predict``: U-Net + one element-wise kernel)."""
        if not x.is_cuda:
            raise _ffi.CindmError("sampling needs ROCm device tensors; there is no CPU execution path")
        if x_self_cond is not None:
            raise NotImplementedError("self-conditioning is outside this build's scope")
        B, nb, Cc, H, W = shape
        ti = self._t_int(t)
        cp = self.model.padded_channels
        xd = to_device_layout(x.float(), cp)
        eps, x0 = torch.zeros_like(xd), torch.zeros_like(xd)
        h, ws = self._prepare(B * nb, x.device)
        with torch.cuda.device(x.device):
            _ffi.check(_ffi.lib().cindm_ddpm2d_predict(h, self.model._h, _ffi.ptr(xd), B, nb,
                                                       int(bool(self.use_average_share)) | (_ffi.OBJECTIVES[self.objective] << 4),
                                                       int(bool(share_noise)), int(bool(clip_x_start)), int(bool(rederive_pred_noise)),
                                                       ti, None, _ffi.ptr(eps), _ffi.ptr(x0), _ffi.ptr(ws), ws.numel(),
                                                       _ffi.current_stream(x.device)))
        return ModelPrediction(from_device_layout(eps, Cc, H, W), from_device_layout(x0, Cc, H, W))

    @torch.no_grad()
    def p_mean_variance(self, shape, x, t, x_self_cond=None, clip_denoised=True):
        """:757-773.  Returns (model_mean, posterior_variance, posterior_log_variance, x_start)."""
        ti = self._t_int(t)
        _, x0, mean = self._step(shape, x, ti, clip_denoised, None, add_noise=False)
        return mean, self.posterior_variance[ti].view(1, 1, 1, 1), self.posterior_log_variance_clipped[ti].view(1, 1, 1, 1), x0

    @torch.no_grad()
    def p_sample(self, shape, x, t: int, x_self_cond=None, clip_denoised=True, design_fn=None, design_guidance="standard",
                 *, noise=None, recur_noise=None):
        """:788-889.  x [B*nb, C, H, W]; returns (x_{t-1}, x_start).  ``recur_noise`` [R, B*nb, C, H, W]: the relaxation
        draws of the "-recurrence-N" branch (else ``sample_noise``)."""
        t = int(t)
        if "recurrence" in design_guidance:
            # :846-889, literally: the posterior mean is computed once; every iteration subtracts the raw design gradient
            # taken at the current relaxed x (model_mean - grad_design) and re-noises; design_fn is required there
            if design_fn is None:
                raise ValueError("the 2-D recurrence guidance needs design_fn (the reference dereferences its gradient)")
            R = int(design_guidance.split("-")[-1])
            _, x_start, mean = self._step(shape, x, t, clip_denoised, None, add_noise=False)
            ratio = self.alphas_cumprod / self.alphas_cumprod_prev
            pred = mean
            xc = x.float()
            for r in range(R):
                with torch.enable_grad():
                    if design_guidance.startswith("standard"):
                        g = design_fn(xc.clone().detach().requires_grad_()).detach()
                    elif design_guidance.startswith("universal-forward-recurrence"):
                        g = design_fn(x_start.clone().detach().requires_grad_()).detach()
                    else:
                        raise NotImplementedError(design_guidance)
                pred = mean - g
                z = recur_noise[r].to(x.device) if recur_noise is not None else \
                    self.sample_noise(shape, x.device).view(-1, shape[2], shape[3], shape[4])
                xc = torch.sqrt(ratio)[t] * pred + torch.sqrt(1 - ratio)[t] * z
            if t > 0:
                z = noise.to(x.device) if noise is not None else self.sample_noise(shape, x.device).view(-1, shape[2], shape[3], shape[4])
                pred = pred + (0.5 * self.posterior_log_variance_clipped[t]).exp() * z
            return pred, x_start
        pred, x_start, _ = self._step(shape, x, t, clip_denoised, noise)
        if design_fn is None:
            return pred, x_start
        eta = (self.coeff_ratio * self.betas.flip(0))[t]

        def grad_of(z):
            with torch.enable_grad():
                return design_fn(z.clone().detach().requires_grad_()).detach()

        if design_guidance == "standard":
            shift = self.standard_fixed_ratio * grad_of(x)
        elif design_guidance == "standard-alpha":
            shift = eta * grad_of(x)
        elif design_guidance == "universal-forward":
            shift = self.forward_fixed_ratio * grad_of(x_start)
        elif design_guidance == "universal-backward":
            xc, shift = x_start.clone(), None
            for kk in range(self.backward_steps):
                gd = grad_of(xc)
                if kk == 1:
                    shift = self.forward_fixed_ratio * gd
                xc = xc - gd * self.backward_lr
            coef = (self.sqrt_alphas_cumprod * self.betas / (torch.sqrt(1 - self.betas) * (1 - self.alphas_cumprod)))[t]
            shift = shift - coef * (xc - x_start)
        else:
            raise ValueError(design_guidance)
        return pred - shift, x_start

    # ------------------------------------------------------------------ loops
    @torch.no_grad()
    def p_sample_loop(self, shape, design_fn=None, design_guidance="standard", return_all_timesteps=None, *,
                      noise=None, seed=0, sample_offset=0, use_graph=True, t_stop=0, device=None, fused=True):
        """:893-907.  Returns [B, nb, C, H, W].  ``noise``: a NoiseTape2D (parity runs); otherwise x_T and the
        per-step draws come from the library's counter-based generator keyed by (seed, sample_offset + design)."""
        B, nb, Cc, H, W = shape
        device = device or self.betas.device
        if device.type != "cuda":
            raise _ffi.CindmError("sampling needs a ROCm device; there is no CPU execution path")
        cp = self.model.padded_channels
        L = _ffi.lib()
        T = self.num_timesteps
        if noise is not None:
            x = to_device_layout(torch.cat([noise.init[0].expand(-1, nb, -1, -1, -1), noise.init[1]], dim=2)
                                 .reshape(B * nb, Cc, H, W).to(device, torch.float32), cp)
        else:
            x = torch.empty((B * nb, H * W, cp), dtype=torch.float32, device=device)
            with torch.cuda.device(device):
                _ffi.check(L.cindm_fill_noise2d(_ffi.ptr(x), B, nb, H * W, Cc, cp, seed, sample_offset, T,
                                                _ffi.current_stream(device)))
        if design_fn is None:
            h, ws = self._prepare(B * nb, device)
            ns = nbnd = None
            if noise is not None:
                ns = _state_cl(noise.step_state.to(device, torch.float32))
                nbnd = _boundary_cl(noise.step_boundary.to(device, torch.float32))
            with torch.cuda.device(device):
                _ffi.check(L.cindm_ddpm2d_sample(h, self.model._h, _ffi.ptr(x), B, nb, self._share_mode(),
                                                 _ffi.ptr(ns), _ffi.ptr(nbnd), seed, sample_offset, T - 1, int(t_stop),
                                                 _ffi.ptr(ws), ws.numel(), _ffi.current_stream(device), int(use_graph)))
            return from_device_layout(x, Cc, H, W).reshape(B, nb, Cc, H, W)
        from .forceunet import ForceObjective
        if isinstance(design_fn, ForceObjective) and design_guidance == "standard-alpha" and fused:
            # the library's own objective: surrogate forward + input gradient, the reverse step and the guidance shift are
            # ONE captured graph per timestep (cindm_ddpm2d_sample_force), as PointObjective is in the 1-D path
            fo = design_fn
            if (fo.B, fo.nb) != (B, nb) or Cc != 3 * fo.frames + 3:
                raise ValueError("ForceObjective was built for another batch / boundary / frame count")
            fo.model.sync_weights()
            h, ws = self._prepare(B * nb, device)
            nfb = L.cindm_airfoil_design_workspace_bytes(fo.model._h, B, nb, fo.frames_per_pass)
            wsf = torch.empty(nfb, dtype=torch.uint8, device=device)
            g = torch.empty_like(x)
            eta = (self.coeff_ratio * self.betas.flip(0)).to(device, torch.float32).contiguous()
            ns = nbnd = None
            if noise is not None:
                ns = _state_cl(noise.step_state.to(device, torch.float32))
                nbnd = _boundary_cl(noise.step_boundary.to(device, torch.float32))
            with torch.cuda.device(device):
                _ffi.check(L.cindm_ddpm2d_sample_force(h, self.model._h, fo.model._h, _ffi.ptr(x), B, nb, self._share_mode(),
                                                       _ffi.ptr(ns), _ffi.ptr(nbnd), seed, sample_offset, T - 1, int(t_stop),
                                                       fo.frames, fo.p_min, fo.p_max, fo.lambda_force, fo.lambda_overlap, fo.factor,
                                                       int(fo.sum_boundary), _ffi.ptr(eta), _ffi.ptr(g), _ffi.ptr(ws), ws.numel(), _ffi.ptr(wsf),
                                                       wsf.numel(), _ffi.current_stream(device), int(use_graph)))
            return from_device_layout(x, Cc, H, W).reshape(B, nb, Cc, H, W)
        img = from_device_layout(x, Cc, H, W)
        for t in reversed(range(int(t_stop), T)):
            nz = None
            if noise is not None and t > 0:
                nz = torch.cat([noise.step_state[t].expand(-1, nb, -1, -1, -1), noise.step_boundary[t]], dim=2)
            img, _ = self.p_sample(shape, img, t, None, design_fn=design_fn, design_guidance=design_guidance, noise=nz)
        return img.reshape(B, nb, Cc, H, W)

    def ddim_sample(self, *a, **k):
        raise NotImplementedError("DDIM sampling is out of scope (SURVEY.md section 8)")

    @torch.no_grad()
    def sample(self, batch_size=16, design_fn=None, design_guidance="standard", num_boundaries=1,
               return_all_timesteps=False, **kw):
        """:960-963."""
        if self.is_ddim_sampling:
            return self.ddim_sample()
        S = self.image_size
        return self.p_sample_loop((batch_size, num_boundaries, self.channels, S, S), design_fn, design_guidance,
                                  return_all_timesteps=return_all_timesteps, **kw)

    def forward(self, *a, **k):
        raise NotImplementedError("training (p_losses) is out of this build's scope (SURVEY.md section 8)")
