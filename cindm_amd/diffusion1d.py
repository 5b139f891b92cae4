"""GaussianDiffusion1D (sampling half) on MI355X: the reference's constructor / buffers / ``sample``
surface (model/diffusion_1d.py:801-2406 of AI4Science-WestlakeU/cindm) over the HIP library.

What runs where
  * every U-Net evaluation, the window / body-pair gather, the score composition, x0 prediction,
    clamp, posterior mean and the noise add run in ``libcindm_hip.so``;
  * with ``design_fn=None`` (and no recurrence / overwrite) the whole reverse loop is ONE call
    (``cindm_ddpm1d_sample``: one hipGraph-captured step replayed per timestep);
  * ``design_fn`` guidance is a user Python callable: its gradient is taken by PyTorch autograd
    between two library calls per step, exactly where the reference takes it.
  * DDIM (``sampling_timesteps < timesteps``, ``ddim_sample`` :1724-1804): unguided = ONE library call
    (``cindm_ddpm1d_sample_ddim``, same captured-step replay with per-step coefficient tables); guided (recurrence
    guidance) = library predictions + the user's gradient per step.
Training (``forward`` / ``p_losses``) and the unreachable ULA/UHMC samplers of the reference are out of this build's
scope and raise NotImplementedError.
"""
import ctypes as C
from collections import namedtuple

import torch
from torch import nn

from . import _ffi
from .objectives import PointObjective
from .schedule import make_schedule

ModelPrediction = namedtuple("ModelPrediction", ["pred_noise", "pred_x_start"])


class NoiseTape:
    """Explicit noise for parity runs, replacing the reference's ``torch.randn`` draws:
    ``init`` [B,L,F] (x_T, :1673 / :1987); ``step`` [T,B,L,F] indexed by timestep (:1281 / :1118);
    ``recur`` [T,R,B,L,F] relaxation draws (:1365); ``cond`` [T,B,Lc,F] inpainting draws (:1717)."""

    def __init__(self, init, step, recur=None, cond=None):
        self.init, self.step, self.recur, self.cond = init, step, recur, cond

    def to(self, device):
        f = lambda t: None if t is None else t.to(device=device, dtype=torch.float32).contiguous()
        return NoiseTape(f(self.init), f(self.step), f(self.recur), f(self.cond))


def _exists(x):
    return x is not None


class GaussianDiffusion1D(nn.Module):
    """Drop-in for the reference's ``GaussianDiffusion1D`` (constructor :802-822)."""

    def __init__(self, model, model_unconditioned=None, betas_inference=None, *, image_size, conditioned_steps,
                 timesteps=1000, sampling_timesteps=None, loss_type="l1", objective="pred_noise",
                 beta_schedule="cosine", ddim_sampling_eta=0., auto_normalize=True, loss_weight_discount=0.95,
                 num_time_steps_UHMC=100, is_diffusion_condition=None, backward_steps=5, backward_lr=1):
        super().__init__()
        self.model = model
        self.model_unconditioned = model_unconditioned
        self.betas_inference = betas_inference
        self.channels = self.model.channels
        self.is_diffusion_condition = is_diffusion_condition
        self.self_condition = False
        self.num_timesteps_UHMC = num_time_steps_UHMC
        self.image_size = image_size
        self.conditioned_steps = conditioned_steps
        self.rollout_steps = image_size
        self.objective = objective
        self.backward_steps = backward_steps
        self.backward_lr = backward_lr
        assert objective in {"pred_noise", "pred_x0", "pred_v"}, "objective must be pred_noise, pred_x0 or pred_v"
        tables = make_schedule(beta_schedule, timesteps, objective)
        self.num_timesteps = int(timesteps)
        # the model's time path is a table with model.timesteps rows (the reference evaluates its time MLP per call)
        mt = getattr(model, "timesteps", None)
        if mt is not None and int(mt) < int(timesteps):
            raise ValueError(f"model was built with timesteps={mt} < diffusion timesteps={timesteps}: pass timesteps={timesteps} to the model")
        self.loss_type = loss_type
        self.loss_weight_discount = loss_weight_discount
        self.sampling_timesteps = sampling_timesteps if _exists(sampling_timesteps) else timesteps
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        for name in _ffi.SCHED_NAMES:               # the 13 buffers, same names and order (:873-910)
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

    # ------------------------------------------------------------------ library handle
    def _handle(self):
        sig = tuple((getattr(self, n).data_ptr(), getattr(self, n)._version) for n in _ffi.SCHED_NAMES)
        if self._h is not None and sig == self._tab_sig:
            return self._h
        L = _ffi.lib()
        if self._h is not None:
            L.cindm_ddpm1d_destroy(self._h)
        d = _ffi.SchedDesc()
        d.timesteps = self.num_timesteps
        keep = []
        for n in _ffi.SCHED_NAMES:
            t = getattr(self, n).detach().to("cpu", torch.float32).contiguous()
            keep.append(t)
            setattr(d, n, t.data_ptr())
        h = C.c_void_p()
        dev = self.betas.device
        if dev.type != "cuda":
            raise _ffi.CindmError("GaussianDiffusion1D is on the CPU: move it to a ROCm device (.to('cuda')); "
                                  "there is no CPU execution path")
        with torch.cuda.device(dev):
            _ffi.check(L.cindm_ddpm1d_create(C.byref(d), C.byref(h)))
        self._h, self._tab_sig = h, sig
        return h

    def _compose_desc(self, mode, n_composed, compose_start_step, window, n_bodies, clip=True, uncond_coef=1.4):
        c = _ffi.ComposeDesc()
        c.mode, c.n_windows, c.compose_start_step, c.window = mode, n_composed + 1, compose_start_step, window
        c.n_bodies, c.cond_steps = n_bodies, self.conditioned_steps
        # (coefficient_unconditioned_grad: 1.4 in the 4-body branch of gradient(), :1900; the 3-body branch subtracts the plain prediction, :1958)
        c.objective, c.clip_denoised, c.uncond_coef = _ffi.OBJECTIVES[self.objective], int(clip), uncond_coef
        return c

    def _desc_for(self, x_shape, compose_mode=None, n_composed=0, compose_start_step=4, single_model_step=-1,
                  compose_n_bodies=2, clip=True, outside=False):
        """Maps the reference's keyword soup onto a compose descriptor."""
        if outside:
            if compose_mode == "mean":
                mode = _ffi.COMPOSE_MEAN_OUTSIDE
            elif compose_mode == "noise_sum":
                mode = _ffi.COMPOSE_NOISESUM_OUTSIDE
            else:
                raise ValueError(f"unknown compose_mode {compose_mode!r}")
            return self._compose_desc(mode, n_composed, compose_start_step, single_model_step, compose_n_bodies, clip)
        if compose_mode is not None and "inside" in compose_mode:
            if compose_mode == "mean-inside":
                mode = _ffi.COMPOSE_MEAN_INSIDE
            elif compose_mode == "sum-inside":
                mode = _ffi.COMPOSE_SUM_INSIDE
            else:
                raise ValueError(f"unknown compose_mode {compose_mode!r}")
            return self._compose_desc(mode, n_composed, compose_start_step, single_model_step, compose_n_bodies, clip)
        nb = x_shape[-1] // 4
        if self.model_unconditioned is not None:           # :1003-1004 -> gradient()
            if nb != 4:
                raise NotImplementedError("model_predictions calls gradient(x, t, 4) whatever the state's width (model/diffusion_1d.py:1004): "
                                          "with model_unconditioned set the state must hold 4 bodies (the 3-body branch: gradient(x_t, t, 3))")
            return self._compose_desc(_ffi.COMPOSE_MULTIBODY, 0, 0, self.model.horizon, nb, clip)
        return self._compose_desc(_ffi.COMPOSE_PLAIN, 0, 0, self.model.horizon, nb, clip)

    def _prepare(self, desc, B, device):
        self.model.sync_weights()
        if desc.mode == _ffi.COMPOSE_MULTIBODY:
            self.model_unconditioned.sync_weights()
        L = _ffi.lib()
        h = self._handle()
        un = self.model_unconditioned._h if desc.mode == _ffi.COMPOSE_MULTIBODY else None
        nbytes = L.cindm_ddpm1d_workspace_bytes(h, self.model._h, un, C.byref(desc), B)
        if nbytes == 0:
            raise _ffi.CindmError(L.cindm_last_error().decode() or "invalid composition")
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return h, un, self._ws

    _warned_crowded = False

    def last_chain_info(self):
        """What the last library chain (``sample`` / ``ddim_sample`` / the built-in guided loop) of this object did:
        ``recovered`` -- an in-kernel exchange timed out and the chain was re-run once on the exchange-free plan;
        ``exchange_free_up_front`` -- another chain was already in flight on this device in this process, so this one ran on the
        exchange-free plan from the start (correct, about 10 % slower); ``chains_in_flight`` when it started, itself included."""
        info = (C.c_int32 * 4)()
        _ffi.check(_ffi.lib().cindm_ddpm1d_last_chain_info(self._handle(), info))
        return {"recovered": bool(info[0]), "exchange_free_up_front": bool(info[1]), "chains_in_flight": int(info[2]),
                "range_fallback": int(info[3])}

    def _chain(self, img, desc, call):
        """One library chain over ``img`` (in place): ``call(h, un, ws)`` issues it.  The FIRST chain after a weight synchronisation
        also carries the range rule on the caller's own data (TemporalUnet1D.range_guard): x_T is kept, the designs are
        checked once, and a chain that came back inf / nan from the split-fp16 kernels is repeated on the exact fp32-MFMA kernels."""
        device, B = img.device, img.shape[0]
        h, un, ws = self._prepare(desc, B, device)
        models = [self.model] + ([self.model_unconditioned] if un is not None else [])
        pend = [m for m in models if m.range_guard_pending()]
        x0 = img.clone() if pend else None
        call(h, un, ws)
        self._note_chain()

        def rerun():
            img.copy_(x0)
            call(*self._prepare(desc, B, device))        # (the fp32 plan sizes its own workspace)
            self._note_chain()
        for m in pend:
            m.range_guard(lambda: img, rerun, device)
        return img

    def _note_chain(self):
        info = self.last_chain_info()
        if info["exchange_free_up_front"] and not GaussianDiffusion1D._warned_crowded:
            GaussianDiffusion1D._warned_crowded = True
            import warnings
            warnings.warn("cindm_amd: a sampling chain started while another was in flight on the same device; the fast kernels exchange "
                          "data between co-resident workgroups and are built for ONE chain per device, so this chain ran on the "
                          "exchange-free plan (correct, slower).  Run chains one after the other, or one process per GPU.", RuntimeWarning)
        return info

    def last_step_info(self):
        """(kernel launches, update fused into the U-Net's last kernel?) of the reverse step emitted last -- as launched
        by the library, not a model of it (``cindm_ddpm1d_last_step_info``)."""
        n, f = C.c_int32(), C.c_int32()
        _ffi.check(_ffi.lib().cindm_ddpm1d_last_step_info(self._handle(), C.byref(n), C.byref(f)))
        return int(n.value), bool(f.value)

    @staticmethod
    def _f32(t, device=None):
        return None if t is None else t.detach().to(device=device or t.device, dtype=torch.float32).contiguous()

    # ------------------------------------------------------------------ reference-named helpers
    def predict_start_from_noise(self, x_t, t, noise):
        return self.sqrt_recip_alphas_cumprod[t].view(-1, 1, 1) * x_t - self.sqrt_recipm1_alphas_cumprod[t].view(-1, 1, 1) * noise

    def q_posterior(self, x_start, x_t, t):
        mean = self.posterior_mean_coef1[t].view(-1, 1, 1) * x_start + self.posterior_mean_coef2[t].view(-1, 1, 1) * x_t
        return mean, self.posterior_variance[t].view(-1, 1, 1), self.posterior_log_variance_clipped[t].view(-1, 1, 1)

    def q_sample(self, x_start, t, noise=None):
        """:2399-2406 (a two-table elementwise expression on the state; plumbing, not on the hot loop)."""
        if noise is None:
            noise = torch.randn_like(x_start)
        return self.sqrt_alphas_cumprod[t].view(-1, 1, 1) * x_start + self.sqrt_one_minus_alphas_cumprod[t].view(-1, 1, 1) * noise

    @staticmethod
    def _t_int(t):
        if torch.is_tensor(t):
            return int(t.reshape(-1)[0])
        return int(t)

    def _models(self, desc):
        return [self.model] + ([self.model_unconditioned] if desc.mode == _ffi.COMPOSE_MULTIBODY else [])

    def _timed_out(self, desc, device):
        """True when an in-kernel exchange of the U-Nets a step with ``desc`` runs timed out since the last poll (every
        model's flag is read and cleared); synchronises.  Raises instead when the model opted out of the recovery."""
        hit, opted_out = False, False
        for m in self._models(desc):                 # every model's flag is read (and cleared) before anything is raised
            rc = m.poll_raw(device)
            hit = hit or rc
            opted_out = opted_out or (rc and not m.recover_exchange_timeouts)
        if opted_out:
            raise _ffi.CindmError(self.model.TIMEOUT_TEXT)
        return hit

    def _rerun_exchange_free(self, fn, desc, device):
        """``fn()`` once more with the step's U-Nets on their exchange-free kernels (TemporalUnet1D.rerun_exchange_free for
        a step that may run two models)."""
        ms = self._models(desc)
        for m in ms:
            m.exchange_free(True)
        try:
            out = fn()
            if self._timed_out(desc, device):
                raise _ffi.CindmError(self.model.TIMEOUT_TEXT)
        finally:
            for m in ms:
                m.exchange_free(False)
        for m in ms:
            m._py_recovered = getattr(m, "_py_recovered", 0) + 1
        return out

    @torch.no_grad()
    def _predict(self, x, cond, t, desc, check=True):
        """(model_mean, x_start, pred_noise) = p_mean_variance(x, cond, t, ...) (:1033-1044) on the device.
        ``check``: read the U-Nets' exchange flags before returning (the Python loops pass False and check once at
        their end: the predictions of a step only feed the next one until then)."""
        if not x.is_cuda:
            raise _ffi.CindmError("sampling needs ROCm device tensors; there is no CPU execution path")
        x = self._f32(x)
        cond_d = self._f32(cond, x.device) if (cond is not None and self.conditioned_steps != 0) else None
        B = x.shape[0]
        h, un, ws = self._prepare(desc, B, x.device)
        mean, x0, eps = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)

        def launch():
            with torch.cuda.device(x.device):
                _ffi.check(_ffi.lib().cindm_ddpm1d_predict(h, self.model._h, un, C.byref(desc), _ffi.ptr(x), _ffi.ptr(cond_d),
                                                           int(t), None, B, _ffi.ptr(mean), _ffi.ptr(x0), _ffi.ptr(eps),
                                                           _ffi.ptr(ws), ws.numel(), _ffi.current_stream(x.device)))

        launch()
        if check and self._timed_out(desc, x.device):
            self._rerun_exchange_free(launch, desc, x.device)
        return mean, x0, eps

    def model_predictions(self, x, cond, t, x_self_cond=None, clip_x_start=False, rederive_pred_noise=False, **kwargs):
        """:951-1031.  Returns ModelPrediction(pred_noise, pred_x_start) (x_start unclamped unless clip_x_start)."""
        desc = self._desc_for(x.shape, kwargs.get("compose_mode"), kwargs.get("n_composed", 0),
                              kwargs.get("compose_start_step", 4), kwargs.get("single_model_step", -1),
                              kwargs.get("compose_n_bodies", 2), clip=clip_x_start)
        _, x0, eps = self._predict(x, cond, self._t_int(t), desc)
        return ModelPrediction(eps, x0)

    def p_mean_variance(self, x, cond, t, x_self_cond=None, clip_denoised=True, **kwargs):
        """:1033-1044.  Returns (model_mean, posterior_variance, posterior_log_variance, x_start, pred_noise)."""
        ti = self._t_int(t)
        desc = self._desc_for(x.shape, kwargs.get("compose_mode"), kwargs.get("n_composed", 0),
                              kwargs.get("compose_start_step", 4), kwargs.get("single_model_step", -1),
                              kwargs.get("compose_n_bodies", 2), clip=clip_denoised)
        mean, x0, eps = self._predict(x, cond, ti, desc)
        return mean, self.posterior_variance[ti].view(1, 1, 1), self.posterior_log_variance_clipped[ti].view(1, 1, 1), x0, eps

    @torch.no_grad()
    def gradient(self, x_t, t, n_bodies, scalar_for_gradient=None):
        """:1857-1982 (t <= 400): pair + unconditioned composition of eps.  ``n_bodies == 4`` (:1865-1926): six pairs, the single-body
        predictions weighted by 1.4 -- what ``model_predictions`` calls (:1004).  ``n_bodies == 3`` (:1927-1982, reached only by a direct
        call): three pairs, weight 1; the reference slices its batched pair output with the literal bounds 0:20 / 20:40 / 40:60, so its
        branch is defined for a batch of 20 only -- here any batch gives what batch 20 gives there (golden at 20:
        tests/golden/gradient3_1d_r6.npz)."""
        if n_bodies not in (3, 4) or self.model_unconditioned is None:
            raise NotImplementedError("gradient(): n_bodies must be 3 or 4, with model_unconditioned set")
        if x_t.shape[-1] != 4 * n_bodies:
            raise ValueError(f"gradient(): x_t has {x_t.shape[-1]} features, n_bodies = {n_bodies} needs {4 * n_bodies}")
        ti = self._t_int(t)
        if ti > 400:
            raise NotImplementedError("gradient(): t > 400 dereferences scalar_for_gradient (unreachable for N <= 401)")
        desc = self._compose_desc(_ffi.COMPOSE_MULTIBODY, 0, 0, self.model.horizon, n_bodies, clip=False,
                                  uncond_coef=1.4 if n_bodies == 4 else 1.0)
        desc.cond_steps = 0                             # x_t here already is cat(cond, x)
        _, _, eps = self._predict(x_t, None, ti, desc)
        return eps

    # ------------------------------------------------------------------ one reverse step
    def _design_shift(self, design_fn, design_guidance, x, x_start, t):
        """Gradient term of the guided step (:1072-1106 / :1235-1269 / :1468-1512), PyTorch autograd on the
        user's callable."""
        eta = self.betas[t] / torch.sqrt(self.alphas_cumprod_prev)[t]
        g = design_guidance

        def grad_of(z):
            with torch.enable_grad():
                zc = z.clone().detach().requires_grad_()
                return torch.autograd.grad(design_fn(zc), zc)[0]

        if g.startswith("standard"):
            gd = grad_of(x)
            if g == "standard" or g.startswith("standard-recurrence"):
                return gd
            if g == "standard-alpha" or g.startswith("standard-alpha-recurrence"):
                return eta * gd
            raise ValueError(g)
        if g.startswith("universal-forward"):
            gd = grad_of(x_start)
            return gd if "pure" in g else eta * gd
        if g.startswith("universal-backward"):
            xc, final = x_start.clone(), None
            for kk in range(self.backward_steps):
                gd = grad_of(xc)
                if kk == 1:
                    final = gd if "pure" in g else eta * gd
                xc = xc - gd * self.backward_lr
            coef = (self.sqrt_alphas_cumprod * self.betas / (torch.sqrt(1 - self.betas) * (1 - self.alphas_cumprod)))[t]
            return final - coef * (xc - x_start)
        raise ValueError(g)

    @torch.no_grad()
    def _guided_step(self, x, cond, t, desc, design_fn, design_guidance, initial_state_overwrite, noise, recur_noise,
                     ddim_return=False, check=True):
        """Shared body of p_sample / p_sample_compose_inside / p_sample_compose_outside.  ``ddim_return``: the
        sampling_timesteps != timesteps convention of the recurrence branch (:1372-1376): returns
        (pred_noise + grad_design_final, x_start) of the last iteration."""
        R = int(design_guidance.split("-")[-1]) if "recurrence" in design_guidance else 0
        x = self._f32(x)
        logvar = self.posterior_log_variance_clipped[t]
        x_start = None
        eps = shift = None
        for r in range(max(R, 1)):
            mean, x_start, eps = self._predict(x, cond, t, desc, check=check)
            pred = mean
            if design_fn is not None:
                shift = self._design_shift(design_fn, design_guidance, x, x_start, t)
                pred = mean - shift
            if initial_state_overwrite is not None:
                k = initial_state_overwrite.shape[1]
                pred = torch.cat([initial_state_overwrite.to(pred), pred[:, k:]], 1)
            if R:
                ratio = self.alphas_cumprod / self.alphas_cumprod_prev
                z = recur_noise[r] if recur_noise is not None else torch.randn_like(pred)
                x = torch.sqrt(ratio)[t] * pred + torch.sqrt(1 - ratio)[t] * z
        if ddim_return:
            return eps + shift, x_start
        if t > 0:
            z = noise if noise is not None else torch.randn_like(x)
            pred = pred + (0.5 * logvar).exp() * z
        return pred, x_start

    @torch.no_grad()
    def p_sample(self, x, cond, t: int, x_self_cond=None, clip_denoised=True, design_fn=None,
                 design_guidance="standard", initial_state_overwrite=None, *, noise=None, recur_noise=None):
        """:1047-1186.  Returns (x_{t-1}, x_start)."""
        desc = self._desc_for(x.shape, None, clip=clip_denoised)
        return self._guided_step(x, cond, int(t), desc, design_fn, design_guidance, initial_state_overwrite, noise, recur_noise)

    @torch.no_grad()
    def p_sample_compose_inside(self, x, cond, t: int, x_self_cond=None, clip_denoised=True, design_fn=None,
                                design_guidance="standard", initial_state_overwrite=None, compose_mode="mean-inside",
                                n_composed=0, compose_start_step=4, single_model_step=-1, compose_n_bodies=2,
                                *, noise=None, recur_noise=None):
        """:1190-1376."""
        ddim = self.sampling_timesteps != self.num_timesteps and "recurrence" in design_guidance     # :1372-1376
        assert "inside" not in compose_mode or single_model_step > 0
        desc = self._desc_for(x.shape, compose_mode, n_composed, compose_start_step, single_model_step, compose_n_bodies,
                              clip=clip_denoised)
        return self._guided_step(x, cond, int(t), desc, design_fn, design_guidance, initial_state_overwrite, noise, recur_noise,
                                 ddim_return=ddim)

    @torch.no_grad()
    def p_sample_compose_outside(self, x, cond, t: int, x_self_cond=None, clip_denoised=True, design_fn=None,
                                 design_guidance="standard", compose_mode="mean", n_composed=0, compose_start_step=4,
                                 single_model_step=-1, compose_n_bodies=2, initial_state_overwrite=None,
                                 *, noise=None, recur_noise=None):
        """:1380-1652."""
        assert single_model_step > 0
        desc = self._desc_for(x.shape, compose_mode, n_composed, compose_start_step, single_model_step, compose_n_bodies,
                              clip=clip_denoised, outside=True)
        return self._guided_step(x, cond, int(t), desc, design_fn, design_guidance, initial_state_overwrite, noise, recur_noise)

    # ------------------------------------------------------------------ loops
    @torch.no_grad()
    def _run_loop(self, img, cond, desc, t_start, t_end, *, noise_steps, seed, sample_offset, inpaint_cond,
                  inpaint_noise_steps, use_graph=True):
        """The unguided reverse loop as one library call (cindm_ddpm1d_sample)."""
        B = img.shape[0]
        cond_d = self._f32(cond, img.device) if (cond is not None and self.conditioned_steps != 0) else None
        inp = self._f32(inpaint_cond, img.device)

        def call(h, un, ws):
            with torch.cuda.device(img.device):
                _ffi.check(_ffi.lib().cindm_ddpm1d_sample(
                    h, self.model._h, un, C.byref(desc), _ffi.ptr(img), _ffi.ptr(cond_d), _ffi.ptr(noise_steps),
                    C.c_uint64(seed), sample_offset, _ffi.ptr(inp), 0 if inp is None else inp.shape[1],
                    _ffi.ptr(inpaint_noise_steps), t_start, t_end, B, _ffi.ptr(ws), ws.numel(),
                    _ffi.current_stream(img.device), int(use_graph)))
        return self._chain(img, desc, call)

    @torch.no_grad()
    def _run_guided_loop(self, img, cond, desc, dz, t_start, t_end, *, noise, seed, sample_offset, inpaint_cond,
                         initial_state_overwrite, use_graph=True):
        """Reverse steps t_start .. t_end guided by the built-in objective as one library call
        (cindm_ddpm1d_sample_guided); ``noise`` rows (step / recur / cond) are indexed by t."""
        device, B = img.device, img.shape[0]
        cond_d = self._f32(cond, device) if (cond is not None and self.conditioned_steps != 0) else None
        inp = self._f32(inpaint_cond, device)
        iso = self._f32(initial_state_overwrite, device)

        def call(h, un, ws):
            with torch.cuda.device(device):
                _ffi.check(_ffi.lib().cindm_ddpm1d_sample_guided(
                    h, self.model._h, un, C.byref(desc), C.byref(dz), _ffi.ptr(img), _ffi.ptr(cond_d),
                    _ffi.ptr(None if noise is None else noise.step), _ffi.ptr(None if noise is None else noise.recur),
                    C.c_uint64(seed), sample_offset, _ffi.ptr(inp), 0 if inp is None else inp.shape[1],
                    _ffi.ptr(None if noise is None else noise.cond), _ffi.ptr(iso), 0 if iso is None else iso.shape[1],
                    int(t_start), int(t_end), B, _ffi.ptr(ws), ws.numel(), _ffi.current_stream(device), int(use_graph)))
        return self._chain(img, desc, call)

    def _init_state(self, shape, device, noise, seed, sample_offset, tag):
        if noise is not None:
            return self._f32(noise.init, device).clone()
        img = torch.empty(shape, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            _ffi.check(_ffi.lib().cindm_fill_normal(_ffi.ptr(img), shape[0], shape[1] * shape[2], C.c_uint64(seed),
                                                    sample_offset, tag, _ffi.current_stream(device)))
        return img

    @staticmethod
    def _draw_seed():
        return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())

    @torch.no_grad()
    def p_sample_loop(self, shape, cond, n_composed=0, compose_start_step=4, compose_n_bodies=2, compose_mode="mean",
                      design_fn=None, design_guidance="standard", initial_state_overwrite=None, initialization_mode=0,
                      initialization_img=None, *, noise=None, seed=None, sample_offset=0, use_graph=True, t_stop=0):
        """:1656-1720.  Build-only keywords: ``noise`` (NoiseTape, explicit draws), ``seed`` /
        ``sample_offset`` (counter-based generator keyed by global sample index), ``t_stop`` (truncate)."""
        device = self.betas.device
        if device.type != "cuda":
            raise _ffi.CindmError("GaussianDiffusion1D is on the CPU: move it to a ROCm device; there is no CPU execution path")
        if seed is None and noise is None:
            seed = self._draw_seed()
        seed = 0 if seed is None else int(seed)
        if noise is not None:
            noise = noise.to(device)
        B, T1 = shape[0], shape[1]
        full = (B, T1 + n_composed * compose_start_step, compose_n_bodies * 4)
        assert compose_start_step < T1
        init = self._init_state(full, device, noise, seed, sample_offset, self.num_timesteps)
        if initialization_mode == 0:
            img = init
        elif initialization_mode == 1:
            img = self._f32(initialization_img, device).reshape(full).clone()
        else:
            img = self._f32(initialization_img, device).reshape(full) + init
        inside = "inside" in compose_mode
        desc = self._desc_for(full, compose_mode, n_composed, compose_start_step, T1, compose_n_bodies, outside=not inside)
        inpaint = cond if (self.conditioned_steps == 0 and cond is not None) else None
        fast = design_fn is None and "recurrence" not in design_guidance and initial_state_overwrite is None
        dz = design_fn.descriptor(design_guidance) if isinstance(design_fn, PointObjective) else None
        if dz is not None and design_fn.last_n_step <= full[1]:
            # built-in objective: the guided loop (gradient, overwrite, relaxations) stays inside the captured step
            return self._run_guided_loop(img, cond, desc, dz, self.num_timesteps - 1, t_stop, noise=noise, seed=seed,
                                         sample_offset=sample_offset, inpaint_cond=inpaint,
                                         initial_state_overwrite=initial_state_overwrite, use_graph=use_graph)
        if fast:
            return self._run_loop(img, cond, desc, self.num_timesteps - 1, t_stop,
                                  noise_steps=None if noise is None else noise.step, seed=seed, sample_offset=sample_offset,
                                  inpaint_cond=inpaint, inpaint_noise_steps=None if noise is None else noise.cond,
                                  use_graph=use_graph)
        img_T = img

        def chain():
            # (the steps' predictions only feed the next step: the exchange flags are read once, after the last one)
            img = img_T
            for t in reversed(range(t_stop, self.num_timesteps)):
                nz = None if noise is None else noise.step[t]
                rn = None if (noise is None or noise.recur is None) else noise.recur[t]
                img, _ = self._guided_step(img, cond, t, desc, design_fn, design_guidance, initial_state_overwrite, nz, rn,
                                           check=False)
                if inpaint is not None:
                    zc = noise.cond[t] if (noise is not None and noise.cond is not None) else torch.randn_like(inpaint)
                    img[:, :inpaint.shape[1], :] = self.q_sample(self._f32(inpaint, device), t, zc)
            return img

        img = chain()
        if self._timed_out(desc, device):
            img = self._rerun_exchange_free(chain, desc, device)
        return img

    @torch.no_grad()
    def sample(self, batch_size=16, cond=None, is_composing_time=False, n_composed=2, compose_start_step=4,
               compose_n_bodies=2, compose_mode="mean", design_fn=None, design_guidance="standard",
               initial_state_overwrite=None, initialization_mode=0, initialization_img=None, **build_kw):
        """:2330-2376.  ``build_kw``: noise=, seed=, sample_offset=, use_graph=, t_stop= (see p_sample_loop)."""
        self.is_ddim_sampling = self.sampling_timesteps < self.num_timesteps
        if self.is_ddim_sampling:               # :2348-2363
            build_kw.pop("t_stop", None)
            return self.ddim_sample((batch_size, self.image_size, self.channels), cond=cond, n_composed=n_composed,
                                    compose_start_step=compose_start_step, compose_n_bodies=compose_n_bodies,
                                    compose_mode=compose_mode, design_fn=design_fn, design_guidance=design_guidance,
                                    initial_state_overwrite=initial_state_overwrite,
                                    initialization_mode=initialization_mode, initialization_img=initialization_img,
                                    **build_kw)
        return self.p_sample_loop((batch_size, self.image_size, self.channels), cond=cond, n_composed=n_composed,
                                  compose_start_step=compose_start_step, compose_n_bodies=compose_n_bodies,
                                  compose_mode=compose_mode, design_fn=design_fn, design_guidance=design_guidance,
                                  initial_state_overwrite=initial_state_overwrite,
                                  initialization_mode=initialization_mode, initialization_img=initialization_img,
                                  **build_kw)

    @torch.no_grad()
    def sample_compose_multibodies(self, cond, N, L, n_bodies, *, noise=None, seed=None, sample_offset=0,
                                   use_graph=True, t_stop=0):
        """:1986-2042 for N <= 401 (ULA branch unreachable): x = cat(cond, noise);
        for i = N-1..0: x[:, cs:] = p_sample(x[:, cs:], cond, i).  Returns [B, rollout_steps, 4*n_bodies]."""
        if N > 401:
            raise NotImplementedError("sample_step_ULA (:2048) is out of scope; use N <= 401")
        if not cond.is_cuda:
            raise _ffi.CindmError("sampling needs ROCm device tensors; there is no CPU execution path")
        device = cond.device
        if seed is None and noise is None:
            seed = self._draw_seed()
        seed = 0 if seed is None else int(seed)
        if noise is not None:
            noise = noise.to(device)
        B = cond.shape[0]
        shape = (B, self.rollout_steps, cond.shape[2])
        img = self._init_state(shape, device, noise, seed, sample_offset, self.num_timesteps)
        desc = self._desc_for(shape, None)
        return self._run_loop(img, cond, desc, N - 1, t_stop, noise_steps=None if noise is None else noise.step,
                              seed=seed, sample_offset=sample_offset, inpaint_cond=None, inpaint_noise_steps=None,
                              use_graph=use_graph)

    # ------------------------------------------------------------------ out of scope
    def forward(self, *a, **k):
        raise NotImplementedError("training loss (p_losses, :2438-2501) is out of this build's scope")

    # ------------------------------------------------------------------ DDIM
    def ddim_schedule(self):
        """(times [S+1] descending to -1, coefs [S,3] = (sqrt(alpha_next), c, sigma)) of ddim_sample (:1743-1777), in the
        reference's fp32 tensor arithmetic (time_next = -1 indexes the last table entry, as the reference's negative
        index does; that step returns x_start and its coefficients are not used)."""
        T, S, eta = self.num_timesteps, self.sampling_timesteps, self.ddim_sampling_eta
        # the table is a pure function of (T, S, eta, alphas_cumprod): built once (250 iterations of scalar tensor arithmetic and
        # a device -> host copy cost 7 ms per ddim_sample call, 8 % of a 250-step chain of 256 designs)
        key = (T, S, float(eta), self.alphas_cumprod.data_ptr(), self.alphas_cumprod._version)
        cached = getattr(self, "_ddim_cache", None)
        if cached is not None and cached[0] == key:
            return list(cached[1]), cached[2].clone()
        times = torch.linspace(-1, T - 1, steps=S + 1)
        times = list(reversed(times.int().tolist()))
        ac = self.alphas_cumprod.detach().to("cpu", torch.float32)
        coefs = torch.zeros((S, 3), dtype=torch.float32)
        for i, (time, time_next) in enumerate(zip(times[:-1], times[1:])):
            alpha, alpha_next = ac[time], ac[time_next]
            sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            c = (1 - alpha_next - sigma ** 2).sqrt()
            coefs[i, 0], coefs[i, 1], coefs[i, 2] = alpha_next.sqrt(), c, sigma
        coefs = torch.nan_to_num(coefs, nan=0.0)
        self._ddim_cache = (key, list(times), coefs.clone())
        return times, coefs

    @torch.no_grad()
    def ddim_sample(self, shape, cond, n_composed=None, clip_denoised=True, compose_start_step=4, compose_n_bodies=2,
                    compose_mode="mean", design_fn=None, design_guidance="standard", initial_state_overwrite=None,
                    initialization_mode=0, initialization_img=None, *, noise=None, seed=None, sample_offset=0,
                    use_graph=True, init_img=None, step_range=None):
        """:1724-1804.  Build-only keywords: ``init_img`` + ``step_range=(i0, i1)`` run DDIM steps i0 .. i1-1 from a given
        state (teacher-forced segments for parity tests: the deterministic sampler amplifies a 1e-6 perturbation of the
        U-Net to 1e-3 .. 1e-2 over 50 .. 250 steps with random-init weights -- measured on the CPU reference itself).
        ``noise``: a NoiseTape whose ``step`` / ``recur`` / ``cond`` rows are indexed by the DDIM STEP index.
        Without ``design_fn`` the whole loop is one library call (``cindm_ddpm1d_sample_ddim``); as in the reference the
        prediction then ignores the compose keywords (:1755).  With ``design_fn`` (recurrence guidance only -- the
        reference's non-recurrence branch does not return a noise prediction, :1283) each step is ``recurrence``
        library predictions + the user's autograd gradient, and the DDIM update of the tiny state runs in torch."""
        device = self.betas.device
        if device.type != "cuda":
            raise _ffi.CindmError("GaussianDiffusion1D is on the CPU: move it to a ROCm device; there is no CPU execution path")
        if seed is None and noise is None:
            seed = self._draw_seed()
        seed = 0 if seed is None else int(seed)
        if noise is not None:
            noise = noise.to(device)
        B = shape[0]
        if init_img is not None:
            img = self._f32(init_img, device).clone()
        else:
            img = self._init_state(tuple(shape), device, noise, seed, sample_offset, self.num_timesteps)
        times, coefs = self.ddim_schedule()
        i0, i1 = (0, len(times) - 1) if step_range is None else step_range
        times, coefs = times[i0:i1 + 1], coefs[i0:i1].contiguous()
        if noise is not None:
            sl = lambda v: None if v is None else v[i0:i1].contiguous()
            noise = NoiseTape(noise.init, sl(noise.step), sl(noise.recur), sl(noise.cond))
        S = len(times) - 1
        inpaint = cond if (self.conditioned_steps == 0 and cond is not None) else None
        if design_fn is None:
            desc = self._desc_for(shape, None, clip=clip_denoised)
            cond_d = self._f32(cond, device) if (cond is not None and self.conditioned_steps != 0) else None
            inp = self._f32(inpaint, device)
            tarr = (C.c_int32 * (S + 1))(*times)
            carr = coefs.contiguous()

            def call(h, un, ws):
                with torch.cuda.device(device):
                    _ffi.check(_ffi.lib().cindm_ddpm1d_sample_ddim(
                        h, self.model._h, un, C.byref(desc), _ffi.ptr(img), _ffi.ptr(cond_d), S, tarr, _ffi.ptr(carr),
                        _ffi.ptr(None if noise is None else noise.step), C.c_uint64(seed), sample_offset, _ffi.ptr(inp),
                        0 if inp is None else inp.shape[1], _ffi.ptr(None if noise is None else noise.cond), B,
                        _ffi.ptr(ws), ws.numel(), _ffi.current_stream(device), int(use_graph)))
            return self._chain(img, desc, call)
        if "recurrence" not in design_guidance:
            raise NotImplementedError("DDIM with design_fn needs a '-recurrence-N' guidance (the reference's other branch "
                                      "returns x_{t-1}, not a noise prediction, :1283)")
        n_composed = 0 if n_composed is None else n_composed
        desc = self._desc_for(shape, compose_mode, n_composed, compose_start_step, shape[1], compose_n_bodies,
                              clip=True)     # p_sample_compose_inside's own default: clip_denoised is not forwarded (:1758-1770)
        coefs = coefs.to(device)
        img_T = img

        def chain():
            img = img_T
            for i, (t, tn) in enumerate(zip(times[:-1], times[1:])):
                rn = None if (noise is None or noise.recur is None) else noise.recur[i]
                eps, x_start = self._guided_step(img, cond, t, desc, design_fn, design_guidance, initial_state_overwrite,
                                                 None, rn, ddim_return=True, check=False)
                if tn < 0:
                    img = x_start
                    continue
                z = noise.step[i] if noise is not None else torch.randn_like(img)
                img = x_start * coefs[i, 0] + coefs[i, 1] * eps + coefs[i, 2] * z
                if inpaint is not None:
                    zc = noise.cond[i] if (noise is not None and noise.cond is not None) else torch.randn_like(inpaint)
                    img[:, :inpaint.shape[1], :] = self.q_sample(self._f32(inpaint, device), t, zc)
            return img

        img = chain()
        if self._timed_out(desc, device):
            img = self._rerun_exchange_free(chain, desc, device)
        return img
