"""CPU oracle for CinDM's compositional diffusion sampling path (1-D n-body).

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement of the reference's
algorithm; it is the *checker* for the HIP path, never the thing shipped or
measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it.  The product package ``cindm_amd`` has no
dependency on it and fails loudly when its HIP library is missing.

Parity status: the reference has no tests, golden vectors or fixtures of its
own, and its arithmetic lives in third-party PyTorch
operators.  This oracle is therefore PINNED AGAINST OUTPUTS OF THE REFERENCE
ITSELF: ``oracle/make_golden.py`` imports ``/root/reference`` in the build
container, runs the reference's classes on seeded synthetic weights / inputs /
injected noise, checks this restatement against them and commits the vectors
under ``tests/golden/`` (``tests/test_oracle_golden.py`` re-checks them on
every run, without the reference).

The restatement is functional: every function takes a flat ``state_dict``
(``name -> torch.Tensor``, the reference's key names) instead of ``nn.Module``
objects, and explicit noise tensors instead of ``torch.randn``.  All arithmetic
is fp32 torch-CPU operators -- the same third-party operators the reference
calls (``F.conv1d``, ``F.group_norm``, ``F.mish``, ``F.linear``, ``einsum``,
``softmax``) at the same call sites, cited per function as
``model/diffusion_1d.py:<line>`` relative to the reference root.
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# schedule tables
# ----------------------------------------------------------------------------

SCHEDULE_BUFFERS = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
    "loss_weight",
)


def linear_beta_schedule(timesteps):
    """model/diffusion_1d.py:464-468."""
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


def cosine_beta_schedule(timesteps, s=0.008):
    """model/diffusion_1d.py:470-480 (fp64, clip to [0, 0.999])."""
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 0, 0.999)


def sigmoid_beta_schedule(timesteps, start=-3, end=3, tau=1):
    """model/diffusion_2d.py:518-531."""
    steps = timesteps + 1
    t = torch.linspace(0, timesteps, steps, dtype=torch.float64) / timesteps
    v_start = torch.tensor(start / tau).sigmoid()
    v_end = torch.tensor(end / tau).sigmoid()
    ac = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 0, 0.999)


def make_schedule(beta_schedule="cosine", timesteps=1000, objective="pred_noise"):
    """The 13 fp32 buffers of GaussianDiffusion1D.__init__, model/diffusion_1d.py:846-910
    (derived in fp64, then cast)."""
    if beta_schedule == "linear":
        betas = linear_beta_schedule(timesteps)
    elif beta_schedule == "cosine":
        betas = cosine_beta_schedule(timesteps)
    elif beta_schedule == "sigmoid":
        betas = sigmoid_beta_schedule(timesteps)
    else:
        raise ValueError(f"unknown beta schedule {beta_schedule}")
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = F.pad(ac[:-1], (1, 0), value=1.0)
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    snr = ac / (1 - ac)
    if objective == "pred_noise":
        lw = torch.ones_like(snr)
    elif objective == "pred_x0":
        lw = snr
    else:
        lw = snr / (snr + 1)
    t64 = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": torch.log(post_var.clamp(min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
        "loss_weight": lw,
    }
    return {k: v.to(torch.float32) for k, v in t64.items()}


# ----------------------------------------------------------------------------
# TemporalUnet1D forward (functional)
# ----------------------------------------------------------------------------

def sinusoidal_pos_emb(t, dim):
    """SinusoidalPosEmb.forward, model/diffusion_1d.py:151-158.  fp32 throughout:
    ``t`` (int64) * fp32 frequency -> fp32 argument -> sin/cos in fp32."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half) * -emb)
    emb = t[:, None] * emb[None, :]
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


def time_mlp(sd, t, dim):
    """TemporalUnet1D.time_mlp, model/diffusion_1d.py:537-542."""
    e = sinusoidal_pos_emb(t, dim)
    e = F.linear(e, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    e = F.mish(e)
    return F.linear(e, sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])


def conv1d_block(sd, p, x, n_groups=8):
    """Conv1dBlock, model/diffusion_1d.py:197-214: Conv1d(k, pad k//2) -> GroupNorm(8) -> Mish.
    The reference applies GroupNorm on a [B,C,1,L] view; identical numerics to [B,C,L]."""
    w = sd[p + ".block.0.weight"]
    x = F.conv1d(x, w, sd[p + ".block.0.bias"], padding=w.shape[-1] // 2)
    x = F.group_norm(x.unsqueeze(2), n_groups, sd[p + ".block.2.weight"], sd[p + ".block.2.bias"], eps=1e-5)
    return F.mish(x.squeeze(2))


def residual_temporal_block(sd, p, x, temb):
    """ResidualTemporalBlock.forward, model/diffusion_1d.py:502-511."""
    tb = F.linear(F.mish(temb), sd[p + ".time_mlp.1.weight"], sd[p + ".time_mlp.1.bias"])
    out = conv1d_block(sd, p + ".blocks.0", x) + tb[:, :, None]
    out = conv1d_block(sd, p + ".blocks.1", out)
    if (p + ".residual_conv.weight") in sd:
        res = F.conv1d(x, sd[p + ".residual_conv.weight"], sd[p + ".residual_conv.bias"])
    else:
        res = x
    return out + res


def linear_attention_temporal(sd, p, x, heads=4, dim_head=32):
    """Residual(PreNorm(LinearAttentionTemporal)), model/diffusion_1d.py:75-81, 123-142, 272-291.
    ``p`` is the Residual's prefix (e.g. ``downs.0.2``)."""
    # LayerNorm over channels, biased var, eps 1e-5 for fp32 (:128-132)
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    y = (x - mean) * (var + 1e-5).rsqrt() * sd[p + ".fn.norm.g"]
    qkv = F.conv1d(y, sd[p + ".fn.fn.to_qkv.weight"]).chunk(3, dim=1)
    b, _, n = x.shape
    q, k, v = (t.reshape(b, heads, dim_head, n) for t in qkv)
    q = q * dim_head ** -0.5
    k = k.softmax(dim=-1)
    context = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", context, q)
    out = out.reshape(b, heads * dim_head, n)
    out = F.conv1d(out, sd[p + ".fn.fn.to_out.weight"], sd[p + ".fn.fn.to_out.bias"])
    return out + x


def unet1d_levels(sd):
    """Number of down levels present in a TemporalUnet1D state dict."""
    n = 0
    while f"downs.{n}.0.blocks.0.block.0.weight" in sd:
        n += 1
    return n


def unet1d_forward(sd, x, t, attention=None, taps=None):
    """TemporalUnet1D.forward, model/diffusion_1d.py:610-646.

    x [B, horizon, F] fp32, t [B] int64 -> eps [B, horizon, F].  The level
    structure (which levels down/up-sample, whether attention exists) is read
    off the state-dict keys, which is how the reference's constructor
    (:549-603) materialises it.  ``taps`` (optional dict) receives named
    intermediate activations ([B,C,L]) for per-block parity tests."""
    dim = sd["time_mlp.3.weight"].shape[0]
    if attention is None:
        attention = "mid_attn.fn.norm.g" in sd
    nl = unet1d_levels(sd)
    x = x.transpose(1, 2)                       # b h t -> b t h
    temb = time_mlp(sd, t, dim)
    if taps is not None:
        taps["temb"] = temb
    h = []
    for i in range(nl):
        x = residual_temporal_block(sd, f"downs.{i}.0", x, temb)
        if taps is not None:
            taps[f"downs.{i}.0"] = x
        x = residual_temporal_block(sd, f"downs.{i}.1", x, temb)
        if taps is not None:
            taps[f"downs.{i}.1"] = x
        if attention:
            x = linear_attention_temporal(sd, f"downs.{i}.2", x)
            if taps is not None:
                taps[f"downs.{i}.2"] = x
        h.append(x)
        if f"downs.{i}.3.conv.weight" in sd:    # Downsample1d :92-98
            x = F.conv1d(x, sd[f"downs.{i}.3.conv.weight"], sd[f"downs.{i}.3.conv.bias"], stride=2, padding=1)
            if taps is not None:
                taps[f"downs.{i}.3"] = x
    x = residual_temporal_block(sd, "mid_block1", x, temb)
    if taps is not None:
        taps["mid_block1"] = x
    if attention:
        x = linear_attention_temporal(sd, "mid_attn", x)
        if taps is not None:
            taps["mid_attn"] = x
    x = residual_temporal_block(sd, "mid_block2", x, temb)
    if taps is not None:
        taps["mid"] = x
        taps["mid_block2"] = x
    for j in range(nl - 1):
        x = torch.cat((x, h.pop()), dim=1)
        x = residual_temporal_block(sd, f"ups.{j}.0", x, temb)
        if taps is not None:
            taps[f"ups.{j}.0"] = x
        x = residual_temporal_block(sd, f"ups.{j}.1", x, temb)
        if taps is not None:
            taps[f"ups.{j}.1"] = x
        if attention:
            x = linear_attention_temporal(sd, f"ups.{j}.2", x)
        if taps is not None:
            taps[f"ups.{j}.2"] = x
        if f"ups.{j}.3.conv.weight" in sd:      # Upsample1d :100-106 (ConvTranspose1d 4,2,1)
            x = F.conv_transpose1d(x, sd[f"ups.{j}.3.conv.weight"], sd[f"ups.{j}.3.conv.bias"], stride=2, padding=1)
            if taps is not None:
                taps[f"ups.{j}.3"] = x
    x = conv1d_block(sd, "final_conv.0", x)
    x = F.conv1d(x, sd["final_conv.1.weight"], sd["final_conv.1.bias"])
    return x.transpose(1, 2)


# ----------------------------------------------------------------------------
# GaussianDiffusion1D sampling half (functional)
# ----------------------------------------------------------------------------

class Diffusion1D:
    """Bundle of what GaussianDiffusion1D holds for sampling (model/diffusion_1d.py:801-910):
    the pair model's state dict, an optional unconditioned (single-body) model,
    schedule tables, ``image_size`` (= rollout steps) and ``conditioned_steps``."""

    def __init__(self, sd, *, image_size, conditioned_steps, sd_uncond=None, timesteps=1000,
                 beta_schedule="cosine", objective="pred_noise", backward_steps=5, backward_lr=1):
        assert objective in ("pred_noise", "pred_x0", "pred_v")
        self.sd = sd
        self.sd_uncond = sd_uncond
        self.image_size = image_size
        self.rollout_steps = image_size
        self.conditioned_steps = conditioned_steps
        self.num_timesteps = timesteps
        self.objective = objective
        self.backward_steps = backward_steps
        self.backward_lr = backward_lr
        self.tab = make_schedule(beta_schedule, timesteps, objective)
        self.channels = sd["final_conv.1.weight"].shape[0]

    def model(self, x, t):
        return unet1d_forward(self.sd, x, t)

    def model_unconditioned(self, x, t):
        return unet1d_forward(self.sd_uncond, x, t)


def _ext(a, t):
    """extract(), model/diffusion_1d.py:454-462, for a scalar python int t: -> 0-d fp32 tensor."""
    return a[t]


def gradient_4body(d, x_t, t):
    """GaussianDiffusion1D.gradient, n_bodies == 4 branch, model/diffusion_1d.py:1865-1926:
    six pair evaluations (batched on dim 0 in the order 12,13,14,23,24,34) plus four
    single-body evaluations weighted by -1.4.  (t <= 400 branch: no scalar_for_gradient.)"""
    B = x_t.shape[0]
    xb = x_t.reshape(B, x_t.shape[1], 4, x_t.shape[2] // 4)
    body = [xb[:, :, i, :] for i in range(4)]
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    x_in = torch.cat([torch.cat([body[i], body[j]], dim=2) for i, j in pairs], dim=0)
    tt = torch.full((x_in.shape[0],), t, dtype=torch.long)
    nc = d.model(x_in, tt)
    nc = nc.reshape(nc.shape[0], nc.shape[1], 2, nc.shape[2] // 2)
    tu = torch.full((B,), t, dtype=torch.long)
    nu = [d.model_unconditioned(body[i].contiguous(), tu) for i in range(4)]
    c = 1.4

    def sl(p, slot):
        return nc[p * B:(p + 1) * B, :, slot, :]

    n1 = sl(0, 0) + sl(1, 0) + sl(2, 0) - c * nu[0]
    n2 = sl(0, 1) + sl(3, 0) + sl(4, 0) - c * nu[1]
    n3 = sl(1, 1) + sl(3, 1) + sl(5, 0) - c * nu[2]
    n4 = sl(2, 1) + sl(4, 1) + sl(5, 1) - c * nu[3]
    return torch.cat([torch.cat([n1, n2], dim=2), torch.cat([n3, n4], dim=2)], dim=2)


def gradient_3body(d, x_t, t):
    """GaussianDiffusion1D.gradient, n_bodies == 3 branch, model/diffusion_1d.py:1927-1982: three pair evaluations (batched on dim 0
    in the order 12, 13, 23) plus three single-body evaluations weighted by -1 (no 1.4 here, :1958-1960).  The reference slices the
    batched pair output with the literal bounds 0:20 / 20:40 / 40:60 -- the branch is only defined for a batch of 20, and this
    restatement keeps the literals.  Reached only by a direct call: model_predictions always passes n_bodies = 4 (:1004).
    (t <= 400 branch: no scalar_for_gradient.)"""
    B = x_t.shape[0]
    assert B == 20, "the reference's 3-body branch hard-codes batch 20 (model/diffusion_1d.py:1958-1960)"
    xb = x_t.reshape(B, x_t.shape[1], 3, x_t.shape[2] // 3)
    body = [xb[:, :, i, :] for i in range(3)]
    x_in = torch.cat([torch.cat([body[0], body[1]], dim=2), torch.cat([body[0], body[2]], dim=2), torch.cat([body[1], body[2]], dim=2)], dim=0)
    tt = torch.full((x_in.shape[0],), t, dtype=torch.long)
    nc = d.model(x_in, tt)
    nc = nc.reshape(nc.shape[0], nc.shape[1], 2, nc.shape[2] // 2)
    tu = torch.full((B,), t, dtype=torch.long)
    nu = [d.model_unconditioned(body[i].contiguous(), tu) for i in range(3)]
    n1 = nc[0:20, :, 0, :] + nc[20:40, :, 0, :] - nu[0]
    n2 = nc[0:20, :, 1, :] + nc[40:60, :, 0, :] - nu[1]
    n3 = nc[20:40, :, 1, :] + nc[40:60, :, 1, :] - nu[2]
    return torch.cat([n1, n2, n3], dim=2)


def compose_inside_eps(d, x, t, *, compose_mode, n_composed, compose_start_step, single_model_step,
                       compose_n_bodies):
    """The "inside" branch of model_predictions, model/diffusion_1d.py:959-1001: one U-Net call
    per (window kk, body pair ii<jj); scatter into [W,B,L,nb(sender),nb(receiver),4]; reduce."""
    nb = compose_n_bodies
    W = n_composed + 1
    B, Ltot, _ = x.shape
    agg = torch.zeros((W, B, Ltot, nb, nb, 4), dtype=x.dtype)
    mask = torch.zeros((W,) + tuple(x.shape), dtype=x.dtype)
    tt = torch.full((B,), t, dtype=torch.long)
    for kk in range(W):
        lo, hi = kk * compose_start_step, kk * compose_start_step + single_model_step
        mask[kk, :, lo:hi] = 1.0
        for ii in range(nb):
            for jj in range(nb):
                if ii < jj:
                    index = torch.cat([torch.arange(ii * 4, (ii + 1) * 4), torch.arange(jj * 4, (jj + 1) * 4)])
                    e = d.model(x[:, lo:hi, index].contiguous(), tt)
                    agg[kk, :, lo:hi, jj, ii] = e[..., :4]
                    agg[kk, :, lo:hi, ii, jj] = e[..., 4:]
    if compose_mode == "mean-inside":
        agg = (agg.sum(-3) / (nb - 1)).flatten(start_dim=3)
        return agg.sum(0) / mask.sum(0)
    elif compose_mode == "sum-inside":
        agg = agg.sum(-3).flatten(start_dim=3)
        return agg.sum(0) / mask.mean(0)
    raise ValueError(compose_mode)


def model_predictions(d, x, cond, t, clip_x_start=False, **kw):
    """model_predictions, model/diffusion_1d.py:951-1031 (clip_x_start=False for the DDPM callers; ddim_sample passes
    clip_denoised, :1755; rederive_pred_noise is never set on the path).  Returns (pred_noise, x_start)."""
    if d.conditioned_steps != 0:
        x = torch.cat([cond, x], dim=1)
    if "compose_mode" in kw and "inside" in kw["compose_mode"]:
        out = compose_inside_eps(d, x, t, compose_mode=kw["compose_mode"], n_composed=kw["n_composed"],
                                 compose_start_step=kw["compose_start_step"],
                                 single_model_step=kw["single_model_step"],
                                 compose_n_bodies=kw["compose_n_bodies"])
    elif d.sd_uncond is not None:
        out = gradient_4body(d, x, t)
    else:
        out = d.model(x, torch.full((x.shape[0],), t, dtype=torch.long))
    T = d.tab
    clipf = (lambda v: v.clamp(-1.0, 1.0)) if clip_x_start else (lambda v: v)
    if d.objective == "pred_noise":
        pred_noise = out
        x_start = clipf(_ext(T["sqrt_recip_alphas_cumprod"], t) * x - _ext(T["sqrt_recipm1_alphas_cumprod"], t) * out)
    elif d.objective == "pred_x0":
        x_start = clipf(out)
        pred_noise = (_ext(T["sqrt_recip_alphas_cumprod"], t) * x - x_start) / _ext(T["sqrt_recipm1_alphas_cumprod"], t)
    else:
        x_start = clipf(_ext(T["sqrt_alphas_cumprod"], t) * x - _ext(T["sqrt_one_minus_alphas_cumprod"], t) * out)
        pred_noise = (_ext(T["sqrt_recip_alphas_cumprod"], t) * x - x_start) / _ext(T["sqrt_recipm1_alphas_cumprod"], t)
    if d.conditioned_steps != 0:
        pred_noise = pred_noise[:, cond.size(1):]
        x_start = x_start[:, cond.size(1):]
    return pred_noise, x_start


def p_mean_variance(d, x, cond, t, clip_denoised=True, **kw):
    """p_mean_variance + q_posterior, model/diffusion_1d.py:1033-1044, 938-949.
    Returns (model_mean, posterior_log_variance (0-d), x_start, pred_noise)."""
    pred_noise, x_start = model_predictions(d, x, cond, t, **kw)
    if clip_denoised:
        x_start = x_start.clamp(-1.0, 1.0)
    T = d.tab
    mean = _ext(T["posterior_mean_coef1"], t) * x_start + _ext(T["posterior_mean_coef2"], t) * x
    return mean, _ext(T["posterior_log_variance_clipped"], t), x_start, pred_noise


def _design_shift(d, design_fn, design_guidance, x, x_start, t):
    """The design-objective gradient term of p_sample*, model/diffusion_1d.py:1072-1106 /
    1235-1269 / 1314-1349 (shared by the recurrence and non-recurrence branches)."""
    T = d.tab
    eta = _ext(T["betas"], t) / torch.sqrt(T["alphas_cumprod_prev"])[t]
    g = design_guidance

    def grad_of(z):
        with torch.enable_grad():
            zc = z.clone().detach().requires_grad_()
            obj = design_fn(zc)
            return torch.autograd.grad(obj, zc)[0]

    if g.startswith("standard"):
        gd = grad_of(x)
        if g == "standard" or g.startswith("standard-recurrence"):
            return gd
        if g == "standard-alpha" or g.startswith("standard-alpha-recurrence"):
            return eta * gd
        raise ValueError(g)
    if g.startswith("universal-forward"):
        gd = grad_of(x_start)
        if "pure" in g:
            return gd
        return eta * gd
    if g.startswith("universal-backward"):
        xc = x_start.clone()
        final = None
        for kk in range(d.backward_steps):
            gd = grad_of(xc)
            if kk == 1:
                final = gd if "pure" in g else eta * gd
            xc = xc - gd * d.backward_lr
        delta = xc - x_start
        coef = (T["sqrt_alphas_cumprod"] * T["betas"] / (torch.sqrt(1 - T["betas"]) * (1 - T["alphas_cumprod"])))[t]
        return final - coef * delta
    raise ValueError(g)


def _recurrence_times(design_guidance):
    return int(design_guidance.split("-")[-1]) if "recurrence" in design_guidance else 0


def _relax_coefs(d, t):
    """sqrt(abar_t/abar_{t-1}), sqrt(1 - abar_t/abar_{t-1}); model/diffusion_1d.py:1181-1182 (fp32 tables)."""
    T = d.tab
    r = T["alphas_cumprod"] / T["alphas_cumprod_prev"]
    return torch.sqrt(r)[t], torch.sqrt(1 - r)[t]


def p_sample(d, x, cond, t, noise, *, design_fn=None, design_guidance="standard",
             initial_state_overwrite=None, recur_noise=None, clip_denoised=True, pmv_kwargs=None, ddim_return=False):
    """p_sample (model/diffusion_1d.py:1047-1186) and, with ``pmv_kwargs`` holding the compose
    arguments, p_sample_compose_inside (:1190-1376): they differ only in what is forwarded to
    p_mean_variance.  ``noise`` replaces ``torch.randn_like(x)`` (ignored at t == 0);
    ``recur_noise[r]`` replaces the r-th relaxation draw (:1180 / :1365).
    Returns (x_{t-1}, x_start)."""
    kw = pmv_kwargs or {}
    R = _recurrence_times(design_guidance)
    if R == 0:
        mean, logvar, x_start, _ = p_mean_variance(d, x, cond, t, clip_denoised, **kw)
        pred = mean
        if design_fn is not None:
            pred = mean - _design_shift(d, design_fn, design_guidance, x, x_start, t)
        if initial_state_overwrite is not None:
            k = initial_state_overwrite.shape[1]
            pred = torch.cat([initial_state_overwrite, pred[:, k:]], 1)
    else:
        for r in range(R):
            mean, logvar, x_start, eps = p_mean_variance(d, x, cond, t, clip_denoised, **kw)
            pred = mean
            shift = None
            if design_fn is not None:
                shift = _design_shift(d, design_fn, design_guidance, x, x_start, t)
                pred = mean - shift
            if initial_state_overwrite is not None:
                k = initial_state_overwrite.shape[1]
                pred = torch.cat([initial_state_overwrite, pred[:, k:]], 1)
            a, b = _relax_coefs(d, t)
            x = a * pred + b * recur_noise[r]
        if ddim_return:
            # sampling_timesteps != 1000 (:1372-1376): (pred_noise + grad_design_final, x_start) of the LAST iteration
            return eps + shift, x_start
    if t > 0:
        pred = pred + (0.5 * logvar).exp() * noise
    return pred, x_start


def p_sample_compose_inside(d, x, cond, t, noise, *, compose_mode="mean-inside", n_composed=0,
                            compose_start_step=4, single_model_step=-1, compose_n_bodies=2, **kw):
    """p_sample_compose_inside, model/diffusion_1d.py:1190-1376."""
    if "inside" in compose_mode:
        pk = dict(compose_mode=compose_mode, n_composed=n_composed, compose_start_step=compose_start_step,
                  single_model_step=single_model_step, compose_n_bodies=compose_n_bodies)
    else:
        pk = dict(compose_mode=compose_mode)
    return p_sample(d, x, cond, t, noise, pmv_kwargs=pk, **kw)


def _outside_aggregate(d, x, cond, t, *, compose_mode, n_composed, compose_start_step, single_model_step,
                       compose_n_bodies, clip_denoised=True):
    """The aggregation half of p_sample_compose_outside, model/diffusion_1d.py:1410-1466:
    run p_mean_variance per (window, pair); "mean": average mu and x0 over senders and windows;
    "noise_sum": sum eps then recompute.  Returns (model_mean, logvar, x_start)."""
    nb = compose_n_bodies
    W = n_composed + 1
    B, Ltot, _ = x.shape
    z = lambda: torch.zeros((W, B, Ltot, nb, nb, 4), dtype=x.dtype)
    if compose_mode == "mean":
        mean_aggr, xs_aggr = z(), z()
    elif compose_mode == "noise_sum":
        eps_aggr = z()
    else:
        raise ValueError(compose_mode)
    mask = torch.zeros((W,) + tuple(x.shape), dtype=x.dtype)
    logvar = None
    for kk in range(W):
        lo, hi = kk * compose_start_step, kk * compose_start_step + single_model_step
        mask[kk, :, lo:hi] = 1.0
        for ii in range(nb):
            for jj in range(nb):
                if ii < jj:
                    index = torch.cat([torch.arange(ii * 4, (ii + 1) * 4), torch.arange(jj * 4, (jj + 1) * 4)])
                    m_e, logvar, xs_e, eps_e = p_mean_variance(d, x[:, lo:hi, index].contiguous(), cond, t, clip_denoised)
                    if compose_mode == "mean":
                        mean_aggr[kk, :, lo:hi, jj, ii] = m_e[..., :4]
                        mean_aggr[kk, :, lo:hi, ii, jj] = m_e[..., 4:]
                        xs_aggr[kk, :, lo:hi, jj, ii] = xs_e[..., :4]
                        xs_aggr[kk, :, lo:hi, ii, jj] = xs_e[..., 4:]
                    else:
                        eps_aggr[kk, :, lo:hi, jj, ii] = eps_e[..., :4]
                        eps_aggr[kk, :, lo:hi, ii, jj] = eps_e[..., 4:]
    if compose_mode == "mean":
        mean_aggr = (mean_aggr.sum(-3) / (nb - 1)).flatten(start_dim=3)
        xs_aggr = (xs_aggr.sum(-3) / (nb - 1)).flatten(start_dim=3)
        x_start = xs_aggr.sum(0) / mask.sum(0)
        mean = mean_aggr.sum(0) / mask.sum(0)
    else:
        eps = eps_aggr.sum(-3).flatten(start_dim=3).sum(0) / mask.mean(0)
        T = d.tab
        x_start = _ext(T["sqrt_recip_alphas_cumprod"], t) * x - _ext(T["sqrt_recipm1_alphas_cumprod"], t) * eps
        if clip_denoised:
            x_start = x_start.clamp(-1.0, 1.0)
        mean = _ext(T["posterior_mean_coef1"], t) * x_start + _ext(T["posterior_mean_coef2"], t) * x
    return mean, logvar, x_start


def p_sample_compose_outside(d, x, cond, t, noise, *, compose_mode="mean", n_composed=0, compose_start_step=4,
                             single_model_step=-1, compose_n_bodies=2, design_fn=None,
                             design_guidance="standard", initial_state_overwrite=None, recur_noise=None,
                             clip_denoised=True):
    """p_sample_compose_outside, model/diffusion_1d.py:1380-1652."""
    assert single_model_step > 0
    agg = dict(compose_mode=compose_mode, n_composed=n_composed, compose_start_step=compose_start_step,
               single_model_step=single_model_step, compose_n_bodies=compose_n_bodies, clip_denoised=clip_denoised)
    R = _recurrence_times(design_guidance)
    if R == 0:
        mean, logvar, x_start = _outside_aggregate(d, x, cond, t, **agg)
        pred = mean
        if design_fn is not None:
            pred = mean - _design_shift(d, design_fn, design_guidance, x, x_start, t)
        if initial_state_overwrite is not None:
            k = initial_state_overwrite.shape[1]
            pred = torch.cat([initial_state_overwrite, pred[:, k:]], 1)
    else:
        for r in range(R):
            mean, logvar, x_start = _outside_aggregate(d, x, cond, t, **agg)
            pred = mean
            if design_fn is not None:
                pred = mean - _design_shift(d, design_fn, design_guidance, x, x_start, t)
            if initial_state_overwrite is not None:
                k = initial_state_overwrite.shape[1]
                pred = torch.cat([initial_state_overwrite, pred[:, k:]], 1)
            a, b = _relax_coefs(d, t)
            x = a * pred + b * recur_noise[r]
    if t > 0:
        pred = pred + (0.5 * logvar).exp() * noise
    return pred, x_start


def q_sample(d, x_start, t, noise):
    """q_sample, model/diffusion_1d.py:2399-2406."""
    T = d.tab
    return _ext(T["sqrt_alphas_cumprod"], t) * x_start + _ext(T["sqrt_one_minus_alphas_cumprod"], t) * noise


class NoiseTape:
    """Explicit stand-in for the reference's on-the-fly ``torch.randn`` draws.  Draw order on the
    reference side (what ``make_golden.py`` patches): one ``randn`` for x_T
    (model/diffusion_1d.py:1673 / :1987), per reverse step R ``randn_like`` relaxation draws
    (:1365 / :1180 / :1646), then one ``randn_like(x)`` if t > 0 (:1281 / :1118 / :1521), then one
    ``randn_like(cond)`` when inpainting (:1717).

    Tensors: ``init`` [B,L,F]; ``step`` [T,B,L,F] indexed by t (row t unused at t == 0);
    ``recur`` [T,R,B,L,F] or None; ``cond`` [T,B,Lc,F] or None."""

    def __init__(self, init, step, recur=None, cond=None):
        self.init, self.step, self.recur, self.cond = init, step, recur, cond

    @staticmethod
    def make(seed, shape, timesteps, recur=0, cond_shape=None):
        g = torch.Generator().manual_seed(seed)
        init = torch.randn(shape, generator=g)
        step = torch.randn((timesteps,) + tuple(shape), generator=g)
        rec = torch.randn((timesteps, recur) + tuple(shape), generator=g) if recur else None
        cn = torch.randn((timesteps,) + tuple(cond_shape), generator=g) if cond_shape else None
        return NoiseTape(init, step, rec, cn)


def p_sample_loop(d, shape, cond, tape, *, n_composed=0, compose_start_step=4, compose_n_bodies=2,
                  compose_mode="mean", design_fn=None, design_guidance="standard",
                  initial_state_overwrite=None, initialization_mode=0, initialization_img=None,
                  t_stop=0, record=None, resume=None):
    """p_sample_loop, model/diffusion_1d.py:1656-1720.  ``tape`` supplies every random draw.
    ``t_stop`` > 0 truncates the chain (for short parity runs); ``record(t, img)`` is called
    after each step; ``resume=(t, img)`` restarts from the state recorded after step t."""
    B, T1 = shape[0], shape[1]
    full = (B, T1 + n_composed * compose_start_step, compose_n_bodies * 4)
    t_first = d.num_timesteps - 1
    if resume is not None:
        t_first, img = resume[0] - 1, resume[1].clone()
    elif initialization_mode == 0:
        img = tape.init.clone()
    elif initialization_mode == 1:
        img = initialization_img.reshape(full)
    else:
        img = initialization_img.reshape(full) + tape.init
    assert tuple(img.shape) == full
    assert compose_start_step < T1
    kw = dict(design_fn=design_fn, design_guidance=design_guidance, compose_mode=compose_mode,
              n_composed=n_composed, compose_start_step=compose_start_step, single_model_step=T1,
              compose_n_bodies=compose_n_bodies, initial_state_overwrite=initial_state_overwrite)
    for t in reversed(range(t_stop, t_first + 1)):
        rn = tape.recur[t] if tape.recur is not None else None
        if "inside" in compose_mode:
            img, _ = p_sample_compose_inside(d, img, cond, t, tape.step[t], recur_noise=rn, **kw)
        else:
            img, _ = p_sample_compose_outside(d, img, cond, t, tape.step[t], recur_noise=rn, **kw)
        if d.conditioned_steps == 0 and cond is not None:
            img = img.clone()
            img[:, :cond.shape[1], :] = q_sample(d, cond, t, tape.cond[t])
        if record is not None:
            record(t, img)
    return img


def sample(d, batch_size, tape, cond=None, n_composed=2, compose_start_step=4, compose_n_bodies=2,
           compose_mode="mean", **kw):
    """GaussianDiffusion1D.sample (non-DDIM branch), model/diffusion_1d.py:2330-2376."""
    return p_sample_loop(d, (batch_size, d.image_size, d.channels), cond, tape, n_composed=n_composed,
                         compose_start_step=compose_start_step, compose_n_bodies=compose_n_bodies,
                         compose_mode=compose_mode, **kw)


def ddim_time_pairs(num_timesteps, sampling_timesteps):
    """The (time, time_next) schedule of ddim_sample, model/diffusion_1d.py:1743-1745."""
    times = torch.linspace(-1, num_timesteps - 1, steps=sampling_timesteps + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


def ddim_coefs(d, time, time_next, eta):
    """(sqrt(alpha_next), c, sigma) of one DDIM update, :1773-1777, in the reference's fp32 tensor arithmetic
    (time_next = -1 indexes the LAST table entry, as the reference's negative index does; that step's img is
    replaced by x_start anyway)."""
    ac = d.tab["alphas_cumprod"]
    alpha, alpha_next = ac[time], ac[time_next]
    sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
    c = (1 - alpha_next - sigma ** 2).sqrt()
    return alpha_next.sqrt(), c, sigma


def ddim_sample(d, shape, cond, tape, *, sampling_timesteps, eta=0.0, clip_denoised=True, n_composed=0,
                compose_start_step=4, compose_n_bodies=2, compose_mode="mean", design_fn=None,
                design_guidance="standard", initial_state_overwrite=None, record=None):
    """ddim_sample, model/diffusion_1d.py:1724-1804.  ``tape``: dict with ``init`` [B,L,F] (x_T, :1749), ``step`` [S,B,L,F]
    (the randn_like(img) of step i, :1779, drawn even when sigma == 0) and, when inpainting, ``cond`` [S,B,Lc,F] (:1792);
    with ``design_fn`` also ``recur`` [S,R,B,L,F] (:1365).  Returns the final img [B,L,F]."""
    img = tape["init"].clone()
    for i, (time, time_next) in enumerate(ddim_time_pairs(d.num_timesteps, sampling_timesteps)):
        if design_fn is None:
            pred_noise, x_start = model_predictions(d, img, cond, time, clip_x_start=clip_denoised)
        else:
            pred_noise, x_start = p_sample_compose_inside(
                d, img, cond, time, None, design_fn=design_fn, design_guidance=design_guidance, compose_mode=compose_mode,
                n_composed=n_composed, compose_start_step=compose_start_step, single_model_step=shape[1],
                compose_n_bodies=compose_n_bodies, initial_state_overwrite=initial_state_overwrite,
                recur_noise=tape["recur"][i], ddim_return=True)
        san, c, sigma = ddim_coefs(d, time, time_next, eta)
        img = x_start * san + c * pred_noise + sigma * tape["step"][i]
        if time_next < 0:
            img = x_start
        elif d.conditioned_steps == 0 and cond is not None:
            img = img.clone()
            img[:, :cond.shape[1], :] = q_sample(d, cond, time, tape["cond"][i])
        if record is not None:
            record(i, img)
    return img


def sample_compose_multibodies(d, cond, N, tape, t_stop=0, record=None, resume=None):
    """sample_compose_multibodies, model/diffusion_1d.py:1986-2042, for N <= 401 (the ULA branch
    :2002-2022 is unreachable then): x = cat(cond, noise); for i = N-1..0:
    x[:, cs:] = p_sample(x[:, cs:], cond=x[:, :cs], i)."""
    assert N <= 401
    cs = d.conditioned_steps
    x = torch.cat([cond, tape.init], dim=1)
    if resume is not None:
        N, x = resume[0], torch.cat([cond, resume[1]], dim=1)
    for i in reversed(range(t_stop, N)):
        new, _ = p_sample(d, x[:, cs:], x[:, :cs], i, tape.step[i])
        x = torch.cat([x[:, :cs], new], dim=1)
        if record is not None:
            record(i, x[:, cs:])
    return x[:, cs:]


# ----------------------------------------------------------------------------
# synthetic, generator-defined weights (shared by oracle-side tests and bench)
# ----------------------------------------------------------------------------

def unet1d_param_shapes(horizon, transition_dim, dim=64, dim_mults=(1, 2, 4, 8), attention=True):
    """State-dict manifest of TemporalUnet1D (model/diffusion_1d.py:519-608): name -> shape,
    in registration order."""
    dims = [transition_dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    nres = len(in_out)
    sh = {}

    def lin(p, i, o):
        sh[p + ".weight"] = (o, i)
        sh[p + ".bias"] = (o,)

    def conv(p, i, o, k):
        sh[p + ".weight"] = (o, i, k)
        sh[p + ".bias"] = (o,)

    def cblock(p, i, o):
        conv(p + ".block.0", i, o, 5)
        sh[p + ".block.2.weight"] = (o,)
        sh[p + ".block.2.bias"] = (o,)

    def rtb(p, i, o):
        cblock(p + ".blocks.0", i, o)
        cblock(p + ".blocks.1", o, o)
        lin(p + ".time_mlp.1", dim, o)
        if i != o:
            conv(p + ".residual_conv", i, o, 1)

    def attn(p, c):
        sh[p + ".fn.fn.to_qkv.weight"] = (384, c, 1)
        conv(p + ".fn.fn.to_out", 128, c, 1)
        sh[p + ".fn.norm.g"] = (1, c, 1)

    lin("time_mlp.1", dim, dim * 4)
    lin("time_mlp.3", dim * 4, dim)
    if horizon % 8 == 0:
        n_plain = 1
    elif horizon % 4 == 0:
        n_plain = 2
    elif horizon % 2 == 0:
        n_plain = 3
    else:
        raise ValueError("horizon must be even")
    for ind, (ci, co) in enumerate(in_out):
        is_last = ind >= nres - n_plain
        rtb(f"downs.{ind}.0", ci, co)
        rtb(f"downs.{ind}.1", co, co)
        if attention:
            attn(f"downs.{ind}.2", co)
        if not is_last:
            conv(f"downs.{ind}.3.conv", co, co, 3)
    # registration order in the reference: self.downs and self.ups are created (empty) before the
    # mid blocks (:544-545), so state_dict() lists downs, ups, mid_*, final_conv.
    for ind, (ci, co) in enumerate(reversed(in_out[1:])):
        rtb(f"ups.{ind}.0", co * 2, co)
        rtb(f"ups.{ind}.1", co, ci)
        if attention:
            attn(f"ups.{ind}.2", ci)
        has_up = ind >= n_plain - 1            # :582 / :591 / :600 (is_last never true: 3 entries < nres-1... )
        if has_up:
            sh[f"ups.{ind}.3.conv.weight"] = (ci, ci, 4)
            sh[f"ups.{ind}.3.conv.bias"] = (ci,)
    mid = dims[-1]
    rtb("mid_block1", mid, mid)
    if attention:
        attn("mid_attn", mid)
    rtb("mid_block2", mid, mid)
    cblock("final_conv.0", dim, dim)
    conv("final_conv.1", dim, transition_dim, 1)
    return sh


def synth_state_dict(shapes, seed=0):
    """Generator-defined random-init weights: for key k, values from numpy's PCG64 seeded by
    (seed, crc32(k)); conv/linear weight & bias uniform(+-1/sqrt(fan_in)) (PyTorch's default
    bound), GroupNorm weight 1+0.1u, bias 0.1u, LayerNorm g 1+0.1u (u in [-1,1))."""
    import zlib
    import numpy as np
    sd = {}
    fan = {}
    for k, s in shapes.items():
        if k.endswith(".weight") and len(s) >= 2:
            f = 1
            for v in s[1:]:
                f *= v
            if ".3.conv." in k and k.startswith("ups."):   # ConvTranspose1d weight [Cin, Cout, k]: fan_in = Cout*k
                f = s[1] * s[2]
            fan[k[:-7]] = f
    for k, s in shapes.items():
        rng = np.random.default_rng([seed, zlib.crc32(k.encode())])
        u = rng.uniform(-1.0, 1.0, size=s).astype(np.float32)
        base = k.rsplit(".", 1)[0]
        if k.endswith(".norm.g"):
            v = 1.0 + 0.1 * u
        elif ".block.2." in k:
            v = (1.0 + 0.1 * u) if k.endswith("weight") else 0.1 * u
        elif base in fan:
            v = u / np.float32(math.sqrt(fan[base]))
        else:
            raise KeyError(k)
        sd[k] = torch.from_numpy(np.ascontiguousarray(v.astype(np.float32)))
    return sd


# ============================================================================
# 2-D airfoil path: Unet (model/diffusion_2d.py:281-408) and GaussianDiffusion sampling (:551-907)
# ============================================================================

def unet2d_param_shapes(dim=64, dim_mults=(1, 2), channels=21, heads=4, dim_head=32):
    """State-dict manifest of the reference's 2-D ``Unet`` (model/diffusion_2d.py:282-367), registration order:
    init_conv, time_mlp, downs, ups, mid_block1, mid_attn, mid_block2, final_res_block, final_conv."""
    dims = [dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    tdim = dim * 4
    hid = heads * dim_head
    sh = {}

    def conv(p, i, o, k, bias=True):
        sh[p + ".weight"] = (o, i, k, k)
        if bias:
            sh[p + ".bias"] = (o,)

    def rb(p, i, o):
        sh[p + ".mlp.1.weight"] = (2 * o, tdim)
        sh[p + ".mlp.1.bias"] = (2 * o,)
        for b, ci in (("block1", i), ("block2", o)):
            conv(f"{p}.{b}.proj", ci, o, 3)
            sh[f"{p}.{b}.norm.weight"] = (o,)
            sh[f"{p}.{b}.norm.bias"] = (o,)
        if i != o:
            conv(p + ".res_conv", i, o, 1)

    def lin_attn(p, c):
        conv(p + ".fn.fn.to_qkv", c, hid * 3, 1, bias=False)
        conv(p + ".fn.fn.to_out.0", hid, c, 1)
        sh[p + ".fn.fn.to_out.1.g"] = (1, c, 1, 1)
        sh[p + ".fn.norm.g"] = (1, c, 1, 1)

    def full_attn(p, c):
        conv(p + ".fn.fn.to_qkv", c, hid * 3, 1, bias=False)
        conv(p + ".fn.fn.to_out", hid, c, 1)
        sh[p + ".fn.norm.g"] = (1, c, 1, 1)

    conv("init_conv", channels, dim, 7)
    sh["time_mlp.1.weight"] = (tdim, dim); sh["time_mlp.1.bias"] = (tdim,)
    sh["time_mlp.3.weight"] = (tdim, tdim); sh["time_mlp.3.bias"] = (tdim,)
    n = len(in_out)
    for ind, (ci, co) in enumerate(in_out):
        p = f"downs.{ind}"
        rb(p + ".0", ci, ci); rb(p + ".1", ci, ci); lin_attn(p + ".2", ci)
        if ind < n - 1:
            conv(p + ".3.1", ci * 4, co, 1)           # Downsample: pixel-unshuffle + 1x1 (:105-109)
        else:
            conv(p + ".3", ci, co, 3)
    for ind, (ci, co) in enumerate(reversed(in_out)):
        p = f"ups.{ind}"
        rb(p + ".0", co + ci, co); rb(p + ".1", co + ci, co); lin_attn(p + ".2", co)
        if ind < n - 1:
            conv(p + ".3.1", co, ci, 3)               # Upsample: nearest x2 + 3x3 (:99-103)
        else:
            conv(p + ".3", co, ci, 3)
    mid = dims[-1]
    rb("mid_block1", mid, mid); full_attn("mid_attn", mid); rb("mid_block2", mid, mid)
    rb("final_res_block", dim * 2, dim)
    conv("final_conv", dim, channels, 1)
    return sh


def synth_state_dict_2d(shapes, seed=0):
    """Generator-defined weights for the 2-D Unet (same recipe as synth_state_dict)."""
    import zlib
    import numpy as np
    sd = {}
    for k, s in shapes.items():
        rng = np.random.default_rng([seed, zlib.crc32(k.encode())])
        u = rng.uniform(-1.0, 1.0, size=s).astype(np.float32)
        if k.endswith(".g"):
            v = 1.0 + 0.1 * u
        elif ".norm." in k:
            v = (1.0 + 0.1 * u) if k.endswith("weight") else 0.1 * u
        else:
            wk = k.rsplit(".", 1)[0] + ".weight"
            ws = shapes[wk]
            fan = 1
            for d_ in ws[1:]:
                fan *= d_
            v = u / np.float32(math.sqrt(fan))
        sd[k] = torch.from_numpy(np.ascontiguousarray(v.astype(np.float32)))
    return sd


def _ws_conv2d(x, w, b, padding):
    """WeightStandardizedConv2d.forward, model/diffusion_2d.py:116-124 (fp32: eps 1e-5, biased variance)."""
    mean = w.mean(dim=(1, 2, 3), keepdim=True)
    var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    return F.conv2d(x, (w - mean) * (var + 1e-5).rsqrt(), b, padding=padding)


def _ln2d(x, g):
    """LayerNorm over channels, model/diffusion_2d.py:126-135."""
    var = torch.var(x, dim=1, unbiased=False, keepdim=True)
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) * (var + 1e-5).rsqrt() * g


def resnet_block_2d(sd, p, x, temb):
    """ResnetBlock.forward, model/diffusion_2d.py:212-224 (+ Block :189-198)."""
    ss = F.linear(F.silu(temb), sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"])[:, :, None, None]
    scale, shift = ss.chunk(2, dim=1)
    h = _ws_conv2d(x, sd[p + ".block1.proj.weight"], sd[p + ".block1.proj.bias"], 1)
    h = F.group_norm(h, 8, sd[p + ".block1.norm.weight"], sd[p + ".block1.norm.bias"], eps=1e-5)
    h = F.silu(h * (scale + 1) + shift)
    h = _ws_conv2d(h, sd[p + ".block2.proj.weight"], sd[p + ".block2.proj.bias"], 1)
    h = F.silu(F.group_norm(h, 8, sd[p + ".block2.norm.weight"], sd[p + ".block2.norm.bias"], eps=1e-5))
    if (p + ".res_conv.weight") in sd:
        x = F.conv2d(x, sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"])
    return h + x


def linear_attention_2d(sd, p, x, heads=4, dim_head=32):
    """Residual(PreNorm(LinearAttention)), model/diffusion_2d.py:226-254."""
    b, c, hh, ww = x.shape
    y = _ln2d(x, sd[p + ".fn.norm.g"])
    qkv = F.conv2d(y, sd[p + ".fn.fn.to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = (t.reshape(b, heads, dim_head, hh * ww) for t in qkv)
    q = q.softmax(dim=-2)
    k = k.softmax(dim=-1)
    q = q * dim_head ** -0.5
    v = v / (hh * ww)
    context = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", context, q).reshape(b, heads * dim_head, hh, ww)
    out = F.conv2d(out, sd[p + ".fn.fn.to_out.0.weight"], sd[p + ".fn.fn.to_out.0.bias"])
    return _ln2d(out, sd[p + ".fn.fn.to_out.1.g"]) + x


def full_attention_2d(sd, p, x, heads=4, dim_head=32):
    """Residual(PreNorm(Attention)), model/diffusion_2d.py:256-278."""
    b, c, hh, ww = x.shape
    y = _ln2d(x, sd[p + ".fn.norm.g"])
    qkv = F.conv2d(y, sd[p + ".fn.fn.to_qkv.weight"]).chunk(3, dim=1)
    q, k, v = (t.reshape(b, heads, dim_head, hh * ww) for t in qkv)
    q = q * dim_head ** -0.5
    sim = torch.einsum("bhdi,bhdj->bhij", q, k)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhdj->bhid", attn, v)
    out = out.permute(0, 1, 3, 2).reshape(b, heads * dim_head, hh, ww)
    return F.conv2d(out, sd[p + ".fn.fn.to_out.weight"], sd[p + ".fn.fn.to_out.bias"]) + x


def unet2d_forward(sd, x, t, taps=None):
    """Unet.forward, model/diffusion_2d.py:369-408 (self_condition False).  x [B, C, H, W], t [B] int64."""
    dim = sd["init_conv.weight"].shape[0]
    nl = 0
    while f"downs.{nl}.0.mlp.1.weight" in sd:
        nl += 1

    def tap(name, v):
        if taps is not None:
            taps[name] = v

    x = F.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=3)
    r = x
    tap("init_conv", x)
    e = sinusoidal_pos_emb(t, dim)
    e = F.linear(e, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    e = F.gelu(e)
    temb = F.linear(e, sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])
    h = []
    for i in range(nl):
        p = f"downs.{i}"
        x = resnet_block_2d(sd, p + ".0", x, temb); tap(p + ".0", x)
        h.append(x)
        x = resnet_block_2d(sd, p + ".1", x, temb); tap(p + ".1", x)
        x = linear_attention_2d(sd, p + ".2", x); tap(p + ".2", x)
        h.append(x)
        if (p + ".3.1.weight") in sd:
            b, c, hh, ww = x.shape                    # 'b c (h p1) (w p2) -> b (c p1 p2) h w'
            xs = x.reshape(b, c, hh // 2, 2, ww // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, hh // 2, ww // 2)
            x = F.conv2d(xs, sd[p + ".3.1.weight"], sd[p + ".3.1.bias"])
        else:
            x = F.conv2d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], padding=1)
        tap(p + ".3", x)
    x = resnet_block_2d(sd, "mid_block1", x, temb); tap("mid_block1", x)
    x = full_attention_2d(sd, "mid_attn", x); tap("mid_attn", x)
    x = resnet_block_2d(sd, "mid_block2", x, temb); tap("mid_block2", x)
    for i in range(nl):
        p = f"ups.{i}"
        x = torch.cat((x, h.pop()), dim=1)
        x = resnet_block_2d(sd, p + ".0", x, temb); tap(p + ".0", x)
        x = torch.cat((x, h.pop()), dim=1)
        x = resnet_block_2d(sd, p + ".1", x, temb); tap(p + ".1", x)
        x = linear_attention_2d(sd, p + ".2", x); tap(p + ".2", x)
        if (p + ".3.1.weight") in sd:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = F.conv2d(x, sd[p + ".3.1.weight"], sd[p + ".3.1.bias"], padding=1)
        else:
            x = F.conv2d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], padding=1)
        tap(p + ".3", x)
    x = torch.cat((x, r), dim=1)
    x = resnet_block_2d(sd, "final_res_block", x, temb); tap("final_res_block", x)
    return F.conv2d(x, sd["final_conv.weight"], sd["final_conv.bias"])


# ----------------------------------------------------------------------------
# ForceUnet (the airfoil design objective's surrogate) and the design gradient built on it
# ----------------------------------------------------------------------------

def _resnet_block_notime_2d(sd, p, x):
    """ResnetBlock.forward with time_emb_dim = None (model/diffusion_2d.py:212-224, Block :189-198)."""
    h = _ws_conv2d(x, sd[p + ".block1.proj.weight"], sd[p + ".block1.proj.bias"], 1)
    h = F.silu(F.group_norm(h, 8, sd[p + ".block1.norm.weight"], sd[p + ".block1.norm.bias"], eps=1e-5))
    h = _ws_conv2d(h, sd[p + ".block2.proj.weight"], sd[p + ".block2.proj.bias"], 1)
    h = F.silu(F.group_norm(h, 8, sd[p + ".block2.norm.weight"], sd[p + ".block2.norm.bias"], eps=1e-5))
    if (p + ".res_conv.weight") in sd:
        x = F.conv2d(x, sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"])
    return h + x


def force_unet_param_shapes(dim=64, dim_mults=(1, 2, 4, 8), channels=4):
    """State-dict manifest of ForceUnet (model/diffusion_2d.py:411-458), registration order."""
    dims = [dim] + [dim * m for m in dim_mults]
    sh = {"init_conv.weight": (dim, channels, 7, 7), "init_conv.bias": (dim,)}

    def rb(p, ci, co):
        for b, (i, o) in (("block1", (ci, co)), ("block2", (co, co))):
            sh[f"{p}.{b}.proj.weight"] = (o, i, 3, 3); sh[f"{p}.{b}.proj.bias"] = (o,)
            sh[f"{p}.{b}.norm.weight"] = (o,); sh[f"{p}.{b}.norm.bias"] = (o,)
        if ci != co:
            sh[f"{p}.res_conv.weight"] = (co, ci, 1, 1); sh[f"{p}.res_conv.bias"] = (co,)

    n = len(dim_mults)
    for ind in range(n):
        ci, co = dims[ind], dims[ind + 1]
        p = f"downs.{ind}"
        rb(p + ".0", ci, ci); rb(p + ".1", ci, ci)
        sh[p + ".2.fn.fn.to_qkv.weight"] = (384, ci, 1, 1)
        sh[p + ".2.fn.fn.to_out.0.weight"] = (ci, 128, 1, 1); sh[p + ".2.fn.fn.to_out.0.bias"] = (ci,)
        sh[p + ".2.fn.fn.to_out.1.g"] = (1, ci, 1, 1)
        sh[p + ".2.fn.norm.g"] = (1, ci, 1, 1)
        if ind < n - 1:
            sh[p + ".3.1.weight"] = (co, ci * 4, 1, 1); sh[p + ".3.1.bias"] = (co,)
        else:
            sh[p + ".3.weight"] = (co, ci, 3, 3); sh[p + ".3.bias"] = (co,)
    mid = dims[-1]
    rb("mid_block1", mid, mid)
    sh["mid_attn.fn.fn.to_qkv.weight"] = (384, mid, 1, 1)
    sh["mid_attn.fn.fn.to_out.weight"] = (mid, 128, 1, 1); sh["mid_attn.fn.fn.to_out.bias"] = (mid,)
    sh["mid_attn.fn.norm.g"] = (1, mid, 1, 1)
    rb("mid_block2", mid, mid)
    sh["final.weight"] = (2, 512); sh["final.bias"] = (2,)
    return sh


def force_unet_forward(sd, x, taps=None):
    """ForceUnet.forward, model/diffusion_2d.py:460-486.  x [N, 4, H, W] -> [N, 2] (drag, lift)."""
    nl = 0
    while f"downs.{nl}.0.block1.proj.weight" in sd:
        nl += 1

    def tap(name, v):
        if taps is not None:
            taps[name] = v

    x = F.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=3); tap("init_conv", x)
    for i in range(nl):
        p = f"downs.{i}"
        x = _resnet_block_notime_2d(sd, p + ".0", x); tap(p + ".0", x)
        x = _resnet_block_notime_2d(sd, p + ".1", x); tap(p + ".1", x)
        x = linear_attention_2d(sd, p + ".2", x); tap(p + ".2", x)
        if (p + ".3.1.weight") in sd:
            b, c, hh, ww = x.shape
            xs = x.reshape(b, c, hh // 2, 2, ww // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, hh // 2, ww // 2)
            x = F.conv2d(xs, sd[p + ".3.1.weight"], sd[p + ".3.1.bias"])
        else:
            x = F.conv2d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], padding=1)
        tap(p + ".3", x)
    x = _resnet_block_notime_2d(sd, "mid_block1", x); tap("mid_block1", x)
    x = full_attention_2d(sd, "mid_attn", x); tap("mid_attn", x)
    x = _resnet_block_notime_2d(sd, "mid_block2", x); tap("mid_block2", x)
    x = x.mean(dim=-1).mean(dim=-1)
    return F.linear(x, sd["final.weight"], sd["final.bias"])


def airfoil_design_grad(sd_force, x, batch_size, num_boundaries, frames, p_min, p_max, lambda_force=1.0, lambda_overlap=1.0,
                        downsampling_factor=4, parts=None, sum_boundary=True):
    """The airfoil design_fn of inference/inverse_design_2d.py:208-214 = force_fn (:98-132) + lambda_overlap * overlap_fn
    (:134-143), restated with torch autograd.  x [B * nb, 3 * frames + 3, 64, 64]; returns the gradient, same shape.
    ``sum_boundary`` selects force_fn's branch: True (:100-120, the script's default) feeds the clamped SUM of a design's
    boundaries and the pressure in simulator units; False (:122-130) each copy's own boundary and the normalised pressure.
    The script itself cannot be imported (it parses arguments and loads data at import time); oracle/make_golden_r3.py
    extracts these functions from its text and pins this restatement against them (PINNING_REPORT_R3.json)."""
    x = x.detach().clone().requires_grad_(True)
    H = x.shape[-1]
    boundary = x[:, -3:]
    if sum_boundary:
        boundary = boundary.view(batch_size, num_boundaries, 3, H, H).sum(dim=1, keepdim=True).clamp(0., 1.) \
            .expand(-1, num_boundaries, -1, -1, -1).reshape(batch_size * num_boundaries, 3, H, H)
    forces = []
    for i in range(frames):
        pressure = x[:, 2 + 3 * i]
        if sum_boundary:
            pressure = (0.5 * pressure + 0.5) * (p_max - p_min) + p_min
        ld = force_unet_forward(sd_force, torch.cat([pressure.unsqueeze(1), boundary], dim=1))
        forces.append(lambda_force * torch.abs(ld[:, 0]) + ld[:, 1])
    summed = torch.sum(torch.stack(forces, dim=0), dim=0)
    if not sum_boundary:         # summed over the boundaries of a design, then expanded back (:128-129): the gradient of
        summed = summed.view(batch_size, num_boundaries).sum(dim=1, keepdim=True).expand(-1, num_boundaries).reshape(-1)
    g_force = torch.autograd.grad(summed, x, grad_outputs=torch.ones_like(summed))[0]
    x2 = x.detach().clone().requires_grad_(True)
    xv = x2.view(batch_size, num_boundaries, -1, H, H)
    bd = xv[:, :, -3].clamp(0., 1.)
    nr = H // downsampling_factor
    dm = bd.view(batch_size, num_boundaries, nr, downsampling_factor, nr, downsampling_factor).mean(dim=(3, 5)) \
        .view(batch_size, num_boundaries, -1)
    ip = torch.matmul(dm, dm.permute(0, 2, 1))
    eye = torch.eye(num_boundaries).unsqueeze(0).expand(batch_size, -1, -1)
    ov = (ip * (1 - eye)).mean(dim=(-2, -1))
    g_ov = torch.autograd.grad(ov, x2, grad_outputs=torch.ones_like(ov))[0]
    if parts is not None:
        parts["force"], parts["overlap"] = g_force, g_ov
    return g_force + lambda_overlap * g_ov


class Diffusion2D:
    """What GaussianDiffusion (2-D) holds for sampling, model/diffusion_2d.py:552-676."""

    def __init__(self, sd, *, image_size=64, frames=6, timesteps=1000, beta_schedule="sigmoid", objective="pred_noise",
                 standard_fixed_ratio=0.01, coeff_ratio=0.1, share_noise=True, use_average_share=True,
                 forward_fixed_ratio=0.01, backward_steps=5, backward_lr=0.01):
        self.sd, self.image_size, self.frames = sd, image_size, frames
        self.num_timesteps, self.objective = timesteps, objective
        self.standard_fixed_ratio, self.coeff_ratio = standard_fixed_ratio, coeff_ratio
        self.forward_fixed_ratio, self.backward_steps, self.backward_lr = forward_fixed_ratio, backward_steps, backward_lr
        self.share_noise, self.use_average_share = share_noise, use_average_share
        self.tab = make_schedule(beta_schedule, timesteps, objective)
        self.channels = sd["final_conv.weight"].shape[0]


def share_states_over_boundaries(x, B, nb, use_average_share=True):
    """share_states_over_boundaries, model/diffusion_2d.py:712-725: the state channels [:-3] are replaced by their
    mean (or sum) over the nb boundary copies of each design.  x [B*nb, C, H, W] -> new tensor."""
    C, H, W = x.shape[1:]
    s = x[:, :-3].reshape(B, nb, C - 3, H, W)
    s = s.mean(dim=1, keepdim=True) if use_average_share else s.sum(dim=1, keepdim=True)
    out = x.clone()
    out[:, :-3] = s.expand(-1, nb, -1, -1, -1).reshape(B * nb, C - 3, H, W)
    return out


def sample_noise_2d(state, boundary):
    """sample_noise, model/diffusion_2d.py:775-785: state noise [B,1,C-3,H,W] shared over boundaries, boundary
    noise [B,nb,3,H,W] independent -> [B, nb, C, H, W]."""
    nb = boundary.shape[1]
    return torch.cat([state.expand(-1, nb, -1, -1, -1), boundary], dim=2)


def model_predictions_2d(d, shape, x, t, clip_x_start=False, rederive_pred_noise=False, share_noise=True):
    """GaussianDiffusion.model_predictions, model/diffusion_2d.py:727-754.  pred_noise (:729-739): the Unet's output with its
    state channels shared over the boundary copies (``share_noise``, :732-733), x_start = predict_start_from_noise
    (:735, clamped with ``clip_x_start``) and, with ``clip_x_start and rederive_pred_noise`` (:738-739), the noise
    re-derived from the clamped x_start (predict_noise_from_start :691-695).  pred_x0 (:741-744): the output IS x_start;
    pred_v (:746-750): x_start = predict_start_from_v (:703-707); both clamp with ``clip_x_start``, share nothing and always
    re-derive the noise.  Returns (pred_noise, x_start)."""
    B, nb = shape[0], shape[1]
    T = d.tab
    tt = torch.full((x.shape[0],), t, dtype=torch.long)
    out = unet2d_forward(d.sd, x, tt)
    if d.objective == "pred_noise":
        eps = out
        if share_noise:
            eps = share_states_over_boundaries(eps, B, nb, d.use_average_share)
        x_start = T["sqrt_recip_alphas_cumprod"][t] * x - T["sqrt_recipm1_alphas_cumprod"][t] * eps
        if clip_x_start:
            x_start = x_start.clamp(-1.0, 1.0)
            if rederive_pred_noise:
                eps = (T["sqrt_recip_alphas_cumprod"][t] * x - x_start) / T["sqrt_recipm1_alphas_cumprod"][t]
        return eps, x_start
    if d.objective == "pred_x0":
        x_start = out
    elif d.objective == "pred_v":
        x_start = T["sqrt_alphas_cumprod"][t] * x - T["sqrt_one_minus_alphas_cumprod"][t] * out
    else:
        raise ValueError(d.objective)
    if clip_x_start:
        x_start = x_start.clamp(-1.0, 1.0)
    eps = (T["sqrt_recip_alphas_cumprod"][t] * x - x_start) / T["sqrt_recipm1_alphas_cumprod"][t]
    return eps, x_start


def p_sample_2d(d, shape, x, t, noise, design_fn=None, design_guidance="standard", clip_denoised=True, recur_noise=None):
    """GaussianDiffusion.p_sample, model/diffusion_2d.py:788-889 (any objective; share_noise either way).
    x [B*nb, C, H, W]; noise [B*nb, C, H, W] (= sample_noise(...).view) or None at t == 0; design_fn returns a GRADIENT
    tensor (:813).  "-recurrence-N" guidance (:846-889, needs design_fn) follows the reference literally: the posterior
    mean is computed ONCE, every iteration subtracts the raw design gradient taken at the current relaxed x
    (``model_mean - grad_design``, not the scaled ``grad_design_final``) and re-noises with ``recur_noise[r]``
    [B*nb, C, H, W].  Returns (x_{t-1}, x_start)."""
    B, nb = shape[0], shape[1]
    T = d.tab
    # p_mean_variance :758-761: model_predictions WITHOUT clip_x_start, then x_start.clamp_ (the same values)
    _, x_start = model_predictions_2d(d, shape, x, t, clip_x_start=False, share_noise=d.share_noise)
    if clip_denoised:
        x_start = x_start.clamp(-1.0, 1.0)
    if not d.share_noise:        # p_mean_variance :762-763: the clamped x_start is shared instead ...
        x_start = share_states_over_boundaries(x_start, B, nb, d.use_average_share)
    mean = T["posterior_mean_coef1"][t] * x_start + T["posterior_mean_coef2"][t] * x
    if not d.share_noise:        # ... and so is the posterior mean (:769-770)
        mean = share_states_over_boundaries(mean, B, nb, d.use_average_share)
    if "recurrence" in design_guidance:
        R = int(design_guidance.split("-")[-1])
        ratio = T["alphas_cumprod"] / T["alphas_cumprod_prev"]
        for r in range(R):
            if design_guidance.startswith("standard"):
                g = design_fn(x.clone())
            elif design_guidance.startswith("universal-forward-recurrence"):
                g = design_fn(x_start.clone())
            else:
                raise ValueError(design_guidance)
            pred = mean - g
            x = torch.sqrt(ratio)[t] * pred + torch.sqrt(1 - ratio)[t] * recur_noise[r]
        if t > 0:
            pred = pred + (0.5 * T["posterior_log_variance_clipped"][t]).exp() * noise
        return pred, x_start
    pred = mean
    if t > 0:
        pred = mean + (0.5 * T["posterior_log_variance_clipped"][t]).exp() * noise
    if design_fn is not None:
        g = design_fn(x.clone())
        if design_guidance == "standard":
            pred = pred - d.standard_fixed_ratio * g
        elif design_guidance == "standard-alpha":
            eta = (d.coeff_ratio * T["betas"].flip(0))[t]
            pred = pred - eta * g
        else:
            raise ValueError(design_guidance)
    return pred, x_start


def p_sample_2d_universal(d, shape, x, t, noise, design_fn, design_guidance, clip_denoised=True):
    """The non-recurrence "universal-forward" / "universal-backward" branches of p_sample, model/diffusion_2d.py:821-843:
    the design gradient is taken at x_start (forward), or after ``backward_steps`` gradient steps on x_start (backward:
    the kk == 1 gradient times forward_fixed_ratio, minus the schedule coefficient times the accumulated delta)."""
    T = d.tab
    base, x_start = p_sample_2d(d, shape, x, t, noise, None, "standard", clip_denoised)
    if design_guidance == "universal-forward":
        shift = d.forward_fixed_ratio * design_fn(x_start.clone())
    elif design_guidance == "universal-backward":
        xc, shift = x_start.clone(), None
        for kk in range(d.backward_steps):
            g = design_fn(xc.clone())
            if kk == 1:
                shift = d.forward_fixed_ratio * g
            xc = xc - g * d.backward_lr
        coef = (T["sqrt_alphas_cumprod"] * T["betas"] / (torch.sqrt(1 - T["betas"]) * (1 - T["alphas_cumprod"])))[t]
        shift = shift - coef * (xc - x_start)
    else:
        raise ValueError(design_guidance)
    return base - shift, x_start


def p_sample_loop_2d(d, shape, tape_init, tape_steps, design_fn=None, design_guidance="standard", t_stop=0, record=None):
    """p_sample_loop, model/diffusion_2d.py:893-907.  tape_init = (state [B,1,C-3,H,W], boundary [B,nb,3,H,W]);
    tape_steps[t] likewise for t > 0."""
    B, nb, C, H, W = shape
    img = sample_noise_2d(*tape_init)
    for t in reversed(range(t_stop, d.num_timesteps)):
        nz = sample_noise_2d(*tape_steps[t]).reshape(B * nb, C, H, W) if t > 0 else None
        out, _ = p_sample_2d(d, shape, img.reshape(B * nb, C, H, W), t, nz, design_fn, design_guidance)
        img = out.reshape(B, nb, C, H, W)
        if record is not None:
            record(t, img)
    return img
