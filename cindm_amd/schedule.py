"""DDPM schedule tables (host, init-time only): the 13 fp32 buffers GaussianDiffusion1D /
GaussianDiffusion register (model/diffusion_1d.py:846-910, model/diffusion_2d.py:600-674 of the
reference).  Everything is derived in fp64 from beta and only then cast to fp32
(SURVEY.md Appendix A.5 / B.4)."""
import math

import torch


def beta_schedule(kind, timesteps):
    T = timesteps
    if kind == "linear":                 # model/diffusion_1d.py:464-468
        s = 1000 / T
        return torch.linspace(s * 1e-4, s * 0.02, T, dtype=torch.float64)
    grid = torch.linspace(0, T, T + 1, dtype=torch.float64)
    if kind == "cosine":                 # :470-480, s = 0.008
        f = torch.cos((grid / T + 0.008) / 1.008 * math.pi * 0.5) ** 2
    elif kind == "sigmoid":              # model/diffusion_2d.py:518-531, start -3, end 3, tau 1
        lo, hi = torch.tensor(-3.0).sigmoid(), torch.tensor(3.0).sigmoid()
        f = (hi - (grid / T * 6.0 - 3.0).sigmoid()) / (hi - lo)
    else:
        raise ValueError(f"unknown beta schedule {kind}")
    f = f / f[0]
    return torch.clip(1 - f[1:] / f[:-1], 0, 0.999)


def make_schedule(kind="cosine", timesteps=1000, objective="pred_noise"):
    beta = beta_schedule(kind, timesteps)
    alpha = 1.0 - beta
    abar = torch.cumprod(alpha, dim=0)
    abar_prev = torch.cat([torch.ones(1, dtype=torch.float64), abar[:-1]])
    pvar = beta * (1.0 - abar_prev) / (1.0 - abar)
    snr = abar / (1 - abar)
    weight = {"pred_noise": torch.ones_like(snr), "pred_x0": snr, "pred_v": snr / (snr + 1)}[objective]
    tab = dict(
        betas=beta, alphas_cumprod=abar, alphas_cumprod_prev=abar_prev,
        sqrt_alphas_cumprod=abar.sqrt(), sqrt_one_minus_alphas_cumprod=(1.0 - abar).sqrt(),
        log_one_minus_alphas_cumprod=(1.0 - abar).log(), sqrt_recip_alphas_cumprod=(1.0 / abar).sqrt(),
        sqrt_recipm1_alphas_cumprod=(1.0 / abar - 1).sqrt(), posterior_variance=pvar,
        posterior_log_variance_clipped=pvar.clamp(min=1e-20).log(),
        posterior_mean_coef1=beta * abar_prev.sqrt() / (1.0 - abar),
        posterior_mean_coef2=(1.0 - abar_prev) * alpha.sqrt() / (1.0 - abar),
        loss_weight=weight,
    )
    return {k: v.to(torch.float32) for k, v in tab.items()}
