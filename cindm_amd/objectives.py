"""Design objectives.  ``PointObjective`` is the paper's objective for the n-body inverse-design task
(``get_design_fn`` in inference/inverse_design_diffusion_1d.py:211-229 of the reference): drive the last
``last_n_step`` positions of every body to ``pos_target``.

It is an ordinary ``design_fn`` callable (``Tensor[B, L, 4*n_bodies] -> scalar``), so it works on every guided
path; in addition ``GaussianDiffusion1D`` recognises it and, for "standard" / "standard-alpha" guidance (with or
without ``-recurrence-N``), evaluates its closed-form gradient inside the library's update kernel so that the whole
guided reverse loop stays one captured-graph replay (``cindm_ddpm1d_sample_guided``) -- no autograd, no host code in
the loop."""
import torch

from . import _ffi


class PointObjective:
    def __init__(self, pos_target, last_n_step, gamma=2, coef=100, time_consistency_coef=0, design_fn_mode="L2"):
        pos_target = torch.as_tensor(pos_target, dtype=torch.float32).reshape(-1)
        assert pos_target.numel() == 2, "pos_target is a 2-D position"
        assert gamma == 2, "the reference asserts gamma == 2"
        if design_fn_mode not in ("L2", "L2square"):
            raise ValueError(design_fn_mode)
        self.pos_target, self.last_n_step, self.gamma = pos_target, int(last_n_step), gamma
        self.coef, self.time_consistency_coef, self.design_fn_mode = float(coef), float(time_consistency_coef), design_fn_mode

    def __call__(self, pos):
        """pos: [B, steps, n_bodies*4] -> scalar loss (summed over the batch, as the reference does)."""
        n_bodies = pos.shape[-1] // 4
        target = self.pos_target.to(pos.device)
        n = self.last_n_step
        per_body = []
        for jj in range(n_bodies):
            sq = ((pos[..., -n:, jj * 4:jj * 4 + 2] - target).abs() ** 2).sum(-1)
            if self.design_fn_mode == "L2":
                sq = sq ** 0.5
            per_body.append(sq.mean(-1).sum(0))
        total = torch.stack(per_body).sum() * self.coef
        if self.time_consistency_coef > 0:
            idx = torch.cat([torch.arange(ii * 4, ii * 4 + 2) for ii in range(n_bodies)]).to(pos.device)
            total = total + (pos[:, 1:, idx] - pos[:, :-1, idx]).square().sum(-1).mean(-1).sum() * self.time_consistency_coef
        return total

    def descriptor(self, design_guidance):
        """cindm_design_desc for this objective under ``design_guidance``, or None when that guidance needs the generic
        (autograd) path."""
        g = design_guidance
        rec = 0
        if "recurrence" in g:
            rec = int(g.split("-")[-1])
            g = g[:g.index("-recurrence")]
            if rec < 1:
                return None
        if g not in ("standard", "standard-alpha"):
            return None
        d = _ffi.DesignDesc()
        d.mode = 1 if self.design_fn_mode == "L2" else 2
        d.alpha = int(g == "standard-alpha")
        d.recurrence, d.last_n_step = rec, self.last_n_step
        d.coef, d.time_consistency_coef = self.coef, self.time_consistency_coef
        d.pos_target[0], d.pos_target[1] = float(self.pos_target[0]), float(self.pos_target[1])
        return d
