"""Data-format helpers at the edges of the sampling path: the n-body layout / scale convention
between the reference's PyG batches and the diffusion state.

``get_item_1d`` mirrors the reference's function of the same name (utils.py:203-222): a field of shape
``[B * n_bodies, n_steps, feature_size]`` in simulator units becomes the diffusion model's
``[B, n_steps, n_bodies * feature_size]`` in units of 1/200 (the arena is 200 wide).  ``to_simulator_units`` is its inverse,
what the inference scripts do by hand before re-simulating a design (``pred * 200``,
inference/inverse_design_diffusion_1d.py:286-300).  Pure tensor reshapes: host or device, no library call.
``eval_simu`` mirrors utils.py:1127-1148 around a REQUIRED ``simulation`` callable: the reference's own simulator
(utils.py:1076-1124) is a pymunk / pygame program and neither package is installed in this image, so the simulator itself stays out
of reach -- everything eval_simu does AROUND it (units, layout, sub-sampling, the objective) is here and pinned against the reference
function run with the same stand-in simulator (tests/golden/eval_simu_r4.npz and the script that made it).  Nothing in this package
imports the reference (tests/test_host_logic.py::test_package_never_imports_the_reference)."""
import torch

NBODY_SCALE = 200.0


def get_item_1d(data, target):
    """``data[target]``: [B * n_bodies, n_steps, feature_size]; ``data.dyn_dims``: one entry per sample of the batch.
    Returns [B, n_steps, n_bodies * feature_size] / 200 (utils.py:203-222)."""
    x = data[target]
    batch_size = len(data.dyn_dims)
    assert x.shape[0] % batch_size == 0
    n_bodies = x.shape[0] // batch_size
    n_steps, feature_size = x.shape[1:]
    x = x.reshape(-1, n_bodies, n_steps, feature_size) / NBODY_SCALE
    return torch.flatten(x.permute(0, 2, 1, 3), -2, -1)


def to_simulator_units(x, n_bodies):
    """Inverse of ``get_item_1d``: [B, n_steps, n_bodies * feature_size] (diffusion units) ->
    [B * n_bodies, n_steps, feature_size] in simulator units."""
    B, n_steps, F = x.shape
    assert F % n_bodies == 0
    y = x.reshape(B, n_steps, n_bodies, F // n_bodies).permute(0, 2, 1, 3) * NBODY_SCALE
    return y.reshape(B * n_bodies, n_steps, F // n_bodies)


def eval_simu(cond_design, design_fn, n_bodies, rollout_steps, time_interval=4, *, simulation):
    """The reference's ``eval_simu`` (utils.py:1127-1148): re-simulate a design from its last conditioning frame and score it.

    cond_design: [batch, conditioned_steps, n_bodies * 4] in diffusion units; returns ``(pred_simu [batch, rollout_steps,
    n_bodies * 4] in diffusion units on cond_design's device, design_fn(pred_simu))``.  ``simulation(features=[batch, n_bodies, 4]
    in simulator units, n_steps=rollout_steps * time_interval) -> [batch, n_steps, n_bodies, 4]`` is REQUIRED: the reference's
    simulator (utils.py:1076-1124) is a pymunk / pygame program outside this package -- a caller that has it passes it in
    (``from utils import simulation`` in the caller's own script); this package never imports the reference."""
    if not callable(simulation):
        raise TypeError("eval_simu: simulation= must be a callable simulator (the reference's utils.simulation needs pymunk / pygame "
                        "and is not part of this package)")
    assert cond_design.shape[-1] // 4 == n_bodies
    cond_simu = cond_design[:, -1, :] * NBODY_SCALE
    cond_simu = cond_simu.reshape(cond_simu.shape[0], n_bodies, -1)
    pred_simu = simulation(features=cond_simu, n_steps=rollout_steps * time_interval)
    pred_simu = pred_simu.reshape(pred_simu.shape[0], pred_simu.shape[1], -1)
    pred_simu = pred_simu[:, time_interval - 1::time_interval]
    pred_simu = pred_simu.to(cond_design.device) / NBODY_SCALE
    return pred_simu, design_fn(pred_simu)
