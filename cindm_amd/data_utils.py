"""Data-format helpers at the edges of the sampling path: the n-body layout / scale convention
between the reference's PyG batches and the diffusion state.

``get_item_1d`` mirrors the reference's function of the same name (utils.py:203-222): a field of shape
``[B * n_bodies, n_steps, feature_size]`` in simulator units becomes the diffusion model's
``[B, n_steps, n_bodies * feature_size]`` in units of 1/200 (the arena is 200 wide).  ``to_simulator_units`` is its inverse,
what the inference scripts do by hand before re-simulating a design (``pred * 200``,
inference/inverse_design_diffusion_1d.py:286-300).  Pure tensor reshapes: host or device, no library call.
``eval_simu`` (utils.py:1127, pymunk re-simulation) is out of reach here: pymunk is not installed."""
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
