"""cindm_amd -- MI355X (gfx950) implementation of CinDM's compositional diffusion sampling path.

Python face = the reference's class surface (``TemporalUnet1D``, ``GaussianDiffusion1D``; 2-D airfoil path: ``Unet``, ``GaussianDiffusion``); all
arithmetic runs in ``libcindm_hip.so`` (hand-written HIP kernels, C ABI in include/cindm_hip.h).
There is no CPU execution path: a missing / unloadable library raises on first use."""
from ._ffi import CindmError
from .checkpoint import Trainer
from .data_utils import eval_simu, get_item_1d, to_simulator_units
from .diffusion1d import GaussianDiffusion1D, NoiseTape
from .diffusion2d import GaussianDiffusion, NoiseTape2D
from .forceunet import ForceObjective, ForceUnet
from .objectives import PointObjective
from .schedule import make_schedule
from .unet1d import TemporalUnet1D
from .unet2d import Unet

__all__ = ["TemporalUnet1D", "GaussianDiffusion1D", "NoiseTape", "Unet", "GaussianDiffusion", "NoiseTape2D",
           "PointObjective", "ForceUnet", "ForceObjective", "Trainer", "make_schedule", "CindmError", "get_item_1d", "to_simulator_units", "eval_simu"]
