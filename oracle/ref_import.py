"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Import harness for the upstream reference (``/root/reference``), usable only in
the build container where that tree exists.  It is used by
``oracle/make_golden.py`` to (a) pin ``oracle/cindm_oracle.py`` against the
reference's own execution and (b) generate the committed golden vectors under
``tests/golden/``.  Nothing of the reference travels to the GPU box.

The reference imports itself as package ``cindm.*`` and pulls in a handful of
third-party packages that are not installed here and are not used by the
sampling arithmetic; they are replaced by empty stubs.
"""
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("CINDM_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "model", "diffusion_1d.py"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch.utils.data as tud

    class _Any:  # permissive placeholder
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return None

    if "ema_pytorch" not in sys.modules:
        _stub("ema_pytorch", EMA=_Any)
    if "imageio" not in sys.modules:
        _stub("imageio", imwrite=lambda *a, **k: None)
    if "termcolor" not in sys.modules:
        _stub("termcolor", colored=lambda s, *a, **k: s)
    if "torch_geometric" not in sys.modules:
        tg = _stub("torch_geometric")
        tgd = _stub("torch_geometric.data", Dataset=tud.Dataset, Data=_Any, DataLoader=tud.DataLoader)
        tgdl = _stub("torch_geometric.data.dataloader", DataLoader=tud.DataLoader)
        tgn = _stub("torch_geometric.nn", GCNConv=_Any)
        tg.data, tg.nn = tgd, tgn
        tgd.dataloader = tgdl
    if "deepsnap" not in sys.modules:
        ds = _stub("deepsnap")
        ds.batch = _stub("deepsnap.batch", Batch=_Any)
    for name in ("pymunk", "pygame"):
        if name not in sys.modules:
            _stub(name)
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.transforms = _stub("torchvision.transforms", Compose=_Any, Lambda=_Any, ToTensor=_Any,
                              Resize=_Any, CenterCrop=_Any, RandomHorizontalFlip=_Any)
        tv.utils = _stub("torchvision.utils", save_image=lambda *a, **k: None, make_grid=_Any)
    if "cindm" not in sys.modules:
        pkg = types.ModuleType("cindm")
        pkg.__path__ = [REFERENCE_ROOT]
        pkg.__spec__ = importlib.machinery.ModuleSpec("cindm", None, is_package=True)
        pkg.__spec__.submodule_search_locations = [REFERENCE_ROOT]
        sys.modules["cindm"] = pkg


def import_reference():
    """Returns (diffusion_1d module, diffusion_2d module) of the reference."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    install_stubs()
    import matplotlib
    matplotlib.use("Agg")
    import importlib
    d1 = importlib.import_module("cindm.model.diffusion_1d")
    d2 = importlib.import_module("cindm.model.diffusion_2d")
    return d1, d2
