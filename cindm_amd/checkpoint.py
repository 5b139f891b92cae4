"""Checkpoint plumbing at the path's edge.

The 1-D drivers load weights with ``diffusion.load_state_dict(torch.load(path)["model"])``
(inference/inverse_design_diffusion_1d.py:179-180), which the classes of this package support directly (same key
names).  The 2-D driver goes through the training harness: ``Trainer(diffusion, ...).load(milestone)``
(inference/inverse_design_2d.py:191-207 -> model/diffusion_2d.py:1213-1231).  ``Trainer`` here is the inference-side
half of that class: the constructor accepts and ignores the training arguments, ``load`` reads
``{results_folder}/model-{milestone}.pt`` and loads its ``"model"`` entry strictly into the diffusion module."""
import os

import torch


class Trainer:
    def __init__(self, diffusion_model, *dataset_args, results_folder="./results", **training_kwargs):
        self.model = diffusion_model
        self.results_folder = str(results_folder)
        self.step = 0
        self.ema = None                      # EMA weights are a training artefact; load(use_ema=True) reads them from the file

    def load(self, milestone, *, use_ema=False, map_location="cpu"):
        """Loads ``data["model"]`` (or, with ``use_ema``, the ``ema_model.*`` entries of ``data["ema"]``) strictly."""
        path = os.path.join(self.results_folder, f"model-{milestone}.pt")
        data = torch.load(path, map_location=map_location)
        sd = data["model"]
        if use_ema:
            pre = "ema_model."
            sd = {k[len(pre):]: v for k, v in data["ema"].items() if k.startswith(pre)}
        self.model.load_state_dict(sd, strict=True)
        self.step = int(data.get("step", 0))
        if "version" in data:
            print(f"loading from version {data['version']}")
        return self
