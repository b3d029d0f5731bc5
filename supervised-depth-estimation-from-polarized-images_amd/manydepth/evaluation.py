"""Evaluation (reference manydepth/evaluation.py:23-311): inference with the five trained modules
and per-material depth metrics, on the HIP forward path.  Settings mirror evaluation.py:25-48."""
import os

import numpy as np
import torch
from torch.utils.data import DataLoader

from manydepth import datasets, networks
from manydepth.utils import readlines
from polardepth import ops
from polardepth import polar as pdpolar
from polardepth._lib import lib, check, ptr, stream_ptr

_MATERIAL_GREY = {"box": 20, "bottle": 40, "can": 60, "cup": 80, "remote": 100, "teapot": 120, "cutlery": 140,
                  "glass": 160, "table": 180, "wall": 200}


class Evaluation:
    def __init__(self, load_weights_folder=None, data_path=None, height=320, width=480, batch_size=12,
                 augment_xolp=True, augment_normals=True, num_workers=0, joint_attention=None):
        """The reference hard-codes its machine's paths (evaluation.py:27-31); here they are arguments, falling back to
        $PD_EVAL_DATA_PATH / $PD_EVAL_WEIGHTS.  ``data_path="synthetic"`` serves seeded synthetic items; anything else
        must be a HAMMER tree (FileNotFoundError otherwise, like the reference on a wrong path)."""
        data_path = data_path if data_path is not None else os.environ.get("PD_EVAL_DATA_PATH")
        load_weights_folder = load_weights_folder if load_weights_folder is not None else os.environ.get("PD_EVAL_WEIGHTS")
        if data_path is None:
            raise FileNotFoundError("Evaluation needs data_path (or $PD_EVAL_DATA_PATH): a HAMMER test tree, or 'synthetic'")
        if not torch.cuda.is_available():
            raise RuntimeError("Evaluation needs the MI355X: there is no CPU fallback")
        self.height, self.width, self.batch_size = height, width, batch_size
        self.min_depth, self.max_depth, self.scales = 0.1, 2.0, [0, 1, 2, 3]
        self.augment_xolp, self.augment_normals = augment_xolp, augment_normals
        self.load_weights_folder = load_weights_folder
        self.device = torch.device("cuda")
        self.models = {"rgb_encoder": networks.ShallowResnetEncoder(18, False)}
        if augment_normals:
            self.models["normals_encoder"] = networks.ShallowNormalsEncoder(9, 0.0)
        if augment_xolp:
            self.models["xolp_encoder"] = networks.ShallowEncoder('XOLP', 2, 0.0)
        self.models["joint_encoder"] = networks.JointEncoder(
            0.0, augment_normals, augment_xolp,
            attention=(os.environ.get("PD_JOINT_ATTENTION") == "1") if joint_attention is None else joint_attention)
        self.models["mono_depth"] = networks.DepthDecoder(self.models["rgb_encoder"].num_ch_enc, self.scales)
        for m in self.models.values():
            m.to(self.device).eval()
        # evaluation.py:96 reads ../splits (cwd = manydepth/); the repository root works too
        files = None
        for root in ("splits", os.path.join("..", "splits")):
            split = os.path.join(root, "HAMMER_unseen", "test_files.txt")
            if os.path.exists(split):
                files = readlines(split)
                break
        if files is None:
            if str(data_path) != datasets.SYNTHETIC:
                raise FileNotFoundError("splits/HAMMER_unseen/test_files.txt not found (evaluation.py:96)")
            files = []
        ds = datasets.HAMMER_Dataset(data_path, files, height, width, [0], 4, is_train=False)
        self.test_loader = DataLoader(ds, batch_size, False, num_workers=num_workers, drop_last=True)

    def load_mono_model(self):
        if self.load_weights_folder is None:
            return
        for n, m in self.models.items():
            path = os.path.join(self.load_weights_folder, f"{n}.pth")
            sd = torch.load(path, map_location="cpu")
            m.load_state_dict({k: v for k, v in sd.items() if k in m.state_dict()})

    @torch.no_grad()
    def predict(self, inputs):
        normals = None
        if ("pol", 0, 0) in inputs:
            out = pdpolar.polar_forward(inputs[("pol", 0, 0)], want=("xolp", "normals") if self.augment_normals else ("xolp",))
            inputs[("xolp", 0, 0)] = out["xolp"]; normals = out.get("normals")
        feats = self.models["rgb_encoder"](inputs["color_aug", 0, 0].float())
        xf = self.models["xolp_encoder"](inputs["xolp", 0, 0].float()) if self.augment_xolp else None
        nf = self.models["normals_encoder"](inputs["xolp", 0, 0].float(), normals=normals) if self.augment_normals else None
        feats = list(feats) + self.models["joint_encoder"](feats[-1], xf, nf)
        disp = self.models["mono_depth"](feats)[("disp", 0)].contiguous()
        N = disp.shape[0]
        depth = torch.empty((N, 1, self.height, self.width), device=disp.device)
        check(lib.pd_disp_to_depth(ptr(disp), ptr(depth), None, N, disp.shape[2], disp.shape[3], self.height, self.width,
                                   self.min_depth, self.max_depth, stream_ptr()), "pd_disp_to_depth")
        return depth.clamp(self.min_depth, self.max_depth)

    def test(self):
        """evaluation.py:120-288: mean over images of the 7 masked depth metrics, for the whole frame and per material
        class (instance-mask grey values :242-264).  The per-image reductions run on the device (pd_depth_metrics);
        11 x 8 numbers per batch leave the GPU instead of every depth map."""
        objects = ["all"] + list(_MATERIAL_GREY)
        sums = {o: torch.zeros(7, dtype=torch.float64, device=self.device) for o in objects}
        counts = {o: torch.zeros((), dtype=torch.float64, device=self.device) for o in objects}
        for inputs in self.test_loader:
            inputs = {k: v.to(self.device) for k, v in inputs.items()}
            depth = self.predict(inputs)
            for o in objects:
                m = ops.depth_metrics(inputs["depth_gt"], depth, self.min_depth, self.max_depth,
                                      mask=None if o == "all" else inputs[("mask", 0, 0)],
                                      mask_value=_MATERIAL_GREY.get(o, 0))
                valid = m[:, 7] > 0
                sums[o] += (m[:, :7].double() * valid[:, None]).sum(0)
                counts[o] += valid.sum()
        results = {}
        for o in objects:
            if counts[o].item() > 0:
                results[o] = (sums[o] / counts[o]).cpu().numpy()
                print(o, ("&{: 8.5f}  " * 7).format(*results[o].tolist()))
        return results
