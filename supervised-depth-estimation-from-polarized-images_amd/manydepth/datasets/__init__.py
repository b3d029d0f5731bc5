"""Datasets of the façade.

``HAMMER_Dataset`` keeps the reference's constructor signature (hammer_dataset.py:23 / indoor_dataset.py:39-57, as
called at trainer.py:276-303).  When ``data_path`` holds a HAMMER tree
(``<scene>/<modality>/{rgb,pol00,pol01,pol10,pol11,_gt,<depth_modality>,_instance}/%06d.png`` + ``intrinsics.txt``,
indoor_dataset.py:118-190, hammer_dataset.py:59-169) the items are decoded from disk with PIL (LANCZOS resize =
the reference's Image.ANTIALIAS, nearest for depth / instance masks, 16-bit depth in mm -> m).  Seeded synthetic
HAMMER-shaped items with the same keys, dtypes and shapes are served ONLY for the literal ``data_path ==
"synthetic"`` (bench.py and the tests pass it); any other path that is missing, or holds no complete frame of the
requested scenes, raises FileNotFoundError like the reference fails on a wrong ``--data_path``.

Difference to the reference by design: the four polarizer grays are handed over raw as ``("pol", 0, 0)`` uint8
[4,H,W] (order 0/45/90/135 deg = pol00, pol01, pol10, pol11, indoor_dataset.py:435-439) and DoLP/AoLP/normals are
computed on the device by K1, instead of the per-pixel ``lstsq`` in the DataLoader workers (:430-442).
Colour augmentation (indoor_dataset.py:92-106, 226-233, 300, 404-407): with probability 0.5 a training item's
``color_aug`` pyramid is the ColorJitter(brightness, contrast, saturation in [0.8, 1.2], hue in [-0.1, 0.1]) of
its ``color`` pyramid -- torchvision 0.8.2's PIL path restated on PIL.ImageEnhance (torchvision is not installed).
"""
import glob
import os
import random

import numpy as np
import torch
from torch.utils.data import Dataset


SYNTHETIC = "synthetic"


def color_jitter_params(brightness=(0.8, 1.2), contrast=(0.8, 1.2), saturation=(0.8, 1.2), hue=(-0.1, 0.1), rng=random):
    """torchvision 0.8.2 ``ColorJitter.get_params``: one factor per property, applied in a random order."""
    ops_ = [("brightness", rng.uniform(*brightness)), ("contrast", rng.uniform(*contrast)),
            ("saturation", rng.uniform(*saturation)), ("hue", rng.uniform(*hue))]
    rng.shuffle(ops_)
    return ops_


def apply_color_jitter(img, params):
    """torchvision.transforms.functional_pil adjust_brightness / _contrast / _saturation / _hue on a PIL RGB image."""
    from PIL import Image, ImageEnhance
    for name, f in params:
        if name == "brightness":
            img = ImageEnhance.Brightness(img).enhance(f)
        elif name == "contrast":
            img = ImageEnhance.Contrast(img).enhance(f)
        elif name == "saturation":
            img = ImageEnhance.Color(img).enhance(f)
        else:
            h, s_, v = img.convert("HSV").split()
            nh = np.array(h, dtype=np.uint8)
            nh += np.uint8(int(f * 255) & 0xff)  # np.uint8(hue_factor * 255) of torchvision: truncation + uint8 wrap-around
            img = Image.merge("HSV", (Image.fromarray(nh, "L"), s_, v)).convert("RGB")
    return img


class HAMMER_Dataset(Dataset):
    def __init__(self, data_path, filenames, height, width, frame_idxs, num_scales, is_train=False, img_ext='.png',
                 offset=10, modality="polarization", supervised_depth=True, supervised_depth_only=True,
                 depth_modality="_gt", items_per_scene=8, raw_pol=None):
        super().__init__()
        # raw_pol: hand the four polarizer images over at their native size; the Trainer resizes them on the device
        # with the Pillow-exact LANCZOS kernels before K1.  Default: $PD_DEVICE_RESIZE == "1".
        self.raw_pol = (os.environ.get("PD_DEVICE_RESIZE") == "1") if raw_pol is None else bool(raw_pol)
        self.data_path, self.modality, self.depth_modality, self.img_ext = data_path, modality, depth_modality, img_ext
        self.height, self.width, self.num_scales = height, width, num_scales
        self.is_train = is_train
        self.synthetic = str(data_path) == SYNTHETIC
        if self.synthetic:
            self.filenames = list(filenames) if filenames else ["synthetic_scene"]
            self.frames = []
            self.items = len(self.filenames) * items_per_scene
        else:
            if not (data_path and os.path.isdir(str(data_path))):
                raise FileNotFoundError(f"HAMMER data_path {data_path!r} is not a directory (pass the literal "
                                        f"{SYNTHETIC!r} for seeded synthetic items)")
            self.filenames = list(filenames or [])
            self.frames = self._discover(self.filenames)
            if self.filenames and not self.frames:
                raise FileNotFoundError(f"no complete HAMMER frame (rgb, pol00/01/10/11, _gt, {depth_modality}) under "
                                        f"{data_path!r} for scenes {self.filenames[:3]}...")
            self.items = len(self.frames)

    # ---- real HAMMER tree -------------------------------------------------------------------------------
    def _discover(self, scenes):
        """(folder, frame_index) of every frame that has rgb, the four polarizer images, _gt and depth_modality."""
        frames = []
        for scene in scenes or []:
            folder = os.path.join(self.data_path, scene, self.modality)
            for f in sorted(glob.glob(os.path.join(folder, "rgb", "*" + self.img_ext))):
                idx = int(os.path.basename(f).split('.')[0])
                need = ["pol00", "pol01", "pol10", "pol11", "_gt", self.depth_modality]
                if all(os.path.isfile(os.path.join(folder, d, "{:06d}.png".format(idx))) for d in need):
                    frames.append((folder, idx))
        return frames

    def _load_item(self, folder, idx):
        from PIL import Image
        H, W = self.height, self.width
        name = "{:06d}{}".format(idx, self.img_ext)
        to_t = lambda im: torch.from_numpy(np.asarray(im, dtype=np.float32).transpose(2, 0, 1) / 255.0)
        inputs = {}
        color = Image.open(os.path.join(folder, "rgb", name)).convert("RGB")
        full_w, full_h = color.size
        prev = color
        do_color_aug = self.is_train and random.random() > 0.5                     # indoor_dataset.py:300
        jitter = color_jitter_params() if do_color_aug else None                   # :404-405, one draw per item
        for s in range(self.num_scales):          # successive LANCZOS resizes (indoor_dataset.py:192-215)
            prev = prev.resize((W >> s, H >> s), Image.LANCZOS)
            inputs[("color", 0, s)] = to_t(prev)
            blank = inputs[("color", 0, s)].sum() == 0                             # :222-225 blank frames stay blank
            inputs[("color_aug", 0, s)] = to_t(apply_color_jitter(prev, jitter)) if (do_color_aug and not blank) \
                else inputs[("color", 0, s)]
        pol_imgs = [Image.open(os.path.join(folder, d, name)).convert("L")
                    for d in ("pol00", "pol01", "pol10", "pol11")]                # 0, 45, 90, 135 degrees
        planes = [np.asarray(im if self.raw_pol else im.resize((W, H), Image.LANCZOS)) for im in pol_imgs]
        inputs[("pol", 0, 0)] = torch.from_numpy(np.stack(planes).astype(np.uint8))

        def depth_of(sub):                        # 16-bit PNG in mm -> metres, nearest resize (hammer_dataset.py:135-169)
            im = Image.open(os.path.join(folder, sub, "{:06d}.png".format(idx)))
            arr = np.asarray(im.resize((W, H), Image.NEAREST)).astype(np.uint16)
            return torch.from_numpy((arr / 1000).astype(np.float32))[None]
        inputs["depth"] = depth_of(self.depth_modality)
        inputs["depth_gt"] = depth_of("_gt")
        mpath = os.path.join(folder, "_instance", "{:06d}.png".format(idx))
        if os.path.isfile(mpath):
            m = np.asarray(Image.open(mpath).convert("L").resize((W, H), Image.NEAREST))
        else:
            m = np.zeros((H, W), np.uint8)
        inputs[("mask", 0, 0)] = torch.from_numpy(m.astype(np.int32))[None]
        K0 = np.eye(4, dtype=np.float32)          # indoor_dataset.py:261-275, 379-388
        with open(os.path.join(folder, "intrinsics.txt")) as f:
            K0[:3, :3] = np.array(f.read().split(), dtype=np.float32).reshape(3, 3)
        K0[0, :] /= full_w
        K0[1, :] /= full_h
        for s in range(self.num_scales):
            K = K0.copy()
            K[0, :] *= W // (2 ** s)
            K[1, :] *= H // (2 ** s)
            inputs[("K", s)] = torch.from_numpy(K)
            inputs[("inv_K", s)] = torch.from_numpy(np.linalg.pinv(K))
        inputs["stereo_T"] = torch.eye(4)
        inputs["stereo_T"][0, 3] = -0.0498921
        return inputs

    def __len__(self):
        return self.items

    def __getitem__(self, index):
        if self.synthetic:
            return self._synthetic_item(index)
        return self._load_item(*self.frames[index])

    # ---- synthetic items --------------------------------------------------------------------------------
    def _synthetic_item(self, index):
        rng = np.random.default_rng(index)
        H, W = self.height, self.width
        inputs = {}
        color = rng.random((3, H, W), dtype=np.float32)
        for s in range(self.num_scales):
            c = color.reshape(3, H >> s, 1 << s, W >> s, 1 << s).mean((2, 4)) if s else color
            inputs[("color", 0, s)] = torch.from_numpy(np.ascontiguousarray(c))
            inputs[("color_aug", 0, s)] = inputs[("color", 0, s)]
            K = np.eye(4, dtype=np.float32)
            K[0, 0] = K[1, 1] = 0.65 * (W >> s); K[0, 2] = (W >> s) / 2; K[1, 2] = (H >> s) / 2
            inputs[("K", s)] = torch.from_numpy(K)
            inputs[("inv_K", s)] = torch.from_numpy(np.linalg.pinv(K))
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        depth = 1.05 + 0.7 * np.sin(xx / (W / 7.0) + index) * np.cos(yy / (H / 5.0))
        depth[rng.random((H, W)) < 0.1] = 0
        inputs["depth"] = torch.from_numpy(depth[None].astype(np.float32))
        inputs["depth_gt"] = inputs["depth"].clone()
        inputs[("mask", 0, 0)] = torch.from_numpy((rng.integers(0, 11, (1, H, W)) * 20).astype(np.int32))
        iun = 120 + 60 * np.sin(xx / 41.0) * np.cos(yy / 37.0)
        rho = 0.02 + 0.25 * (0.5 + 0.5 * np.sin(xx / 29.0 + yy / 53.0)) ** 2
        phi = (np.pi / 2) * np.sin(xx / 61.0 - yy / 43.0)
        pol = np.stack([iun * (1 + rho * np.cos(2 * a - 2 * phi)) for a in (0, np.pi / 4, np.pi / 2, 3 * np.pi / 4)])
        pol = np.clip(np.rint(pol + rng.normal(0, 1.5, pol.shape)), 0, 255).astype(np.uint8)
        inputs[("pol", 0, 0)] = torch.from_numpy(pol)
        inputs["stereo_T"] = torch.eye(4)
        return inputs


KITTIRAWDataset = CityscapesPreprocessedDataset = KITTIOdomDataset = None   # other datasets: out of scope
