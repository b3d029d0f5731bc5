"""Datasets of the façade.

``HAMMER_Dataset`` keeps the reference's constructor signature (hammer_dataset.py:23, as called at
trainer.py:276-303).  File decoding of the real HAMMER tree is outside the hot path (SURVEY.md §8f
rank 1); this build serves seeded synthetic HAMMER-shaped items (same keys, dtypes and shapes as
indoor_dataset.py:277-425) and additionally hands over the four raw polarizer planes as
``("pol", 0, 0)`` uint8 so that DoLP/AoLP/normals are computed on the device by K1 instead of in the
DataLoader workers (indoor_dataset.py:430-442).
"""
import numpy as np
import torch
from torch.utils.data import Dataset


class HAMMER_Dataset(Dataset):
    def __init__(self, data_path, filenames, height, width, frame_idxs, num_scales, is_train=False, img_ext='.png',
                 offset=10, modality="polarization", supervised_depth=True, supervised_depth_only=True,
                 depth_modality="_gt", items_per_scene=8):
        super().__init__()
        self.height, self.width, self.num_scales = height, width, num_scales
        self.filenames = list(filenames) if filenames else ["synthetic_scene"]
        self.items = len(self.filenames) * items_per_scene
        self.is_train = is_train

    def __len__(self):
        return self.items

    def __getitem__(self, index):
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
