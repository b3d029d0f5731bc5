"""Seeded synthetic "HAMMER-shaped" batches generated directly on the device.

Polarizer planes come from a physical model I_theta = Iun (1 + rho cos(2 theta - 2 phi)) + noise with
smooth low-DoLP fields (realistic, theta mostly inside the tables); RGB is uniform noise with a 2x2
averaged pyramid; GT depth is a smooth field in [0.3, 1.8] m with ~10 % invalid (zero) pixels and the
right ``pad_cols`` columns invalid (512x612 frames padded to 512x640, BASELINE.md shape note).
"""
import math

import torch


def _smooth(shape, gen, device, cutoff=24):
    """Low-pass noise in [0,1]: bilinear upsampling of a coarse random grid."""
    B, H, W = shape
    coarse = torch.rand((B, 1, max(H // cutoff, 2), max(W // cutoff, 2)), generator=gen, device=device)
    return torch.nn.functional.interpolate(coarse, size=(H, W), mode="bilinear", align_corners=True)[:, 0]


def make_batch(B, H=512, W=640, frame_w=612, device="cuda", seed=0, scales=(0, 1, 2, 3), with_pol=True,
               with_xolp=False):
    """Batch dict as IndoorDataset.__getitem__ + collate would deliver it (indoor_dataset.py:277-425),
    plus the optional raw planes ("pol",0,0) uint8 [B,4,H,W] for the fused on-device XOLP path."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev).manual_seed(seed)
    inputs = {}
    color = torch.rand((B, 3, H, W), generator=gen, device=dev)
    inputs[("color", 0, 0)] = color
    inputs[("color_aug", 0, 0)] = color
    c = color
    for s in scales:
        if s == 0:
            continue
        c = torch.nn.functional.avg_pool2d(inputs[("color", 0, s - 1)] if ("color", 0, s - 1) in inputs else c, 2)
        inputs[("color", 0, s)] = c
        inputs[("color_aug", 0, s)] = c
    depth = 0.3 + 1.5 * _smooth((B, H, W), gen, dev)
    invalid = torch.rand((B, H, W), generator=gen, device=dev) < 0.10
    depth = torch.where(invalid, torch.zeros_like(depth), depth)
    if frame_w < W:
        depth[:, :, frame_w:] = 0.0
    inputs["depth"] = depth[:, None].contiguous()
    inputs["depth_gt"] = inputs["depth"]
    inputs[("mask", 0, 0)] = (torch.rand((B, 1, H, W), generator=gen, device=dev) * 11).int() * 20
    for s in scales:
        K = torch.eye(4, device=dev)[None].repeat(B, 1, 1)
        w_s, h_s = W // (2 ** s), H // (2 ** s)
        K[:, 0, 0] = 0.65 * w_s; K[:, 1, 1] = 0.65 * w_s; K[:, 0, 2] = w_s / 2; K[:, 1, 2] = h_s / 2
        inputs[("K", s)] = K
        inputs[("inv_K", s)] = torch.linalg.inv(K)
    if with_pol or with_xolp:
        iun = 20 + 200 * _smooth((B, H, W), gen, dev)
        rho = 0.6 * _smooth((B, H, W), gen, dev, cutoff=32) ** 3
        phi = math.pi * (_smooth((B, H, W), gen, dev, cutoff=40) - 0.5)
        planes = []
        for a in (0.0, math.pi / 4, math.pi / 2, 3 * math.pi / 4):
            planes.append(iun * (1 + rho * torch.cos(2 * a - 2 * phi)))
        pol = torch.stack(planes, 1) + 1.5 * torch.randn((B, 4, H, W), generator=gen, device=dev)
        pol = pol.round().clamp(0, 255).to(torch.uint8)
        inputs[("pol", 0, 0)] = pol
    return inputs
