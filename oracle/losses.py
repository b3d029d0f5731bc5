"""Oracle (plain PyTorch-CPU fp32 restatement) of the supervised loss path and metrics.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Never imported by the product.

* ``manydepth/layers.py:62-71``    disp_to_depth
* ``manydepth/layers.py:452-465``  get_smooth_loss
* ``manydepth/layers.py:468-499``  SSIM (constructed by the trainer, inactive under
  ``--depth_supervision_only``; kept because north_star names it)
* ``manydepth/layers.py:539-577``  compute_depth_errors
* ``manydepth/trainer.py:531-545`` per-scale upsample + disp_to_depth (process_batch)
* ``manydepth/trainer.py:1126-1150,1241-1265,1294`` compute_losses, supervised branch
* ``manydepth/trainer.py:1298-1309`` compute_supervised_normals_losses
* ``kornia.geometry.depth.depth_to_normals`` (kornia==0.5.11, environment.yml:20 -- NOT installed in
  the build container; restated from the published 0.5.11 algorithm: unproject pixels with K,
  ``spatial_gradient`` = 3x3 Sobel normalised by 8 with replicate padding (cross-correlation),
  cross product of the x- and y-gradients, L2 normalise with eps 1e-12).  **Parity unpinned**: no
  reference fixture can pin it; the GPU kernel is tested against this restatement only.
"""
import torch
import torch.nn.functional as F


def disp_to_depth(disp, min_depth, max_depth):
    min_disp, max_disp = 1 / max_depth, 1 / min_depth
    scaled = min_disp + (max_disp - min_disp) * disp
    return scaled, 1 / scaled


def get_smooth_loss(disp, img):
    gdx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gdy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gix = torch.mean(torch.abs(img[:, :, :, :-1] - img[:, :, :, 1:]), 1, keepdim=True)
    giy = torch.mean(torch.abs(img[:, :, :-1, :] - img[:, :, 1:, :]), 1, keepdim=True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


def ssim(x, y):
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    y = F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x, mu_y = F.avg_pool2d(x, 3, 1), F.avg_pool2d(y, 3, 1)
    sx = F.avg_pool2d(x ** 2, 3, 1) - mu_x ** 2
    sy = F.avg_pool2d(y ** 2, 3, 1) - mu_y ** 2
    sxy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + C1) * (2 * sxy + C2)
    d = (mu_x ** 2 + mu_y ** 2 + C1) * (sx + sy + C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def depth_to_3d(depth, K):
    """kornia 0.5.11 depth_to_3d(normalize_points=False): [B,1,H,W], K [B,3,3] -> [B,3,H,W]."""
    B, _, H, W = depth.shape
    v, u = torch.meshgrid(torch.arange(H, dtype=depth.dtype), torch.arange(W, dtype=depth.dtype), indexing="ij")
    fx, fy = K[:, 0, 0].view(B, 1, 1), K[:, 1, 1].view(B, 1, 1)
    cx, cy = K[:, 0, 2].view(B, 1, 1), K[:, 1, 2].view(B, 1, 1)
    x = (u[None] - cx) / fx
    y = (v[None] - cy) / fy
    xyz = torch.stack([x, y, torch.ones_like(x)], 1)
    return xyz * depth


def spatial_gradient_sobel(t):
    """kornia spatial_gradient(mode='sobel', order=1, normalized=True): [B,C,H,W] -> [B,C,2,H,W]."""
    kx = torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]], dtype=t.dtype) / 8.0
    k = torch.stack([kx, kx.t()])[:, None]                      # [2,1,3,3]
    B, C, H, W = t.shape
    p = F.pad(t.reshape(B * C, 1, H, W), (1, 1, 1, 1), mode="replicate")
    return F.conv2d(p, k).view(B, C, 2, H, W)


def depth_to_normals(depth, K):
    xyz = depth_to_3d(depth, K)
    g = spatial_gradient_sobel(xyz)
    a, b = g[:, :, 0], g[:, :, 1]
    return F.normalize(torch.cross(a, b, dim=1), dim=1, p=2)


def normals_loss(depth_gt, depth_pred, intrinsics, mask):
    """trainer.py:1298-1309 (note 2 - cos, not 1 - cos)."""
    Km = intrinsics[:, :3, :3]
    cos = F.cosine_similarity(depth_to_normals(depth_gt, Km), depth_to_normals(depth_pred, Km), dim=1).unsqueeze(1)
    return ((2 * torch.ones_like(cos) - cos) * mask).sum() / mask.sum()


def normals_pred_loss(normals_pred, depth_gt, intrinsics, min_depth=0.1, max_depth=2.0):
    """Loss of the `arch1++_separate_normals_dec` variant (README.md:54): the formula of trainer.py:1298-1309 with the
    network's 3-channel output in place of depth_to_normals(depth_pred), masked by the depth range (trainer.py:1242-1243)."""
    mask = ((depth_gt >= min_depth) & (depth_gt <= max_depth)).float()
    cos = F.cosine_similarity(depth_to_normals(depth_gt, intrinsics[:, :3, :3]), normals_pred, dim=1).unsqueeze(1)
    return ((2 * torch.ones_like(cos) - cos) * mask).sum() / mask.sum()


def upsample_disp_to_depth(disp, H, W, min_depth, max_depth):
    """trainer.py:538-543."""
    up = F.interpolate(disp, [H, W], mode="bilinear", align_corners=False)
    return disp_to_depth(up, min_depth, max_depth)[1]


def compute_losses(inputs, outputs, scales=(0, 1, 2, 3), min_depth=0.1, max_depth=2.0,
                   normals_loss_weight=0.35, disparity_smoothness=1e-3, normals_fn=normals_loss):
    """Supervised branch of Trainer.compute_losses (depth_supervision=True, is_multi=False)."""
    losses, total = {}, 0
    gt = inputs["depth"]
    for s in scales:
        disp = outputs[("disp", s)]
        color = inputs[("color", 0, s)]
        mask = (gt >= min_depth).float() * (gt <= max_depth).float()
        depth = outputs[("depth", 0, s)]
        sup = (torch.abs(gt - depth) * mask).sum() / mask.sum()
        loss = sup
        nl = normals_fn(gt, depth, inputs[("K", 0)], mask)
        losses[f"supervised_depth_loss/{s}"] = sup
        losses[f"normals_loss/{s}"] = nl
        loss = loss + normals_loss_weight * nl
        mean_disp = disp.mean(2, True).mean(3, True)
        norm_disp = disp / (mean_disp + 1e-7)
        loss = loss + disparity_smoothness * get_smooth_loss(norm_disp, color) / (2 ** s)
        total = total + loss
        losses[f"loss/{s}"] = loss
    losses["loss"] = total / len(scales)
    return losses


def compute_depth_errors(gt, pred):
    thresh = torch.max(gt / pred, pred / gt)
    a1 = (thresh < 1.25).float().mean()
    a2 = (thresh < 1.25 ** 2).float().mean()
    a3 = (thresh < 1.25 ** 3).float().mean()
    rmse = torch.sqrt(((gt - pred) ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    abs_rel = torch.mean(torch.abs(gt - pred) / gt)
    sq_rel = torch.mean((gt - pred) ** 2 / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3
