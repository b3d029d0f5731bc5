"""The kernels the bench line is graded on, at the tile sizes they run in production, against plain PyTorch fp32 on the
CPU (convolutions, BatchNorm) and against the CPU oracle (one whole training step at 512x640).

The small-size suites (test_conv_gpu.py, test_modules_gpu.py, test_step_gpu.py: M <= 3072 pixels) take the 64x64 /
128x32 tiles, the few-slice weight gradient and the single-workgroup BatchNorm reduction.  Production
(BASELINE.json configs[2]: batch 16 at 512x640) takes
  * ``conv_igemm_uni_kernel<128,64,64,32,{ZERO,TRANSPOSED,REFLECT},2>`` for every layer with M >= 65 536 output pixels
    and more than 32 output channels (csrc/conv.hip: pd_conv2d_tile_m) -- its own TM = 2 fragment layout and the
    64x32 transposed epilogue;
  * ``conv_wgrad_uni_kernel`` with 15..153 pixel slices per tile (wgrad_plan: one or two rounds of the 768 resident
    workgroups) and the ordered ``reduce_rows`` pass over them;
  * ``bn_fwd_stats_kernel`` / ``bn_bwd_stats_kernel`` (more than 4096 partial rows: fp64 atomics + last-arrival ticket).
Every case asserts which kernel label the call took, so a change of the dispatch rule cannot silently move these
tests back onto the small tiles.

Tolerances: fp32 MFMA accumulation in a fixed k order vs the CPU's blocked order: |err| <= ~1e-6 * sum|a b|, checked
as 2e-5 of the output scale (same bar as test_conv_gpu.py); whole step: disparities 2e-5 abs, loss 1e-4 relative
(north_star), per-tensor gradient: at least as close to the fp64 oracle as the fp32 oracle is (x1.5)."""
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from polardepth import ops

pytestmark = pytest.mark.gpu


def _close(got, ref, tol=2e-5, what=""):
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"{what} max err {err:.3e} vs scale {scale:.3e}"


def _labels(fn):
    ops.PROFILE = []
    try:
        out = fn()
        return out, [p[0] for p in ops.PROFILE]
    finally:
        ops.PROFILE = None


PROD_CASES = [
    # N, C, H, W, Co, k, s, p                                                (production layer, per-image M)
    (1, 64, 256, 320, 64, 3, 1, 1),        # ShallowEncoder ResBlock1 conv: M = 81 920, K = 576 (4.5 k-tiles: split-k wgrad tile)
    (2, 64, 256, 320, 64, 5, 1, 2),        # ShallowEncoder Conv2 5x5: the most expensive layer of the net, K = 1600
    (3, 64, 150, 150, 64, 3, 1, 1),        # M = 67 500 = 527 tiles + 44 rows: M tail, tiles that span two images
    (16, 128, 64, 80, 256, 5, 1, 2),       # JointEncoder Conv1 at batch 16: M = 81 920, 4 Cout tiles, 100 wgrad tiles (two rounds)
    (16, 128, 64, 80, 128, 3, 1, 1),       # JointEncoder ResBlock1/2 at batch 16
    (16, 64, 128, 160, 128, 3, 2, 1),      # resnet layer2.0.conv1 3x3/s2 at batch 16 (dgrad: four parity-class launches), M = 81 920
    (1, 192, 256, 320, 64, 1, 1, 0),       # 1x1 on the 128-row tile (fc1-like, pad 0)
]


@pytest.mark.parametrize("x3", ["0", "1"])
@pytest.mark.parametrize("case", PROD_CASES)
def test_production_tile_conv_forward_dgrad_wgrad(case, x3, monkeypatch):
    """x3 = "0": every launch on the fp32-MFMA production tiles; "1" (the default): the shapes pd_conv2d_uses_x3 accepts go
    to the bf16-split kernel (256 x 64 tiles), the others stay where they were."""
    fl = ops.CONV_AUTO if x3 == "1" else ops.CONV_FP32_MFMA       # the `flags` word of pd_conv2d* / pd_conv2d_wgrad
    monkeypatch.setattr(ops, "CONV_FLAGS", fl)
    monkeypatch.setattr(ops, "WGRAD_FLAGS", fl)   # weight gradient: conv_wgrad_x3c_kernel | the fp32-MFMA scalar-pixel kernel
    N, C, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, k, k, generator=g) / (C * k * k) ** 0.5
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, stride=s, padding=p)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    M = ref.shape[0] * ref.shape[2] * ref.shape[3]
    assert M >= 65536 and Co > 32, "not a production-tile case"
    xd = x.cuda().contiguous(memory_format=torch.channels_last)
    wd = w.cuda().contiguous(memory_format=torch.channels_last)

    def expect(Mx, Cox, Cx, sx, mode, ho, wo):
        rb = ops.lib.pd_conv2d_uses_x3(Mx, Cox, Cx, k, k, sx, p, mode, 0, 0, ho, wo, fl)
        want = 0
        if x3 == "1" and Cox % 64 == 0 and Cx % 4 == 0 and -(-Cx // 16) * 16 <= 2 * Cx:
            # (a data gradient -- mode 2, no BatchNorm statistics -- may end in a partial 128-row tile)
            want = 2 if (Mx % 256 == 0 and (Mx // 256) * (Cox // 64) >= 512) else 1 if ((Mx % 128 == 0 or mode == 2) and (Mx // 128) * (Cox // 64) >= 320) else 0
            tw = 32 if (wo % 32 == 0 and ho % 8 == 0) else 16 if (wo % 16 == 0 and ho % 16 == 0) else 8 if (wo % 8 == 0 and ho % 32 == 0) else 0
            # the halo-tile kernel: 64-column workgroups where they are >= 512, else 32-column ones (3x3 off the 32 x 8 tiles; 5x5 on them)
            if k in (3, 5) and sx == 1 and Cx % 16 == 0 and tw and Mx >= 65536:
                if (Mx // 256) * (Cox // 64) >= 512 or ((Mx // 256) * (Cox // 32) >= 512 and ((k == 3 and tw != 8) or (k == 5 and tw == 8))):
                    want = 3
        assert rb == want, (rb, want)
        return ["conv_halo_x3_kernel<8x32,64>" if rb == 3 else f"conv_igemm_x3_kernel<{128 * rb},64>" if rb else "conv_igemm_uni_kernel<128,64>"]

    (y, stats), lab = _labels(lambda: ops.conv2d_fwd(xd, wd, None, stride=s, pad=p, want_stats=True))
    assert lab == expect(M, Co, C, s, 0, ref.shape[2], ref.shape[3]), lab
    _close(y.cpu(), ref.detach(), what="fwd")
    # BatchNorm partials of the transposed epilogue: one row per 128-pixel tile
    assert stats.shape[0] == (M + 127) // 128
    tot = stats.double().sum(0).cpu()
    rf = ref.detach().double()
    # The column sums are sums of M zero-mean outputs: their own magnitude (~ sqrt(M)) is no yardstick.  (a) the epilogue's
    # sums against the sums of the tensor it stored; (b) the mean of the output against the reference's, in units of the
    # output's rms: the bf16 MFMA truncates (two's complement) where it aligns its addends, which shifts every output of
    # the split kernel by about -1e-7 rms at K = 1600 (fp32 MFMA: 2e-10) -- far below what a BatchNorm mean can see, but
    # M times that in the sum
    yd = y.cpu().double()
    _close(tot[:, 0], yd.sum((0, 2, 3)), 1e-6, "stats sum vs stored tensor")
    assert ((tot[:, 0] - rf.sum((0, 2, 3))).abs().max().item() / M) <= 2.5e-7 * rf.pow(2).mean().sqrt().item(), "stats mean"
    _close(tot[:, 1], (rf ** 2).sum((0, 2, 3)), 1e-5, "stats sumsq")
    # ... and the epilogue without statistics, with bias + ReLU (element-wise epilogue of the same tile)
    b = torch.randn(Co, generator=g)
    _close(ops.conv2d_fwd(xd, wd, b.cuda(), stride=s, pad=p, act=ops.ACT_RELU).cpu(),
           F.relu(ref.detach() + b[None, :, None, None]), what="fwd bias relu")

    dyd = dy.cuda().contiguous(memory_format=torch.channels_last)
    S = ops.lib.pd_conv2d_wgrad_workspace(M, Co, k * k * C, fl) // (4 * (Co * k * k * C + Co))
    # (the bf16-split kernel plans for its 512 resident workgroups: one round when that leaves >= 4 slices per tile)
    assert S >= (4 if x3 == "1" else 15), f"weight gradient not in the many-slice regime (S = {S})"
    dw, lab = _labels(lambda: ops.conv2d_wgrad(xd, dyd, w.shape, stride=s, pad=p))
    ho_, wo_ = ref.shape[2], ref.shape[3]
    halo_w = (x3 == "1" and k in (3, 5) and s == 1 and C % 64 == 0 and Co % 64 == 0 and
              ((wo_ % 32 == 0 and ho_ % 2 == 0) or (wo_ % 16 == 0 and ho_ % 4 == 0) or (wo_ % 8 == 0 and ho_ % 8 == 0)))
    assert lab == (["conv_wgrad_halo_x3_kernel"] if halo_w else ["conv_wgrad_x3c_kernel"] if x3 == "1" else ["conv_wgrad_kernel"]), lab
    _close(dw.cpu(), wr.grad, what="wgrad")

    dx, lab = _labels(lambda: ops.conv2d_dgrad(dyd, wd, (H, W), stride=s, pad=p))
    if s == 1:
        assert lab == expect(N * H * W, C, Co, 1, 2, H, W), lab
    else:
        assert lab == ["conv_dgrad_s2_phases"], lab
    _close(dx.cpu(), xr.grad, what="dgrad")
    if s == 1:      # the residual blocks' skip gradient rides in the data-gradient epilogue
        add = torch.randn(N, C, H, W, generator=g)
        addd = add.cuda().contiguous(memory_format=torch.channels_last)
        _close(ops.conv2d_dgrad(dyd, wd, (H, W), stride=s, pad=p, addend=addd).cpu(), xr.grad + add, what="dgrad + addend")


@pytest.mark.parametrize("x3", ["0", "1"])
@pytest.mark.parametrize("case", [(4, 192, 128, 160, 64), (1, 96, 256, 320, 64), (3, 128, 150, 158, 64)])
def test_production_tile_reflection_padded_conv(case, x3, monkeypatch):
    """Decoder upconvs (ReflectionPad2d(1) + Conv3x3 + ELU) on the 128x64 REFLECT instantiation + the reflect weight /
    bias gradient in its many-slice regime + the data gradient (pad-1 dgrad + border strips) through the autograd node."""
    fl = ops.CONV_AUTO if x3 == "1" else ops.CONV_FP32_MFMA
    monkeypatch.setattr(ops, "CONV_FLAGS", fl)
    monkeypatch.setattr(ops, "WGRAD_FLAGS", fl)   # reflect weight gradient: bf16-split | fp32 MFMA kernel
    from polardepth import functional as PF
    N, C, H, W, Co = case
    g = torch.Generator().manual_seed(sum(case) + 3)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(Co, generator=g) * 0.3
    xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    ref = F.elu(F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), wr, br))
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    assert N * H * W >= 65536
    conv = torch.nn.Conv2d(C, Co, 3).cuda()
    conv.weight.data = w.cuda().contiguous(memory_format=torch.channels_last); conv.bias.data = b.cuda()
    xc = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)

    def run():
        y = PF.reflect_conv_act(xc, conv, ops.ACT_ELU)
        (y * dy.cuda()).sum().backward()
        PF.sync_wgrad_stream()
        return y
    y, lab = _labels(run)
    # (forward on the REFLECT instantiation; the pad-1 data gradient is a zero-padding launch: fp32 or bf16-split kernel)
    assert sum(l.startswith(("conv_igemm_uni_kernel<128,64>", "conv_igemm_x3_kernel", "conv_halo_x3_kernel")) for l in lab) == 2 and (({"conv_wgrad_x3c_kernel", "conv_wgrad_halo_x3_kernel", "conv_wgrad_roll_x3_kernel"} & set(lab)) if x3 == "1" else "conv_wgrad_kernel" in lab), lab
    _close(y.detach().cpu(), ref.detach(), 3e-5, "fwd")
    _close(xc.grad.cpu(), xr.grad, 3e-5, "dgrad")
    _close(conv.weight.grad.cpu(), wr.grad, 3e-5, "wgrad")
    _close(conv.bias.grad.cpu(), br.grad, 3e-5, "bias grad")


@pytest.mark.parametrize("R,C", [(5000, 64), (10240, 64), (4097, 96), (20480, 256)])
def test_batchnorm_finalize_ticket_kernels(R, C):
    """pd_bn_fwd_finalize / pd_bn_bwd_finalize above 4096 partial rows (bn_fwd_stats_kernel / bn_bwd_stats_kernel:
    per-workgroup fp64 atomics, the last workgroup by ticket finalises) against fp64 NumPy, twice in a row on the same
    accumulator: the kernel must leave the sums and the ticket word zero for the next layer."""
    from polardepth._lib import lib, check, ptr, stream_ptr
    rng = np.random.default_rng(R + C)
    part = rng.standard_normal((R, C, 2)).astype(np.float32)
    part[:, :, 1] = np.abs(part[:, :, 1]) * 3 + 2.0          # sum of squares >= (sum)^2 / n for a plausible variance
    gamma = rng.standard_normal(C).astype(np.float32); beta = rng.standard_normal(C).astype(np.float32)
    rm0 = rng.standard_normal(C).astype(np.float32); rv0 = (rng.random(C) + 0.5).astype(np.float32)
    count = float(R * 128)
    s = part.astype(np.float64).sum(0)
    mean = s[:, 0] / count
    var = np.maximum(s[:, 1] / count - mean * mean, 0.0)
    invstd = 1.0 / np.sqrt(var + 1e-5)
    dev = torch.device("cuda")
    acc = torch.zeros(2 * C + 2, dtype=torch.float64, device=dev)
    pd_ = torch.from_numpy(part).to(dev)
    gd, bd = torch.from_numpy(gamma).to(dev), torch.from_numpy(beta).to(dev)
    for rep in range(2):
        rm, rv = torch.from_numpy(rm0.copy()).to(dev), torch.from_numpy(rv0.copy()).to(dev)
        scale, shift, smean, sinv = (torch.empty(C, device=dev) for _ in range(4))
        check(lib.pd_bn_fwd_finalize(ptr(pd_), R, C, count, ptr(gd), ptr(bd),
                                     ptr(rm), ptr(rv), 0.1, 1e-5, ptr(acc), acc.numel(), ptr(scale), ptr(shift), ptr(smean), ptr(sinv), 1,
                                     stream_ptr()), "pd_bn_fwd_finalize")
        torch.cuda.synchronize()
        np.testing.assert_allclose(smean.cpu().numpy(), mean, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(sinv.cpu().numpy(), invstd, rtol=1e-6)
        np.testing.assert_allclose(scale.cpu().numpy(), gamma * invstd, rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(shift.cpu().numpy(), beta - mean * (gamma * invstd), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(rm.cpu().numpy(), 0.9 * rm0 + 0.1 * mean, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(rv.cpu().numpy(), 0.9 * rv0 + 0.1 * var * count / (count - 1), rtol=1e-6, atol=1e-7)
        assert acc.abs().max().item() == 0.0, "accumulator / ticket not left zero"
        dgamma, dbeta = torch.ones(C, device=dev), torch.ones(C, device=dev)
        coef = torch.empty(2 * C, device=dev)
        check(lib.pd_bn_bwd_finalize(ptr(pd_), R, C, count, ptr(acc), acc.numel(), ptr(dgamma), ptr(dbeta), ptr(coef), 1, stream_ptr()),
              "pd_bn_bwd_finalize")
        torch.cuda.synchronize()
        np.testing.assert_allclose(dbeta.cpu().numpy(), 1.0 + s[:, 0], rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(dgamma.cpu().numpy(), 1.0 + s[:, 1], rtol=1e-6, atol=1e-4)
        np.testing.assert_allclose(coef.cpu().numpy(), np.concatenate([s[:, 0], s[:, 1]]) / count, rtol=1e-6, atol=1e-9)
        assert acc.abs().max().item() == 0.0
    # the accumulator length is part of the contract: one double short of the ticket word is refused
    from polardepth._lib import PolarDepthError
    with pytest.raises(PolarDepthError):
        check(lib.pd_bn_bwd_finalize(ptr(pd_), R, C, count, ptr(acc), 2 * C, ptr(dgamma), ptr(dbeta), ptr(coef), 1, stream_ptr()),
              "pd_bn_bwd_finalize")


def test_conv_block_with_more_than_4096_stat_rows_matches_torch():
    """pre_encoders.ConvBlock (conv(bias) -> BN(train) -> ReLU) at 7 x 64 x 256 x 320: 4480 partial rows -> the ticket branch of
    the statistics kernel inside the real layer, forward and backward, against torch.nn on the CPU."""
    from manydepth.networks.pre_encoders import ConvBlock
    torch.manual_seed(3)
    N, C, H, W = 7, 64, 256, 320
    blk = ConvBlock(C, C, 3, 'none', 1, 0.0)
    ref = torch.nn.Sequential(torch.nn.Conv2d(C, C, 3, 1, 1), torch.nn.BatchNorm2d(C), torch.nn.ReLU())
    ref[0].load_state_dict(blk.conv.state_dict()); ref[1].load_state_dict(blk.bn.state_dict())
    ref.train()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(N, C, H, W, generator=g)
    gy = torch.randn(N, C, H, W, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    assert ops.lib.pd_conv2d_stats_rows(N * H * W, C) > 4096
    blk = blk.cuda().train()
    blk.conv.weight.data = blk.conv.weight.data.contiguous(memory_format=torch.channels_last)
    xc = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    from polardepth import functional as PF
    y = blk(xc)
    _close(y.detach().cpu(), yr.detach(), 2e-5, "fwd")
    _close(blk.bn.running_mean.cpu(), ref[1].running_mean, 1e-5, "running_mean")
    _close(blk.bn.running_var.cpu(), ref[1].running_var, 1e-5, "running_var")
    # A ReLU whose pre-activation is within fp32 rounding of zero (~1e-6 of the 36.7 M elements) may switch the other
    # way on the two sides, and one switched element moves 576 entries of dW by O(1): the seed gradient is zeroed within
    # 1e-4 of the kink (on either side's output), so that the comparison does not depend on rounding-decided branches
    near = ((yr.detach() > 0) & (yr.detach() < 1e-4)) | ((y.detach().cpu() > 0) & (y.detach().cpu() < 1e-4))
    assert 0 < near.float().mean().item() < 1e-3
    gy = torch.where(near, torch.zeros_like(gy), gy)
    (yr * gy).sum().backward()
    (y * gy.cuda()).sum().backward()
    PF.sync_wgrad_stream()
    torch.cuda.synchronize()
    _close(xc.grad.cpu(), xr.grad, 1e-4, "dx")
    _close(blk.conv.weight.grad.cpu(), ref[0].weight.grad, 1e-4, "dw")
    _close(blk.bn.weight.grad.cpu(), ref[1].weight.grad, 1e-4, "dgamma")
    _close(blk.bn.bias.grad.cpu(), ref[1].bias.grad, 1e-4, "dbeta")


@pytest.mark.parametrize("B", [4, 16])
def test_full_resolution_training_step_matches_oracle(B, tmp_path, monkeypatch):
    """BASELINE configs[2] at batch 4 and at its own batch 16: K1 -> three encoders -> joint encoder -> decoder -> multi-scale
    loss -> backward on 512x640 frames (dropout 0, BatchNorm in training mode), HIP path vs the CPU oracle with identical
    weights and batch.  At this size every 256x320 / 128x160 layer runs the 128x64 tile, the weight gradients their
    many-slice plans, the 16-channel halo kernels and tiled disparity heads their multi-tile grids.  Batch 16 is the shape
    bench.py times: exactly its tile sizes (256- vs 128-row split tiles follow M = B H W), slice plans and launch labels;
    it runs the default (bf16-split) family only -- two oracle passes at that size are ~2 minutes of CPU."""
    sys.path.insert(0, GOLDEN)
    from synth_weights import fill_state_dict
    import bench
    from manydepth.options import MonodepthOptions
    from manydepth.trainer import Trainer
    from polardepth import synthetic
    from polardepth import functional as PF
    from oracle import nets as onets
    from oracle_step import oracle_grads
    H, W = bench.H, bench.W
    opts = MonodepthOptions().parse([
        "--png", "--batch_size", str(B), "--height", str(H), "--width", str(W), "--dataset", "HAMMER", "--split", "HAMMER",
        "--eval_split", "HAMMER_unseen", "--min_depth", "0.1", "--max_depth", "2.0", "--depth_supervision_only", "True",
        "--depth_supervision", "True", "--normals_loss_weight", "0.35", "--augment_xolp", "--augment_normals",
        "--log_dir", str(tmp_path), "--data_path", "synthetic", "--data_path_val", "synthetic", "--num_workers", "0",
        "--weights_init", "scratch", "--learning_rate", "1e-4", "--dropout_rate", "0.0"])
    tr = Trainer(opts)
    ref = onets.build_models(True, True, 0.0)
    for name, m in ref.items():
        fill_state_dict(m, 0, prefix=name + ".")
        tr.models[name].load_state_dict(m.state_dict())
        m.train()
    tr.set_train()
    batch = synthetic.make_batch(B, H, W, frame_w=bench.FRAME_W, device="cuda", seed=21)
    cpu = {k: v.cpu() for k, v in batch.items()}

    def step():
        outputs, losses, _ = tr.process_batch(dict(batch), is_train=True)
        losses["loss"].backward()
        PF.sync_wgrad_stream()
        return outputs, losses

    # Three-way comparison (tools/grad_calibration.py prints the table): the oracle in fp32 is the reference's arithmetic,
    # the oracle in fp64 the yardstick.  Training-mode BatchNorm through ~25 layers amplifies fp32 rounding of the
    # encoder gradients to ~1e-2 relative on BOTH fp32 paths (decoder gradients, in front of the first BatchNorm: 1e-5),
    # so "equal to the fp32 oracle within 5e-3" is not a meaningful bar at this size; "as close to exact arithmetic as the
    # fp32 oracle is" is.  Both kernel families run the same step from the same weights:
    #   fp32 MFMA (flags PD_CONV_FP32_MFMA): every tensor within 1.5x the fp32 oracle's distance (measured: median of
    #     the ratios 0.84, worst 1.15);
    #   bf16-split kernels (the default): median ratio <= 1.25, every tensor within 2x (measured: 1.07, worst 1.52 -- the
    #     split kernels are as close to fp64 as the CPU's fp32 arithmetic, the fp32-MFMA kernels a little closer).
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(nthreads, 16))       # the GPU box's CPU share; 256 oversubscribed threads are slower than 16
    try:
        g64, L64, d64 = oracle_grads(ref, cpu, H, W, torch.float64)
        g32, L32, d32 = oracle_grads(ref, cpu, H, W, torch.float32)
    finally:
        torch.set_num_threads(nthreads)
    import os
    import statistics
    families = (("bf16-split", "1", 2.0, 1.25), ("fp32 MFMA", "0", 1.5, 1.25))
    if B == 16:
        # Batch 16: the fp32 oracle's own distance from fp64 shrinks (oneDNN's blocked sums average rounding noise over 4x the
        # pixels) while the kernels' fp32 accumulators run over slices that are 4x longer; the per-tensor bar stays, the
        # median bar is the kernel-level one of tests/test_conv_gpu.py (1.5x the fp32 arithmetic's error).  Measured round 4:
        # median 1.27 with the sign-alternating accumulation (1.31 without it), worst tensor < 2.
        families = (("bf16-split", "1", 2.0, 1.5),)
    for family, knob, per_tensor, median_bar in families:
        monkeypatch.setattr(ops, "CONV_FLAGS", ops.CONV_AUTO if knob == "1" else ops.CONV_FP32_MFMA)
        monkeypatch.setattr(ops, "WGRAD_FLAGS", ops.CONV_AUTO if knob == "1" else ops.CONV_FP32_MFMA)
        tr.model_optimizer.zero_grad()
        (outputs, losses), lab = _labels(step)
        torch.cuda.synchronize()
        n_prod = sum(l.startswith(("conv_igemm_uni_kernel<128,64>", "conv_igemm_x3_kernel", "conv_halo_x3_kernel")) for l in lab)
        assert n_prod >= 30, f"production tiles not exercised: {n_prod}"
        assert any(l.startswith(("conv_igemm_x3_kernel", "conv_halo_x3_kernel")) for l in lab) == (knob == "1")
        if B == 16 and knob == "1":          # the launch labels of bench.py's step: both split tile sizes and the split weight gradient
            assert {"conv_halo_x3_kernel<8x32,64>", "conv_igemm_x3_kernel<128,64>", "conv_wgrad_x3c_kernel", "conv_wgrad_halo_x3_kernel",
                    "conv_wgrad_roll_x3_kernel"} <= set(lab), sorted(set(lab))
        gpu_grads = {f"{mn}.{k}": v.grad.detach().cpu().clone() for mn in tr.models for k, v in tr.models[mn].named_parameters()
                     if v.grad is not None}
        for s in range(4):
            d = outputs[("disp", s)].detach().cpu()
            assert (d - d32[s]).abs().max().item() < 2e-5, f"{family}: disp {s} vs fp32 oracle"
            assert (d.double() - d64[s]).abs().max().item() < 2e-5, f"{family}: disp {s} vs fp64 oracle"
        for k in ("loss", "loss/0", "loss/1", "loss/2", "loss/3", "supervised_depth_loss/0", "supervised_depth_loss/3"):
            assert abs(losses[k].item() - L32[k]) <= 1e-4 * abs(L32[k]), (family, k, losses[k].item(), L32[k])
            assert abs(losses[k].item() - L64[k]) <= 1e-4 * abs(L64[k]), (family, k, losses[k].item(), L64[k])
        bad, rows = [], []
        for k, g in g64.items():
            if k.endswith("conv.bias") and not k.startswith("mono_depth"):
                continue                         # bias in front of BatchNorm: exactly 0 here, rounding noise in torch
            n = g.norm().item() + 1e-30
            e_hip = (gpu_grads[k].double() - g).norm().item() / n
            e_cpu = (g32[k] - g).norm().item() / n
            rows.append((k, e_hip, e_cpu))
            if e_hip > max(1e-4, per_tensor * e_cpu):
                bad.append((k, e_hip, e_cpu))
        assert len(rows) > 150
        if os.environ.get("PD_TEST_VERBOSE"):
            for r in rows:
                print("%-12s %-60s hip-fp64 %.2e  cpu32-fp64 %.2e" % ((family,) + r))
        assert not bad, (family, bad)
        med = statistics.median(r[1] / r[2] for r in rows if r[2] > 1e-6)
        assert med <= median_bar, (family, med)
        dec = [r for r in rows if r[0].startswith("mono_depth")]
        assert max(r[1] for r in dec) < 2e-4, f"{family}: decoder gradients (no BatchNorm between them and the loss)"
        print("full-resolution step, %s kernels: %d gradient tensors; worst hip-fp64 %.2e, worst cpu32-fp64 %.2e; median ratio %.2f; "
              "hip closer on %d" % (family, len(rows), max(r[1] for r in rows), max(r[2] for r in rows), med, sum(r[1] <= r[2] for r in rows)))
