"""Test helper: the CPU oracle's version of one supervised training step (forward, multi-scale loss, backward) in fp32
-- what the reference's PyTorch-CPU path computes -- or with every module and input cast to fp64, the yardstick that
tells rounding noise amplified by the network from a wrong kernel (tests/test_prodsize_gpu.py, tools/grad_calibration.py).
Test infrastructure only."""
import copy

import torch


def oracle_grads(ref, cpu, H, W, dtype):
    from oracle import nets as onets, losses as ol, polar as opolar
    xolp, _, _, _ = opolar.polar_forward(cpu[("pol", 0, 0)].numpy())
    models = ref if dtype == torch.float32 else {k: copy.deepcopy(m).double() for k, m in ref.items()}
    for m in models.values():
        m.train()
        for p in m.parameters():
            p.grad = None
    color = cpu[("color_aug", 0, 0)].to(dtype)
    feats = models["rgb_encoder"](color)
    xf = models["xolp_encoder"](xolp.float().to(dtype))
    normals = opolar.get_normals(xolp.float()).float().to(dtype)           # the fp32 values the network consumes
    nf = onets.ShallowEncoder.forward(models["normals_encoder"], normals)
    feats = list(feats) + models["joint_encoder"](feats[-1], xf, nf)
    outs = dict(models["mono_depth"](feats))
    inputs = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in cpu.items()}
    for s in range(4):
        outs[("depth", 0, s)] = ol.upsample_disp_to_depth(outs[("disp", s)], H, W, 0.1, 2.0)
    L = ol.compute_losses(inputs, outs, normals_loss_weight=0.35)
    L["loss"].backward()
    grads = {f"{mn}.{k}": p.grad.detach().double() for mn, m in models.items() for k, p in m.named_parameters()
             if p.grad is not None}
    return grads, {k: float(v.detach()) for k, v in L.items()}, {s: outs[("disp", s)].detach() for s in range(4)}
