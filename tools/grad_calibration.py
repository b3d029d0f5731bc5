#!/usr/bin/env python3
"""How far is each parameter gradient of one full-resolution training step from exact arithmetic?

Three computations of the same step (same weights, same batch, dropout 0, BatchNorm in training mode):
  hip   -- the product path on the MI355X (fp32),
  cpu32 -- the oracle in fp32 (what the reference's PyTorch-CPU path computes),
  cpu64 -- the oracle with every module and input cast to fp64 (the yardstick).
Prints per tensor: rel L2 of hip vs cpu64, cpu32 vs cpu64, hip vs cpu32.  The test
tests/test_prodsize_gpu.py::test_full_resolution_training_step_matches_oracle uses the same three-way scheme.

    python tools/grad_calibration.py [batch]          (on the GPU box)
"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"), os.path.join(ROOT, "tests"),
          os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import torch  # noqa: E402


from oracle_step import oracle_grads  # noqa: E402


def main():
    from synth_weights import fill_state_dict
    import bench
    from manydepth.options import MonodepthOptions
    from manydepth.trainer import Trainer
    from polardepth import synthetic
    from polardepth import functional as PF
    from oracle import nets as onets
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    H, W = bench.H, bench.W
    opts = MonodepthOptions().parse([
        "--png", "--batch_size", str(B), "--height", str(H), "--width", str(W), "--dataset", "HAMMER", "--split", "HAMMER",
        "--eval_split", "HAMMER_unseen", "--min_depth", "0.1", "--max_depth", "2.0", "--depth_supervision_only", "True",
        "--depth_supervision", "True", "--normals_loss_weight", "0.35", "--augment_xolp", "--augment_normals",
        "--log_dir", tempfile.mkdtemp(), "--data_path", "synthetic", "--data_path_val", "synthetic", "--num_workers", "0",
        "--weights_init", "scratch", "--learning_rate", "1e-4", "--dropout_rate", "0.0"])
    tr = Trainer(opts)
    ref = onets.build_models(True, True, 0.0)
    for name, m in ref.items():
        fill_state_dict(m, 0, prefix=name + ".")
        tr.models[name].load_state_dict(m.state_dict())
    tr.set_train()
    batch = synthetic.make_batch(B, H, W, frame_w=bench.FRAME_W, device="cuda", seed=21)
    cpu = {k: v.cpu() for k, v in batch.items()}
    tr.model_optimizer.zero_grad()
    outputs, losses, _ = tr.process_batch(dict(batch), is_train=True)
    losses["loss"].backward()
    PF.sync_wgrad_stream()
    torch.cuda.synchronize()
    hip = {f"{mn}.{k}": v.grad.detach().cpu().double() for mn in tr.models for k, v in tr.models[mn].named_parameters()
           if v.grad is not None}
    g64, L64, d64 = oracle_grads(ref, cpu, H, W, torch.float64)
    g32, L32, d32 = oracle_grads(ref, cpu, H, W, torch.float32)
    print(f"loss hip {losses['loss'].item():.8f} cpu32 {L32['loss']:.8f} cpu64 {L64['loss']:.8f}")
    for s in range(4):
        dh = outputs[("disp", s)].detach().cpu().double()
        print(f"disp {s}: hip-cpu64 {(dh - d64[s]).abs().max():.2e}  cpu32-cpu64 {(d32[s].double() - d64[s]).abs().max():.2e}")
    print(f"{'tensor':58s} {'hip-64':>9s} {'cpu32-64':>9s} {'hip-cpu32':>9s}")
    for k, g in g64.items():
        if k.endswith("conv.bias") and not k.startswith("mono_depth"):
            continue
        n = g.norm() + 1e-30
        print(f"{k:58s} {float((hip[k] - g).norm() / n):9.2e} {float((g32[k] - g).norm() / n):9.2e} "
              f"{float((hip[k] - g32[k]).norm() / n):9.2e}")


if __name__ == "__main__":
    main()
