"""Child process of tests/test_dp_gpu.py: two training steps through the Trainer on cuda:0, flat parameters saved to argv[1].
With PD_DIST_TEST=1 the process group is initialised (world 1, backend "nccl" = RCCL): the reducer's bucket
bookkeeping, the side comm stream and the RCCL all-reduce are all live, as with N ranks."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main(out_path):
    if os.environ.get("PD_DIST_TEST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1)
    from manydepth.options import MonodepthOptions
    from manydepth.trainer import Trainer
    from polardepth import synthetic
    from polardepth import functional as PF
    torch.manual_seed(0)
    opts = MonodepthOptions().parse([
        "--png", "--batch_size", "2", "--height", "64", "--width", "96", "--dataset", "HAMMER", "--split", "HAMMER",
        "--eval_split", "HAMMER_unseen", "--depth_supervision_only", "True", "--depth_supervision", "True",
        "--normals_loss_weight", "0.35", "--augment_xolp", "--augment_normals", "--log_dir", tempfile.mkdtemp(),
        "--data_path", "synthetic", "--data_path_val", "synthetic", "--num_workers", "0", "--weights_init", "scratch"])
    tr = Trainer(opts)
    PF.DropoutState.manual_seed(7)
    tr.set_train()
    losses = []
    for step in range(2):
        batch = synthetic.make_batch(2, 64, 96, frame_w=92, device="cuda", seed=step)
        tr.model_optimizer.zero_grad()
        _, L, _ = tr.process_batch(dict(batch), is_train=True)
        L["loss"].backward()
        tr.model_optimizer.step()
        losses.append(L["loss"].detach().cpu())
    torch.cuda.synchronize()
    info = {"flat": tr.store.flat.cpu(), "losses": torch.stack(losses), "distributed": tr.distributed,
            "reducer_active": bool(tr.reducer is not None and tr.reducer.active),
            "buckets": 0 if tr.reducer is None else len(tr.reducer.buckets),
            "global_norm": tr.loss_cfg.global_norm}
    torch.save(info, out_path)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
