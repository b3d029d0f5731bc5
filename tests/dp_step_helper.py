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
    # PD_DP_GRAPH=segmented|capture: two eager steps, then the step as a hipGraph (polardepth/graph.py) for three more --
    # PD_DP_STEPS=5 without it runs the same five steps eagerly
    graph_comm = os.environ.get("PD_DP_GRAPH")
    n_steps = int(os.environ.get("PD_DP_STEPS", 5 if graph_comm else 2))
    gs = None
    host_ms = None
    for step in range(n_steps):
        batch = synthetic.make_batch(2, 64, 96, frame_w=92, device="cuda", seed=step)
        if graph_comm and step >= 2:
            if gs is None:
                from polardepth.graph import GraphedTrainStep
                gs = GraphedTrainStep(tr, batch, warmup=1, restore_state=True, comm=graph_comm)
            import time
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loss = gs.step(batch)
            host_ms = (time.perf_counter() - t0) * 1e3
            losses.append(loss.detach().cpu().clone())
            continue
        tr.model_optimizer.zero_grad()
        _, L, _ = tr.process_batch(dict(batch), is_train=True)
        L["loss"].backward()
        tr.model_optimizer.step()
        losses.append(L["loss"].detach().cpu())
    torch.cuda.synchronize()
    probe = None
    if tr.reducer is not None and tr.reducer.active and os.environ.get("PD_DP_STALL_PROBE") == "1":
        # A busy encoder stream must not hold back a decoder bucket: bucket 0 (decoder parameters, produced on the main
        # and weight-gradient streams) is marked ready while an encoder stream sits in a ~0.3 s kernel with one parameter
        # of a LATER bucket marked behind it; bucket 0's all-reduce has to complete while that kernel still runs.
        import time
        red = tr.reducer
        red.reset()
        enc = tr._enc_streams[0]
        last = tr.store.used_params()[-1]
        assert red.bucket_of[id(last)] == len(red.buckets) - 1 and len(red.buckets) > 1
        with torch.cuda.stream(enc):
            torch.cuda._sleep(int(1e8))        # 50 ms at 2 GHz, 1 s if the counter ticks at 100 MHz
            red.mark_ready(last)
            enc_done = torch.cuda.Event(); enc_done.record()
        for p in tr.store.used_params():
            if red.bucket_of[id(p)] == 0:
                red.mark_ready(p)
        assert red.launched[0] and not red.launched[-1]
        comm_done = torch.cuda.Event(); comm_done.record(red.comm_stream)
        t0 = time.time()
        while not comm_done.query() and time.time() - t0 < 5.0:
            time.sleep(0.001)
        probe = {"comm_done_while_encoder_busy": bool(comm_done.query() and not enc_done.query()),
                 "bucket0_streams": len(red.deps[0]), "waited_s": time.time() - t0}
        red.finish()
        torch.cuda.synchronize()
    info = {"probe": probe, "graph_host_ms": host_ms, "adam_steps": tr.model_optimizer.step_count, "flat": tr.store.flat.cpu(), "losses": torch.stack(losses), "distributed": tr.distributed,
            "reducer_active": bool(tr.reducer is not None and tr.reducer.active),
            "buckets": 0 if tr.reducer is None else len(tr.reducer.buckets),
            "global_norm": tr.loss_cfg.global_norm}
    torch.save(info, out_path)
    # teardown in dependency order: the captured graph (it may hold RCCL nodes), then the process group, then the interpreter
    del gs
    import gc
    gc.collect()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
