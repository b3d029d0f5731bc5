"""The training step as a hipGraph (polardepth/graph.py) against the eager step: same weights, same batches, same dropout
seed -> bit-identical parameters, Adam moments and losses after several steps (dropout 0.1 active: the masks come from the
device-side step counter in both modes), and a host cost per step that no longer scales with the ~550 launches."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer(tmp_path, tag):
    from test_step_gpu import _opts
    from manydepth.trainer import Trainer
    torch.manual_seed(0)
    return Trainer(_opts(tmp_path / tag, ["--dropout_rate", "0.1"]))


def test_graphed_step_is_bit_identical_to_the_eager_step(tmp_path):
    from polardepth import functional as PF
    from polardepth import synthetic
    from polardepth.graph import GraphedTrainStep
    batches = [synthetic.make_batch(2, 64, 96, frame_w=92, device="cuda", seed=s) for s in range(7)]

    PF.DropoutState.manual_seed(99)
    tr_e = _trainer(tmp_path, "eager")
    tr_e.set_train()
    losses_e = []
    for b in batches:
        tr_e.model_optimizer.zero_grad()
        _, L, _ = tr_e.process_batch(dict(b), is_train=True)
        L["loss"].backward()
        tr_e.model_optimizer.step()
        losses_e.append(L["loss"].detach().clone())
    torch.cuda.synchronize()

    PF.DropoutState.manual_seed(99)
    tr_g = _trainer(tmp_path, "graph")
    assert torch.equal(tr_g.store.flat, _trainer(tmp_path, "init").store.flat)
    # three eager steps, then capture and replay the rest
    losses_g = []
    gs = None
    for i, b in enumerate(batches):
        if i < 3:
            tr_g.set_train()
            tr_g.model_optimizer.zero_grad()
            _, L, _ = tr_g.process_batch(dict(b), is_train=True)
            L["loss"].backward()
            tr_g.model_optimizer.step()
            losses_g.append(L["loss"].detach().clone())
        else:
            if gs is None:       # built in the middle of the run: the warm-up step runs on a snapshot that is restored
                gs = GraphedTrainStep(tr_g, b, warmup=1, restore_state=True)
            losses_g.append(gs.step(b).detach().clone())
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(losses_e, losses_g)):
        assert torch.equal(a, b), f"loss of step {i}: eager {a.item()!r} graph {b.item()!r}"
    assert torch.equal(tr_e.store.flat, tr_g.store.flat)
    assert torch.equal(tr_e.model_optimizer.exp_avg, tr_g.model_optimizer.exp_avg)
    assert torch.equal(tr_e.model_optimizer.exp_avg_sq, tr_g.model_optimizer.exp_avg_sq)
    assert tr_e.model_optimizer.step_count == tr_g.model_optimizer.step_count == len(batches)
    for (n1, b1), (n2, b2) in zip(tr_e.models["joint_encoder"].named_buffers(), tr_g.models["joint_encoder"].named_buffers()):
        assert torch.equal(b1, b2), n1
    # dropout masks do change from replay to replay (the step counter in device memory advances)
    assert len({round(x.item(), 7) for x in losses_g[3:]}) > 1

    # host cost of one replay: a few hundred microseconds, not the ~10 ms of 550 Python launches
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gs.step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    assert host < 5e-3, f"replay took {host * 1e3:.2f} ms of host time"       # (0.3-1 ms here; headroom for a slow host)


def test_run_epoch_with_the_graphed_step_matches_the_eager_epoch(tmp_path, monkeypatch):
    """Trainer.run_epoch (trainer.py:430-442) with PD_STEP_GRAPH=1: the graph is captured on the first batch of the loader
    (warm-up on a snapshot), every batch is copied into the static inputs and replayed -- same parameters after the epoch
    as the eager loop, step / epoch bookkeeping unchanged."""
    from polardepth import functional as PF
    out = []
    for graph in ("0", "1"):
        monkeypatch.setenv("PD_STEP_GRAPH", graph)
        PF.DropoutState.manual_seed(5)
        tr = _trainer(tmp_path, "epoch" + graph)
        tr.opt.log_frequency = 10 ** 9          # no logging / validation inside the epoch
        tr.step = 1
        tr.run_epoch()
        torch.cuda.synchronize()
        assert tr.step == 1 + len(tr.train_loader)
        out.append((tr.store.flat.clone(), tr.model_optimizer.step_count, tr.model_optimizer.exp_avg.clone()))
    assert out[0][1] == out[1][1] > 0
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][2], out[1][2])


def test_graphed_step_follows_the_learning_rate_schedule(tmp_path, monkeypatch):
    """trainer.py:238-240,467: StepLR multiplies the learning rate by 0.1 every scheduler_step_size epochs.  lr (and the
    reducer's grad_scale) are device words of the optimizer (pd_step_set_hyper), not arguments frozen at capture: two
    epochs with scheduler_step_size = 1 through the captured step leave the parameters of the eager run, bit for bit --
    and they differ from a run whose learning rate stayed at its first value."""
    from polardepth import functional as PF
    out = {}
    for mode in ("eager", "graph", "eager_const_lr"):
        monkeypatch.setenv("PD_STEP_GRAPH", "1" if mode == "graph" else "0")
        PF.DropoutState.manual_seed(11)
        from test_step_gpu import _opts
        from manydepth.trainer import Trainer
        torch.manual_seed(0)
        tr = Trainer(_opts(tmp_path / mode, ["--dropout_rate", "0.1", "--scheduler_step_size", "1"]))
        tr.opt.log_frequency = 10 ** 9
        tr.step = 1
        lrs = []
        for _ in range(2):
            lrs.append(tr.model_optimizer.param_groups[0]["lr"])
            tr.run_epoch()                       # ends with model_lr_scheduler.step()
            if mode == "eager_const_lr":
                tr.model_optimizer.param_groups[0]["lr"] = lrs[0]
        torch.cuda.synchronize()
        if mode != "eager_const_lr":
            assert abs(lrs[1] - 0.1 * lrs[0]) < 1e-12 * lrs[0], lrs
        out[mode] = (tr.store.flat.clone(), tr.model_optimizer.exp_avg.clone())
    assert torch.equal(out["eager"][0], out["graph"][0]) and torch.equal(out["eager"][1], out["graph"][1])
    assert not torch.equal(out["eager"][0], out["eager_const_lr"][0])


def test_two_optimizers_keep_their_own_device_step_words(tmp_path):
    """Adam's device-side t / lr / grad_scale belong to the optimizer, not to the device: a second trainer stepping in
    between does not disturb the first one's bias corrections."""
    from polardepth import synthetic
    tr_a, tr_b = _trainer(tmp_path, "a"), _trainer(tmp_path, "b")
    for tr in (tr_a, tr_b):
        tr.set_train()
        tr.model_optimizer.use_device_step(True)
    assert tr_a.model_optimizer.dev_state.data_ptr() != tr_b.model_optimizer.dev_state.data_ptr()
    b = synthetic.make_batch(2, 64, 96, frame_w=92, device="cuda", seed=3)

    def one(tr):
        tr.model_optimizer.zero_grad()
        _, L, _ = tr.process_batch(dict(b), is_train=True)
        L["loss"].backward()
        tr.model_optimizer.step()

    one(tr_a); one(tr_b); one(tr_b); one(tr_b); one(tr_a)
    torch.cuda.synchronize()
    assert int(tr_a.model_optimizer.dev_state[1]) == 2 and int(tr_b.model_optimizer.dev_state[1]) == 3
    assert tr_a.model_optimizer.step_count == 2 and tr_b.model_optimizer.step_count == 3
