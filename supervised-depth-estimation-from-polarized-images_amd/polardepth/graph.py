"""One training step (zero_grad -> K1 -> encoders -> decoder -> loss -> backward -> Adam) as a hipGraph.

The step launches ~550 kernels from Python; the host needs 12-35 ms to enqueue them (a fraction of the 65 ms of GPU time
on one GPU, but at N ranks the slowest host sets the pace of every all-reduce).  Captured once, the step is replayed with
one call.  What made the step capturable: nothing per-step is a kernel argument any more -- the dropout masks' Philox
offset and Adam's step count live in device memory (functional.DropoutState, pd_step_tick) -- no host synchronisation
inside the step, one-time driver calls (LDS size attributes) hoisted out of the launch path.  Side streams (the two
shallow encoders, the weight-gradient stream) fork from and rejoin the capture stream through events, so the graph keeps
their concurrency.  The eager path stays as it was and produces the same bits (tests/test_graph_gpu.py).

Reference hot loop this replaces: manydepth/trainer.py:430-442 (run_epoch body).
"""
import torch


class GraphedTrainStep:
    """tr: manydepth.trainer.Trainer (single process); example_batch: dict of device tensors with the shapes every later
    batch will have.  ``step(batch)`` copies the batch into the static input buffers, replays the graph and returns the
    (static) loss tensor; ``outputs`` / ``losses`` are the static dictionaries of the captured step."""

    def __init__(self, tr, example_batch, warmup=3, restore_state=False):
        """restore_state: run the warm-up steps on a snapshot (parameters, Adam moments, BatchNorm buffers, step counters are
        put back afterwards), so that building the graph in the middle of a run does not consume training steps."""
        if getattr(tr, "distributed", False):
            raise NotImplementedError("GraphedTrainStep: the RCCL gradient exchange is not captured; use the eager step "
                                      "for multi-process runs")
        self.tr = tr
        self.opt = tr.model_optimizer
        dev = tr.device
        self.static = {k: v.to(dev).clone() for k, v in example_batch.items()}
        self.opt.use_device_step(True)
        tr.set_train()
        snap = None
        if restore_state:
            from . import functional as PF
            bufs = [t for m in tr.models.values() for t in m.buffers()]
            snap = (tr.store.flat.clone(), self.opt.exp_avg.clone(), self.opt.exp_avg_sq.clone(), [t.clone() for t in bufs],
                    PF.DropoutState.state(dev).clone(), self.opt.step_count)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                       # warm-up on a side stream, as stream capture requires
            for _ in range(max(int(warmup), 1)):
                self._eager_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs, self.losses = self._eager_step()
        self.opt.step_count -= 1           # the capture recorded a step without executing it
        if snap is not None:
            from . import functional as PF
            flat, m, v, bvals, st, count = snap
            tr.store.flat.copy_(flat); self.opt.exp_avg.copy_(m); self.opt.exp_avg_sq.copy_(v)
            for t, b in zip([t for mod in tr.models.values() for t in mod.buffers()], bvals):
                t.copy_(b)
            PF.DropoutState.state(dev).copy_(st)
            PF.DropoutState.state(dev)[1] = count
            self.opt.step_count = count
            tr.store.grad.zero_(); tr.store.mark_zeroed()
            tr.store.weights_changed()
        self.loss = self.losses["loss"]
        self.replays = 0

    def _eager_step(self):
        self.opt.zero_grad()
        outputs, losses, _ = self.tr.process_batch(dict(self.static), is_train=True)
        losses["loss"].backward()
        self.opt.step()
        return outputs, losses

    def load(self, batch):
        for k, v in batch.items():
            dst = self.static.get(k)
            if dst is not None and dst.data_ptr() != v.data_ptr():
                dst.copy_(v, non_blocking=True)

    def step(self, batch=None):
        if batch is not None:
            self.load(batch)
        self.graph.replay()
        self.opt.step_count += 1           # the device counter advanced inside the graph; keep the host's view in step
        self.tr.store.weights_changed()
        self.replays += 1
        return self.loss
