"""One training step (zero_grad -> K1 -> encoders -> decoder -> loss -> backward -> Adam) as a hipGraph.

The step launches ~520 kernels from Python; the host needs 12-35 ms to enqueue them (a fraction of the 53 ms of GPU time
on one GPU, but at N ranks the slowest host sets the pace of every all-reduce).  Captured once, the step is replayed with
one call.  What made the step capturable: nothing per-step is a kernel argument any more -- the dropout masks' Philox
offset, Adam's step count, the learning rate and the gradient scale live in device memory (functional.DropoutState,
FusedAdam.dev_state: pd_step_tick / pd_step_set_hyper) -- no host synchronisation inside the step, one-time driver calls
(LDS size attributes) hoisted out of the launch path.  Side streams (the two shallow encoders, the weight-gradient stream)
fork from and rejoin the capture stream through events, so the graph keeps their concurrency.  The eager path stays as it
was and produces the same bits (tests/test_graph_gpu.py).

Data-parallel runs (one process per GPU, engine.GradReducer): two ways to combine the replay with the RCCL exchange --
  * "segmented" (default): the graph holds zero_grad .. backward; behind the replay the reducer all-reduces every bucket
    (the same per-bucket calls as the overlapped eager path, hence the same bits) and the Adam kernel is launched eagerly:
    one replay + ~8 collectives + 2 launches per step on the host instead of ~520.  What it gives up is the overlap of the
    exchange with backward (85 MB over xGMI: ~1 ms of a 53 ms step);
  * "capture" (PD_GRAPH_COMM=capture + PD_GRAPH_COMM_CAPTURE_UNSAFE=1): the all-reduces are captured with everything else
    (torch's ProcessGroupNCCL records them as graph nodes), overlap included.  NOT usable on this PyTorch / ROCm build: the
    Work object of a collective issued during capture still enters the process group's watchdog list, and when the watchdog
    thread polls its end event -- recorded in the capturing stream -- HIP answers hipErrorCapturedEvent, the thread throws and
    the rank aborts (seen as an intermittent SIGABRT of the world-1 test, round 4).  The mode is refused unless the second
    variable says otherwise; bit-equality with eager DP steps was verified on the runs the watchdog did not hit.
Reference hot loop this replaces: manydepth/trainer.py:430-442 (run_epoch body).
"""
import os

import torch


class GraphedTrainStep:
    """tr: manydepth.trainer.Trainer; example_batch: dict of device tensors with the shapes every later batch will have.
    ``step(batch)`` copies the batch into the static input buffers, replays the graph (and, data-parallel, exchanges the
    gradients and steps the optimizer) and returns the (static) loss tensor; ``outputs`` / ``losses`` are the static
    dictionaries of the captured step."""

    def __init__(self, tr, example_batch, warmup=3, restore_state=False, comm=None):
        """restore_state: run the warm-up steps on a snapshot (parameters, Adam moments, BatchNorm buffers, step counters are
        put back afterwards), so that building the graph in the middle of a run does not consume training steps.
        comm: None (PD_GRAPH_COMM or "segmented") | "segmented" | "capture" -- only read when tr runs data-parallel."""
        from . import functional as PF
        self.tr = tr
        self.opt = tr.model_optimizer
        red = getattr(tr, "reducer", None)
        self.dp = bool(getattr(tr, "distributed", False)) and red is not None and red.active
        self.comm = (comm or os.environ.get("PD_GRAPH_COMM") or "segmented") if self.dp else None
        if self.comm not in (None, "segmented", "capture"):
            raise ValueError(f"GraphedTrainStep: comm must be 'segmented' or 'capture', got {self.comm!r}")
        if self.comm == "capture" and os.environ.get("PD_GRAPH_COMM_CAPTURE_UNSAFE") != "1":
            raise NotImplementedError(
                "GraphedTrainStep(comm='capture'): collectives captured into the graph leave Work objects whose events the process "
                "group's watchdog thread polls (hipErrorCapturedEvent -> the rank aborts) on this PyTorch/ROCm build; use "
                "comm='segmented' (default), or set PD_GRAPH_COMM_CAPTURE_UNSAFE=1 to try it anyway")
        self.segmented = self.comm == "segmented"
        if self.segmented and getattr(getattr(tr, "loss_cfg", None), "global_norm", False):
            raise NotImplementedError("GraphedTrainStep (segmented): the global loss normalisation all-reduces in the middle of "
                                      "the forward pass; use PD_GRAPH_COMM=capture or the eager step")
        dev = tr.device
        self.static = {k: v.to(dev).clone() for k, v in example_batch.items()}
        self.opt.use_device_step(not self.segmented)        # segmented: Adam is launched eagerly, with host arguments
        tr.set_train()
        snap = None
        if restore_state:
            bufs = [t for m in tr.models.values() for t in m.buffers()]
            snap = (tr.store.flat.clone(), self.opt.exp_avg.clone(), self.opt.exp_avg_sq.clone(), [t.clone() for t in bufs],
                    PF.DropoutState.state(dev).clone(), self.opt.step_count)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                       # warm-up on a side stream, as stream capture requires
            for _ in range(max(int(warmup), 1)):
                self._warm_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        if self.segmented:
            red.deferred = True
        try:
            # capture_error_mode: under torch.distributed the process group's watchdog / heartbeat threads query events
            # while this thread captures; in the default "global" mode such a query invalidates the capture or throws on
            # THEIR thread (std::terminate: an intermittent SIGABRT of the rank).  Only this thread's calls are policed.
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.outputs, self.losses = self._fwd_bwd()
                if self.segmented:
                    PF.sync_wgrad_stream()                  # every forked stream rejoins the capture stream
                else:
                    self.opt.step()
        except Exception:
            if self.segmented:
                red.deferred = False
            raise
        if self.segmented:
            red.deferred = False           # (no Python runs during a replay; eager steps on this trainer keep their reducer)
        else:
            self.opt.step_count -= 1       # the capture recorded an optimizer step without executing it
        if snap is not None:
            flat, m, v, bvals, st, count = snap
            tr.store.flat.copy_(flat); self.opt.exp_avg.copy_(m); self.opt.exp_avg_sq.copy_(v)
            for t, b in zip([t for mod in tr.models.values() for t in mod.buffers()], bvals):
                t.copy_(b)
            PF.DropoutState.state(dev).copy_(st)
            if self.opt.dev_state is not None:
                self.opt.dev_state[1] = count
            self.opt.step_count = count
            tr.store.grad.zero_(); tr.store.mark_zeroed()
            tr.store.weights_changed()
        self.loss = self.losses["loss"]
        self.replays = 0

    def _fwd_bwd(self):
        self.opt.zero_grad()
        outputs, losses, _ = self.tr.process_batch(dict(self.static), is_train=True)
        losses["loss"].backward()
        return outputs, losses

    def _warm_step(self):
        """One eager step with the semantics the replayed step will have (segmented: deferred exchange)."""
        red = getattr(self.tr, "reducer", None)
        if self.segmented:
            red.deferred = True
        try:
            out = self._fwd_bwd()
        finally:
            if self.segmented:
                red.deferred = False
        if self.segmented:
            red.exchange_now()
        self.opt.step()
        return out

    def load(self, batch):
        for k, v in batch.items():
            dst = self.static.get(k)
            if dst is not None and dst.data_ptr() != v.data_ptr():
                dst.copy_(v, non_blocking=True)

    def step(self, batch=None):
        if batch is not None:
            self.load(batch)
        if self.segmented:
            self.graph.replay()                     # zero_grad .. backward; the reducer saw nothing (deferred at capture)
            self.tr.store.grad_is_zero = False      # (the replay wrote the gradients behind the store's back)
            self.tr.reducer.exchange_now()          # RCCL all-reduce of every bucket behind the replay
            self.opt.step()                         # eager Adam: host-side lr / t / grad_scale, clears the gradient
        else:
            self.opt.sync_hyper()                   # lr (StepLR, trainer.py:467) / grad_scale changed since the last replay?
            self.graph.replay()
            self.opt.step_count += 1                # the device counter advanced inside the graph; keep the host's view in step
            self.tr.store.weights_changed()
        self.replays += 1
        return self.loss
