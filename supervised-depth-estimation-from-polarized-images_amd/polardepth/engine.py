"""Parameter store, fused Adam and the data-parallel gradient reducer.

* ``ParamStore`` flattens every parameter of the model dict into one fp32 buffer (and one gradient
  buffer of the same layout), laid out in *backward completion order* so that contiguous buckets
  become ready progressively.  Parameters become views (channels_last strides are kept), so
  ``state_dict`` / ``load_state_dict`` / checkpoints work unchanged.
* ``FusedAdam`` = torch.optim.Adam semantics (trainer.py:238) as one kernel over the flat buffer.
* ``GradReducer`` = one process per GPU; bucketed RCCL all-reduce of the flat gradient over xGMI on a
  side stream, launched as soon as the kernels have written the last gradient of a bucket, overlapped
  with the rest of backward; the 1/world_size scaling is folded into the Adam kernel.
  The reference has no distributed code: this component is new.
"""
import os

import torch
import torch.distributed as dist

from ._lib import lib, check, ptr, stream_ptr


class ParamStore:
    def __init__(self, models, order=None, unused=lambda model_name, param_name: False, device=None):
        names = list(order) if order is not None else list(models.keys())
        fwd, tail = [], []
        for mn in names:
            for pn, p in models[mn].named_parameters():
                (tail if unused(mn, pn) else fwd).append((f"{mn}.{pn}", p))
        used = list(reversed(fwd))             # backward completion order
        self.entries = used + tail
        self.n_used_params = len(used)
        device = device or self.entries[0][1].device
        align = 4                              # keep every parameter 16-byte aligned
        offs, total = [], 0
        for _, p in self.entries:
            offs.append(total)
            total += (p.numel() + align - 1) // align * align
        self.models = models
        self.numel = total
        self.n_used = offs[self.n_used_params] if self.n_used_params < len(offs) else total
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=device)
        # FusedAdam.step clears the gradient in the pass that consumes it; zero_grad() then has nothing to do unless
        # something wrote since: our kernels (functional.grad_buf resets the flag) or a torch in-place op on a
        # .grad view (bumps the version counter of the shared storage)
        self.grad_is_zero = True
        self._zero_version = self.grad._version
        self.offsets = {}
        for (name, p), off in zip(self.entries, offs):
            self.offsets[name] = (off, p.numel())
            self._adopt(p, off)
        # Data-gradient operands: every 4-D weight transposed ([Co][T][Ci] -> [Ci][T][Co]) into flat_t at the same
        # offset by ONE launch per backward pass (the first request after a forward convolution, ops.FWD_EPOCH) --
        # instead of one launch per layer and step.
        self.flat_t = None
        self._wt_epoch = -1
        self._wt_index, rows, blk = {}, [], [0]
        for (name, p), off in zip(self.entries, offs):
            if p.dim() == 4 and p.numel() < 2 ** 31:
                co, ci, kh, kw = p.shape
                self._wt_index[self.flat.data_ptr() + 4 * off] = (off, co, ci, kh, kw)
                rows.append([off, co, kh * kw, ci])
                blk.append(blk[-1] + kh * kw * ((co + 31) // 32) * ((ci + 31) // 32))   # one workgroup per 32x32 tile and tap
        if rows:
            self._wt_table = torch.tensor(rows, dtype=torch.int32, device=device)
            self._wt_blk = torch.tensor(blk, dtype=torch.int32, device=device)
            self._wt_n, self._wt_blocks = len(rows), blk[-1]
            from . import ops as _ops
            _ops.WT_PROVIDERS[:] = [pr for pr in _ops.WT_PROVIDERS if pr.flat.data_ptr() != self.flat.data_ptr()]
            _ops.WT_PROVIDERS.append(self)

    def weights_changed(self):
        """Drop the transposed copies (FusedAdam.step calls it; a forward convolution has the same effect)."""
        self._wt_epoch = -1

    def transposed(self, w):
        e = self._wt_index.get(w.data_ptr())
        if e is None or tuple(w.shape) != (e[1], e[2], e[3], e[4]) or not w.is_cuda:
            return None
        off, co, ci, kh, kw = e
        from . import ops as _ops
        if self._wt_epoch != _ops.FWD_EPOCH:
            if self.flat_t is None:
                self.flat_t = torch.empty_like(self.flat)
            from ._lib import lib, check, ptr, stream_ptr
            check(lib.pd_weight_transpose_batched(ptr(self.flat), ptr(self.flat_t), ptr(self._wt_table), ptr(self._wt_blk),
                                                  self._wt_n, self._wt_blocks, stream_ptr()), "pd_weight_transpose_batched")
            self._wt_epoch = _ops.FWD_EPOCH
            self._wt_event = torch.cuda.Event()
            self._wt_event.record()
        else:
            torch.cuda.current_stream().wait_event(self._wt_event)   # (backward nodes of other streams ask too)
        return self.flat_t[off:off + co * ci * kh * kw].view(ci, kh, kw, co).permute(0, 3, 1, 2)

    def _view(self, buf, p, off):
        flat = buf[off:off + p.numel()]
        if p.dim() == 4 and p.data.is_contiguous(memory_format=torch.channels_last) and not p.data.is_contiguous():
            co, ci, kh, kw = p.shape
            return flat.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return flat.view(p.shape)

    def _adopt(self, p, off):
        src = p.data
        if p.dim() == 4:
            src = src.contiguous(memory_format=torch.channels_last)
            pv = self.flat[off:off + p.numel()].view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2)
            gv = self.grad[off:off + p.numel()].view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2)
        else:
            pv = self.flat[off:off + p.numel()].view(p.shape)
            gv = self.grad[off:off + p.numel()].view(p.shape)
        pv.copy_(src.to(self.flat.device))
        p.data = pv
        p.grad = gv
        p._pd_store = self            # functional.grad_buf marks the gradient buffer as written

    def zero_grad(self):
        if not (self.grad_is_zero and self.grad._version == self._zero_version):
            self.grad.zero_()
        self.mark_zeroed()

    def mark_zeroed(self):
        self.grad_is_zero = True
        self._zero_version = self.grad._version

    def used_params(self):
        return [p for _, p in self.entries[:self.n_used_params]]


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr) semantics over a ParamStore, one kernel per step."""

    def __init__(self, store, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, reducer=None,
                 zero_grad_in_step=True):
        self.store = store
        self.reducer = reducer
        # step() clears the used gradient range in the pass that consumed it, so that the next zero_grad() is free
        # (unlike torch.optim.Adam, p.grad reads zero after step(); PD_ADAM_KEEP_GRAD=1 keeps torch's behaviour)
        self.zero_grad_in_step = zero_grad_in_step and os.environ.get("PD_ADAM_KEEP_GRAD") != "1"
        params = [p for _, p in store.entries]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.exp_avg = torch.zeros_like(store.flat)
        self.exp_avg_sq = torch.zeros_like(store.flat)
        self.step_count = 0
        self.grad_scale = 1.0
        self.device_step = False     # graph mode: t, lr and grad_scale live in THIS optimizer's device words (dev_state)
        self.dev_state = None        # int64[4]: [1] Adam's t, [2] lr | grad_scale << 32 (fp32 bit patterns); pd_step_set_hyper
        self._dev_hyper = None       # (lr, grad_scale) last written to dev_state[2]

    def use_device_step(self, on=True):
        """Keep Adam's step count, learning rate and gradient scale in device memory so that step() has no per-step kernel
        argument and can be replayed from a hipGraph: t is incremented by pd_step_tick in front of the update, lr and
        grad_scale are rewritten by sync_hyper() whenever the scheduler (trainer.py:467 StepLR) or the reducer changed
        them.  Same bits as the host-side arguments.  The words belong to this optimizer (a second optimizer or trainer on
        the same device has its own); only the dropout stream's step counter is per device (functional.DropoutState)."""
        self.device_step = bool(on)
        if on:
            if self.dev_state is None:
                self.dev_state = torch.zeros(4, dtype=torch.int64, device=self.store.flat.device)
            self.dev_state[1] = self.step_count
            self._dev_hyper = None
            self.sync_hyper()

    def sync_hyper(self):
        """Write lr / grad_scale to the device words if they changed since the last write (one tiny launch, outside any
        capture: GraphedTrainStep.step calls it in front of every replay)."""
        if not self.device_step:
            return
        cur = (float(self.param_groups[0]["lr"]), float(self.grad_scale))
        if cur != self._dev_hyper:
            check(lib.pd_step_set_hyper(ptr(self.dev_state), cur[0], cur[1], stream_ptr()), "pd_step_set_hyper")
            self._dev_hyper = cur

    def zero_grad(self, set_to_none=False):
        from . import functional as PF
        PF.sync_wgrad_stream()
        PF.reset_backward_state()
        self.store.zero_grad()
        if self.store.flat.is_cuda:
            PF.DropoutState.begin_step(self.store.flat.device)       # a training step begins: fresh dropout masks
        if self.reducer is not None:
            self.reducer.reset()

    @torch.no_grad()
    def step(self, closure=None):
        from . import functional as PF
        PF.sync_wgrad_stream()                   # weight gradients are produced on a side stream
        if self.reducer is not None:
            self.reducer.finish()
        g = self.param_groups[0]
        self.step_count += 1
        n = self.store.n_used
        b1, b2 = g["betas"]
        state = None
        if self.device_step:
            state = self.dev_state
            if not torch.cuda.is_current_stream_capturing():
                self.sync_hyper()              # (a capture must not bake the write in: it would pin lr to today's value)
            check(lib.pd_step_tick(ptr(state), 0, 1, stream_ptr()), "pd_step_tick")
        check(lib.pd_adam_step(ptr(self.store.flat), ptr(self.store.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), n,
                               float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]),
                               self.step_count, ptr(state), float(self.grad_scale), int(self.zero_grad_in_step), stream_ptr()),
              "pd_adam_step")
        self.store.weights_changed()           # (raw-pointer write: the transposed copies are stale)
        if self.zero_grad_in_step:
            # (the tail behind n_used -- parameters that never receive gradients, e.g. ResNet layer3/4/fc -- is never read)
            self.store.mark_zeroed()

    # ---- checkpoint interop with torch.optim.Adam ("adam.pth", trainer.py:1614-1617, 1681-1691)
    # torch numbers the parameters in the order they were handed to the optimizer: the reference's
    # ``parameters_to_train`` = normals_encoder, xolp_encoder, joint_encoder, rgb_encoder, mono_depth, each in
    # ``model.parameters()`` order (trainer.py:194-219) -- NOT the backward-completion order of the flat buffer.
    REFERENCE_MODEL_ORDER = ("normals_encoder", "xolp_encoder", "joint_encoder", "rgb_encoder", "mono_depth")

    def reference_order(self):
        """Parameter names in the reference optimizer's index order (models missing from this run are skipped)."""
        by_model = {}
        for name, _ in self.store.entries:
            by_model.setdefault(name.split(".", 1)[0], []).append(name)
        names = []
        for m in list(self.REFERENCE_MODEL_ORDER) + [k for k in by_model if k not in self.REFERENCE_MODEL_ORDER]:
            if m in by_model:
                if m in getattr(self.store, "models", {}):      # model.parameters() order
                    names += [f"{m}.{pn}" for pn, _ in self.store.models[m].named_parameters()]
                else:
                    names += by_model[m]
        return names

    def state_dict(self):
        names = self.reference_order()
        entries = dict(self.store.entries)
        used = {n for n, _ in self.store.entries[:self.store.n_used_params]}
        state = {}
        for i, name in enumerate(names):
            if name in used and self.step_count > 0:
                off, n = self.store.offsets[name]
                p = entries[name]
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.store._view(self.exp_avg, p, off).clone(),
                            "exp_avg_sq": self.store._view(self.exp_avg_sq, p, off).clone()}
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        g["params"] = list(range(len(names)))
        return {"state": state, "param_groups": [g], "pd_order": names}

    def load_state_dict(self, sd):
        """Accepts this build's files and a plain ``torch.optim.Adam`` state of the reference (no ``pd_order``: its
        indices follow ``parameters_to_train``).  Shapes are validated before anything is copied; a mismatch raises
        ValueError, which Trainer.load_model answers like the reference does ("Can't load Adam - using random")."""
        g = sd["param_groups"][0]
        names = sd.get("pd_order") or self.reference_order()
        if len(g.get("params", names)) != len(names):
            raise ValueError(f"adam.pth holds {len(g['params'])} parameters, this model has {len(names)}")
        entries = dict(self.store.entries)
        todo, steps = [], []
        for i, st in sd["state"].items():
            name = names[int(i)]
            if name not in self.store.offsets:
                raise ValueError(f"adam.pth: unknown parameter {name!r}")
            p = entries[name]
            for k in ("exp_avg", "exp_avg_sq"):
                if tuple(st[k].shape) != tuple(p.shape):
                    raise ValueError(f"adam.pth: {k} of parameter {int(i)} has shape {tuple(st[k].shape)}, "
                                     f"{name} needs {tuple(p.shape)} (parameter order mismatch?)")
            todo.append((name, p, st))
            steps.append(int(float(st["step"])))
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in g:
                self.param_groups[0][k] = g[k]
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        for name, p, st in todo:
            off, n = self.store.offsets[name]
            self.store._view(self.exp_avg, p, off).copy_(st["exp_avg"])
            self.store._view(self.exp_avg_sq, p, off).copy_(st["exp_avg_sq"])
        self.step_count = max(steps) if steps else 0


class GradReducer:
    """Bucketed, overlapped all-reduce (sum) of ``store.grad[:n_used]``; scale is applied by Adam."""

    def __init__(self, store, bucket_bytes=16 << 20, process_group=None):
        self.store = store
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # PD_DIST_TEST=1 exercises the full bucket / side-stream / RCCL path even with a single rank
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("PD_DIST_TEST") == "1")
        self.buckets = []            # (start, end)
        self.bucket_of = {}
        cap = max(bucket_bytes // 4, 1)
        start, cur = 0, 0
        idx = 0
        for name, p in store.entries[:store.n_used_params]:
            off, n = store.offsets[name]
            if off + n - start > cap and off > start:
                self.buckets.append((start, off)); start = off; idx += 1
            self.bucket_of[id(p)] = idx
            cur = off + n
            p._pd_grad_ready = (lambda pp=p: self.mark_ready(pp))
        self.buckets.append((start, store.n_used))
        self.counts = [0] * len(self.buckets)
        for name, p in store.entries[:store.n_used_params]:
            self.counts[self.bucket_of[id(p)]] += 1
        self.cuda = store.grad.is_cuda
        self.comm_stream = torch.cuda.Stream(device=store.grad.device) if self.cuda and self.active else None
        self._events = {}            # (bucket, stream id) -> reusable event: the stream's position behind the bucket's last kernel
        self._t0 = self._t1 = None   # events around finish(): the part of the exchange the step actually waits for
        # deferred: the backward pass runs inside a hipGraph replay (polardepth/graph.py, segmented mode) -- no Python runs
        # while the gradients are produced, so nothing is launched from mark_ready(); exchange_now() reduces every bucket
        # behind the replay, with the same per-bucket calls (hence the same bits) as the overlapped path
        self.deferred = False
        self.reset()

    def reset(self):
        self.pending = list(self.counts)
        self.seen = set()
        self.launched = [False] * len(self.buckets)
        self.works = []
        self.deps = [dict() for _ in self.buckets]      # per bucket: stream id -> event recorded behind its last producer

    def _producer_streams(self):
        """Streams that hold kernels writing the gradient of the parameter being marked: the stream the backward node runs
        on (BatchNorm / bias / head gradients; autograd replays a node on its forward stream, i.e. an encoder stream for an
        encoder layer) and the weight-gradient side stream."""
        from . import functional as PF
        streams = [torch.cuda.current_stream()]
        if PF.USE_WGRAD_STREAM:
            side = PF._WGRAD_STREAMS.get(self.store.grad.device.index)
            if side is not None:
                streams.append(side)
        return streams

    def mark_ready(self, p, streams=None):
        """Called by the backward node that produced the last gradient kernel of p (after enqueueing it).  Records, per
        producer stream, an event behind that kernel; the bucket's all-reduce waits for exactly these events -- not for
        whole streams: an encoder stream that is still busy with layers of another bucket does not hold back a decoder
        bucket whose kernels have finished."""
        if not self.active or self.deferred or id(p) in self.seen:
            return
        self.seen.add(id(p))
        b = self.bucket_of[id(p)]
        if self.comm_stream is not None or streams is not None:
            for st in (streams if streams is not None else self._producer_streams()):
                key = (b, getattr(st, "cuda_stream", st))
                ev = self._events.get(key)
                if ev is None and self.comm_stream is not None:
                    ev = self._events[key] = torch.cuda.Event()
                if ev is not None:
                    ev.record(st)
                self.deps[b][key[1]] = ev           # a later record on the same stream supersedes the earlier one
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        s, e = self.buckets[b]
        view = self.store.grad[s:e]
        if self.comm_stream is not None:
            if self.pending[b] == 0:
                for ev in self.deps[b].values():
                    self.comm_stream.wait_event(ev)
            else:
                # finish() flushes a bucket with a parameter that received no gradient this step: nothing recorded for
                # it, wait for everything enqueued so far
                from . import functional as PF
                self.comm_stream.wait_stream(torch.cuda.current_stream())
                PF.sync_wgrad_stream(self.comm_stream)
            with torch.cuda.stream(self.comm_stream):
                self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def exchange_now(self):
        """All buckets, now: the current stream holds every gradient (a graph replay of forward + backward has been
        enqueued on it); the all-reduces run on the comm stream behind it, the current stream waits for them."""
        if not self.active:
            return
        self.reset()
        was, self.deferred = self.deferred, False
        try:
            self.finish()                        # every bucket has pending > 0: _launch waits for the current stream
        finally:
            self.deferred = was

    def finish(self):
        if not self.active or self.deferred:
            return
        for b in range(len(self.buckets)):       # buckets holding a parameter that got no gradient this step
            self._launch(b)
        timed = self.comm_stream is not None and not torch.cuda.is_current_stream_capturing()
        if timed:
            if self._t0 is None:
                self._t0, self._t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self._t0.record()
        for w in self.works:
            w.wait()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if timed:
            self._t1.record()
        self.works = []

    def exposed_wait_ms(self):
        """GPU time the last step's optimizer stream spent waiting for the exchange (after a synchronize): the part of the
        all-reduce that backward did not hide."""
        if self._t0 is None:
            return 0.0
        return float(self._t0.elapsed_time(self._t1))
