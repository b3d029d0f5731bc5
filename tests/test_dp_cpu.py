"""Data-parallel gradient reducer on CPU: world_size 2, gloo backend (the GPU path uses the same
code with backend "nccl" = RCCL and a side stream).  Checks bucketing, overlap bookkeeping
(buckets fire when their last gradient is marked ready), the final sum, and that parameters
without a gradient do not dead-lock the step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from polardepth.engine import ParamStore, GradReducer
        torch.manual_seed(0)
        models = {"a": nn.Sequential(nn.Linear(64, 64), nn.Linear(64, 32)), "b": nn.Sequential(nn.Conv2d(8, 16, 3))}
        store = ParamStore(models, order=["a", "b"], device=torch.device("cpu"))
        red = GradReducer(store, bucket_bytes=8 * 1024)
        assert len(red.buckets) >= 3
        assert red.buckets[0][0] == 0 and red.buckets[-1][1] == store.n_used
        for (s0, e0), (s1, e1) in zip(red.buckets[:-1], red.buckets[1:]):
            assert e0 == s1
        for step in range(2):
            store.zero_grad(); red.reset()
            params = store.used_params()
            skip = params[1]                                  # one parameter never gets a gradient this step
            launched_before_finish = 0
            for i, p in enumerate(params):
                if p is skip:
                    continue
                p.grad.copy_(torch.full_like(p, float(rank + 1) * (i + 1)))
                p._pd_grad_ready()
                p._pd_grad_ready()                            # idempotent
            launched_before_finish = sum(red.launched)
            assert 0 < launched_before_finish < len(red.buckets)   # overlapped buckets, one held back by `skip`
            red.finish()
            for i, p in enumerate(params):
                exp = 0.0 if p is skip else 3.0 * (i + 1)          # (1 + 2) * (i + 1)
                assert torch.all(p.grad == exp), (step, i)
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_grad_reducer_gloo_world2():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_reducer_is_noop_single_process():
    from polardepth.engine import ParamStore, GradReducer
    models = {"a": nn.Linear(8, 8)}
    store = ParamStore(models, device=torch.device("cpu"))
    red = GradReducer(store)
    assert red.world == 1
    for p in store.used_params():
        p._pd_grad_ready()
    red.finish()
