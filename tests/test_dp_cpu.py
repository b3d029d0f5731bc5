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
        # deferred mode (the backward pass runs inside a hipGraph replay: polardepth/graph.py, segmented): hooks that fire
        # while it is set launch nothing; exchange_now() reduces every bucket with the per-bucket calls of the overlapped
        # path; finish() afterwards (FusedAdam.step calls it) must not reduce a second time
        store.zero_grad(); red.reset()
        red.deferred = True
        for i, p in enumerate(store.used_params()):
            p.grad.copy_(torch.full_like(p, float(rank + 1) * (i + 1)))
            p._pd_grad_ready()
        assert sum(red.launched) == 0 and not red.works
        red.finish()                                               # still deferred: a no-op
        assert sum(red.launched) == 0
        red.deferred = False
        red.exchange_now()
        assert all(red.launched) and not red.works
        red.finish()
        for i, p in enumerate(store.used_params()):
            assert torch.all(p.grad == 3.0 * (i + 1)), i
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


def _worker_real_models(rank, world, port, ret):
    """The reducer over the REAL parameter store of the five models (CPU tensors): buckets follow backward completion
    (decoder first), the unused ResNet tail is neither reduced nor counted, the reduced flat gradient equals the sum
    over ranks, and exchange_loss_sums yields the global mask normalisation (trainer.py:1247,1308)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from manydepth import networks
        from polardepth.engine import ParamStore, GradReducer
        from polardepth.functional import exchange_loss_sums
        torch.manual_seed(0)
        models = {"rgb_encoder": networks.ShallowResnetEncoder(18, False), "xolp_encoder": networks.ShallowEncoder('XOLP', 2, 0.1),
                  "normals_encoder": networks.ShallowNormalsEncoder(9, 0.1), "joint_encoder": networks.JointEncoder(0.1, True, True),
                  "mono_depth": networks.DepthDecoder(np.array([64, 64, 128, 256, 512]), range(4))}
        order = ["rgb_encoder", "xolp_encoder", "normals_encoder", "joint_encoder", "mono_depth"]
        unused = lambda m, p: m == "rgb_encoder" and p.split(".")[1] in ("layer3", "layer4", "fc")
        store = ParamStore(models, order=order, unused=unused, device=torch.device("cpu"))
        red = GradReducer(store)                       # 16 MB buckets
        assert 85e6 < store.n_used * 4 < 86e6 and 5 <= len(red.buckets) <= 9      # 85 MB in <= 16 MB buckets cut at parameter boundaries
        first = [n for n, _ in store.entries[:3]]
        assert all(n.startswith("mono_depth.") for n in first)           # decoder gradients complete first
        assert red.buckets[-1][1] == store.n_used
        # backward order: mark parameters ready front to back; early buckets must fire before the last one is complete
        g = torch.Generator().manual_seed(100 + rank)
        store.grad[:store.n_used].copy_(torch.randn(store.n_used, generator=g))
        tail_before = store.grad[store.n_used:].clone()
        local = store.grad[:store.n_used].clone()
        fired = []
        names = [n for n, _ in store.entries[:store.n_used_params]]
        for i, p in enumerate(store.used_params()):
            # producer streams as the trainer has them: decoder / joint / ResNet layers on the main stream, the two shallow
            # encoders on their own streams, every weight gradient on the side stream (tokens stand in for HIP streams)
            own = {"xolp_encoder": "enc0", "normals_encoder": "enc1"}.get(names[i].split(".", 1)[0], "main")
            red.mark_ready(p, streams=[own, "wgrad"])
            fired.append(sum(red.launched))
        assert fired[len(fired) // 3] >= 1 and fired[-1] == len(red.buckets)
        # a bucket waits for the streams of its own parameters only: the decoder buckets never for an encoder stream
        for b in range(len(red.buckets)):
            models_in_b = {n.split(".", 1)[0] for n, p in store.entries[:store.n_used_params] if red.bucket_of[id(p)] == b}
            expect = {"wgrad"} | {{"xolp_encoder": "enc0", "normals_encoder": "enc1"}.get(m, "main") for m in models_in_b}
            assert set(red.deps[b]) == expect, (b, models_in_b, set(red.deps[b]))
        assert set(red.deps[0]) == {"main", "wgrad"}
        red.finish()
        other = torch.randn(store.n_used, generator=torch.Generator().manual_seed(100 + (1 - rank)))
        assert torch.allclose(store.grad[:store.n_used], local + other, rtol=0, atol=1e-6)
        assert torch.equal(store.grad[store.n_used:], tail_before)       # unused ResNet layers: never touched
        # loss scalar exchange: rank r holds (sum|d|m, sum(2-cos)m, sum m, sx, sy) per scale
        S = 4
        sums = torch.tensor([[10.0 * (rank + 1) + s, 3.0 * (rank + 1) + s, 100.0 * (rank + 1) + 7 * s, 5.0 + rank, 6.0 + rank]
                             for s in range(S)], dtype=torch.float64).reshape(-1)
        val, bwd = exchange_loss_sums(sums)
        v, b = val.view(S, 5), bwd.view(S, 5)
        for s in range(S):
            assert v[s, 0].item() == 10.0 * 3 + 2 * s and v[s, 2].item() == 100.0 * 3 + 14 * s     # global sums
            assert v[s, 3].item() == 5.0 + rank and b[s, 3].item() == 5.0 + rank                   # smoothness stays local
            assert b[s, 2].item() == v[s, 2].item() / world                                        # gradient denominator
            # mean over ranks of num_r / (den_global / world) == sum_r num_r / den_global
            num_r = [10.0 * (r + 1) + s for r in range(world)]
            assert abs(sum(n / b[s, 2].item() for n in num_r) / world - sum(num_r) / v[s, 2].item()) < 1e-15
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_reducer_over_the_real_five_model_store_and_loss_scalar_exchange():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker_real_models, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}
