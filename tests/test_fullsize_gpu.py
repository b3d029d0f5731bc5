"""The network at BASELINE.json's full size (configs[2]: batch 16, 512x612 frames -> 512x640 grid, three encoders,
four scales) through size-independent properties -- the oracle cannot finish this size in seconds:

* idempotence: the same forward/backward pass twice gives bit-identical gradients (every reduction is ordered);
* route equivalence: the fused routes of the decoder / loss (ActGrad hand-over, one-pass disparity-head gradient,
  ground-truth normal cache, parity-split stride-2 data gradient, 16-channel halo kernels, skip gradients in the
  data-gradient epilogue, reflect border strips) against the plain routes they replace, which the small-size tests
  pin to the oracle: same loss to 1e-6, same gradient to 1e-5 of its L2 norm."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_full_size_step_is_reproducible_and_route_independent(tmp_path, monkeypatch):
    import bench
    from polardepth import functional as PF
    from polardepth import ops, synthetic

    B = 16
    tr = bench.build_trainer(B, bench.H, bench.W, str(tmp_path))
    batch = synthetic.make_batch(B, bench.H, bench.W, frame_w=bench.FRAME_W, device="cuda")
    store = tr.model_optimizer.store
    for m in tr.models.values():
        m.train()

    def grads():
        PF.DropoutState.manual_seed(77)
        store.grad.zero_()
        store.mark_zeroed()
        _, losses, _ = tr.process_batch({k: v.clone() for k, v in batch.items()}, is_train=True)
        losses["loss"].backward()
        PF.sync_wgrad_stream()
        torch.cuda.synchronize()
        return store.grad.clone(), float(losses["loss"].detach())

    g0, l0 = grads()
    assert torch.isfinite(g0).all() and l0 == l0 and 0.0 < l0 < 100.0
    g1, l1 = grads()
    assert l0 == l1
    assert torch.equal(g0, g1), "full-size gradients are not bit-reproducible"

    for name in ("USE_ACT_FUSION", "USE_GT_NORMAL_CACHE", "USE_DISPHEAD_FUSED", "USE_SKIP_FUSION", "USE_REFLECT_BORDER"):
        monkeypatch.setattr(PF, name, False)
    for name in ("USE_S2_PHASES", "USE_CONV16"):
        monkeypatch.setattr(ops, name, False)
    g2, l2 = grads()
    assert abs(l2 - l0) <= 1e-6 * abs(l0), (l0, l2)
    rel = float((g2 - g0).norm() / g0.norm())
    assert rel <= 1e-5, f"plain routes differ from the fused ones by {rel:.3e} of the gradient norm"
