"""TB/s of the chain kernels (BN-apply + ReLU + pool + dropout: forward, backward reduce, backward apply) on the
step's big layers, HIP events, buffers rotated beyond the Infinity Cache."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd"))
from polardepth._lib import lib, check, ptr, stream_ptr  # noqa: E402


def run(N, H, W, C, pool, drop=0.1, sets=3, iters=12):
    dev = "cuda"
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    xs = [torch.randn(N, H, W, C, device=dev) for _ in range(sets)]
    outs = [torch.empty(N, Ho, Wo, C, device=dev) for _ in range(sets)]
    dys = [torch.randn(N, Ho, Wo, C, device=dev) for _ in range(sets)]
    dxs = [torch.empty(N, H, W, C, device=dev) for _ in range(sets)]
    scale, shift, mean, invstd = (torch.rand(C, device=dev) + 0.5 for _ in range(4))
    coef = torch.rand(2 * C, device=dev) * 0.01
    rows = lib.pd_chain_bwd_rows(N, H, W, C)
    part = torch.empty(rows, C, 2, device=dev)
    st = stream_ptr()
    res = {}

    def timeit(fn):
        for i in range(sets):
            fn(i)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for i, (a, b) in enumerate(ev):
            a.record(); fn(i % sets); b.record()
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in ev)[iters // 2]
    nin, nout = N * H * W * C * 4, N * Ho * Wo * C * 4
    t = timeit(lambda i: check(lib.pd_chain_fwd(ptr(xs[i]), ptr(scale), ptr(shift), None, ptr(outs[i]), N, H, W, C, 0, C, 1,
                                                int(pool), drop, 1, 2, None, 0, st), "fwd"))
    res["fwd"] = (t, (nin + nout) / t / 1e6)
    t = timeit(lambda i: check(lib.pd_chain_bwd_reduce(ptr(dys[i]), C, ptr(xs[i]), ptr(outs[i]), C, ptr(scale), ptr(shift),
                                                       ptr(mean), ptr(invstd), ptr(part), N, H, W, C, 1, int(pool), drop, 1, 2,
                                                       None, 0, st), "reduce"))
    res["bwd_reduce"] = (t, (nin + nout) / t / 1e6)
    t = timeit(lambda i: check(lib.pd_chain_bwd_apply(ptr(dys[i]), C, ptr(xs[i]), ptr(outs[i]), C, ptr(scale), ptr(shift),
                                                      ptr(mean), ptr(invstd), ptr(coef), ptr(dxs[i]), None, N, H, W, C, 1,
                                                      int(pool), drop, 1, 2, None, 0, st), "apply"))
    res["bwd_apply"] = (t, (2 * nin + nout) / t / 1e6)
    print(json.dumps({"shape": [N, H, W, C], "pool": pool,
                      **{k: {"ms": round(v[0], 4), "GBps": round(v[1])} for k, v in res.items()}}), flush=True)


if __name__ == "__main__":
    run(16, 256, 320, 64, False)
    run(16, 256, 320, 64, True)
    run(16, 128, 160, 64, False)
    run(16, 64, 80, 128, False)
