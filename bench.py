#!/usr/bin/env python3
"""Benchmark of the hot path: training images/s of the 3-encoder polarimetric depth network.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = zero_grad + K1 polar preprocessing + 3 encoders + joint encoder + decoder + multi-scale
loss + backward + fused Adam on one synthetic HAMMER-shaped batch that is already resident in HBM.
Workload (BASELINE.json configs[2], the configuration the metric is quoted on): full 3-encoder net
(augment_xolp + augment_normals), scales [0,1,2,3], batch 16 per GPU, 512x612 frames -- the polar
kernel runs on the true 512x612 planes, network and loss on 512x640 (28 zero-padded, mask-invalid
columns; the reference itself cannot run W=612, trainer.py:107-108); images/s counts frames.
Weak scaling: every rank processes its own batch of 16; gradients are all-reduced with RCCL.

The JSON line also carries
  roofline     -- the dominant kernel (implicit-GEMM conv: fp32 MFMA, or fp32 products from six bf16 MFMAs), algorithmic FLOPs / HIP-event time
                  of its launches in instrumented steps of the same workload (run serially on one stream:
                  in the timed steps weight-gradient kernels overlap data-gradient kernels, which would
                  stretch per-kernel durations), vs the 157.3 TFLOP/s fp32 matrix peak -- or, for the bf16x3 kernel, the
                  dense bf16 peak / 6 = 416.7 fp32-equivalent TFLOP/s (MI355X_MICROARCH.md); `kernels` lists the top four;
  precision    -- which products run where, the test that bounds the error, and the same step with PD_CONV_X3=0;
  cpu_baseline -- the CPU oracle (oracle/: plain PyTorch-CPU restatement of the reference) timed on
                  this box's host cores on a bounded sample (rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "supervised-depth-estimation-from-polarized-images_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

# the host driver of this pool only supports dmabuf IPC: RCCL between processes needs this (a no-op when already set)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

H, W, FRAME_W, BATCH = 512, 640, 612, 16
FP32_MFMA_PEAK_TF = 157.3
X3_PEAK_TF = 2500.0 / 6.0          # fp32 products as six bf16 MFMAs (conv_igemm_x3_kernel), in fp32-equivalent FLOPs


def build_trainer(batch, height, width, log_dir):
    from manydepth.options import MonodepthOptions
    from manydepth.trainer import Trainer
    opts = MonodepthOptions().parse([
        "--png", "--batch_size", str(batch), "--height", str(height), "--width", str(width), "--dataset", "HAMMER",
        "--split", "HAMMER", "--eval_split", "HAMMER_unseen", "--min_depth", "0.1", "--max_depth", "2.0",
        "--disparity_smoothness", "1e-3", "--depth_supervision_only", "True", "--depth_supervision", "True",
        "--normals_loss_weight", "0.35", "--augment_xolp", "--augment_normals", "--learning_rate", "1e-4",
        "--weights_init", "scratch", "--num_workers", "0", "--log_dir", log_dir, "--data_path", "synthetic",
        "--data_path_val", "synthetic", "--model_name", "bench"] +
        (["--dropout_rate", os.environ["PD_BENCH_DROPOUT"]] if "PD_BENCH_DROPOUT" in os.environ else []))   # tuning aid only
    torch.manual_seed(0)        # reproducible random-init weights (final_loss is then comparable across runs and builds)
    return Trainer(opts)


def train_step(tr, batch):
    tr.model_optimizer.zero_grad()
    _, losses, _ = tr.process_batch(dict(batch), is_train=True)
    losses["loss"].backward()
    tr.model_optimizer.step()
    return losses["loss"]


def _oracle_step_factory(sample_batch):
    """One oracle train step (fwd + loss + bwd + Adam) of the same 3-encoder 512x640 workload as a closure."""
    import numpy as np
    from oracle import nets as onets, losses as ol, polar as opolar
    torch.manual_seed(0)
    models = onets.build_models(True, True, 0.1)
    params = [p for m in models.values() for p in m.parameters()]
    opt = torch.optim.Adam(params, 1e-4)
    for m in models.values():
        m.train()
    rng = np.random.default_rng(0)
    pol = rng.integers(0, 256, (sample_batch, 4, H, FRAME_W), dtype=np.uint8)
    color = torch.rand(sample_batch, 3, H, W)
    inputs = {("color", 0, 0): color}
    for s in range(1, 4):
        inputs[("color", 0, s)] = torch.nn.functional.avg_pool2d(inputs[("color", 0, s - 1)], 2)
    gt = 0.3 + 1.5 * torch.rand(sample_batch, 1, H, W)
    gt[..., FRAME_W:] = 0
    K = torch.eye(4)[None].repeat(sample_batch, 1, 1)
    K[:, 0, 0] = K[:, 1, 1] = 0.65 * W; K[:, 0, 2] = W / 2; K[:, 1, 2] = H / 2
    inputs["depth"] = gt; inputs[("K", 0)] = K

    def step():
        opt.zero_grad()
        xolp, _, _, _ = opolar.polar_forward(pol)                       # DataLoader-side XOLP of the reference
        xolp = torch.nn.functional.pad(xolp, (0, W - FRAME_W))
        outs = dict(onets.forward_models(models, color, xolp))          # includes the scipy normals bounce
        for s in range(4):
            outs[("depth", 0, s)] = ol.upsample_disp_to_depth(outs[("disp", s)], H, W, 0.1, 2.0)
        L = ol.compute_losses(inputs, outs, normals_loss_weight=0.35)
        L["loss"].backward()
        opt.step()
    return step


def _median_time(fn, warmup, reps):
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def cpu_baseline():
    """CPU baseline per BASELINE.md §2 on this box's host cores, bounded: the oracle train step at the best thread count
    of a sweep {8, 16, 32, 64, all} (warm-up 2, median of 5, batch 2) and on one thread (the reference pins
    OMP/MKL_NUM_THREADS=1, trainer.py:9-11; batch 1, warm-up 1, median of 3), and the per-frame K1 paths of the reference on one 512x612 frame (warm-up 2, median of 5): closed-form
    fp64 XOLP, the literal lstsq XOLP (xolp.py:20) and the SciPy-table normals (normals_vec.py:11-60)."""
    import numpy as np
    from oracle import polar as opolar
    host = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    t_begin = time.perf_counter()

    def note(msg):               # progress on stderr: a silent multi-minute CPU phase reads as a hung run
        print(f"[cpu_baseline +{time.perf_counter() - t_begin:5.1f}s] {msg}", file=sys.stderr, flush=True)

    # thread sweep at batch 1: one probe step per count after one warm-up step (oversubscribed intra-op threading makes
    # "all cores" SLOWER than one thread on the 128-thread hosts of this pool); the sweep stops growing the thread count
    # once a step gets slower or ~60 s are spent.  Then the best count: batch 2, warm-up 2, median of up to 5 steps within
    # ~45 s.  One thread (the reference pins OMP/MKL_NUM_THREADS=1): batch 1, warm-up 1, median of 3.
    probe = _oracle_step_factory(1)
    sweep = {}
    for t in sorted({1, 8, 16, 32, 64, host} & set(range(1, host + 1))):
        torch.set_num_threads(t)
        sweep[t] = _median_time(probe, 1, 1)
        note(f"{t} threads: {sweep[t]:.2f} s / step (batch 1)")
        if time.perf_counter() - t_begin > 60 or (t > 1 and sweep[t] > 1.5 * min(sweep.values())):
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    step = _oracle_step_factory(2)
    ts = []
    for i in range(7):
        t0 = time.perf_counter()
        step()
        if i >= 2:
            ts.append(time.perf_counter() - t0)
        if i >= 4 and time.perf_counter() - t_begin > 60 + 45:
            break
    t_best = sorted(ts)[len(ts) // 2]
    note(f"best: {best} threads, batch 2: {t_best:.2f} s / step (median of {len(ts)})")
    torch.set_num_threads(1)
    try:
        t_one = _median_time(probe, 1, 3) if 1 not in sweep or sweep[1] < 8 else sweep[1]
    finally:
        torch.set_num_threads(default_threads)
    note(f"1 thread, batch 1: {t_one:.2f} s / step")
    rng = np.random.default_rng(1)
    frame = rng.integers(0, 256, (H, FRAME_W, 4), dtype=np.uint8)
    P = H * FRAME_W
    t_closed = _median_time(lambda: opolar.iun_and_xolp(frame), 2, 5)
    t_lstsq = _median_time(lambda: opolar.iun_and_xolp_lstsq(frame), 1, 3)
    xolp = torch.from_numpy(opolar.xolp_planes(np.ascontiguousarray(np.moveaxis(frame, -1, 0)[None]))[0])
    t_norm = _median_time(lambda: opolar.get_normals(xolp).float(), 2, 5)
    return {"value": round(2 / t_best, 4), "unit": "images/s", "cores": best, "kind": "port",
            "sample": f"oracle/ (PyTorch-CPU + NumPy/SciPy restatement of the reference) train step of the same 3-encoder "
                      f"512x640 workload at batch 2 on the best thread count of a sweep ({best} of {host} host threads): "
                      f"warm-up 2, median of {len(ts)} = {t_best:.2f} s/step",
            "thread_sweep_s_per_step_batch1": {str(k): round(v, 2) for k, v in sweep.items()},
            "host_threads": host,
            "one_thread": {"value": round(1 / t_one, 4), "unit": "images/s", "cores": 1,
                           "sample": f"same step at batch 1, torch.set_num_threads(1) (the reference pins OMP/MKL_NUM_THREADS=1, "
                                     f"trainer.py:9-11): warm-up 1, median of 3 = {t_one:.1f} s/step"},
            "k1_per_frame_1thread": {
                "frame": "512x612 uint8 x 4",
                "xolp_closed_form_fp64_ms": round(t_closed * 1e3, 1), "xolp_closed_form_GBps_12Bpx": round(P * 12 / t_closed / 1e9, 3),
                "xolp_lstsq_ms": round(t_lstsq * 1e3, 1),
                "normals_scipy_ms": round(t_norm * 1e3, 1),
                "xolp_plus_normals_GBps_48Bpx": round(P * 48 / (t_closed + t_norm) / 1e9, 3)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_graph", action="store_true",
                    help="time the eager step only (default on one GPU: the step is also captured into a hipGraph and reported "
                         "beside the eager loop; `value` is always the eager loop)")
    ap.add_argument("--dp_graph", action="store_true",
                    help="with --gpus N > 1: also time the captured step under the gradient reducer (polardepth/graph.py, "
                         "segmented mode).  Off by default at N > 1: the scaling run's line must not depend on a capture "
                         "beside a live process group, which only world-1 tests have exercised on this pool")
    ap.add_argument("--attention", action="store_true",
                    help="BASELINE configs[4] variant (single-head attention at the joint-encoder merge, fp32); "
                         "not the headline workload")
    ap.add_argument("--normals_decoder", action="store_true",
                    help="the arch1++_separate_normals_dec variant (README.md:54: a decoder behind the normals encoder predicts "
                         "normals, supervised by the normals of the ground truth); not the headline workload")
    ap.add_argument("--bf16", action="store_true",
                    help="with --attention: the attention block on the bf16 matrix cores (configs[4] as specified: "
                         "v_mfma_f32_32x32x16_bf16, fp32 softmax statistics and accumulators); the JSON line then carries "
                         "an `attention` object graded against the bf16 MFMA peak")
    args = ap.parse_args()
    if args.attention:
        os.environ["PD_JOINT_ATTENTION"] = "1"
    if args.normals_decoder:
        os.environ["PD_NORMALS_DECODER"] = "1"
    if args.bf16:
        assert args.attention, "--bf16 selects the bf16 attention kernels: use it with --attention"
        os.environ["PD_ATTENTION_BF16"] = "1"

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path is HIP-only (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("PD_DIST_TEST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world)         # "nccl" is RCCL on ROCm
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from polardepth import synthetic, ops
    from polardepth import functional as PF
    log_dir = tempfile.mkdtemp(prefix="pd_bench_")
    tr = build_trainer(args.batch, H, W, log_dir)
    tr.set_train()
    PF.DropoutState.manual_seed(1234 + rank)
    batch = synthetic.make_batch(args.batch, H, W, frame_w=FRAME_W, device=f"cuda:{local_rank}", seed=rank)
    batch[("pol", 0, 0)] = batch[("pol", 0, 0)][..., :FRAME_W].contiguous()      # true 512x612 planes for K1
    batch.pop("depth_gt"); batch.pop(("mask", 0, 0))

    def timed_loop(step_fn):
        """W warm-up steps, then exactly K timed steps between barrier + synchronize on both sides; also the host time
        of enqueueing one step into an empty queue."""
        t_host = None
        for i in range(args.warmup):
            if i == args.warmup - 1:
                torch.cuda.synchronize()
                t_h0 = time.perf_counter()
            step_fn()
            if i == args.warmup - 1:
                t_host = time.perf_counter() - t_h0
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step_fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, t_host, loss

    dt, t_host, loss = timed_loop(lambda: train_step(tr, batch))
    eager = {"images_per_s": round(args.batch * args.steps / dt, 3), "ms_per_step": round(dt / args.steps * 1e3, 3),
             "host_enqueue_ms_per_step": None if t_host is None else round(t_host * 1e3, 2), "final_loss": round(float(loss.detach()), 6)}
    # (before the graph section: a second trainer and the graph's memory pool slow this trainer's allocations afterwards)
    # the same eager step with every convolution on the fp32 MFMA (flags PD_CONV_FP32_MFMA of pd_conv2d* / pd_conv2d_wgrad):
    # what the bf16x3 kernels buy, measured in this process on this GPU
    fp32_only = None
    from polardepth import ops as _ops
    if world == 1 and _ops.CONV_FLAGS == _ops.CONV_AUTO and _ops.WGRAD_FLAGS == _ops.CONV_AUTO and not args.attention and not args.normals_decoder:
        with _ops.conv_flags(conv=_ops.CONV_FP32_MFMA, wgrad=_ops.CONV_FP32_MFMA):
            n_ref = min(args.steps, 10)
            for _ in range(2):
                train_step(tr, batch)
            torch.cuda.synchronize()
            t_r0 = time.perf_counter()
            for _ in range(n_ref):
                train_step(tr, batch)
            torch.cuda.synchronize()
            dt_ref = time.perf_counter() - t_r0
            fp32_only = {"images_per_s": round(args.batch * n_ref / dt_ref, 3), "ms_per_step": round(dt_ref / n_ref * 1e3, 3),
                         "steps": n_ref, "launch": "eager"}
    graph_info = None
    if world > 1 and not args.dp_graph and not args.no_graph:
        graph_info = {"skipped": "N > 1: pass --dp_graph to time the captured step under the gradient reducer "
                                 "(tests/test_dp_gpu.py covers it bit for bit on a world-1 RCCL group)"}
    elif not args.no_graph:         # (data-parallel: graph of zero_grad..backward + bucketed all-reduce + eager Adam, polardepth/graph.py)
        # the same step replayed from a hipGraph (polardepth/graph.py).  A second trainer with the same seed, so that both
        # loops run the same number of steps from the same initial weights: final_loss must agree bit for bit.
        try:
            from polardepth.graph import GraphedTrainStep
            PF.DropoutState.manual_seed(1234 + rank)
            tr_g = build_trainer(args.batch, H, W, log_dir)
            tr_g.set_train()
            t_c0 = time.perf_counter()
            gstep = GraphedTrainStep(tr_g, batch, warmup=3)
            torch.cuda.synchronize()
            t_capture = time.perf_counter() - t_c0
            # the eager loop ran warmup + steps steps before its final loss; the graphed trainer has run its 3 eager warm-up
            # steps (the capture itself executes nothing)
            extra = args.warmup - 3
            for _ in range(max(extra, 0)):
                gstep.step()
            saved_warmup, args.warmup = args.warmup, 0
            dt_g, _, loss_g = timed_loop(lambda: gstep.step())
            args.warmup = saved_warmup
            torch.cuda.synchronize()
            loss_g_val = float(loss_g.detach())      # (the loss tensor is a static buffer of the graph: read it before the next replay)
            t_h0 = time.perf_counter()
            gstep.step()
            t_host_g = time.perf_counter() - t_h0
            torch.cuda.synchronize()
            graph_info = {"images_per_s": round(args.batch * args.steps / dt_g, 3), "ms_per_step": round(dt_g / args.steps * 1e3, 3),
                          "host_enqueue_ms_per_step": round(t_host_g * 1e3, 3), "final_loss": round(loss_g_val, 6),
                          "capture_s": round(t_capture, 2),
                          "final_loss_equals_eager": bool(extra >= 0 and loss_g_val == float(loss.detach())),
                          "comm": gstep.comm}
            # `value` stays the EAGER loop (the Trainer's default path); the replay is reported beside it
        except Exception as exc:           # the bench line must not depend on the capture
            graph_info = {"error": f"{type(exc).__name__}: {exc}"[:400]}
    dp_info = None
    if dist.is_initialized():
        # per-rank view for diagnosing a scaling run: every rank's own loop time (before the MAX) and the part of the last
        # step's all-reduce that backward did not hide (GPU time the optimizer stream waited for the comm stream)
        mine = torch.tensor([dt, tr.reducer.exposed_wait_ms() if tr.reducer is not None else 0.0], device="cuda", dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = torch.stack(allr).cpu()
        dp_info = {"step_ms_per_rank_min": round(per_rank[:, 0].min().item() / args.steps * 1e3, 3),
                   "step_ms_per_rank_max": round(per_rank[:, 0].max().item() / args.steps * 1e3, 3),
                   "allreduce_exposed_ms_last_step_per_rank": [round(v, 3) for v in per_rank[:, 1].tolist()],
                   "gradient_MB": round(tr.store.n_used * 4 / 1e6, 1),
                   "buckets": 0 if tr.reducer is None else len(tr.reducer.buckets)}
    if world > 1:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss_val = float(loss)

    # ---- roofline of the dominant kernel: instrumented steps of the same workload (HIP events on the
    # launch stream around every implicit-GEMM conv launch), outside the timed region
    # (weight-gradient kernels normally overlap the data-gradient kernels on a side stream; for per-kernel
    # durations the instrumented steps run them serially on one stream)
    overlap, PF.USE_WGRAD_STREAM = PF.USE_WGRAD_STREAM, False
    enc_streams, tr.encoder_streams = tr.encoder_streams, False      # (likewise: the three encoders on one stream)
    ops.PROFILE = []
    from polardepth import polar as pdpolar
    k1_step_events, k1_real = [], pdpolar.polar_forward

    def k1_timed(*a, **kw):           # K1 inside the step: HIP events around the launch, on the launch stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = k1_real(*a, **kw)
        e1.record()
        k1_step_events.append((e0, e1))
        return out

    pdpolar.polar_forward = k1_timed
    for _ in range(4):
        train_step(tr, batch)
    torch.cuda.synchronize()
    pdpolar.polar_forward = k1_real
    prof, ops.PROFILE = ops.PROFILE, None
    PF.USE_WGRAD_STREAM = overlap
    tr.encoder_streams = enc_streams
    INSTR_STEPS = 4
    k1_in_step_ms = sorted(e0.elapsed_time(e1) for e0, e1 in k1_step_events)[len(k1_step_events) // 2]
    by_kernel = {}
    for name, flops, e0, e1, _shape in prof:
        k = by_kernel.setdefault(name, [0.0, 0.0, 0])
        k[0] += flops; k[1] += e0.elapsed_time(e1) * 1e-3; k[2] += 1
    dom = max(by_kernel.items(), key=lambda kv: kv[1][1])
    dname, (dflops, dtime, dcount) = dom
    achieved = dflops / dtime / 1e12
    step_conv_time = sum(v[1] for v in by_kernel.values()) / INSTR_STEPS
    # HBM traffic per launch of that kernel: rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, gfx950
    # FETCH_SIZE x2 correction) cannot run inside this process -- the committed summary of those passes is quoted
    traffic, traffic_src = None, None
    for pmc_name in ("r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):
        pmc = os.path.join(ROOT, "profiles", pmc_name)
        if os.path.exists(pmc):
            with open(pmc) as f:
                pmc_all = json.load(f)
            dom_pmc = pmc_all.get("kernels", {}).get(dname) or (pmc_all.get("dominant", {}) if pmc_all.get("dominant", {}).get("kernel") == dname else None)
            if dom_pmc:
                traffic, traffic_src = dom_pmc.get("hbm_bytes_per_launch"), "profiles/" + pmc_name
                break
    # peak of a kernel family: the fp32 MFMA peak, or -- for the kernel that forms every fp32 product from six bf16 MFMAs
    # (three-way split, conv_igemm_x3_kernel) -- the dense bf16 MFMA peak / 6: the ceiling of that scheme in
    # fp32-equivalent (algorithmic) FLOPs
    def family_peak(name):
        return X3_PEAK_TF if "x3" in name else FP32_MFMA_PEAK_TF
    dpeak = family_peak(dname)
    kernels = [{"kernel": n, "ms_per_step": round(v[1] / INSTR_STEPS * 1e3, 3), "launches_per_step": v[2] // INSTR_STEPS,
                "achieved": round(v[0] / v[1] / 1e12, 2), "peak": round(family_peak(n), 1),
                "frac": round(v[0] / v[1] / 1e12 / family_peak(n), 4)}
               for n, v in sorted(by_kernel.items(), key=lambda kv: -kv[1][1])[:4]]
    roofline = {"bound": "mfma", "kernel": dname, "achieved": round(achieved, 2), "peak": round(dpeak, 1),
                "unit": "TFLOP/s", "frac": round(achieved / dpeak, 4), "traffic": traffic,
                "peak_note": ("dense bf16 MFMA peak 2500 / 6 products per fp32 product (fp32-equivalent FLOPs)"
                              if "x3" in dname else "fp32 MFMA peak"),
                "kernels": kernels,
                "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": traffic_src,
                "algorithmic_flops_per_launch": round(dflops / dcount),
                "launches_per_step": dcount // INSTR_STEPS, "avg_launch_ms": round(dtime / dcount * 1e3, 4),
                "all_conv_kernels_ms_per_step": round(step_conv_time * 1e3, 2),
                "all_conv_kernels_TFLOPs": round(sum(v[0] for v in by_kernel.values()) / INSTR_STEPS /
                                                 max(step_conv_time, 1e-9) / 1e12, 2)}

    # ---- second half of the metric ("XOLP-kernel GB/s"): K1 exactly as the step calls it (512x612 planes in,
    # XOLP + 9-channel normals out on the 512x640 pitch), kernel-only time by HIP events, algorithmic bytes
    # = 4 B read + (8 + 36) B written per frame pixel.  Every launch works on another of
    # K1_SETS buffer sets (3 x 248 MB: beyond the 256 MB Infinity Cache), as in the step, where 79 ms of other
    # traffic separate two K1 launches; the cache-resident figure (one set) is reported beside it.
    K1_SETS = 3
    pol = batch[("pol", 0, 0)]
    pols = [pol] + [pol.roll(7 * i, dims=3).contiguous() for i in range(1, K1_SETS)]
    k1_outs = [pdpolar.polar_forward(p, want=("xolp", "normals"), out_width=W) for p in pols]

    def time_k1(nsets, launches=24):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for i, (e0, e1) in enumerate(evs):
            e0.record()
            pdpolar.polar_forward(pols[i % nsets], want=("xolp", "normals"), out_width=W, out=k1_outs[i % nsets])
            e1.record()
        torch.cuda.synchronize()
        return sorted(e0.elapsed_time(e1) for e0, e1 in evs)[launches // 2]

    time_k1(K1_SETS, 6)
    k1_ms = time_k1(K1_SETS)
    k1_ms_warm = time_k1(1)
    # K1 on a batch large enough that launch / prologue / drain no longer weigh (SURVEY.md §7 hard part 7: "report GB/s vs B")
    k1_big = None
    try:
        BIG = 128
        big_pol = pol.repeat(BIG // args.batch + 1, 1, 1, 1)[:BIG].contiguous()
        big_out = pdpolar.polar_forward(big_pol, want=("xolp", "normals"), out_width=W)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
        for e0, e1 in evs:
            e0.record()
            pdpolar.polar_forward(big_pol, want=("xolp", "normals"), out_width=W, out=big_out)
            e1.record()
        torch.cuda.synchronize()
        big_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)[len(evs) // 2]
        k1_big = {"batch": BIG, "avg_launch_ms": round(big_ms, 4), "GBps": round(BIG * H * FRAME_W * 48 / big_ms / 1e6, 1),
                  "frac": round(BIG * H * FRAME_W * 48 / big_ms / 1e6 / 8000.0, 4)}
        del big_pol, big_out
    except Exception as exc:
        k1_big = {"error": str(exc)[:200]}
    k1_bytes = args.batch * H * FRAME_W * 48
    # `achieved` is the launch INSIDE the training step (what a rocprof trace of this command shows): the step streams
    # gigabytes between two K1 launches, so K1 starts with its LUT / table image evicted and the caches holding the
    # predecessor's lines; back to back on rotating buffers (HBM-resident data, warm LUT) the same kernel is faster.
    xolp_kernel = {"kernel": "polar_kernel<LS,fast normals,1024,nt,hot>", "bound": "hbm",
                   "achieved": round(k1_bytes / k1_in_step_ms / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                   "frac": round(k1_bytes / k1_in_step_ms / 1e6 / 8000.0, 4), "algorithmic_bytes_per_launch": k1_bytes,
                   "avg_launch_ms": round(k1_in_step_ms, 4), "where": "inside the train step (HIP events, median of 4 steps)",
                   "back_to_back": {"buffer_sets": K1_SETS, "avg_launch_ms": round(k1_ms, 4),
                                    "GBps": round(k1_bytes / k1_ms / 1e6, 1), "frac": round(k1_bytes / k1_ms / 1e6 / 8000.0, 4),
                                    "cache_resident_GBps": round(k1_bytes / k1_ms_warm / 1e6, 1)},
                   "batch_128": k1_big,
                   "streaming_ceiling_note": "K1's byte mix (4 B in, 44 B out per pixel as eleven planar fp32 planes, nontemporal "
                                             "stores) with TRIVIAL compute streams at 5.3-5.7 TB/s at B=128 and takes 37-41 us at B=16 "
                                             "back to back on this pool; a float4 copy 5.3-5.9, pure reads 6.1-6.3, pure nontemporal "
                                             "writes 4.8-5.8 TB/s; every other layout of the 48 B/px (float3 planes, 36 B / 44 B records) "
                                             "0.4-3.4 TB/s (tools/membench5.hip, profiles/r04_membench5_k1_ceiling.log).  At B=16 the "
                                             "kernel's issue phase ends after ~36 us; the rest is launch, table prologue and HBM write "
                                             "drain (profiles/r03_k1_timeline.log)"}
    del pols, k1_outs

    attention = None
    if args.attention:
        # the attention kernels alone at the step's size (16 x 5120 tokens, C = 128): HIP events, algorithmic FLOPs
        # 4 N T^2 C forward, 10 N T^2 C backward (five products), against the matrix peak of the operand type
        Tn, Cn = (H // 8) * (W // 8), 128
        qa, ka, va = (torch.randn(args.batch, Cn, H // 8, W // 8, device="cuda").contiguous(memory_format=torch.channels_last)
                      .requires_grad_(True) for _ in range(3))
        PF.self_attention(qa, ka, va).sum().backward()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        reps = 5
        ev[0].record()
        outs = [PF.self_attention(qa, ka, va) for _ in range(reps)]
        ev[1].record()
        for o_ in outs:
            o_.backward(torch.ones_like(o_))
        ev[2].record()
        torch.cuda.synchronize()
        t_f, t_b = ev[0].elapsed_time(ev[1]) / reps, ev[1].elapsed_time(ev[2]) / reps
        fl = args.batch * Tn * Tn * Cn
        peak = 2500.0 if args.bf16 else FP32_MFMA_PEAK_TF
        attention = {"dtype": "bf16 operands, fp32 softmax/accumulate" if args.bf16 else "f32",
                     "kernels": "attn_*_bf16_kernel (v_mfma_f32_32x32x16_bf16)" if args.bf16 else "attn_*_kernel (v_mfma_f32_32x32x2_f32)",
                     "tokens": Tn, "fwd_ms": round(t_f, 3), "bwd_ms": round(t_b, 3),
                     "fwd_TFLOPs": round(4 * fl / t_f / 1e9, 1), "bwd_TFLOPs": round(10 * fl / t_b / 1e9, 1),
                     "peak_TFLOPs": peak, "fwd_frac": round(4 * fl / t_f / 1e9 / peak, 4),
                     "bwd_frac": round(10 * fl / t_b / 1e9 / peak, 4)}

    result = {
        "metric": "train images/sec (512x612, 3-encoder)", "value": round(args.batch * world * args.steps / dt, 3),
        "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: full 3-encoder (augment_xolp+augment_normals), multi-scale loss "
                               "scales=[0,1,2,3], batch 16 per GPU, 512x612 frames (network/loss on 512x640 padded), "
                               "dropout 0.1, Adam lr 1e-4, fp32" + (" + joint-encoder attention (configs[4] variant)"
                                                                   if args.attention else "") +
                               (" + separate normals decoder (arch1++_separate_normals_dec variant)" if args.normals_decoder else ""),
                   "global_batch": args.batch * world, "height": H, "width": W, "frame_width": FRAME_W,
                   "parallelism": f"dp{world}"},
        "final_loss": round(loss_val, 6), "host_enqueue_ms_per_step": None if t_host is None else round(t_host * 1e3, 2),
        "step_launch": "eager",
        "eager": eager, "graph": graph_info,
        "roofline": roofline, "xolp_kernel": xolp_kernel,
        "precision": {"accumulate": "f32",
                      "conv_products": "fp32 MFMA; the layers with >= 32 output channels and enough tiles to fill the chip (forward with zero "
                                       "or 3x3 reflection padding, stride-1 data gradient, the 4x4 space-to-depth stems, weight gradients) form "
                                       "each fp32 product from a three-way bf16 split of both operands (six bf16 MFMAs, dropped terms <= 2^-23 "
                                       "of the product; odd tiles / slices accumulate the negated result so that the bf16 MFMA's truncation "
                                       "bias cancels in sums)" if _ops.CONV_FLAGS == _ops.CONV_AUTO else "fp32 MFMA",
                      "evidence": "tests/test_conv_gpu.py::test_bf16x3_kernel_keeps_fp32_accuracy, test_rolling_row_weight_gradient, "
                                  "test_row_window_form_of_the_space_to_depth_stems (forward / data gradient / weight gradient error vs an "
                                  "fp64 convolution <= 1.5x the fp32-MFMA kernel's on the same input), test_bf16x3_kernels_nonfinite_and_"
                                  "extreme_inputs, _denormal_range, _cancellation_heavy_contraction, and tests/test_prodsize_gpu.py::"
                                  "test_full_resolution_training_step_matches_oracle at B = 4 and B = 16 (512x640 step vs the fp32 and fp64 "
                                  "oracle: disparities 2e-5, losses 1e-4, every gradient tensor within 2x the fp32 oracle's own distance "
                                  "from fp64)",
                      "fp32_mfma_only": fp32_only},
    }
    if dp_info is not None:
        result["dp"] = dp_info
    if attention is not None:
        result["attention"] = attention
        if args.bf16:
            result["dtype"] = "f32 (network) + bf16 attention operands"
            result["config"]["workload"] += " on the bf16 matrix cores"
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.attention and not args.normals_decoder:
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
