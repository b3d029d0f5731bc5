"""manydepth.trainer.Trainer -- the reference's supervised training loop (manydepth/trainer.py:73)
on the MI355X-native engine.

Kept from the reference: constructor contract (options Namespace), attributes (models,
model_optimizer, model_lr_scheduler, opt, device, epoch, step), method names and return values of
train / run_epoch / process_batch / compute_losses / compute_supervised_normals_losses / val / test /
set_train / set_eval / save_opts / save_model / load_model / load_mono_model / log / log_time, the
per-model ``<name>.pth`` + ``adam.pth`` checkpoint layout and the loss dictionary keys.

Implemented scope: the supervised single-frame branch (``--depth_supervision_only``,
``--depth_supervision``) -- trainer.py:192-216, 469-477, 497-513, 531-545, 567, 1126-1150, 1241-1265,
1298-1309.  Pose / matching / reprojection / DPT / stereo branches raise NotImplementedError.

What runs where: every tensor operation of a training step is a hand-written HIP kernel from
libpolardepth.so (polar preprocessing K1, implicit-GEMM convolutions K2, fused BN/ReLU/pool chains
K3, decoder glue K4, multi-scale loss K5, Adam); torch supplies memory, streams, the autograd tape
and torch.distributed (RCCL).  There is no CPU fallback: ``--no_cuda`` is rejected.
"""
import json
import logging
import os
import time
from datetime import datetime

import numpy as np
import torch
from torch.utils.data import DataLoader

from .utils import readlines, sec_to_hm_str
from .layers import SSIM, compute_depth_errors, compute_depth_errors_numpy
from manydepth import datasets, networks
from polardepth import functional as PF
from polardepth import polar as pdpolar
from polardepth import resize as pdresize
from polardepth import ops
from polardepth.engine import ParamStore, FusedAdam, GradReducer
from polardepth._lib import lib, check, ptr, stream_ptr

try:
    from torch.utils.tensorboard import SummaryWriter
except Exception:                                   # tensorboard is optional: keep the interface
    class SummaryWriter:
        def __init__(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def add_image(self, *a, **k): pass
        def close(self): pass

_MATERIAL_GREY = {"box": 20, "bottle": 40, "can": 60, "cup": 80, "remote": 100, "teapot": 120, "cutlery": 140,
                  "glass": 160, "table": 180, "wall": 200}     # instance-mask grey values, trainer.py:1388-1407


def _unused_resnet_param(model_name, param_name):
    """ShallowResnetEncoder builds a full resnet18 but never runs layer3/layer4/fc (resnet_encoder.py:819-820)."""
    return model_name == "rgb_encoder" and param_name.split(".")[1] in ("layer3", "layer4", "fc")


def _attention_variant(opt):
    """BASELINE config 5 ("arch1++_attention"): the reference's master parser has no flag for it, so the option
    table stays identical to manydepth/options.py and the variant is switched by an attribute set on the options
    object (``opt.joint_attention = True``) or by ``PD_JOINT_ATTENTION=1``."""
    return bool(getattr(opt, "joint_attention", False)) or os.environ.get("PD_JOINT_ATTENTION") == "1"


def _normals_decoder_variant(opt):
    """`arch1++_separate_normals_dec` (reference README.md:54): like the attention variant the reference's master parser has
    no flag for it; selected on the options object (``opt.normals_decoder = True``) or by ``PD_NORMALS_DECODER=1``."""
    return bool(getattr(opt, "normals_decoder", False)) or os.environ.get("PD_NORMALS_DECODER") == "1"


class Trainer:
    def __init__(self, options):
        self.opt = options
        if self.opt.no_cuda:
            raise RuntimeError("--no_cuda: this build has no CPU path (the CPU restatement lives in oracle/ and "
                               "is test infrastructure only)")
        if not torch.cuda.is_available():
            raise RuntimeError("no MI355X visible: the HIP hot path cannot run and there is no CPU fallback")
        for flag in ("train_stereo_only", "train_dpt", "train_student", "use_stereo", "res_pose", "supervise_pose"):
            if getattr(self.opt, flag, False):
                raise NotImplementedError(f"--{flag}: outside the supervised hot path of this build")
        if not (self.opt.depth_supervision_only and self.opt.depth_supervision):
            raise NotImplementedError("only --depth_supervision_only True --depth_supervision True is implemented "
                                      "(the path of train_supervised_GT.sh)")
        local_rank = int(os.environ.get("LOCAL_RANK", 0))
        self.device = torch.device("cuda", local_rank)
        torch.cuda.set_device(self.device)
        self.distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.rank = torch.distributed.get_rank() if self.distributed else 0

        timestamp = datetime.now()
        self.data_path, self.data_path_val, self.log_dir = self.opt.data_path, self.opt.data_path_val, self.opt.log_dir
        self.log_path = os.path.join(self.opt.log_dir, self.opt.model_name + '_' + timestamp.strftime("%m-%d_%H-%M-%S"))
        self.log_args(timestamp)

        assert self.opt.height % 32 == 0, "'height' must be a multiple of 32"
        assert self.opt.width % 32 == 0, "'width' must be a multiple of 32"

        self.models = {}
        self.parameters_to_train = []
        self.num_scales = len(self.opt.scales)
        self.train_teacher_and_pose = not self.opt.freeze_teacher_and_pose

        # MODEL SETUP (trainer.py:192-216)
        self.models["rgb_encoder"] = networks.ShallowResnetEncoder(18, self.opt.weights_init == "pretrained")
        if self.opt.augment_normals:
            self.models["normals_encoder"] = networks.ShallowNormalsEncoder(in_channels=9,
                                                                            dropout_rate=self.opt.dropout_rate)
        if self.opt.augment_xolp:
            self.models["xolp_encoder"] = networks.ShallowEncoder(mode='XOLP', in_channels=2,
                                                                  dropout_rate=self.opt.dropout_rate)
        self.models["joint_encoder"] = networks.JointEncoder(dropout_rate=self.opt.dropout_rate,
                                                             include_normals=self.opt.augment_normals,
                                                             include_xolp=self.opt.augment_xolp,
                                                             attention=_attention_variant(self.opt))
        self.models["mono_depth"] = networks.DepthDecoder(self.models["rgb_encoder"].num_ch_enc, self.opt.scales)
        if _normals_decoder_variant(self.opt):
            if not self.opt.augment_normals:
                raise ValueError("the separate normals decoder sits behind the normals encoder: it needs --augment_normals")
            self.models["normals_decoder"] = networks.NormalsDecoder(64)
        for m in self.models.values():
            m.to(self.device)
        if self.distributed:                    # identical replicas: broadcast rank 0's initialisation
            for m in self.models.values():
                for t in list(m.parameters()) + list(m.buffers()):
                    torch.distributed.broadcast(t.data, 0)

        order = [n for n in ("rgb_encoder", "xolp_encoder", "normals_encoder", "joint_encoder", "mono_depth",
                             "normals_decoder") if n in self.models]
        self.store = ParamStore(self.models, order=order, unused=_unused_resnet_param, device=self.device)
        for n in order:
            self.parameters_to_train += list(self.models[n].parameters())
        self.reducer = GradReducer(self.store) if self.distributed else None
        self.model_optimizer = FusedAdam(self.store, self.opt.learning_rate, reducer=self.reducer)
        if self.distributed:
            self.model_optimizer.grad_scale = 1.0 / torch.distributed.get_world_size()
        self.model_lr_scheduler = torch.optim.lr_scheduler.StepLR(self.model_optimizer, self.opt.scheduler_step_size, 0.1)

        if self.opt.load_weights_folder is not None:
            self.load_model()
        if self.opt.mono_weights_folder is not None:
            self.load_mono_model()

        # DATA (trainer.py:254-303)
        if self.opt.dataset != "HAMMER":
            raise NotImplementedError("only --dataset HAMMER is on the hot path")
        self.dataset = datasets.HAMMER_Dataset

        def _split(split, which, data_path):
            path = os.path.join("splits", split, f"{which}_files.txt")
            if os.path.exists(path):
                return readlines(path)
            if str(data_path) == datasets.SYNTHETIC:      # seeded synthetic items need no scene list
                return []
            raise FileNotFoundError(f"{path} not found (run from the repository root, trainer.py:261-263; "
                                    f"--data_path {datasets.SYNTHETIC} serves synthetic items)")
        train_files = [self.opt.overfit_scene] if self.opt.overfit else _split(self.opt.split, "train", self.data_path)
        val_files = [self.opt.overfit_scene] if self.opt.overfit else _split(self.opt.split, "val", self.data_path)
        test_files = _split(self.opt.eval_split, "test", self.data_path_val)
        mk = lambda path, files, tr: self.dataset(path, files, self.opt.height, self.opt.width, [0], 4, is_train=tr,
                                                  img_ext='.png', offset=self.opt.offset, modality=self.opt.modality,
                                                  supervised_depth=True, supervised_depth_only=True,
                                                  depth_modality=self.opt.depth_modality)
        train_dataset, val_dataset, test_dataset = mk(self.data_path, train_files, True), \
            mk(self.data_path, val_files, False), mk(self.data_path_val, test_files, False)
        sampler = torch.utils.data.distributed.DistributedSampler(train_dataset) if self.distributed else None
        self.train_loader = DataLoader(train_dataset, self.opt.batch_size, sampler is None, sampler=sampler,
                                       num_workers=self.opt.num_workers, pin_memory=True, drop_last=True)
        self.val_loader = DataLoader(val_dataset, self.opt.batch_size, False, num_workers=self.opt.num_workers,
                                     pin_memory=True, drop_last=True)
        self.val_iter = iter(self.val_loader)
        self.test_loader = DataLoader(test_dataset, self.opt.batch_size, False, num_workers=self.opt.num_workers,
                                      pin_memory=True, drop_last=True)
        self.num_total_steps = len(train_dataset) // self.opt.batch_size * self.opt.num_epochs

        self.writers = {}
        modes = ["train", "val", "val_mono", "test", "test_mono"] + ["test_mono_" + m for m in
                                                                     ("glass", "cutlery", "can", "bottle", "cup", "teapot",
                                                                      "remote", "box", "table", "wall")]
        for mode in modes:
            self.writers[mode] = SummaryWriter(os.path.join(self.log_path, mode)) if self.rank == 0 else SummaryWriter()
        if not self.opt.no_ssim:
            self.ssim = SSIM()
        self.depth_metric_names = ["de/abs_rel", "de/sq_rel", "de/rms", "de/log_rms", "da/a1", "da/a2", "da/a3"]
        self.depth_metric_names_mono = [n.replace("/", "_mono/", 1) for n in self.depth_metric_names]
        # PD_GLOBAL_LOSS_NORM=1 (or opt.global_loss_norm): mask-normalise over the global batch like the reference's
        # single process; default = per replica (standard DDP semantics, no exchange on the loss path)
        global_norm = self.distributed and (bool(getattr(self.opt, "global_loss_norm", False))
                                            or os.environ.get("PD_GLOBAL_LOSS_NORM") == "1")
        # independent encoders on separate HIP streams (PD_ENCODER_STREAMS=0: everything on the current stream)
        self.encoder_streams = os.environ.get("PD_ENCODER_STREAMS", "1") != "0" and self.device.type == "cuda"
        self.step_graph = os.environ.get("PD_STEP_GRAPH") == "1" or bool(getattr(self.opt, "step_graph", False))
        self._graphed = None
        self._enc_streams = []
        self.loss_cfg = PF.LossCfg(self.opt.scales, self.opt.min_depth, self.opt.max_depth, self.opt.normals_loss_weight,
                                   self.opt.disparity_smoothness, self.opt.height, self.opt.width, global_norm=global_norm)
        self.epoch, self.step = 0, 0
        self.start_time = time.time()
        if self.rank == 0:
            print("There are {:d} training items and {:d} validation items and {:d} test items\n".format(
                len(train_dataset), len(val_dataset), len(test_dataset)))
            self.save_opts()

    # ------------------------------------------------------------------ bookkeeping
    def log_args(self, timestamp):
        if int(os.environ.get("RANK", 0)) != 0:
            return
        os.makedirs(self.log_path, exist_ok=True)
        logging.basicConfig(filename=os.path.join(self.log_path, 'args.log'), level=logging.INFO, format='%(message)s')
        logging.info('Run started at: %s', str(timestamp))
        for arg, value in sorted(vars(self.opt).items()):
            logging.info("%s: %r", arg, value)

    def set_train(self):
        for m in self.models.values():
            m.train()

    def set_eval(self):
        for m in self.models.values():
            m.eval()

    # ------------------------------------------------------------------ training loop (trainer.py:379-467)
    def train(self):
        # a loaded trainer_state.pth continues where the checkpoint stopped (the reference restarts at epoch 0)
        first_epoch = getattr(self, "resume_epoch", 0)
        self.epoch, self.step = first_epoch, getattr(self, "resume_step", 0)
        self.test()
        self.start_time = time.time()
        for self.epoch in range(first_epoch, self.opt.num_epochs):
            self.run_epoch()
            if (self.epoch + 1) % self.opt.save_frequency == 0:
                self.save_model(epoch_complete=True)     # rank 0 writes, every rank waits (no torn files under torchrun)
                self.test()

    def run_epoch(self):
        self.set_train()
        if self.distributed and hasattr(self.train_loader.sampler, "set_epoch"):
            self.train_loader.sampler.set_epoch(self.epoch)
        # a checkpoint written in the middle of an epoch resumes in that epoch, behind the batches already consumed
        skip = getattr(self, "resume_batch", 0) if self.epoch == getattr(self, "resume_epoch", -1) else 0
        self.resume_batch = 0
        for batch_idx, inputs in enumerate(self.train_loader):
            if batch_idx < skip:
                continue
            before_op_time = time.time()
            if self.step_graph:
                # PD_STEP_GRAPH=1: the whole step (zero_grad .. Adam) replayed from a hipGraph captured on the first batch
                # (polardepth/graph.py; same bits as the eager step, ~2 ms of host time instead of 13-35)
                dev_inputs = {k: v.to(self.device, non_blocking=True) for k, v in inputs.items()}
                if self._graphed is None:
                    from polardepth.graph import GraphedTrainStep
                    self._graphed = GraphedTrainStep(self, dev_inputs, warmup=1, restore_state=True)
                self._graphed.step(dev_inputs)
                outputs, losses = self._graphed.outputs, self._graphed.losses
                inputs = {**dev_inputs, **{k: v for k, v in self._graphed.static.items() if k not in dev_inputs}}
            else:
                self.model_optimizer.zero_grad()
                outputs, losses, mono_outputs = self.process_batch(inputs, is_train=True)
                losses["loss"].backward()
                self.model_optimizer.step()
            early_phase = batch_idx % self.opt.log_frequency == 0 and self.step < 2000
            late_phase = self.step % 2000 == 0
            if early_phase or late_phase:
                duration = time.time() - before_op_time
                self.log_time(batch_idx, duration, losses["loss"].cpu().data)    # the only host sync of the loop
                if "depth_gt" in inputs:
                    self.compute_depth_losses(inputs, outputs, losses)
                self.log("train", inputs, outputs, losses)
                self.val()
                self.save_model(epoch_complete=False, batch_idx=batch_idx)
            self.step += 1
        self.model_lr_scheduler.step()

    # ------------------------------------------------------------------ forward (trainer.py:469-648)
    def _polar_inputs(self, inputs):
        """On-device XOLP / normals from the raw planes when the loader hands them over (K1)."""
        normals = None
        if ("pol_mosaic", 0, 0) in inputs and ("pol", 0, 0) not in inputs:
            # un-split sensor frame (four polarizer images as quadrants): split on the device
            inputs[("pol", 0, 0)] = pdpolar.split_mosaic(inputs[("pol_mosaic", 0, 0)])
        if ("pol", 0, 0) in inputs and (self.opt.augment_xolp or self.opt.augment_normals):
            want = ["xolp"] + (["normals"] if self.opt.augment_normals else [])
            pol = inputs[("pol", 0, 0)]
            if pol.shape[2] != self.opt.height or pol.shape[3] > self.opt.width:
                # raw frames from the loader (HAMMER_Dataset(raw_pol=True)): Pillow-exact LANCZOS resize on the device
                pol = pdresize.resize_lanczos_u8(pol, (self.opt.height, self.opt.width))
            # planes narrower than the network width (512x612 frames -> 512x640): K1 pads on the fly
            out = pdpolar.polar_forward(pol, want=tuple(want),
                                        out_width=self.opt.width if pol.shape[3] < self.opt.width else None)
            inputs[("xolp", 0, 0)] = out["xolp"]
            normals = out.get("normals")
        return normals

    def _forward_models(self, inputs):
        normals = self._polar_inputs(inputs)
        xolp_feats = normals_feats = None
        side = []
        if self.encoder_streams and (self.opt.augment_xolp or self.opt.augment_normals):
            # The three encoders are independent until the joint encoder: the shallow XOLP / normals encoders run on
            # two side streams next to the ResNet on the main one, so the HBM-bound kernels of one encoder (BatchNorm /
            # ReLU / pool chains) overlap the matrix-bound convolutions of another.  autograd replays every backward
            # node on its forward stream and inserts the cross-stream waits itself.
            main = torch.cuda.current_stream()
            x_in = inputs["xolp", 0, 0].float()
            if self.opt.augment_xolp:
                sx = self._encoder_stream(0)
                sx.wait_stream(main)
                with torch.cuda.stream(sx):
                    xolp_feats = self.models["xolp_encoder"](x_in)
                x_in.record_stream(sx)
                side.append((sx, xolp_feats))
            if self.opt.augment_normals:
                sn = self._encoder_stream(1)
                sn.wait_stream(main)
                with torch.cuda.stream(sn):
                    normals_feats = self.models["normals_encoder"](x_in, normals=normals)
                x_in.record_stream(sn)
                if normals is not None:
                    normals.record_stream(sn)
                side.append((sn, normals_feats))
        feats = self.models["rgb_encoder"](inputs["color_aug", 0, 0].float())
        if side:
            main = torch.cuda.current_stream()
            for st, out in side:
                main.wait_stream(st)
                for t in (out if isinstance(out, (list, tuple)) else [out]):
                    if torch.is_tensor(t):
                        t.record_stream(main)
        else:
            if self.opt.augment_xolp:
                xolp_feats = self.models["xolp_encoder"](inputs["xolp", 0, 0].float())
            if self.opt.augment_normals:
                normals_feats = self.models["normals_encoder"](inputs["xolp", 0, 0].float(), normals=normals)
        enc_feats = self.models["joint_encoder"](feats[-1], xolp_feats, normals_feats)
        feats = list(feats) + enc_feats
        outputs = dict(self.models['mono_depth'](feats))
        if "normals_decoder" in self.models:      # `arch1++_separate_normals_dec`: normals straight from the normals encoder
            outputs[("normals_pred", 0)] = self.models["normals_decoder"](normals_feats)
        return outputs

    def _encoder_stream(self, i):
        while len(self._enc_streams) <= i:
            self._enc_streams.append(torch.cuda.Stream(device=self.device))
            PF.register_producer_stream(self._enc_streams[-1])      # the gradient reducer / Adam wait for it too
        return self._enc_streams[i]

    def process_batch(self, inputs, is_train=False):
        for key, ipt in inputs.items():
            inputs[key] = ipt.to(self.device, non_blocking=True)
        if self.train_teacher_and_pose:
            mono_outputs = self._forward_models(inputs)
        else:
            with torch.no_grad():
                mono_outputs = self._forward_models(inputs)
        for scale in self.opt.scales:
            mono_outputs[("disp", 0, scale)] = mono_outputs[("disp", scale)]
        losses = self.compute_losses(inputs, mono_outputs, is_multi=False)
        return mono_outputs, losses, mono_outputs

    def compute_losses(self, inputs, outputs, is_multi=False):
        """Supervised branch of trainer.py:1126-1296; also fills ("depth",0,s) / ("mono_depth",0,s)
        (trainer.py:538-545) because K5 produces the full-resolution depth maps on the way."""
        scales = list(self.opt.scales)
        disps = [outputs[("disp", s)] for s in scales]
        colors = [inputs[("color", 0, s)] for s in scales]
        vals, depths = PF.multiscale_loss(self.loss_cfg, inputs["depth"], inputs[("K", 0)], disps, colors)
        losses = {"loss": vals[0]}
        for i, s in enumerate(scales):
            losses[f"loss/{s}"] = vals[1 + 3 * i]
            losses[f"supervised_depth_loss/{s}"] = vals[2 + 3 * i]
            losses[f"supervised_normals_loss/{s}"] = vals[3 + 3 * i]
            outputs[("depth", 0, s)] = depths[i]
            outputs[("mono_depth", 0, s)] = depths[i]
        if ("normals_pred", 0) in outputs:
            # `arch1++_separate_normals_dec`: the predicted normals against the normals of the ground truth, weighted like
            # the normals term of the depth loss (README.md:54,67: normals_loss_weight)
            nl = PF.normals_pred_loss(outputs[("normals_pred", 0)], inputs["depth"], inputs[("K", 0)], self.opt.min_depth,
                                      self.opt.max_depth)
            losses["normals_decoder_loss"] = nl
            losses["loss"] = losses["loss"] + self.opt.normals_loss_weight * nl
        return losses

    def compute_supervised_normals_losses(self, depth_gt, depth_pred, intrinsics, mask=None):
        """trainer.py:1298-1309 (value only): sum((2 - cos(n_gt, n_pred)) * mask) / sum(mask) with the CALLER's mask
        [N,1,H,W] (the reference passes the depth-range mask, trainer.py:1242-1243,1250; None selects it here)."""
        N, _, H, W = depth_gt.shape
        dev = depth_gt.device
        K = torch.eye(4, device=dev)[None].repeat(N, 1, 1)
        K[:, :3, :3] = intrinsics[:, :3, :3]
        gt, pred = depth_gt.contiguous().float(), depth_pred.detach().contiguous().float()
        if mask is None:
            mask = (gt >= self.opt.min_depth) & (gt <= self.opt.max_depth)
        mask = mask.to(device=dev, dtype=torch.float32).expand(N, 1, H, W).contiguous()
        part = torch.empty((lib.pd_loss_rows(N * H * W), 2), dtype=torch.float32, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        check(lib.pd_normals_loss_masked(ptr(pred), ptr(gt), ptr(K.contiguous()), ptr(mask), ptr(part), ptr(out), N, H, W,
                                         stream_ptr()), "pd_normals_loss_masked")
        return out[0]

    # ------------------------------------------------------------------ validation / test
    def _next_val(self):
        try:
            return next(self.val_iter)
        except StopIteration:
            self.val_iter = iter(self.val_loader)
            return next(self.val_iter)

    def val(self):
        self.set_eval()
        inputs = self._next_val()
        with torch.no_grad():
            outputs, losses, _ = self.process_batch(inputs)
            losses = {}
            if "depth_gt" in inputs:
                self.compute_depth_losses(inputs, outputs, losses, mono=True)
            self.log("val_mono", inputs, outputs, losses, log_images=False, log_essential_images=True)
        self.set_train()

    def test(self):
        """trainer.py:782-980: full test split, "all" + per-material metrics.  The per-image masked metrics
        (compute_depth_losses_from_list, trainer.py:1357-1434) are reduced on the device by pd_depth_metrics;
        only 11 x 7 numbers per batch leave the GPU."""
        if self.rank == 0:
            print("Running full test set at Epoch: ", self.epoch)
        self.set_eval()
        objects = ["all"] + list(_MATERIAL_GREY)
        sums = {o: torch.zeros(7, dtype=torch.float64, device=self.device) for o in objects}
        counts = {o: torch.zeros((), dtype=torch.float64, device=self.device) for o in objects}
        with torch.no_grad():
            for inputs in self.test_loader:
                for key, ipt in inputs.items():
                    inputs[key] = ipt.to(self.device)
                outputs = self._forward_models(inputs)
                depth = torch.empty_like(inputs["depth_gt"])
                N, _, H, W = depth.shape
                d0 = outputs[("disp", 0)].contiguous()
                check(lib.pd_disp_to_depth(ptr(d0), ptr(depth), None, N, d0.shape[2], d0.shape[3], H, W,
                                           self.opt.min_depth, self.opt.max_depth, stream_ptr()), "pd_disp_to_depth")
                for o in objects:
                    m = ops.depth_metrics(inputs["depth_gt"], depth, self.opt.min_depth, self.opt.max_depth,
                                          mask=None if o == "all" else inputs[("mask", 0, 0)],
                                          mask_value=_MATERIAL_GREY.get(o, 0))
                    valid = m[:, 7] > 0
                    sums[o] += (m[:, :7].double() * valid[:, None]).sum(0)
                    counts[o] += valid.sum()
        if self.rank == 0:
            for o in objects:
                if counts[o].item() == 0:
                    continue
                mean_errors = (sums[o] / counts[o]).cpu().numpy()
                print("\n  " + o + "\n  " + ("{:>8} | " * 7).format("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3"))
                print(("&{: 8.5f}  " * 7).format(*mean_errors.tolist()) + "\\\\")
                losses = {metric: np.array(mean_errors[i]) for i, metric in enumerate(self.depth_metric_names)}
                self.log("test_mono" if o == "all" else "test_mono_" + o, None, None, losses, log_images=False)
        self.set_train()

    def compute_depth_losses(self, inputs, outputs, losses, mono=False):
        """trainer.py:1311-1355."""
        depth_pred = outputs[("depth", 0, 0)].detach().clamp(self.opt.min_depth, self.opt.max_depth)
        depth_gt = inputs["depth_gt"]
        mask = (depth_gt > self.opt.min_depth) * (depth_gt < self.opt.max_depth)
        errs = compute_depth_errors(depth_gt[mask], depth_pred[mask])
        for i, metric in enumerate(self.depth_metric_names):
            losses[metric] = np.array(errs[i].cpu())

    def compute_depth_losses_from_list(self, gts, preds, losses, masks, object="all"):
        """trainer.py:1357-1434: per-image metrics, optionally restricted to one material class."""
        errors = []
        lo, hi = self.opt.min_depth, self.opt.max_depth
        for k in range(len(preds)):
            pred_b = preds[k].clamp(lo, hi)[:, 0].numpy()
            gt_b = gts[k][:, 0].numpy()
            m_b = masks[k][:, 0].numpy()
            for b in range(pred_b.shape[0]):
                mask = np.logical_and(gt_b[b] > lo, gt_b[b] < hi)
                if object != "all":
                    mask = np.logical_and(mask, m_b[b] == _MATERIAL_GREY[object])
                if not mask.any():
                    continue
                errors.append(compute_depth_errors_numpy(gt_b[b][mask], np.clip(pred_b[b][mask], lo, hi)))
        if not errors:
            return
        mean_errors = np.array(errors).mean(0)
        print("\n  " + ("{:>8} | " * 7).format("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3"))
        print(("&{: 8.5f}  " * 7).format(*mean_errors.tolist()) + "\\\\")
        for i, metric in enumerate(self.depth_metric_names):
            losses[metric] = np.array(mean_errors[i])

    # ------------------------------------------------------------------ logging
    def log_time(self, batch_idx, duration, loss):
        samples_per_sec = self.opt.batch_size / duration
        time_sofar = time.time() - self.start_time
        left = (self.num_total_steps / self.step - 1.0) * time_sofar if self.step > 0 else 0
        if self.rank == 0:
            print("epoch {:>3} | batch {:>6} | examples/s: {:5.1f} | loss: {:.5f} | time elapsed: {} | time left: {}".format(
                self.epoch, batch_idx, samples_per_sec, float(loss), sec_to_hm_str(time_sofar), sec_to_hm_str(left)))

    def log(self, mode, inputs, outputs, losses, log_images=True, log_essential_images=False, mono_depth=False):
        writer = self.writers[mode]
        for l, v in losses.items():
            try:
                value = float(v)
            except (TypeError, ValueError):      # non-scalar entries (images, per-pixel maps) are not logged by this build
                continue
            writer.add_scalar("{}".format(l), value, self.step)

    # ------------------------------------------------------------------ checkpoints (trainer.py:1586-1691)
    def save_opts(self):
        models_dir = os.path.join(self.log_path, "models")
        os.makedirs(models_dir, exist_ok=True)
        with open(os.path.join(models_dir, 'opt.json'), 'w') as f:
            json.dump(self.opt.__dict__.copy(), f, indent=2)

    def save_model(self, epoch_complete=True, batch_idx=-1):
        """trainer.py:1597-1617 layout (<model>.pth + adam.pth) plus trainer_state.pth.  Only rank 0 writes; every file
        goes through a temporary name + os.replace, and all ranks meet at a barrier afterwards."""
        save_folder = os.path.join(self.log_path, "models", "weights_{}".format(self.epoch))
        if self.rank == 0:
            os.makedirs(save_folder, exist_ok=True)

            def _save(obj, name):
                tmp = os.path.join(save_folder, name + ".tmp")
                torch.save(obj, tmp)
                os.replace(tmp, os.path.join(save_folder, name))
            for model_name, model in self.models.items():
                _save({k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}, "{}.pth".format(model_name))
            _save(self.model_optimizer.state_dict(), "adam.pth")
            # resume state (not in the reference, which restarts epoch/step/LR schedule on load).
            # run_epoch saves on its logging steps BEFORE self.step += 1 and before lr_scheduler.step(): such a
            # checkpoint is marked incomplete and resumes inside the same epoch.
            _save({"epoch": self.epoch, "step": self.step, "epoch_complete": bool(epoch_complete),
                   "batch_idx": int(batch_idx), "lr_scheduler": self.model_lr_scheduler.state_dict(),
                   "dropout_seed": PF.DropoutState.seed,
                   "dropout_step": PF.DropoutState.get_step(getattr(self, "device", None))}, "trainer_state.pth")
        if self.distributed:
            torch.distributed.barrier()

    def _load_into(self, name, path, strict=False):
        model_dict = self.models[name].state_dict()
        pretrained = torch.load(path, map_location="cpu")
        pretrained = {k: v for k, v in pretrained.items() if k in model_dict}
        model_dict.update(pretrained)
        self.models[name].load_state_dict(model_dict, strict=strict)     # copies in place: flat-buffer views stay valid

    def load_mono_model(self):
        for n in ['rgb_encoder', 'mono_depth', 'normals_encoder', 'xolp_encoder', 'joint_encoder', 'normals_decoder']:
            path = os.path.join(self.opt.mono_weights_folder, "{}.pth".format(n))
            if n in self.models and os.path.isfile(path):
                self._load_into(n, path, strict=True)

    def load_model(self):
        folder = os.path.expanduser(self.opt.load_weights_folder)
        assert os.path.isdir(folder), "Cannot find folder {}".format(folder)
        names = [n for n in self.opt.models_to_load if n in self.models] or list(self.models)
        for n in names:
            path = os.path.join(folder, "{}.pth".format(n))
            if os.path.isfile(path):
                self._load_into(n, path)
        opt_path = os.path.join(folder, "adam.pth")
        if os.path.isfile(opt_path):
            try:
                self.model_optimizer.load_state_dict(torch.load(opt_path, map_location=self.device))
            except (ValueError, KeyError, RuntimeError) as e:
                print("Can't load Adam - using random ({})".format(e))
        state_path = os.path.join(folder, "trainer_state.pth")
        if os.path.isfile(state_path) and getattr(self.opt, "resume_state", True):
            st = torch.load(state_path, map_location="cpu")
            if st.get("epoch_complete", True):       # written by train() after run_epoch (scheduler already stepped)
                self.resume_epoch, self.resume_step, self.resume_batch = int(st["epoch"]) + 1, int(st["step"]), 0
            else:                                    # written inside run_epoch: same epoch, next batch, next step
                self.resume_epoch, self.resume_step = int(st["epoch"]), int(st["step"]) + 1
                self.resume_batch = int(st.get("batch_idx", -1)) + 1
            self.model_lr_scheduler.load_state_dict(st["lr_scheduler"])
            PF.DropoutState.seed = int(st["dropout_seed"])
            PF.DropoutState.set_step(getattr(self, "device", None), int(st.get("dropout_step", 0)))
