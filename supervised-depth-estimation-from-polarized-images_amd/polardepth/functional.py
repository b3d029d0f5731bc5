"""Autograd glue: each Function's forward/backward is a short sequence of HIP kernels.

torch.autograd is used only as the tape; no ATen compute kernel runs inside these Functions.
Parameter gradients are *accumulated by the kernels straight into* ``param.grad`` (a view of the
flat gradient buffer when a ParamStore owns the parameters) and the Functions return ``None``
for them, so there is no per-parameter AccumulateGrad pass; ``zero_grad`` is one memset.
"""
import ctypes
import os

import torch

from . import ops
from ._lib import lib, check, ptr, stream_ptr

CL = torch.channels_last
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


# ------------------------------------------------------------------ dropout RNG stream (Philox)
class DropoutState:
    """Philox4x32-10 stream of the dropout masks: (seed, offset) with offset = site + 4096 * step, where ``site`` numbers
    the dropout call sites since the step began (host side, a kernel argument) and ``step`` counts the training steps in
    DEVICE memory (``state(device)[0]``, incremented by pd_step_tick when the optimizer's zero_grad opens a step).  Nothing
    per-step is a kernel argument, so a captured hipGraph of the step draws fresh masks on every replay; the eager path
    uses the same scheme and therefore the same masks.  (Adam's device-side words are the optimizer's own:
    engine.FusedAdam.dev_state.)"""
    seed = 0x5EEDC0DE
    site = 0
    _state = {}

    @classmethod
    def state(cls, device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        st = cls._state.get(idx)
        if st is None:
            st = cls._state[idx] = torch.zeros(4, dtype=torch.int64, device=torch.device("cuda", idx))
        return st

    @classmethod
    def manual_seed(cls, seed):
        cls.seed, cls.site = int(seed) & (2 ** 63 - 1), 0
        for st in cls._state.values():
            st[0] = 0

    @classmethod
    def get_step(cls, device=None):
        """Training steps begun so far (host copy of the device counter: synchronises; checkpoints only)."""
        if device is None or not torch.cuda.is_available():
            return 0
        return int(cls.state(device)[0].item())

    @classmethod
    def set_step(cls, device, step):
        if device is not None and torch.cuda.is_available():
            cls.state(device)[0] = int(step)

    @classmethod
    def begin_step(cls, device):
        """A training step begins (FusedAdam.zero_grad): site numbering restarts, the device step counter advances."""
        cls.site = 0
        with torch.cuda.device(device):
            check(lib.pd_step_tick(ptr(cls.state(device)), 1, 0, stream_ptr()), "pd_step_tick")

    @classmethod
    def next(cls):
        cls.site += 1
        if cls.site >= 4096:
            raise RuntimeError("more than 4095 dropout sites since the last optimizer.zero_grad(): the Philox offsets of "
                               "consecutive steps would overlap")
        return cls.seed, cls.site


# ------------------------------------------------------------------ side stream for weight gradients
# The weight gradient and the data gradient of a layer both read dz and are independent of each other:
# wgrad kernels run on a side stream next to the dgrad / elementwise kernels of the main stream, which
# fills the tails of the MFMA kernels.  Consumers of the parameter gradients (Adam, the RCCL reducer)
# wait for the side stream through ``sync_wgrad_stream``.
_WGRAD_STREAMS = {}
USE_WGRAD_STREAM = os.environ.get("PD_WGRAD_STREAM", "1") == "1"
USE_FLASH_ATTENTION = os.environ.get("PD_FLASH_ATTENTION", "1") != "0"   # fused attention kernels (config 5)
USE_BF16_ATTENTION = os.environ.get("PD_ATTENTION_BF16", "0") == "1"     # ... on the bf16 matrix cores (configs[4] as specified)
USE_ATTN_BWD_STREAMS = os.environ.get("PD_ATTN_BWD_STREAMS", "1") != "0"  # bf16 attention backward: dK/dV and dQ on two streams
USE_BN_FOLDING = os.environ.get("PD_BN_FOLDING", "1") != "0"   # inference: BatchNorm folded into the conv epilogue
USE_DISP_HEADS = os.environ.get("PD_DISP_HEADS", "1") != "0"     # direct kernels for the 1-channel disparity heads
USE_S2D_STEMS = os.environ.get("PD_S2D_STEMS", "1") == "1"


def wgrad_stream(device):
    st = _WGRAD_STREAMS.get(device.index)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _WGRAD_STREAMS[device.index] = st
    return st


_ATTN_STREAMS = {}


def _attn_side_stream(device):
    """A stream of its own for the dQ kernel of the bf16 attention backward (the weight-gradient stream may hold a queue of
    earlier kernels, which would delay the join)."""
    st = _ATTN_STREAMS.get(device.index)
    if st is None:
        st = _ATTN_STREAMS[device.index] = torch.cuda.Stream(device=device)
    return st


_PRODUCER_STREAMS = []      # further streams that write parameter gradients (the trainer's encoder streams)


def register_producer_stream(st):
    if all(st is not q for q in _PRODUCER_STREAMS):
        _PRODUCER_STREAMS.append(st)


def sync_wgrad_stream(stream=None):
    """Make `stream` (default: the current one) wait for all gradient-producing side streams: the weight-gradient
    stream and the registered encoder streams (a gradient bucket of the reducer may span encoders)."""
    target = stream or torch.cuda.current_stream()
    for st in list(_WGRAD_STREAMS.values()) + _PRODUCER_STREAMS:
        if st is not target and st != target:
            target.wait_stream(st)


_join_queued = False


def _join_after_backward():
    """Autograd callback: when the backward pass ends, the main stream waits for the side stream, so that
    anything enqueued afterwards (reads of .grad, the optimizer) sees complete weight gradients."""
    global _join_queued
    _join_queued = False
    sync_wgrad_stream()


def _wgrad_async(x, dz, fn):
    """Run fn() (which launches wgrad kernels reading x and dz) on the side stream."""
    global _join_queued
    if not USE_WGRAD_STREAM:
        fn()
        return
    if not _join_queued:
        torch.autograd.Variable._execution_engine.queue_callback(_join_after_backward)
        _join_queued = True
    side = wgrad_stream(dz.device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    x.record_stream(side)
    dz.record_stream(side)


# ------------------------------------------------------------------ helpers
def grad_buf(p):
    """The tensor that receives the gradient of parameter p (allocated zeroed on first use)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)          # preserve_format: channels_last weights stay channels_last
    store = getattr(p, "_pd_store", None)
    if store is not None:
        store.grad_is_zero = False            # the next zero_grad() has something to clear (unless Adam clears it)
    return p.grad


def _ready(p):
    hook = getattr(p, "_pd_grad_ready", None)
    if hook is not None:
        hook()


def nhwc_view(t):
    """Return (tensor, row_stride) with channel stride 1 and regular pixel strides; copies only if needed."""
    n, c, h, w = t.shape
    s = t.stride()
    if s[1] == 1 and s[2] == w * s[3] and s[0] == h * w * s[3] and s[3] >= c and s[3] % 4 == 0:
        return t, s[3]
    if c == 1 and t.is_contiguous():
        return t, 1
    t = t.contiguous(memory_format=CL)
    return t, c


def _f32(dev, *shape):
    return torch.empty(shape, dtype=torch.float32, device=dev)


class BNParams:
    """Bundle handed to ConvBNChain: affine params, running buffers and mode."""
    __slots__ = ("gamma", "beta", "running_mean", "running_var", "training", "momentum", "eps")

    def __init__(self, bn, training):
        self.gamma, self.beta = bn.weight, bn.bias
        self.running_mean, self.running_var = bn.running_mean, bn.running_var
        self.training = training
        self.momentum = BN_MOMENTUM if bn.momentum is None else bn.momentum
        self.eps = bn.eps


class SkipGrad:
    """(also used between the decoder's skip connection and the max-pool of the ResNet stem, see attach_skip_mail)
    Mailbox of a residual block: the backward pass of the block's LAST convolution deposits the gradient of the skip
    connection here instead of returning it; the backward pass of the block's FIRST convolution (which autograd runs
    later: its output feeds the last one) adds it in the epilogue of its data-gradient kernel.  Both convolutions read
    the same block input, so the sum is exactly what autograd would have formed with a separate add kernel."""
    __slots__ = ("grad", "closed")

    def __init__(self):
        self.grad = None
        self.closed = False


class ActGrad:
    """Hand-over between the backward nodes around one ELU output y of the depth decoder (depth_decoder.py:60-71).
    The node that produced y (ReflectConvActFn) needs dL/dz = dL/dy * ELU'(z); the consumers of y can deliver exactly
    that from their own kernels instead of a separate pass over the tensor:
      * y has ONE consumer (upconv(i,0) -> upsample): pd_up_bwd_elu multiplies by ELU'(.) and sets `activated`;
      * y has two (upconv(i,1) -> disparity head and next level's first convolution): the convolution deposits its
        data gradient in `grad` instead of returning it, the head (which autograd runs later, its node is older) adds
        it, multiplies by ELU'(.) and sets `activated` -- autograd's add kernel and pd_act_bwd are both gone.
    Any other order falls back to the plain route: a head that finds no deposit although one is expected sets
    `closed`, after which the convolution returns its gradient to autograd as usual."""
    __slots__ = ("grad", "activated", "closed", "expect_deposit")

    def __init__(self, expect_deposit=False):
        self.grad = None
        self.activated = False
        self.closed = False
        self.expect_deposit = expect_deposit


_DEPOSITS = []          # ActGrad mailboxes that received a data gradient in the running backward pass
_deposit_check_queued = False


def _check_deposits_after_backward():
    """Autograd callback at the end of a backward pass: a deposit nobody collected means the disparity head of that tensor
    did not take part in this pass, and everything upstream of it received no gradient -- fail loudly instead of training
    on silent zeros.  (DepthDecoder ties its heads together with JoinHeadsFn so that this cannot happen through its
    outputs; the check covers hand-built graphs.)"""
    global _deposit_check_queued
    _deposit_check_queued = False
    left = [m for m in _DEPOSITS if m.grad is not None]
    for m in _DEPOSITS:
        m.grad = None
    _DEPOSITS.clear()
    if left:
        raise RuntimeError("ActGrad: a convolution handed its input gradient to the disparity head of the same tensor, but "
                           "that head did not run in this backward pass (loss over a subset of the decoder's scales on a "
                           "hand-built graph?); the gradients upstream of it are missing.  Set PD_ACT_FUSION=0.")


def reset_backward_state():
    """A training step begins (FusedAdam.zero_grad): forget what an ABORTED backward pass left behind.  The two end-of-
    backward callbacks above reset their own flags, but autograd skips final callbacks when a node raises -- the flags would
    then stay set for the rest of the process (no join, no deposit check ever queued again) and _DEPOSITS would keep
    gradient tensors alive."""
    global _join_queued, _deposit_check_queued
    _join_queued = False
    _deposit_check_queued = False
    for m in _DEPOSITS:
        m.grad = None
        m.closed = False
    _DEPOSITS.clear()


class JoinHeadsFn(torch.autograd.Function):
    """Identity over the disparity maps of one decoder pass.  Its node makes every head an ancestor of every returned map:
    a backward pass that starts from ANY subset of them (``outputs[("disp", 0)].sum().backward()``, a loss over fewer
    scales than the decoder has) still runs all heads -- autograd materialises the missing output gradients as zeros --
    so each head collects the data gradient the next level's first convolution deposited for it (ActGrad)."""

    @staticmethod
    def forward(ctx, *disps):
        return tuple(d.view_as(d) for d in disps)

    @staticmethod
    def backward(ctx, *grads):
        return grads


def join_heads(disps):
    return JoinHeadsFn.apply(*disps) if len(disps) > 1 else tuple(disps)


USE_ACT_FUSION = os.environ.get("PD_ACT_FUSION", "1") != "0"
USE_GT_NORMAL_CACHE = os.environ.get("PD_GT_NORMAL_CACHE", "1") != "0"
USE_EDGE_WEIGHT_CACHE = os.environ.get("PD_EDGE_WEIGHT_CACHE", "1") != "0"   # smoothness edge weights: forward -> backward
USE_SKIP_FUSION = os.environ.get("PD_SKIP_FUSION", "1") != "0"
USE_DISPHEAD_FUSED = os.environ.get("PD_DISPHEAD_FUSED", "1") != "0"   # disparity heads: data + weight gradient in one pass
USE_REFLECT_BORDER = os.environ.get("PD_REFLECT_BORDER", "1") != "0"   # reflect-conv dX = pad-1 dgrad + border strips (no fold pass)


class ChainCfg:
    __slots__ = ("stride", "pad", "relu_pre", "pool", "drop_p", "relu_post", "affine", "bn", "seed", "offset", "infer",
                 "skip_out", "skip_in")

    def __init__(self, stride=1, pad=0, relu_pre=True, pool=False, drop_p=0.0, relu_post=False, affine=None, bn=None,
                 skip_out=None, skip_in=None):
        self.stride, self.pad, self.relu_pre, self.pool = stride, pad, relu_pre, pool
        self.drop_p, self.relu_post, self.affine, self.bn = float(drop_p), relu_post, affine, bn
        self.seed = self.offset = 0
        self.infer = False
        # SkipGrad mailboxes: skip_out on the convolution whose `res` is the block input, skip_in on the convolution
        # that consumes the block input directly (only when both see the SAME tensor)
        self.skip_out, self.skip_in = (skip_out, skip_in) if USE_SKIP_FUSION else (None, None)


# ------------------------------------------------------------------ conv -> BN -> ReLU -> pool -> dropout -> +res -> ReLU
_BN_ACC = {}


def _bn_acc(dev, C):
    """fp64 accumulator of the BatchNorm partial sums (2 C sums + the ticket word behind them): zero on entry, left zero by
    pd_bn_{fwd,bwd}_finalize, so one long-lived buffer per (device, stream) replaces a memset per layer and direction."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = _BN_ACC.get(key)
    if buf is None or buf.numel() < 2 * C + 2:
        buf = torch.zeros(max(2 * C + 2, 8192), dtype=torch.float64, device=dev)
        _BN_ACC[key] = buf
    return buf


class ConvBNChainFn(torch.autograd.Function):
    """pre_encoders.ConvBlock (+ ResidualBlock add) and torchvision conv/bn/relu(/add) as one node."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, res, cfg):
        bn = cfg.bn
        dev = x.device
        N = x.shape[0]
        Co = weight.shape[0]
        if not (x.stride(1) == 1) and x.shape[1] % 4 == 0:
            x = ops.as_nhwc(x)
        training = bn.training
        # 7x7 / stride-2 / pad-3 stems run as a 4x4 / stride-1 conv over the space-to-depth input (vector gather)
        s2d = (USE_S2D_STEMS and weight.shape[2] == 7 and weight.shape[3] == 7 and cfg.stride == 2 and cfg.pad == 3
               and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)
        ctx.s2d = s2d
        alg_k = None
        if s2d:
            alg_k = 49 * x.shape[1]                               # FLOPs are counted for the 7x7 filter, not the padded 4x4x4C
            x = ops.s2d_input(x, cfg.affine)                      # saved for the weight gradient
            w_eff, stride, pad, aff = ops.s2d_weight(weight), 1, 2, None
            out_hw = (x.shape[2], x.shape[3])
        else:
            w_eff, stride, pad, aff, out_hw = weight, cfg.stride, cfg.pad, cfg.affine, None
        if getattr(cfg, "infer", False):
            return ConvBNChainFn._inference(x, w_eff, bias, gamma, beta, res, cfg, stride, pad, aff, out_hw)
        if training:
            z, part = ops.conv2d_fwd(x, w_eff, bias, stride, pad, want_stats=True, affine=aff, out_hw=out_hw, alg_k=alg_k)
        else:
            z, part = ops.conv2d_fwd(x, w_eff, bias, stride, pad, affine=aff, out_hw=out_hw, alg_k=alg_k), None
        _, _, Hz, Wz = z.shape
        scale, shift, mean, invstd = _f32(dev, Co), _f32(dev, Co), _f32(dev, Co), _f32(dev, Co)
        acc = _bn_acc(dev, Co) if training else None
        check(lib.pd_bn_fwd_finalize(ptr(part), 0 if part is None else part.shape[0], Co, float(N * Hz * Wz),
                                     ptr(gamma), ptr(beta), ptr(bn.running_mean), ptr(bn.running_var),
                                     bn.momentum, bn.eps, ptr(acc), 0 if acc is None else acc.numel(), ptr(scale), ptr(shift),
                                     ptr(mean), ptr(invstd), int(training), stream_ptr()), "pd_bn_fwd_finalize")
        Ho, Wo = (Hz // 2, Wz // 2) if cfg.pool else (Hz, Wz)
        out = ops.empty_nhwc(N, Co, Ho, Wo, dev)
        ld_res = 0
        if res is not None:
            res, ld_res = nhwc_view(res)
        drop = cfg.drop_p if training else 0.0
        if drop > 0:
            cfg.seed, cfg.offset = DropoutState.next()
        check(lib.pd_chain_fwd(ptr(z), ptr(scale), ptr(shift), ptr(res), ptr(out), N, Hz, Wz, Co, ld_res, Co,
                               int(cfg.relu_pre), int(cfg.pool), drop, cfg.seed, cfg.offset,
                               ptr(DropoutState.state(dev)) if drop > 0 else None, int(cfg.relu_post),
                               stream_ptr()), "pd_chain_fwd")
        ctx.cfg, ctx.drop, ctx.training, ctx.has_res = cfg, drop, training, res is not None
        ctx.params = (weight, bias, gamma, beta)
        ctx.save_for_backward(x, z, out, scale, shift, mean, invstd)
        return out

    @staticmethod
    def _inference(x, w, bias, gamma, beta, res, cfg, stride, pad, aff, out_hw):
        """Inference (eval mode, no autograd): the BatchNorm is folded into the conv epilogue,
        y = relu(conv * scale + (shift + bias * scale)); plain blocks are ONE kernel, blocks with a max-pool, a
        residual add or a post-add ReLU keep a (now affine-free) chain pass behind it."""
        bn = cfg.bn
        dev, N, Co = x.device, x.shape[0], w.shape[0]
        scale, shift = _f32(dev, Co), _f32(dev, Co)
        check(lib.pd_bn_fwd_finalize(None, 0, Co, 1.0, ptr(gamma), ptr(beta), ptr(bn.running_mean), ptr(bn.running_var),
                                     bn.momentum, bn.eps, None, 0, ptr(scale), ptr(shift), None, None, 0, stream_ptr()),
              "pd_bn_fwd_finalize")
        if bias is not None:
            shift = torch.addcmul(shift, bias.detach(), scale)
        act = ops.ACT_RELU if cfg.relu_pre else ops.ACT_NONE
        y = ops.conv2d_fwd(x, w, shift, stride, pad, act=act, affine=aff, out_hw=out_hw, out_scale=scale)
        if not (cfg.pool or res is not None or cfg.relu_post):
            return y
        _, _, Hz, Wz = y.shape
        Ho, Wo = (Hz // 2, Wz // 2) if cfg.pool else (Hz, Wz)
        out = ops.empty_nhwc(N, Co, Ho, Wo, dev)
        ld_res = 0
        if res is not None:
            res, ld_res = nhwc_view(res)
        check(lib.pd_chain_fwd(ptr(y), None, None, ptr(res), ptr(out), N, Hz, Wz, Co, ld_res, Co, 0, int(cfg.pool), 0.0,
                               0, 0, None, int(cfg.relu_post), stream_ptr()), "pd_chain_fwd")
        return out

    @staticmethod
    def backward(ctx, dy):
        cfg = ctx.cfg
        x, z, out, scale, shift, mean, invstd = ctx.saved_tensors
        weight, bias, gamma, beta = ctx.params
        dev = dy.device
        N, Co, Hz, Wz = z.shape
        dy, ld_dy = nhwc_view(dy)
        st = stream_ptr()
        mean_p, invstd_p, coef = (mean, invstd, _f32(dev, 2 * Co)) if ctx.training else (None, None, None)
        step_state = ptr(DropoutState.state(dev)) if ctx.drop > 0 else None
        if ctx.training:
            rows = lib.pd_chain_bwd_rows(N, Hz, Wz, Co)
            part = _f32(dev, rows, Co, 2)
            check(lib.pd_chain_bwd_reduce(ptr(dy), ld_dy, ptr(z), ptr(out), Co, ptr(scale), ptr(shift), ptr(mean),
                                          ptr(invstd), ptr(part), N, Hz, Wz, Co, int(cfg.relu_pre), int(cfg.pool),
                                          ctx.drop, cfg.seed, cfg.offset, step_state, int(cfg.relu_post), st),
                  "pd_chain_bwd_reduce")
            acc = _bn_acc(dev, Co)
            dgamma = grad_buf(gamma) if gamma is not None and gamma.requires_grad else None
            dbeta = grad_buf(beta) if beta is not None and beta.requires_grad else None
            check(lib.pd_bn_bwd_finalize(ptr(part), rows, Co, float(N * Hz * Wz), ptr(acc), acc.numel(), ptr(dgamma),
                                         ptr(dbeta), ptr(coef), 1, st), "pd_bn_bwd_finalize")
        dz = ops.empty_nhwc(N, Co, Hz, Wz, dev)
        want_dres = ctx.has_res and cfg.relu_post and ctx.needs_input_grad[5]
        dres = ops.empty_nhwc(*out.shape, dev) if want_dres else None
        check(lib.pd_chain_bwd_apply(ptr(dy), ld_dy, ptr(z), ptr(out), Co, ptr(scale), ptr(shift), ptr(mean_p),
                                     ptr(invstd_p), ptr(coef), ptr(dz), ptr(dres), N, Hz, Wz, Co, int(cfg.relu_pre),
                                     int(cfg.pool), ctx.drop, cfg.seed, cfg.offset, step_state, int(cfg.relu_post), st),
              "pd_chain_bwd_apply")
        if ctx.has_res and not cfg.relu_post and ctx.needs_input_grad[5]:
            dres = dy                     # out = f(x) + res: the residual gradient is dy itself
        if dres is not None and cfg.skip_out is not None and dres.stride(1) == 1:
            cfg.skip_out.grad = dres      # summed into the data gradient of the block's first convolution instead
            dres = None
        if weight.requires_grad:
            # the bias feeds a BatchNorm: its gradient is identically zero (mean subtraction)
            gw = grad_buf(weight)
            if ctx.s2d:
                def _stem_wgrad():
                    dw2 = ops.conv2d_wgrad(x, dz, (Co, x.shape[1], 4, 4), 1, 2, alg_k=49 * (x.shape[1] // 4))
                    ops.s2d_weight_grad(dw2, gw, accumulate=True)
                _wgrad_async(x, dz, _stem_wgrad)
            else:
                _wgrad_async(x, dz, lambda: ops.conv2d_wgrad(x, dz, weight.shape, cfg.stride, cfg.pad,
                                                             affine=cfg.affine, dw=gw, accumulate=True))
            if bias is not None and bias.requires_grad:
                grad_buf(bias)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.s2d:
                raise NotImplementedError("input gradient of a space-to-depth stem (stems read data, not activations)")
            skip = None
            if cfg.skip_in is not None:
                skip, cfg.skip_in.grad = cfg.skip_in.grad, None
                if skip is not None and tuple(skip.shape) != tuple(x.shape):
                    raise RuntimeError("SkipGrad: the skip gradient does not have the block input's shape")
            dx = ops.conv2d_dgrad(dz, weight, (x.shape[2], x.shape[3]), cfg.stride, cfg.pad, addend=skip)
        elif cfg.skip_in is not None and cfg.skip_in.grad is not None:
            raise RuntimeError("SkipGrad: a skip gradient was deposited but the block input needs no gradient")
        for p in (weight, bias, gamma, beta):
            if p is not None:
                _ready(p)
        return dx, None, None, None, None, dres, None


def conv_bn_chain(x, conv, bn, cfg, res=None, training=True):
    cfg.bn = BNParams(bn, training)
    # inference = eval mode outside autograd (decided here: inside Function.forward grad mode is always off)
    cfg.infer = (not training) and (not torch.is_grad_enabled()) and USE_BN_FOLDING
    if cfg.infer:
        return ConvBNChainFn.forward(_NoCtx(), x, conv.weight, conv.bias, bn.weight, bn.bias, res, cfg)
    return ConvBNChainFn.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, res, cfg)


class _NoCtx:
    """Stand-in for the autograd context on the inference path (nothing is saved)."""


# ------------------------------------------------------------------ decoder: reflect conv + activation
class ReflectConvActFn(torch.autograd.Function):
    """layers.Conv3x3 (ReflectionPad2d(1) + Conv2d(3)) followed by ELU (ConvBlock) or sigmoid (dispconv)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, act_mail=None, dx_mail=None):
        x = ops.as_nhwc(x)
        y = ops.conv2d_fwd(x, weight, bias, 1, 1, mode=ops.MODE_REFLECT, act=act)
        ctx.act = act
        ctx.mails = (act_mail, dx_mail)         # ActGrad of this node's output / of its input
        ctx.params = (weight, bias)
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        weight, bias = ctx.params
        N, Co, H, W = y.shape
        dy = dy.contiguous(memory_format=CL) if not (dy.is_contiguous(memory_format=CL) or Co == 1) else dy
        if Co == 1 and not dy.is_contiguous():
            dy = dy.contiguous()
        act_mail, dx_mail = ctx.mails
        pre_activated = act_mail is not None and act_mail.activated
        if pre_activated:
            act_mail.activated = False          # the consumer already multiplied by the activation derivative
        if ctx.act != ops.ACT_NONE and not pre_activated:
            dz = torch.empty_like(y)
            check(lib.pd_act_bwd(ptr(dy), ptr(y), ptr(dz), y.numel(), ctx.act, stream_ptr()), "pd_act_bwd")
        else:
            dz = dy
        if weight.requires_grad:
            db = grad_buf(bias) if bias is not None and bias.requires_grad else None
            gw = grad_buf(weight)
            _wgrad_async(x, dz, lambda: ops.conv2d_wgrad(x, dz, weight.shape, 1, 1, mode=ops.MODE_REFLECT, dw=gw,
                                                         dbias=db, accumulate=True))
        dx = None
        if ctx.needs_input_grad[0]:
            Ci = x.shape[1]
            if USE_REFLECT_BORDER and 1 < Co < 128:
                # (from 128 output channels on the fold pass over the small deep tensor is cheaper than the border kernel,
                #  whose every target pixel re-streams its slice of a multi-megabyte filter)
                # the interior of the padded-grid gradient IS the zero-padding (pad 1) data gradient: it goes straight
                # into dx; the four folded border strips are added by a kernel that touches 2(H+W) pixels per image
                dx = ops.conv2d_dgrad(dz, weight, (H, W), 1, 1)
                dzv, ld_dz = nhwc_view(dz)
                check(lib.pd_reflect_dgrad_border(ptr(dzv), ld_dz, ptr(ops.weight_cl(weight)), ptr(dx), N, H, W, Co, Ci,
                                                  stream_ptr()), "pd_reflect_dgrad_border")
            else:
                # gradient on the reflection-padded grid (a zero-pad transposed conv), then fold the border
                dxp = ops.conv2d_dgrad(dz, weight, (H + 2, W + 2), 1, 0)
                dx = ops.empty_nhwc(N, Ci, H, W, dy.device)
                check(lib.pd_reflect_fold(ptr(dxp), ptr(dx), N, H, W, Ci, stream_ptr()), "pd_reflect_fold")
            if dx_mail is not None and not dx_mail.closed:
                if dx_mail.grad is not None:
                    raise RuntimeError("ActGrad: a data gradient was deposited twice")
                dx_mail.grad, dx = dx, None     # the disparity head of the same tensor sums and activates it
                global _deposit_check_queued
                _DEPOSITS.append(dx_mail)
                if not _deposit_check_queued:
                    torch.autograd.Variable._execution_engine.queue_callback(_check_deposits_after_backward)
                    _deposit_check_queued = True
        for p in (weight, bias):
            if p is not None:
                _ready(p)
        return dx, None, None, None, None, None


class DispHeadFn(torch.autograd.Function):
    """sigmoid(Conv3x3(x)) with one output channel (depth_decoder.py:52-53,69-71): direct memory-bound kernels
    (pd_disphead_*) instead of an MFMA tile with a single useful column."""

    @staticmethod
    def forward(ctx, x, weight, bias, head_mail=None):
        x = ops.as_nhwc(x)
        N, C, H, W = x.shape
        y = torch.empty((N, 1, H, W), dtype=torch.float32, device=x.device)
        check(lib.pd_disphead_fwd(ptr(x), ptr(ops.weight_cl(weight)), ptr(bias), ptr(y), N, H, W, C, stream_ptr()),
              "pd_disphead_fwd")
        ctx.mail = head_mail                    # ActGrad of x (an ELU output with, possibly, a second consumer)
        ctx.params = (weight, bias)
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        weight, bias = ctx.params
        N, C, H, W = x.shape
        dy = dy.contiguous()
        mail = ctx.mail
        if weight.requires_grad and USE_DISPHEAD_FUSED:
            # one pass: x read once, dx written once, dw / dbias from the same folded neighbour sums
            db = grad_buf(bias) if bias is not None and bias.requires_grad else None
            gw = grad_buf(weight)
            dx = ops.empty_nhwc(N, C, H, W, x.device) if ctx.needs_input_grad[0] else None
            add, elu = None, 0
            if mail is not None and dx is not None:
                if mail.grad is not None:
                    add, mail.grad = mail.grad, None
                    if tuple(add.shape) != tuple(x.shape) or not add.is_contiguous(memory_format=CL):
                        raise RuntimeError("ActGrad: the deposited gradient does not match the head's input")
                if add is not None or not mail.expect_deposit:
                    elu, mail.activated = 1, True
                else:
                    mail.closed = True          # the other consumer has not run yet: autograd sums, pd_act_bwd activates
            ws = ops._workspace(lib.pd_disphead_workspace(C), x.device)
            check(lib.pd_disphead_bwd(ptr(dy), ptr(y), ptr(x), ptr(ops.weight_cl(weight)), ptr(add), elu, ptr(dx), ptr(gw),
                                      ptr(db), ptr(ws), ws.numel(), N, H, W, C, 1, stream_ptr()), "pd_disphead_bwd")
            for p in (weight, bias):
                if p is not None:
                    _ready(p)
            return dx, None, None, None
        if mail is not None:
            mail.closed = True                  # unfused kernels: plain route (a gradient already deposited is added below)
        if weight.requires_grad:
            db = grad_buf(bias) if bias is not None and bias.requires_grad else None
            gw = grad_buf(weight)

            def wgrad():
                nbytes = lib.pd_disphead_workspace(C)
                ws = ops._workspace(nbytes, x.device)
                check(lib.pd_disphead_bwd_weight(ptr(dy), ptr(y), ptr(x), ptr(gw), ptr(db), ptr(ws), ws.numel(), N, H, W, C,
                                                 1, stream_ptr()), "pd_disphead_bwd_weight")
            _wgrad_async(x, dy, wgrad)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.empty_nhwc(N, C, H, W, x.device)
            check(lib.pd_disphead_bwd_data(ptr(dy), ptr(y), ptr(ops.weight_cl(weight)), ptr(dx), N, H, W, C, stream_ptr()),
                  "pd_disphead_bwd_data")
            if mail is not None and mail.grad is not None:
                dx, mail.grad = dx + mail.grad, None
        elif mail is not None and mail.grad is not None:
            raise RuntimeError("ActGrad: a gradient was deposited but the head's input needs none")
        for p in (weight, bias):
            if p is not None:
                _ready(p)
        return dx, None, None, None


def reflect_conv_act(x, conv, act, act_mail=None, dx_mail=None, head_mail=None):
    """act_mail / dx_mail / head_mail: ActGrad mailboxes wired by DepthDecoder.forward (None = plain autograd route);
    act_mail belongs to this convolution's ELU output, dx_mail and head_mail to its input."""
    w = conv.weight
    if act_mail is not None and act != ops.ACT_ELU:
        raise ValueError("ActGrad hand-over is implemented for ELU outputs")
    if (act == ops.ACT_SIGMOID and w.shape[0] == 1 and w.shape[1] in (16, 32, 64, 128) and USE_DISP_HEADS
            and x.shape[2] >= 2 and x.shape[3] >= 2):
        return DispHeadFn.apply(x, w, conv.bias, head_mail)
    if head_mail is not None:
        head_mail.closed = True                 # no direct head kernel for this shape: plain route
    return ReflectConvActFn.apply(x, w, conv.bias, act, act_mail, dx_mail)


class PaddedConvFn(torch.autograd.Function):
    """ReflectionPad2d(p) / ZeroPad2d(p) + Conv2d(k, bias) for any odd k = 2p + 1 (layers.Conv5x5; not on the training
    step): general implicit-GEMM forward, data gradient on the padded grid + pd_reflect_fold_pad, general weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad, reflect, act=ops.ACT_NONE):
        x = ops.as_nhwc(x)
        mode = ops.MODE_REFLECT if reflect else ops.MODE_ZERO
        y = ops.conv2d_fwd(x, weight, bias, 1, pad, mode=mode, act=act)
        ctx.geom = (pad, mode, act)
        ctx.params = (weight, bias)
        ctx.save_for_backward(x, y if act != ops.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        weight, bias = ctx.params
        pad, mode, act = ctx.geom
        N, Ci, H, W = x.shape
        Co = weight.shape[0]
        dy = dy.contiguous(memory_format=CL) if Co > 1 else dy.contiguous()
        if act != ops.ACT_NONE:                 # the epilogue's activation (sigmoid of the uncertainty heads), through its output
            dz = torch.empty_like(dy)
            check(lib.pd_act_bwd(ptr(dy), ptr(y), ptr(dz), dy.numel(), act, stream_ptr()), "pd_act_bwd")
            dy = dz
        if weight.requires_grad:
            db = grad_buf(bias) if bias is not None and bias.requires_grad else None
            gw = grad_buf(weight)
            _wgrad_async(x, dy, lambda: ops.conv2d_wgrad(x, dy, weight.shape, 1, pad, mode=mode, dw=gw, dbias=db, accumulate=True))
        dx = None
        if ctx.needs_input_grad[0]:
            if mode == ops.MODE_ZERO:
                dx = ops.conv2d_dgrad(dy, weight, (H, W), 1, pad)
            else:
                dxp = ops.conv2d_dgrad(dy, weight, (H + 2 * pad, W + 2 * pad), 1, 0)
                dx = ops.empty_nhwc(N, Ci, H, W, dy.device)
                check(lib.pd_reflect_fold_pad(ptr(dxp), ptr(dx), N, H, W, Ci, pad, stream_ptr()), "pd_reflect_fold_pad")
        for p in (weight, bias):
            if p is not None:
                _ready(p)
        return dx, None, None, None, None, None


def padded_conv(x, conv, pad, reflect=True, act=ops.ACT_NONE):
    return PaddedConvFn.apply(x, conv.weight, conv.bias, int(pad), bool(reflect), int(act))


class UpCatFn(torch.autograd.Function):
    """cat([bilinear_x2(a), skip], 1)  (layers.upsample + depth_decoder.py:64-67)."""

    @staticmethod
    def forward(ctx, a, skip, act_mail=None, skip_mail=None):
        a = ops.as_nhwc(a)
        N, Ca, H, W = a.shape
        Cs, ld_s = 0, 0
        if skip is not None:
            skip, ld_s = nhwc_view(skip)
            Cs = skip.shape[1]
        out = ops.empty_nhwc(N, Ca + Cs, 2 * H, 2 * W, a.device)
        check(lib.pd_upcat_fwd(ptr(a), ptr(skip), ld_s, ptr(out), N, H, W, Ca, Cs, stream_ptr()), "pd_upcat_fwd")
        ctx.dims = (N, Ca, Cs, H, W)
        ctx.skip_mail = skip_mail
        ctx.mail = act_mail                     # ActGrad of a (ELU output whose only consumer this node is)
        if act_mail is not None:
            ctx.save_for_backward(a)            # the same tensor the producing convolution keeps: no extra memory
        return out

    @staticmethod
    def backward(ctx, dout):
        N, Ca, Cs, H, W = ctx.dims
        dout, ld = nhwc_view(dout)
        da = ops.empty_nhwc(N, Ca, H, W, dout.device)
        elu_y = None
        if ctx.mail is not None:
            (elu_y,) = ctx.saved_tensors
            ctx.mail.activated = True
        check(lib.pd_up_bwd_elu(ptr(dout), ld, ptr(elu_y), ptr(da), N, H, W, Ca, stream_ptr()), "pd_up_bwd")
        dskip = dout[:, Ca:] if Cs and ctx.needs_input_grad[1] else None
        smail = ctx.skip_mail
        if dskip is not None and smail is not None and USE_SKIP_FUSION and not smail.closed and smail.grad is None:
            # the skip tensor's other consumer (the ResNet max-pool) adds this gradient in its own backward kernel
            global _deposit_check_queued
            smail.grad, dskip = dskip, None
            _DEPOSITS.append(smail)
            if not _deposit_check_queued:
                torch.autograd.Variable._execution_engine.queue_callback(_check_deposits_after_backward)
                _deposit_check_queued = True
        return da, dskip, None, None


def upcat(a, skip=None, act_mail=None):
    # a skip tensor that carries a mailbox (set by the encoder: its other consumer can add the skip gradient in its own
    # backward kernel) receives the gradient by deposit instead of through autograd's accumulation pass
    skip_mail = getattr(skip, "_pd_skip_mail", None) if skip is not None and torch.is_grad_enabled() else None
    return UpCatFn.apply(a, skip, act_mail, skip_mail)


class MaxPool3s2Fn(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) (resnet_encoder.py:814)."""

    @staticmethod
    def forward(ctx, x, mail=None):
        ctx.mail = mail                         # SkipGrad of x: the gradient of x's other consumer (decoder skip)
        x = ops.as_nhwc(x)
        N, C, H, W = x.shape
        y = ops.empty_nhwc(N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, x.device)
        # the window position of every maximum (one byte per element) replaces x on the tape
        idx = torch.empty(y.numel(), dtype=torch.uint8, device=x.device) if ctx.needs_input_grad[0] else None
        check(lib.pd_maxpool3s2_fwd(ptr(x), ptr(y), ptr(idx), N, H, W, C, stream_ptr()), "pd_maxpool3s2_fwd")
        ctx.shape = (N, C, H, W)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        dy = ops.as_nhwc(dy)
        if not dy.is_contiguous(memory_format=CL):
            dy = dy.contiguous(memory_format=CL)
        dx = ops.empty_nhwc(N, C, H, W, dy.device)
        add, ld_add = None, 0
        if ctx.mail is not None and ctx.mail.grad is not None:
            add, ctx.mail.grad = ctx.mail.grad, None
            if tuple(add.shape) != (N, C, H, W) or add.stride(1) != 1 or add.stride(3) % 4:
                raise RuntimeError("SkipGrad: the deposited skip gradient does not match the pooled tensor")
            ld_add = add.stride(3)
        if ctx.mail is not None:
            ctx.mail.closed = True              # a deposit arriving after this node ran goes back to autograd
        check(lib.pd_maxpool3s2_bwd_add(ptr(idx), ptr(dy), ptr(add), ld_add, ptr(dx), N, H, W, C, stream_ptr()),
              "pd_maxpool3s2_bwd")
        return dx, None


def maxpool3s2(x, mail=None):
    return MaxPool3s2Fn.apply(x, mail)


# ------------------------------------------------------------------ plain convolution with bias (no BatchNorm)
class ConvBiasFn(torch.autograd.Function):
    """nn.Conv2d (zero padding, bias) without a normalisation layer behind it, e.g. the 1x1 q/k/v/o projections."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad):
        x = ops.as_nhwc(x)
        y = ops.conv2d_fwd(x, weight, bias, stride, pad)
        ctx.geom = (stride, pad)
        ctx.params = (weight, bias)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        weight, bias = ctx.params
        stride, pad = ctx.geom
        dy = ops.as_nhwc(dy)
        if weight.requires_grad:
            db = grad_buf(bias) if bias is not None and bias.requires_grad else None
            gw = grad_buf(weight)
            _wgrad_async(x, dy, lambda: ops.conv2d_wgrad(x, dy, weight.shape, stride, pad, dw=gw, dbias=db, accumulate=True))
        dx = ops.conv2d_dgrad(dy, weight, (x.shape[2], x.shape[3]), stride, pad) if ctx.needs_input_grad[0] else None
        for p in (weight, bias):
            if p is not None:
                _ready(p)
        return dx, None, None, None, None


def conv_bias(x, conv):
    return ConvBiasFn.apply(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0])


# ------------------------------------------------------------------ single-head self-attention (config 5 variant)
class SelfAttentionFn(torch.autograd.Function):
    """o = softmax(q k^T / sqrt(C)) v over the T = H*W tokens of each image; q, k, v: [N,C,H,W] NHWC.

    The score matrix is materialised in fp32 per image ([T,T]; 105 MB at T = 5120) and every matrix product
    runs on the fp32-MFMA GEMM kernels (exact fp32, same parity bar as the rest of the network)."""

    @staticmethod
    def forward(ctx, q, k, v):
        q, k, v = ops.as_nhwc(q), ops.as_nhwc(k), ops.as_nhwc(v)
        N, C, H, W = q.shape
        T = H * W
        scale = 1.0 / (C ** 0.5)
        qm, km, vm = (t.permute(0, 2, 3, 1).reshape(N, T, C) for t in (q, k, v))      # views of the NHWC storage
        P = torch.empty((N, T, T), dtype=torch.float32, device=q.device)
        o = ops.empty_nhwc(N, C, H, W, q.device)
        om = o.permute(0, 2, 3, 1).reshape(N, T, C)
        for n in range(N):
            ops.gemm_nt(qm[n], km[n], out=P[n])
            ops.softmax_rows_(P[n], scale)
            ops.gemm_nt(P[n], ops.transpose2d(vm[n]), out=om[n])
        ctx.scale = scale
        ctx.save_for_backward(qm, km, vm, P)
        return o

    @staticmethod
    def backward(ctx, do):
        qm, km, vm, P = ctx.saved_tensors
        N, T, C = qm.shape
        do = ops.as_nhwc(do)
        H, W = do.shape[2], do.shape[3]
        dom = do.permute(0, 2, 3, 1).reshape(N, T, C)
        dq, dk, dv = (ops.empty_nhwc(N, C, H, W, do.device) for _ in range(3))
        dqm, dkm, dvm = (t.permute(0, 2, 3, 1).reshape(N, T, C) for t in (dq, dk, dv))
        for n in range(N):
            ops.transpose2d(ops.gemm_tn(dom[n], P[n]), out=dvm[n])             # dV = P^T dO
            dP = ops.gemm_nt(dom[n], vm[n])                                    # dP = dO V^T
            ops.softmax_rows_bwd_(P[n], dP, ctx.scale)                         # dS (in place)
            ops.gemm_nt(dP, ops.transpose2d(km[n]), out=dqm[n])                # dQ = dS K
            ops.transpose2d(ops.gemm_tn(qm[n], dP), out=dkm[n])                # dK = dS^T Q
        return dq, dk, dv


class FlashAttentionFn(torch.autograd.Function):
    """The same attention as one fused kernel per direction (pd_attn_fwd / pd_attn_bwd): scores never leave the
    registers (flash-attention recurrence on the fp32 matrix cores); needs C == 128 and T % 32 == 0."""

    @staticmethod
    def forward(ctx, q, k, v):
        q, k, v = ops.as_nhwc(q), ops.as_nhwc(k), ops.as_nhwc(v)
        N, C, H, W = q.shape
        T = H * W
        scale = 1.0 / (C ** 0.5)
        o = ops.empty_nhwc(N, C, H, W, q.device)
        lse = torch.empty((N, T), dtype=torch.float32, device=q.device)
        ctx.bf16 = USE_BF16_ATTENTION
        if ctx.bf16:
            ws = ops._workspace(lib.pd_attn_bf16_workspace(N, T, C, 0), q.device)
            check(lib.pd_attn_bf16_fwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(ws), ws.numel(), N, T, C, scale,
                                       stream_ptr()), "pd_attn_bf16_fwd")
        else:
            check(lib.pd_attn_fwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), N, T, C, scale, stream_ptr()), "pd_attn_fwd")
        ctx.scale = scale
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        N, C, H, W = q.shape
        T = H * W
        do = ops.as_nhwc(do)
        dq, dk, dv = (ops.empty_nhwc(N, C, H, W, do.device) for _ in range(3))
        delta = torch.empty_like(lse)
        if ctx.bf16:
            ws = ops._workspace(lib.pd_attn_bf16_workspace(N, T, C, 1), q.device)
            args = (ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv), ptr(ws), ws.numel(),
                    N, T, C, ctx.scale)
            if USE_ATTN_BWD_STREAMS:
                # dK / dV and dQ are independent: on two streams behind the packing pass the workgroups of one kernel fill the
                # last, partly empty round of the other (2.5 and 1.25 rounds of the chip)
                main = torch.cuda.current_stream()
                side = _attn_side_stream(q.device)
                check(lib.pd_attn_bf16_bwd_parts(*args, 1, stream_ptr()), "pd_attn_bf16_bwd_parts")
                side.wait_stream(main)
                check(lib.pd_attn_bf16_bwd_parts(*args, 2, stream_ptr()), "pd_attn_bf16_bwd_parts")
                with torch.cuda.stream(side):
                    check(lib.pd_attn_bf16_bwd_parts(*args, 4, stream_ptr()), "pd_attn_bf16_bwd_parts")
                for t in (q, k, v, do, lse, delta, dq, ws):
                    t.record_stream(side)
                main.wait_stream(side)
            else:
                check(lib.pd_attn_bf16_bwd(*args, stream_ptr()), "pd_attn_bf16_bwd")
        else:
            check(lib.pd_attn_bwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(do), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv),
                                  N, T, C, ctx.scale, stream_ptr()), "pd_attn_bwd")
        return dq, dk, dv


def self_attention(q, k, v):
    if USE_FLASH_ATTENTION and q.shape[1] == 128 and (q.shape[2] * q.shape[3]) % 32 == 0:
        return FlashAttentionFn.apply(q, k, v)
    return SelfAttentionFn.apply(q, k, v)


# ------------------------------------------------------------------ multi-scale loss
class LossCfg:
    def __init__(self, scales, min_depth, max_depth, normals_loss_weight, disparity_smoothness, height, width,
                 global_norm=False, group=None):
        self.scales = list(scales)
        self.min_depth, self.max_depth = float(min_depth), float(max_depth)
        self.w_normals, self.w_smooth = float(normals_loss_weight), float(disparity_smoothness)
        self.H, self.W = int(height), int(width)
        # data parallel: normalise the masked L1 / normals terms by the mask count of the GLOBAL batch, as the
        # reference's single process does (trainer.py:1247,1308), instead of per replica
        self.global_norm, self.group = bool(global_norm), group


def exchange_loss_sums(sums, group=None):
    """All-reduce the three masked sums per scale (sum|d|m, sum(2-cos)m, sum m) over the data-parallel ranks.

    sums: fp64 [S*5] of this rank (pd_loss_finalize).  Returns (sums_value, sums_bwd): the first carries the global
    masked sums (the loss value every rank reports); the second carries sum m / world instead, so that the usual
    1/world averaging of the gradients yields exactly d/dtheta of (sum_r num_r) / (sum_r den_r).  The smoothness
    sums stay local (its mean over equal-sized replicas is what the gradient averaging already computes)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    v = sums.view(-1, 5).clone()
    if world > 1 or dist.is_initialized():
        red = v[:, :3].contiguous()
        dist.all_reduce(red, op=dist.ReduceOp.SUM, group=group)
        v[:, :3] = red
    b = v.clone()
    b[:, 2] = v[:, 2] / world
    return v.reshape(-1), b.reshape(-1)


def _iarr(vals):
    return (ctypes.c_int * len(vals))(*vals)


def _parr(tensors):
    """Host array of device pointers (NULL for None) for the multi-scale entry points."""
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


USE_MULTISCALE_LAUNCH = os.environ.get("PD_MULTISCALE_LAUNCH", "1") != "0"   # all scales of the loss per launch
SUP_BWD_TWO_PASS = os.environ.get("PD_SUP_BWD_TWO_PASS") == "1"   # per-scale backward with the [N,H,W,6] intermediate (its test assigns it)


class MultiScaleLossFn(torch.autograd.Function):
    """trainer.py:531-545 + 1126-1150,1241-1265,1298-1309 as one node.

    Inputs: gt [N,1,H,W], K [N,4,4], then per scale (disp_s [N,1,h,w], color_s [N,3,h,w]).
    Outputs: vals [1+3S] = (loss, then per scale loss/s, supervised_depth_loss/s, normals_loss/s)
    and the S full-resolution depth maps (non-differentiable by-products, ("depth",0,s)).
    """

    @staticmethod
    def forward(ctx, cfg, gt, K, *disp_color):
        S = len(cfg.scales)
        disps = [d.contiguous() for d in disp_color[:S]]
        colors = [c.contiguous() for c in disp_color[S:]]
        gt = gt.contiguous().float()
        K = K.contiguous().float()
        dev = gt.device
        N, _, H, W = gt.shape
        st = stream_ptr()
        stride = 2048
        sup_part, sm_part = _f32(dev, S, stride, 3), _f32(dev, S, stride, 2)
        sup_rows, sm_rows, dims = [], [], []
        depths, means, edge_ws = [], [], []
        with_n = 1
        # unit normals of the ground truth: once per step instead of once per scale and direction
        gtn = None
        if USE_GT_NORMAL_CACHE and S > 1:
            gtn = _f32(dev, N, H, W, 4)
            check(lib.pd_gt_normals(ptr(gt), ptr(K), ptr(gtn), N, H, W, cfg.min_depth, cfg.max_depth, st), "pd_gt_normals")
        multi = USE_MULTISCALE_LAUNCH and S <= 8 and all(H % d.shape[2] == 0 and H // d.shape[2] in (1, 2, 4, 8) and
                                                          W // d.shape[3] == H // d.shape[2] and W % d.shape[3] == 0 for d in disps)
        for i, s in enumerate(cfg.scales):
            d = disps[i]
            hs, ws = d.shape[2], d.shape[3]
            depths.append(_f32(dev, N, 1, H, W)); means.append(_f32(dev, N))
            edge_ws.append(_f32(dev, N, hs, ws, 2) if (USE_EDGE_WEIGHT_CACHE and ctx.needs_input_grad[3 + i]) else None)
            sup_rows.append(lib.pd_loss_rows(N * H * W)); sm_rows.append(lib.pd_loss_rows(N * hs * ws))
            dims += [N, hs, ws]
        if multi:
            check(lib.pd_multiscale_loss_fwd(_parr(disps), _parr(colors), _iarr([d.shape[2] for d in disps]),
                                             _iarr([d.shape[3] for d in disps]), S, ptr(gt), ptr(K), ptr(gtn), _parr(depths),
                                             _parr(means), _parr(edge_ws), ptr(sup_part), ptr(sm_part), stride, N, H, W,
                                             cfg.min_depth, cfg.max_depth, with_n, st), "pd_multiscale_loss_fwd")
        else:
            for i in range(S):
                d = disps[i]
                hs, ws = d.shape[2], d.shape[3]
                check(lib.pd_disp_to_depth(ptr(d), ptr(depths[i]), None, N, hs, ws, H, W, cfg.min_depth, cfg.max_depth, st),
                      "pd_disp_to_depth")
                check(lib.pd_sup_loss_fwd(ptr(depths[i]), ptr(gt), ptr(K), ptr(gtn), ptr(sup_part[i]), N, H, W, cfg.min_depth,
                                          cfg.max_depth, with_n, st), "pd_sup_loss_fwd")
                check(lib.pd_smooth_fwd(ptr(d), ptr(colors[i]), ptr(means[i]), ptr(sm_part[i]), ptr(edge_ws[i]), N, hs, ws, st),
                      "pd_smooth_fwd")
        ctx.multi = multi
        sums = torch.empty(S * 5, dtype=torch.float64, device=dev)
        vals = _f32(dev, 1 + 3 * S)
        check(lib.pd_loss_finalize(ptr(sup_part), _iarr(sup_rows), ptr(sm_part), _iarr(sm_rows), _iarr(dims),
                                   _iarr(cfg.scales), S, stride, cfg.w_normals, cfg.w_smooth, ptr(sums), ptr(vals), st),
              "pd_loss_finalize")
        if cfg.global_norm:       # 3 x S doubles over RCCL, then the division again on the exchanged sums
            sums_val, sums = exchange_loss_sums(sums, cfg.group)
            check(lib.pd_loss_from_sums(ptr(sums_val), _iarr(dims), _iarr(cfg.scales), S, cfg.w_normals, cfg.w_smooth,
                                        ptr(vals), st), "pd_loss_from_sums")
        ctx.cfg, ctx.S = cfg, S
        ctx.gtn = gtn
        ctx.edge_ws = edge_ws
        ctx.save_for_backward(gt, K, sums, *disps, *colors, *depths, *means)
        ctx.mark_non_differentiable(*depths)
        return (vals, *depths)

    @staticmethod
    def backward(ctx, gvals, *_unused):
        cfg, S = ctx.cfg, ctx.S
        saved = ctx.saved_tensors
        gt, K, sums = saved[:3]
        disps, colors = saved[3:3 + S], saved[3 + S:3 + 2 * S]
        depths, means = saved[3 + 2 * S:3 + 3 * S], saved[3 + 3 * S:3 + 4 * S]
        dev = gt.device
        N, _, H, W = gt.shape
        st = stream_ptr()
        gvals = gvals.contiguous().float()
        wts = _f32(dev, 3 * S)
        check(lib.pd_loss_weights(ptr(gvals), _iarr(cfg.scales), S, cfg.w_normals, cfg.w_smooth, ptr(wts), st),
              "pd_loss_weights")
        if ctx.multi and not SUP_BWD_TWO_PASS:
            grads = [torch.empty_like(d) for d in disps]
            gup = _f32(dev, S, N, H, W)
            gws = _f32(dev, sum(d.numel() for d in disps))
            gacc = torch.empty(S * N, dtype=torch.float64, device=dev)
            check(lib.pd_multiscale_loss_bwd(_parr(disps), _parr(colors), _parr(depths), _parr(means), _parr(ctx.edge_ws),
                                             _iarr([d.shape[2] for d in disps]), _iarr([d.shape[3] for d in disps]), S, ptr(gt),
                                             ptr(K), ptr(ctx.gtn), ptr(wts), ptr(sums), ptr(gup), ptr(gws), ptr(gacc),
                                             _parr(grads), N, H, W, cfg.min_depth, cfg.max_depth, st), "pd_multiscale_loss_bwd")
            return (None, None, None, *grads, *([None] * S))
        # (the [N,H,W,6] intermediate of the two-pass form; the fused kernel keeps it in LDS)
        ab = _f32(dev, N, H, W, 6) if SUP_BWD_TWO_PASS else None
        gup = _f32(dev, N, H, W)
        grads = []
        for i in range(S):
            d = disps[i]
            hs, ws = d.shape[2], d.shape[3]
            check(lib.pd_sup_loss_bwd(ptr(depths[i]), ptr(gt), ptr(K), ptr(ctx.gtn), ptr(wts[3 * i:]), ptr(sums[5 * i:]), ptr(ab),
                                      ptr(gup), N, H, W, cfg.min_depth, cfg.max_depth, 1, 1, int(SUP_BWD_TWO_PASS), st), "pd_sup_loss_bwd")
            gd = torch.empty_like(d)
            check(lib.pd_up_gather_bwd(ptr(gup), ptr(gd), N, hs, ws, H, W, 0, 0, st), "pd_up_gather_bwd")
            gws = _f32(dev, N, hs, ws)
            gacc = torch.empty(N, dtype=torch.float64, device=dev)
            check(lib.pd_smooth_bwd(ptr(d), ptr(colors[i]), ptr(means[i]), ptr(wts[3 * i:]), ptr(ctx.edge_ws[i]), ptr(gws), ptr(gacc),
                                    ptr(gd), N, hs, ws, 1, st), "pd_smooth_bwd")
            grads.append(gd)
        return (None, None, None, *grads, *([None] * S))


class NormalsPredLossFn(torch.autograd.Function):
    """sum((2 - cos(n_pred, n_gt)) m) / sum(m) for a PREDICTED normal map [N,3,H,W] (the `arch1++_separate_normals_dec`
    variant, README.md:54; formula of trainer.py:1298-1309, mask of trainer.py:1242-1243): pd_gt_normals +
    pd_normals_pred_loss_fwd / _bwd."""

    @staticmethod
    def forward(ctx, pred, gt, K, min_depth, max_depth):
        pred, ld = nhwc_view(pred)
        N, C, H, W = pred.shape
        if C != 3:
            raise ValueError(f"normals_pred_loss: the prediction must have 3 channels, got {C}")
        gt = gt.contiguous().float()
        K = K.contiguous().float()
        dev = pred.device
        st = stream_ptr()
        gtn = _f32(dev, N, H, W, 4)
        check(lib.pd_gt_normals(ptr(gt), ptr(K), ptr(gtn), N, H, W, float(min_depth), float(max_depth), st), "pd_gt_normals")
        part = _f32(dev, lib.pd_loss_rows(N * H * W), 2)
        out = _f32(dev, 2)
        check(lib.pd_normals_pred_loss_fwd(ptr(pred), ld, ptr(gtn), ptr(gt), ptr(part), ptr(out), N, H, W, float(min_depth),
                                           float(max_depth), st), "pd_normals_pred_loss_fwd")
        ctx.geom = (ld, float(min_depth), float(max_depth))
        ctx.save_for_backward(pred, gtn, gt, out)
        return out[:1].clone().reshape(())

    @staticmethod
    def backward(ctx, g):
        pred, gtn, gt, out = ctx.saved_tensors
        ld, min_d, max_d = ctx.geom
        N, _, H, W = pred.shape
        gout = g.reshape(1).contiguous().float()
        dpred = ops.empty_nhwc(N, 3, H, W, pred.device)
        check(lib.pd_normals_pred_loss_bwd(ptr(pred), ld, ptr(gtn), ptr(gt), ptr(gout), ptr(out), ptr(dpred), 3, N, H, W, min_d,
                                           max_d, stream_ptr()), "pd_normals_pred_loss_bwd")
        return dpred, None, None, None, None


def normals_pred_loss(pred, gt, K, min_depth, max_depth):
    return NormalsPredLossFn.apply(pred, gt, K, min_depth, max_depth)


def multiscale_loss(cfg, gt, K, disps, colors):
    out = MultiScaleLossFn.apply(cfg, gt, K, *disps, *colors)
    return out[0], list(out[1:])


# ------------------------------------------------------------------ layers.get_smooth_loss as a function of its own
class SmoothLossFn(torch.autograd.Function):
    """layers.py:452-465 on the K5 kernels: mean |dx disp| e^{-|dx I|} + mean |dy disp| e^{-|dy I|} of the disparity AS GIVEN
    (pd_smooth_fwd / pd_smooth_bwd with mean = NULL: no mean normalisation -- the caller of the reference function applies
    it first, trainer.py:1256-1258).  Differentiable in disp; the image is data."""

    @staticmethod
    def forward(ctx, disp, img):
        N, _, h, w = disp.shape
        dev = disp.device
        st = stream_ptr()
        rows = lib.pd_loss_rows(N * h * w)
        part = _f32(dev, rows, 2)
        edge = _f32(dev, N, h, w, 2) if disp.requires_grad else None
        check(lib.pd_smooth_fwd(ptr(disp), ptr(img), None, ptr(part), ptr(edge), N, h, w, st), "pd_smooth_fwd")
        sx, sy = part.double().sum(0)                      # ordered fp64 sum of the per-workgroup partials
        ctx.save_for_backward(disp, img, edge)
        return (sx / (N * h * (w - 1)) + sy / (N * (h - 1) * w)).float()

    @staticmethod
    def backward(ctx, g):
        disp, img, edge = ctx.saved_tensors
        N, _, h, w = disp.shape
        dev = disp.device
        wts = torch.zeros(3, dtype=torch.float32, device=dev)
        wts[2] = g.float()
        gws = _f32(dev, N, h, w)
        gacc = torch.empty(N, dtype=torch.float64, device=dev)
        gd = torch.empty_like(disp)
        check(lib.pd_smooth_bwd(ptr(disp), ptr(img), None, ptr(wts), ptr(edge), ptr(gws), ptr(gacc), ptr(gd), N, h, w, 0,
                                stream_ptr()), "pd_smooth_bwd")
        return gd, None


def smooth_loss(disp, img):
    if not (disp.is_cuda and img.is_cuda):
        raise RuntimeError("get_smooth_loss runs on the MI355X; there is no CPU fallback")
    if disp.shape[1] != 1 or img.shape[1] != 3 or disp.shape[2] < 2 or disp.shape[3] < 2:
        raise ValueError(f"get_smooth_loss: disp [N,1,h,w] and img [N,3,h,w] expected, got {tuple(disp.shape)} / {tuple(img.shape)}")
    return SmoothLossFn.apply(disp.float().contiguous(), img.float().contiguous())


# ------------------------------------------------------------------ DPT decoder glue (manydepth/dpt/blocks.py; --train_dpt)
class ConvBiasActFn(torch.autograd.Function):
    """nn.Conv2d (zero padding, bias) with the activation of the NEXT module in its epilogue (ReLU: the residual units of
    the DPT fusion blocks, blocks.py:289-300); backward through the activation output (pd_act_bwd)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act):
        x = ops.as_nhwc(x)
        y = ops.conv2d_fwd(x, weight, bias, stride, pad, act=act)
        ctx.geom = (stride, pad, act)
        ctx.params = (weight, bias)
        ctx.save_for_backward(x, y if act != ops.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        weight, bias = ctx.params
        stride, pad, act = ctx.geom
        dy = ops.as_nhwc(dy)
        if act != ops.ACT_NONE:
            dz = torch.empty_like(dy)
            check(lib.pd_act_bwd(ptr(dy), ptr(y), ptr(dz), dy.numel(), act, stream_ptr()), "pd_act_bwd")
            dy = dz
        if weight.requires_grad:
            db = grad_buf(bias) if bias is not None and bias.requires_grad else None
            gw = grad_buf(weight)
            _wgrad_async(x, dy, lambda: ops.conv2d_wgrad(x, dy, weight.shape, stride, pad, dw=gw, dbias=db, accumulate=True))
        dx = ops.conv2d_dgrad(dy, weight, (x.shape[2], x.shape[3]), stride, pad) if ctx.needs_input_grad[0] else None
        for p in (weight, bias):
            if p is not None:
                _ready(p)
        return dx, None, None, None, None, None


def conv_bias_act(x, conv, act=ops.ACT_NONE):
    return ConvBiasActFn.apply(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], int(act))


class ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = ops.as_nhwc(x)
        y = torch.empty_like(x)
        check(lib.pd_relu_add(ptr(x), None, ptr(y), x.numel(), 1, stream_ptr()), "pd_relu_add")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = ops.as_nhwc(dy)
        dz = torch.empty_like(dy)
        check(lib.pd_act_bwd(ptr(dy), ptr(y), ptr(dz), dy.numel(), ops.ACT_RELU, stream_ptr()), "pd_act_bwd")
        return dz


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = ops.as_nhwc(a), ops.as_nhwc(b)
        y = torch.empty_like(a)
        check(lib.pd_relu_add(ptr(a), ptr(b), ptr(y), a.numel(), 0, stream_ptr()), "pd_relu_add")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class Upsample2xAlignedFn(torch.autograd.Function):
    """F.interpolate(scale_factor=2, mode="bilinear", align_corners=True) (blocks.py:138-172, 375-377)."""

    @staticmethod
    def forward(ctx, x):
        x = ops.as_nhwc(x)
        N, C, H, W = x.shape
        y = ops.empty_nhwc(N, C, 2 * H, 2 * W, x.device)
        check(lib.pd_up2x_ac_fwd(ptr(x), ptr(y), N, H, W, C, stream_ptr()), "pd_up2x_ac_fwd")
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W = ctx.shape
        dy = ops.as_nhwc(dy)
        dx = ops.empty_nhwc(N, C, H, W, dy.device)
        check(lib.pd_up2x_ac_bwd(ptr(dy), ptr(dx), N, H, W, C, stream_ptr()), "pd_up2x_ac_bwd")
        return dx


def _need_cuda4(x, what):
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 4 and x.shape[1] % 4 == 0):
        raise RuntimeError(f"{what} needs a CUDA(HIP) NCHW tensor with a multiple of 4 channels; there is no CPU fallback")
    return x.float()


def relu(x):
    return ReluFn.apply(_need_cuda4(x, "relu"))


def add(a, b):
    return AddFn.apply(_need_cuda4(a, "add"), _need_cuda4(b, "add"))


def upsample2x_aligned(x):
    return Upsample2xAlignedFn.apply(_need_cuda4(x, "upsample2x_aligned"))
