"""ctypes loader for libpolardepth.so (C ABI in include/polardepth.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PD_LIB: another build of the same ABI (the timing-probe libraries of tools/build_probe.sh); read once, at import
_SO = os.environ.get("PD_LIB") or os.path.join(_HERE, "libpolardepth.so")


class LibraryMissing(RuntimeError):
    pass


class PolarDepthError(RuntimeError):
    pass


_c = ctypes
_vp, _i, _sz, _dbl = _c.c_void_p, _c.c_int, _c.c_size_t, _c.c_double
_dp = _c.POINTER(_c.c_double)
_l, _f = _c.c_long, _c.c_float
_u64 = _c.c_uint64
_ip = _c.POINTER(_c.c_int)
_u = _c.c_uint
_vpp = _c.POINTER(_c.c_void_p)

# name -> (restype, argtypes); mirrors include/polardepth.h one to one
SIGNATURES = {
    "pd_last_error": (_c.c_char_p, []),
    "pd_version": (_i, []),
    "pd_polar_tables_bytes": (_sz, [_i, _i, _i]),
    "pd_polar_tables_pack": (_i, [_dp, _dp, _i, _dp, _dp, _i, _dp, _dp, _i, _vp, _sz]),
    "pd_polar_tables_build": (_i, [_dbl, _vp, _sz, _c.POINTER(_sz)]),
    "pd_polar_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _vp]),
    "pd_polar_normals_from_xolp": (_i, [_vp, _vp, _vp, _sz, _i, _i, _i, _i, _vp]),
    "pd_polar_theta": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _l, _vp]),
    "pd_polar_calc_normals": (_i, [_vp, _vp, _vp, _i, _l, _i, _i, _vp]),
    "pd_conv2d_tile_m": (_i, [_l, _i]),
    "pd_conv2d_stats_rows": (_l, [_l, _i]),
    "pd_conv2d_uses_x3": (_i, [_l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _u]),
    "pd_conv2d_wgrad_uses_x3": (_i, [_l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _u]),
    "pd_conv2d": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                       _i, _f, _f, _l, _u, _vp]),
    "pd_conv16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _l, _l, _i, _i, _i, _l, _i, _i, _vp]),
    "pd_conv16_wgrad_workspace": (_sz, [_i]),
    "pd_conv16_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _l, _l, _l, _l, _i, _vp]),
    "pd_dgrad_s2_filters": (_i, [_vp, _vp, _i, _i, _vp]),
    "pd_conv2d_rect": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _i, _l, _u, _vp]),
    "pd_interleave4": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "pd_conv2d_add": (_i, [_vp, _vp, _vp, _l, _vp, _i, _i, _i, _i, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _i, _l, _u, _vp]),
    "pd_conv2d_wgrad_workspace": (_sz, [_l, _i, _i, _u]),
    "pd_conv2d_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i,
                             _i, _i, _f, _f, _l, _i, _u, _vp]),
    "pd_weight_transpose": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "pd_weight_transpose_batched": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "pd_stem_s2d_input": (_i, [_vp, _vp, _i, _i, _i, _i, _l, _l, _l, _l, _i, _f, _f, _vp]),
    "pd_stem_s2d_weight": (_i, [_vp, _vp, _i, _i, _vp]),
    "pd_stem_s2d_weight_grad": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "pd_bn_fwd_finalize": (_i, [_vp, _l, _i, _dbl, _vp, _vp, _vp, _vp, _f, _f, _vp, _l, _vp, _vp, _vp, _vp, _i, _vp]),
    "pd_bn_bwd_finalize": (_i, [_vp, _l, _i, _dbl, _vp, _l, _vp, _vp, _vp, _i, _vp]),
    "pd_chain_bwd_rows": (_l, [_i, _i, _i, _i]),
    "pd_chain_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _l, _l, _i, _i, _f, _u64, _u64, _vp, _i, _vp]),
    "pd_chain_bwd_reduce": (_i, [_vp, _l, _vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _u64,
                                 _u64, _vp, _i, _vp]),
    "pd_chain_bwd_apply": (_i, [_vp, _l, _vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f,
                                _u64, _u64, _vp, _i, _vp]),
    "pd_maxpool3s2_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pd_maxpool3s2_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pd_maxpool3s2_bwd_add": (_i, [_vp, _vp, _vp, _l, _vp, _i, _i, _i, _i, _vp]),
    "pd_upcat_fwd": (_i, [_vp, _vp, _l, _vp, _i, _i, _i, _i, _i, _vp]),
    "pd_up_bwd": (_i, [_vp, _l, _vp, _i, _i, _i, _i, _vp]),
    "pd_up_bwd_elu": (_i, [_vp, _l, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pd_act_bwd": (_i, [_vp, _vp, _vp, _l, _i, _vp]),
    "pd_up2x_ac_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "pd_up2x_ac_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "pd_relu_add": (_i, [_vp, _vp, _vp, _l, _i, _vp]),
    "pd_reflect_fold": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "pd_reflect_fold_pad": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "pd_reflect_dgrad_border": (_i, [_vp, _l, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "pd_adam_step": (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _l, _vp, _f, _i, _vp]),
    "pd_step_tick": (_i, [_vp, _i, _i, _vp]),
    "pd_step_set_hyper": (_i, [_vp, _f, _f, _vp]),
    "pd_loss_rows": (_i, [_l]),
    "pd_disp_to_depth": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp]),
    "pd_up_gather_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "pd_gt_normals": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _vp]),
    "pd_sup_loss_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp]),
    "pd_normals_loss_masked": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "pd_normals_pred_loss_fwd": (_i, [_vp, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp]),
    "pd_normals_pred_loss_bwd": (_i, [_vp, _l, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _f, _f, _vp]),
    "pd_multiscale_loss_fwd": (_i, [_vpp, _vpp, _ip, _ip, _i, _vp, _vp, _vp, _vpp, _vpp, _vpp, _vp, _vp, _i, _i, _i, _i, _f, _f,
                                    _i, _vp]),
    "pd_multiscale_loss_bwd": (_i, [_vpp, _vpp, _vpp, _vpp, _vpp, _ip, _ip, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vpp,
                                    _i, _i, _i, _f, _f, _vp]),
    "pd_sup_loss_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _i, _i, _vp]),
    "pd_smooth_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "pd_smooth_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pd_loss_finalize": (_i, [_vp, _ip, _vp, _ip, _ip, _ip, _i, _i, _f, _f, _vp, _vp, _vp]),
    "pd_loss_from_sums": (_i, [_vp, _ip, _ip, _i, _f, _f, _vp, _vp]),
    "pd_loss_weights": (_i, [_vp, _ip, _i, _f, _f, _vp, _vp]),
    "pd_softmax_rows_fwd": (_i, [_vp, _l, _l, _f, _vp]),
    "pd_softmax_rows_bwd": (_i, [_vp, _vp, _l, _l, _f, _vp]),
    "pd_resize_u8_pass": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "pd_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "pd_attn_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "pd_attn_bf16_workspace": (_sz, [_i, _i, _i, _i]),
    "pd_attn_bf16_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _f, _vp]),
    "pd_attn_bf16_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _f, _vp]),
    "pd_attn_bf16_bwd_parts": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _f, _i, _vp]),
    "pd_disphead_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pd_disphead_bwd_data": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "pd_disphead_workspace": (_sz, [_i]),
    "pd_disphead_bwd_weight": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    "pd_disphead_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    "pd_ssim_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "pd_ssim_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "pd_depth_metrics": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _l, _f, _f, _vp]),
}


class _Lib:
    """Lazy handle; attribute access loads the library or raises LibraryMissing."""

    def __init__(self):
        self._h = None

    def _load(self):
        if self._h is None:
            if not os.path.exists(_SO):
                raise LibraryMissing(
                    f"{_SO} not found: build it with `make -C {os.path.join(os.path.dirname(_HERE), 'csrc')}` "
                    "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
            # torch bundles its own libamdhip64; it must be the HIP runtime of this process, so that
            # torch's streams / allocations and this library's kernels live in one runtime.
            import torch  # noqa: F401
            h = ctypes.CDLL(_SO)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(h, name)
                fn.restype = res
                fn.argtypes = args
            self._h = h
        return self._h

    def __getattr__(self, name):
        return getattr(self._load(), name)

    @property
    def path(self):
        return _SO


lib = _Lib()


def check(rc, what=""):
    if rc != 0:
        msg = lib.pd_last_error()
        raise PolarDepthError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device/host pointer of a tensor or None."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
