"""Command-line contract of the reference (manydepth/options.py:13-380), flag for flag.

Same names, types and defaults -- including the ``type=bool`` quirk (any non-empty string is
True, options.py:33-86,205) -- so that train_supervised_GT.sh works unchanged.  Declared as a
table instead of ~80 add_argument calls.
"""
import argparse
import os

_B, _I, _F, _S = bool, int, float, str
_SPLITS = ["eigen_zhou", "eigen_full", "odom", "benchmark", "cityscapes_preprocessed", "ppp", "HAMMER",
           "HAMMER_unseen", "eccv_depth_stereo"]
_DATASETS = ["kitti", "kitti_odom", "kitti_depth", "kitti_test", "cityscapes_preprocessed", "ppp", "HAMMER",
             "eccv_depth_stereo"]
_EVAL_SPLITS = ["eigen", "eigen_benchmark", "benchmark", "odom_9", "odom_10", "cityscapes", "ppp", "HAMMER",
                "HAMMER_unseen", "eccv_depth_stereo"]

# (flag, kwargs).  'flag' entries are store_true switches.
_TYPED = [
    ("data_path", dict(type=_S, default=os.path.join("/media/patrick/LargeSSD/Datasets", "kitti"))),
    ("data_path_val", dict(type=_S, default=os.path.join("/media/patrick/LargeSSD/Datasets", "kitti"))),
    ("log_dir", dict(type=_S, default=os.path.join("/media/patrick/LargeSSD/experiments", "manyattention2"))),
    ("use_polyaxon", dict(type=_B, default=False)), ("res_pose", dict(type=_B, default=False)),
    ("train_stereo_only", dict(type=_B, default=False)), ("use_stereo", dict(type=_B, default=False)),
    ("depth_supervision_only", dict(type=_B, default=False)), ("train_dpt", dict(type=_B, default=False)),
    ("midas", dict(type=_B, default=False)), ("depth_supervision", dict(type=_B, default=False)),
    ("train_student", dict(type=_B, default=False)), ("pose_input", dict(type=_B, default=False)),
    ("supervise_pose", dict(type=_B, default=False)), ("max_disparity", dict(type=_I, default=192)),
    ("post_process_mono_while_training", dict(type=_B, default=False)),
    ("motion_masking_begin", dict(type=_I, default=0)), ("motion_masking_end", dict(type=_I, default=25)),
    ("motion_masking_mask", dict(type=_I, default=0)), ("pose_attach_epoch", dict(type=_I, default=30)),
    ("offset", dict(type=_I, default=1)), ("modality", dict(type=_S, default="polarization")),
    ("depth_modality", dict(type=_S, default="_gt")), ("model_name", dict(type=_S, default="mdp")),
    ("split", dict(type=_S, choices=_SPLITS, default="eigen_zhou")),
    ("num_layers", dict(type=_I, default=18, choices=[18, 34, 50, 101, 152])),
    ("depth_binning", dict(type=_S, choices=["linear", "inverse"], default="linear")),
    ("num_depth_bins", dict(type=_I, default=96)),
    ("dataset", dict(type=_S, default="kitti", choices=_DATASETS)),
    ("height", dict(type=_I, default=192)), ("width", dict(type=_I, default=640)),
    ("disparity_smoothness", dict(type=_F, default=1e-3)),
    ("scales", dict(nargs="+", type=_I, default=[0, 1, 2, 3])),
    ("min_depth", dict(type=_F, default=0.1)), ("max_depth", dict(type=_F, default=2.0)),
    ("frame_ids", dict(nargs="+", type=_I, default=[0, -1, 1])),
    ("overfit", dict(type=_B, default=False)), ("overfit_scene", dict(type=_S, default="scene2_traj1_2")),
    ("batch_size", dict(type=_I, default=12)), ("learning_rate", dict(type=_F, default=1e-4)),
    ("num_epochs", dict(type=_I, default=30)), ("scheduler_step_size", dict(type=_I, default=10)),
    ("freeze_teacher_epoch", dict(type=_I, default=20)), ("normals_loss_weight", dict(type=_F, default=0.35)),
    ("weights_init", dict(type=_S, default="pretrained", choices=["pretrained", "scratch"])),
    ("num_matching_frames", dict(type=_I, default=1)), ("dropout_rate", dict(type=_F, default=0.1)),
    ("num_workers", dict(type=_I, default=12)), ("load_weights_folder", dict(type=_S)),
    ("mono_weights_folder", dict(type=_S)),
    ("models_to_load", dict(nargs="+", type=_S, default=["depth", "pose_encoder", "pose"])),
    ("log_frequency", dict(type=_I, default=500)), ("save_frequency", dict(type=_I, default=2)),
    ("pred_depth_scale_factor", dict(type=_F, default=1)), ("ext_disp_to_eval", dict(type=_S)),
    ("eval_split", dict(type=_S, default="eigen", choices=_EVAL_SPLITS)), ("eval_out_dir", dict(type=_S)),
]
_SWITCHES = ["png", "freeze_teacher_and_pose", "v1_multiscale", "avg_reprojection", "disable_automasking", "no_ssim",
             "use_future_frame", "disable_motion_masking", "no_matching_augmentation", "augment_normals",
             "augment_xolp", "no_cuda", "eval_stereo", "eval_mono", "disable_median_scaling", "save_pred_disps",
             "no_eval", "eval_eigen_to_benchmark", "post_process", "zero_cost_volume", "static_camera", "eval_teacher"]


class MonodepthOptions:
    def __init__(self):
        self.parser = argparse.ArgumentParser(description="ManyDepth options")
        for name, kw in _TYPED:
            self.parser.add_argument("--" + name, **kw)
        for name in _SWITCHES:
            self.parser.add_argument("--" + name, action="store_true")

    def parse(self, args=None):
        self.options = self.parser.parse_args(args)
        return self.options
