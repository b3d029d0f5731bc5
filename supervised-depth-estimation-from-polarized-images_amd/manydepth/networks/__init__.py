"""Network classes re-exported like reference manydepth/networks/__init__.py:2-6 (hot-path subset).

ResnetEncoder / ResnetEncoderMatching / PoseDecoder / PoseCNN belong to the self-supervised
multi-frame path that ``--depth_supervision_only`` disables (trainer.py:222); they are placeholders
that raise on construction.
"""
from .resnet_encoder import ShallowResnetEncoder
from .pre_encoders import ShallowEncoder, ShallowNormalsEncoder, JointEncoder
from .depth_decoder import DepthDecoder
from .normals_decoder import NormalsDecoder      # `arch1++_separate_normals_dec` variant (README.md:54)


def _out_of_scope(name):
    class _Placeholder:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"networks.{name} is part of the self-supervised / multi-frame path, "
                                      "which is outside the MI355X hot path of this build")
    _Placeholder.__name__ = name
    return _Placeholder


ResnetEncoder = _out_of_scope("ResnetEncoder")
ResnetEncoderMatching = _out_of_scope("ResnetEncoderMatching")
PoseDecoder = _out_of_scope("PoseDecoder")
PoseCNN = _out_of_scope("PoseCNN")
