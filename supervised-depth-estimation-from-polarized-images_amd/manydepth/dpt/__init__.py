"""manydepth.dpt facade: the convolutional decoder blocks of the vendored DPT / MiDaS model (reference manydepth/dpt/blocks.py)
on the HIP kernels.  The transformer backbone of DPTDepthModel (dpt/models.py:89, dpt/vit.py) lives in timm 0.5.4, which is
not part of the reference checkout's importable code and not in this image: only what the reference itself defines -- the
fusion blocks, residual units and Interpolate -- is built and pinned (tests/golden/g8_dpt_fusion.npz)."""
from .blocks import FeatureFusionBlock_custom, Interpolate, ResidualConvUnit_custom  # noqa: F401
