"""polardepth -- MI355X-native engine behind the manydepth / polarisation façade.

Thin host layer over ``libpolardepth.so`` (hand-written HIP for gfx950, C ABI declared in
``include/polardepth.h``).  PyTorch is used for device memory, streams and
``torch.distributed`` only.  There is NO CPU fallback: every op raises if the HIP
library is missing or the tensors are not on the GPU.
"""
from ._lib import lib, LibraryMissing, check  # noqa: F401
