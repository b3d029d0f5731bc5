"""CPU oracle for the polarimetric depth hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (NumPy / SciPy-free
NumPy / plain PyTorch-CPU fp32) of the reference algorithm on the hot path that
``BASELINE.json:north_star`` names.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it, and there only as the
checker -- never as the thing measured as the product or shipped.  The product
path (``supervised-depth-estimation-from-polarized-images_amd/``) never imports
``oracle`` and fails loudly when the HIP library is missing.

Pinning: the reference has no tests or golden vectors of its own.
The restatement is pinned by fixtures under ``tests/golden/`` that were generated
by importing the reference's own Python in the build container
(``tests/golden/make_golden.py``, which needs ``/root/reference``); the fixtures
are data only.  Pieces whose arithmetic lives in third-party code that is absent
from the container (torchvision resnet18, kornia ``depth_to_normals``) are
restated from their published algorithm and are marked "parity unpinned" where
they are defined.

Every function cites the reference ``file:line`` it follows (paths relative to
the reference checkout).
"""
