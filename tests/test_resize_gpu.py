"""Device LANCZOS resize of uint8 planes vs PIL (Image.resize(..., Image.LANCZOS) = the reference's Image.ANTIALIAS,
indoor_dataset.py:335-349): bit-exact."""
import numpy as np
import pytest
import torch

from polardepth import resize as pdresize

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(832, 1088, 512, 612), (100, 130, 64, 96), (64, 96, 100, 130), (300, 200, 77, 311),
                                   (512, 612, 512, 612), (480, 640, 480, 320)])
def test_resize_matches_pillow_bit_exactly(shape):
    from PIL import Image
    Hs, Ws, Hd, Wd = shape
    rng = np.random.default_rng(sum(shape))
    planes = rng.integers(0, 256, (2, 4, Hs, Ws), dtype=np.uint8)
    planes[0, 0] = 255 * (rng.random((Hs, Ws)) > 0.5)          # hard edges: ringing must clip like Pillow
    got = pdresize.resize_lanczos_u8(torch.from_numpy(planes).cuda(), (Hd, Wd)).cpu().numpy()
    assert got.shape == (2, 4, Hd, Wd)
    for b in range(2):
        for c in range(4):
            ref = np.asarray(Image.fromarray(planes[b, c], "L").resize((Wd, Hd), Image.LANCZOS))
            np.testing.assert_array_equal(got[b, c], ref)


def test_resize_then_k1_equals_host_resize_then_k1():
    """The device path of the loader hand-over: raw planes -> resize -> K1 equals PIL resize -> K1."""
    from PIL import Image
    from polardepth import polar as pdpolar
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 256, (1, 4, 208, 272), dtype=np.uint8)
    host = np.stack([np.asarray(Image.fromarray(raw[0, c], "L").resize((96, 64), Image.LANCZOS)) for c in range(4)])[None]
    dev = pdresize.resize_lanczos_u8(torch.from_numpy(raw).cuda(), (64, 96))
    a = pdpolar.polar_forward(dev, want=("xolp", "normals"))
    b = pdpolar.polar_forward(torch.from_numpy(host).cuda(), want=("xolp", "normals"))
    assert torch.equal(a["xolp"], b["xolp"]) and torch.equal(a["normals"], b["normals"])
