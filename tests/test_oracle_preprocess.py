"""Pins oracle/preprocess_ref.py (the restatement of Pillow's 8-bit bilinear resampling + the reference's pad-to-square geometry) against
Pillow itself and against torch's float32 arithmetic for ToTensor + Normalize.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import preprocess_ref as ref

PIL = pytest.importorskip("PIL")
from PIL import Image, ImageOps  # noqa: E402


def _pil_pipeline(arr: np.ndarray, size: int) -> np.ndarray:
    """What dataset/transforms.py:336-360 does, written against Pillow directly (the committed golden fixture comes from the reference's own class)."""
    image = Image.fromarray(arr)
    width, height = image.size
    scale_factor = size / max(width, height)
    nw, nh = int(width * scale_factor), int(height * scale_factor)
    image = image.resize((nw, nh), Image.BILINEAR)
    pw, ph = (size - nw) // 2, (size - nh) // 2
    return np.asarray(ImageOps.expand(image, (pw, ph, size - nw - pw, size - nh - ph), fill=(0, 0, 0)))


GEOMS = [(500, 375, 224), (375, 500, 224), (224, 224, 224), (224, 100, 224), (64, 48, 224), (1, 1, 32), (2, 7, 16), (1023, 517, 224), (49, 49, 224),
         (3000, 200, 224), (97, 4001, 96), (640, 480, 384), (31, 33, 64), (256, 255, 128), (7, 5, 224)]


@pytest.mark.parametrize("w,h,size", GEOMS)
def test_resize_and_padding_equals_pillow(w, h, size):
    rng = np.random.default_rng(w * 7919 + h)
    arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if (w + h) % 3 == 0:                         # saturated blocks exercise the clip
        arr[: h // 2, : w // 2] = 255
        arr[h // 2:, w // 2:] = 0
    np.testing.assert_array_equal(ref.resize_and_padding(arr, size), _pil_pipeline(arr, size))


def test_random_geometries_equal_pillow():
    rng = np.random.default_rng(0)
    for _ in range(40):
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 700))
        size = int(rng.choice([32, 96, 224]))
        nw, nh, _, _ = ref.output_geometry(w, h, size)
        arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if nw == 0 or nh == 0:
            with pytest.raises(ValueError):
                ref.resize_and_padding(arr, size)
            with pytest.raises(ValueError):
                _pil_pipeline(arr, size)
            continue
        np.testing.assert_array_equal(ref.resize_and_padding(arr, size), _pil_pipeline(arr, size), err_msg=f"{w}x{h}->{size}")


def test_long_side_can_truncate_below_size():
    """int(width * (size / width)) is not always `size` in double arithmetic: the geometry must follow Python's float expression, not the ideal."""
    hits = [(w, ref.output_geometry(w, w, 224)[0]) for w in range(1, 3000)]
    short = [w for w, nw in hits if nw != 224]
    assert short, "expected at least one width whose long side truncates to size-1"
    w = short[0]
    arr = np.random.default_rng(1).integers(0, 256, (w, w, 3), dtype=np.uint8)
    np.testing.assert_array_equal(ref.resize_and_padding(arr, 224), _pil_pipeline(arr, 224))


def test_to_tensor_normalize_equals_torch_float32():
    rng = np.random.default_rng(2)
    u8 = rng.integers(0, 256, (17, 19, 3), dtype=np.uint8)
    u8.reshape(-1)[:256] = np.arange(256, dtype=np.uint8)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    t = torch.from_numpy(u8).permute(2, 0, 1).contiguous().to(torch.float32).div(255)            # torchvision F.to_tensor
    t.sub_(torch.as_tensor(mean, dtype=torch.float32)[:, None, None]).div_(torch.as_tensor(std, dtype=torch.float32)[:, None, None])   # F.normalize
    got = ref.to_tensor_normalize(u8, mean, std)
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), t.numpy().view(np.uint32))
