"""TEST INFRASTRUCTURE — CPU restatement of the reference's deterministic validation-time input pipeline:

    resize_and_padding(size, training=False)  ->  to_tensor  ->  normalize(mean, std)
    (configs/classification/pet.yaml:94-101, configs/faceX/cbir.yaml:92-99; dataset/transforms.py:325-362, 466-477)

Only tests/, __graft_entry__.smoke() and bench tools' cpu_baseline legs may import this module; the product path
(visiondk_amd/preprocess.py -> vdk_preprocess_*) never does.

What the reference executes lives in two un-vendored dependencies, restated here from their published algorithms:

* Pillow (requirements: `Pillow`; 12.2.0 in this image) `Image.resize(size, Image.BILINEAR)` = `ImagingResample` (src/libImaging/Resample.c):
  separable convolution resampling, horizontal pass first, then vertical, each in 8-bit fixed point:
    - `precompute_coeffs`: scale = in/out, filterscale = max(scale, 1), support = 1.0 * filterscale (bilinear), ksize = ceil(support)*2+1;
      per output coordinate: center = (xx + .5)*scale, xmin = max(0, int(center - support + .5)), xmax = min(in, int(center + support + .5)) - xmin,
      w[x] = triangle((x + xmin - center + .5) / filterscale), normalised by their sum (all in double);
    - `normalize_coeffs_8bpc`: k = int(±0.5 + w * 2^22) (PRECISION_BITS = 32 - 8 - 2);
    - per pass: acc = 2^21 + sum(pixel * k)  (int32),  out = clip(acc >> 22, 0, 255): the HORIZONTAL result is rounded to uint8 before the vertical pass;
    - a pass whose size does not change is skipped (identical to running it: its coefficients are exactly [1, 0]).
  `ImageOps.expand(image, (l, t, r, b), fill=(0,0,0))` pastes the image into a zero canvas.
* torchvision `ToTensor` (uint8 HWC -> float32 CHW, `.div(255)`) and `Normalize` (`sub_(mean).div_(std)`, both float32 tensors).

Pinned by tests/test_oracle_preprocess.py against Pillow itself (installed here and on the GPU box) over random and extreme geometries, and against
tests/golden/preprocess.npz, which tests/golden/make_golden.py produced by executing the reference's own `ResizeAndPadding2Square` class."""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def output_geometry(width: int, height: int, size: int) -> Tuple[int, int, int, int]:
    """dataset/transforms.py:343-357: (new_width, new_height, pad_left, pad_top) — Python float arithmetic, truncating int()."""
    max_side = max(width, height)
    scale_factor = size / max_side
    new_width = int(width * scale_factor)
    new_height = int(height * scale_factor)
    return new_width, new_height, (size - new_width) // 2, (size - new_height) // 2


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter and the full-image box (in0 = 0, in1 = in_size)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.empty(xmax, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            if v < 0.0:
                v = -v
            v = 1.0 - v if v < 1.0 else 0.0
            w[x] = v
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def _pass(img: np.ndarray, kk: np.ndarray, bounds: np.ndarray) -> np.ndarray:
    """One 8-bit resampling pass along axis 0 of `img` [n_in, ...] -> [n_out, ...]."""
    out = np.empty((kk.shape[0],) + img.shape[1:], dtype=np.uint8)
    src = img.astype(np.int32)
    for xx in range(kk.shape[0]):
        xmin, xmax = bounds[xx]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int32)
        for x in range(xmax):
            acc += src[xmin + x] * kk[xx, x]
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bilinear(img: np.ndarray, new_width: int, new_height: int) -> np.ndarray:
    """PIL `Image.resize((new_width, new_height), Image.BILINEAR)` on an HWC uint8 array."""
    if new_width <= 0 or new_height <= 0:
        raise ValueError("height and width must be > 0")            # Image.resize -> ImagingResample error text
    h, w = img.shape[:2]
    if w != new_width:                                                # horizontal first (Resample.c ImagingResample)
        kk, b = precompute_coeffs(w, new_width)
        img = _pass(img.transpose(1, 0, 2), kk, b).transpose(1, 0, 2)
    if h != new_height:
        kk, b = precompute_coeffs(h, new_height)
        img = _pass(img, kk, b)
    return np.ascontiguousarray(img)


def resize_and_padding(img: np.ndarray, size: int) -> np.ndarray:
    """ResizeAndPadding2Square(size, training=False).__call__ (dataset/transforms.py:336-360): HWC uint8 -> [size, size, 3] uint8."""
    h, w = img.shape[:2]
    nw, nh, pl, pt = output_geometry(w, h, size)
    small = resize_bilinear(img, nw, nh)
    out = np.zeros((size, size, img.shape[2]), dtype=np.uint8)
    out[pt:pt + nh, pl:pl + nw] = small
    return out


def to_tensor_normalize(img_u8: np.ndarray, mean: Sequence[float], std: Sequence[float]) -> np.ndarray:
    """torchvision ToTensor + Normalize in float32: (u8 / 255 - mean) / std, HWC -> CHW."""
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    m = np.asarray(mean, dtype=np.float32)[:, None, None]
    s = np.asarray(std, dtype=np.float32)[:, None, None]
    return (x - m) / s


def preprocess(img: np.ndarray, size: int, mean: Sequence[float] = (0.485, 0.456, 0.406), std: Sequence[float] = (0.229, 0.224, 0.225)) -> np.ndarray:
    return to_tensor_normalize(resize_and_padding(img, size), mean, std)
