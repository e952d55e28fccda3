"""Host side of the validation-time input pipeline (the step in front of the hot path).

The reference builds a per-image CPU chain from the yaml's `augment` list with `create_AugTransforms` (dataset/transforms.py:530-555); for the
validation / gallery loaders that list is always

    - resize_and_padding: {size: S, training: False}       (transforms.py:325-362, :499-500)
    - to_tensor: no_params                                  (:466-468)
    - normalize: {mean: [...], std: [...]}                  (:474-477)

`create_AugTransforms` here accepts the same list and returns a `ValPipeline` that runs the three steps for a WHOLE BATCH of decoded images of
different sizes in one kernel (`vdk_preprocess_resize_pad_normalize`), bit-exact with what PIL + torchvision produce per image on the CPU.  Anything
else in the list (the random training augmentations) raises: those stay host-side data-loader work and are out of scope.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
MAX_SIDE = 8192


def _as_hwc_u8(img) -> np.ndarray:
    if isinstance(img, torch.Tensor):
        img = img.cpu().numpy()
    a = np.asarray(img)                      # PIL.Image supports the array protocol (`Image.open(path).convert('RGB')`, dataset/basedataset.py read_image)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise TypeError(f"expected a decoded RGB image as uint8 [H, W, 3], got {a.dtype} {a.shape}")
    return np.ascontiguousarray(a)


class ValPipeline:
    """ResizeAndPadding2Square(size, training=False) -> ToTensor -> Normalize(mean, std), batched on the device."""

    def __init__(self, size: int = 224, mean: Sequence[float] = IMAGENET_MEAN, std: Sequence[float] = IMAGENET_STD, device=None, backend=None):
        self.size = int(size)
        self.mean = tuple(float(np.float32(m)) for m in mean)
        self.std = tuple(float(np.float32(s)) for s in std)
        if len(self.mean) != 3 or len(self.std) != 3:
            raise ValueError("mean and std must have three entries (RGB)")
        if any(s == 0 for s in self.std):
            raise ValueError("std evaluated to zero, leading to division by zero.")   # torchvision F.normalize
        self.be = backend if backend is not None else _lib.load()
        self.device = torch.device(device) if device is not None else torch.device("cuda" if self.be.device_only else "cpu")
        self._ws: Optional[torch.Tensor] = None

    def __repr__(self) -> str:
        return f"ValPipeline(resize_and_padding(size={self.size}) -> to_tensor -> normalize(mean={self.mean}, std={self.std}))"

    # ---- packing: ragged uint8 images back to back, offsets, (w, h) -----------------------------------------------------------------------
    def pack(self, images: Sequence) -> tuple:
        arrs = [_as_hwc_u8(im) for im in images]
        if not arrs:
            raise ValueError("empty batch")
        wh = np.array([[a.shape[1], a.shape[0]] for a in arrs], dtype=np.int32)
        for w, h in wh:
            # PIL's own error for a side that truncates to zero (Image.resize), raised here before anything is launched
            sf = self.size / max(int(w), int(h))
            if int(w * sf) <= 0 or int(h * sf) <= 0:
                raise ValueError("height and width must be > 0")
            if max(w, h) > MAX_SIDE:
                raise ValueError(f"image side {max(w, h)} above the supported {MAX_SIDE}")
        sizes = np.array([a.size for a in arrs], dtype=np.int64)
        offsets = np.zeros(len(arrs), dtype=np.int64)
        offsets[1:] = np.cumsum(sizes)[:-1]
        total = int(sizes.sum())
        pinned = self.device.type == "cuda"
        buf = torch.empty((total + 15) // 16 * 16 + 16, dtype=torch.uint8, pin_memory=pinned)     # the kernel stages rows with aligned 16-byte loads
        nb = buf.numpy()
        for a, o in zip(arrs, offsets):
            nb[o:o + a.size] = a.reshape(-1)
        return buf, torch.from_numpy(offsets), torch.from_numpy(wh), int(wh.max())

    def run_packed(self, pixels: torch.Tensor, offsets: torch.Tensor, wh: torch.Tensor, max_side: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Device-resident inputs (uint8 [bytes], int64 [B], int32 [B, 2]) -> float32 [B, 3, S, S]."""
        be, S, B = self.be, self.size, wh.shape[0]
        need = C.c_size_t(0)
        be.check(be.lib.vdk_preprocess_workspace_bytes(B, S, max_side, C.byref(need)), "vdk_preprocess_workspace_bytes")
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != pixels.device:
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=pixels.device)
        if out is None:
            out = torch.empty((B, 3, S, S), dtype=torch.float32, device=pixels.device)
        m, s = self.mean, self.std
        be.check(be.lib.vdk_preprocess_resize_pad_normalize(be.ptr(pixels), be.ptr(offsets), be.ptr(wh), B, S, max_side, m[0], m[1], m[2], s[0], s[1], s[2],
                                                            be.ptr(out), None, be.ptr(self._ws), self._ws.numel(), be.stream()),
                 "vdk_preprocess_resize_pad_normalize")
        return out

    def __call__(self, images: Union[Sequence, np.ndarray], out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """images: a list of decoded RGB images (PIL.Image / uint8 [H, W, 3] arrays), any sizes -> float32 [B, 3, S, S] on the device,
        what `torch.stack([transform(img) for img in images])` (the reference's collate_fn, dataset/basedataset.py:180-196) gives."""
        if isinstance(images, np.ndarray) and images.ndim == 3:
            images = [images]
        pixels, offsets, wh, max_side = self.pack(images)
        dev = self.device
        return self.run_packed(pixels.to(dev, non_blocking=True), offsets.to(dev, non_blocking=True), wh.to(dev, non_blocking=True), max_side, out)


def create_AugTransforms(augments: List[dict], device=None, backend=None) -> ValPipeline:
    """dataset/transforms.py:530-555 for the deterministic validation list.  Same input format: a list of single-key dicts, params dict or 'no_params'."""
    names = [next(iter(a.keys())) for a in augments]
    if names != ["resize_and_padding", "to_tensor", "normalize"]:
        raise NotImplementedError(f"only the validation chain resize_and_padding -> to_tensor -> normalize runs on the device, got {names}")
    rp = augments[0]["resize_and_padding"]
    rp = {} if rp == "no_params" else dict(rp)
    if rp.get("training", False):
        raise NotImplementedError("resize_and_padding(training=True) draws its resampling filter at random per image (transforms.py:338-339); host-side only")
    nm = augments[2]["normalize"]
    nm = {} if nm == "no_params" else dict(nm)
    return ValPipeline(size=rp.get("size", 224), mean=nm.get("mean", IMAGENET_MEAN), std=nm.get("std", IMAGENET_STD), device=device, backend=backend)


def set_label_transforms(label, num_classes: int, label_smooth: float) -> torch.Tensor:
    """Soft target vector for the BCE / focal losses (`vdk_bce_logits` consumes it as is); same contract as the reference's
    ImageDatasets.set_label_transforms (dataset/basedataset.py:198-231): with a = label_smooth, negatives sit at a / 2 and positives at 1 - a / 2.
    Accepted label forms: a one-hot / multi-hot float tensor of length num_classes (csv datasets), a class index, or a list of 0/1 flags."""
    half = 0.5 * label_smooth
    if isinstance(label, torch.Tensor) and label.size(0) == num_classes:
        return label if label_smooth <= 0 else label * (1 - label_smooth) + half
    hot = torch.zeros(num_classes, dtype=torch.bool)
    if isinstance(label, int):
        hot[label] = True
    elif isinstance(label, (list, tuple)):
        flags = torch.as_tensor(label)
        hot[: flags.numel()] = flags != 0
    target = torch.full((num_classes,), half)
    target[hot] = 1 - half
    return target
