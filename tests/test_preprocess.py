"""csrc/preprocess.hip (vdk_preprocess_resize_pad_normalize) vs oracle/preprocess_ref.py and the fixture the reference's own
ResizeAndPadding2Square produced (tests/golden/preprocess.npz): bit-exact, on the emulator and on the MI355X.  Ragged batches, up- and down-scaling,
extreme aspect ratios, sides that truncate below `size`, 1x1 images, sizes above 256 / 512 columns."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import preprocess_ref as ref
from visiondk_amd import preprocess

G = Path(__file__).resolve().parent / "golden"


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _images(rng, geoms):
    out = []
    for w, h in geoms:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if (w + h) % 2 == 0:
            a[: max(h // 2, 1), : max(w // 3, 1)] = 255          # saturated / black blocks: the 8-bit clip
            a[h // 2:, w // 2:] = 0
        out.append(a)
    return out


def test_golden_fixture_from_reference_class(be, dev):
    z = np.load(G / "preprocess.npz")
    for i in range(int(z["n"])):
        size = int(z[f"size{i}"])
        pipe = preprocess.ValPipeline(size=size, mean=z["mean"], std=z["std"], device=dev, backend=be)
        got = pipe([z[f"in{i}"]]).cpu().numpy()[0]
        # the oracle and the kernel both reproduce the reference's uint8 canvas and float tensor exactly
        assert np.array_equal(ref.resize_and_padding(z[f"in{i}"], size), z[f"u8_{i}"]), i
        assert np.array_equal(_bits(got), _bits(z[f"f32_{i}"])), (i, np.abs(got - z[f"f32_{i}"]).max())


@pytest.mark.parametrize("size,geoms", [
    (64, [(61, 47), (47, 61), (64, 64), (130, 20), (20, 200), (1, 1), (7, 5), (33, 33), (64, 10), (300, 299)]),
    (224, [(500, 375), (49, 49), (224, 100), (97, 1001)]),
    (32, [(2000, 70), (30, 700), (31, 33), (1500, 120)]),                      # 60x downscale: wide tap windows, short side of one or two rows
    (300, [(123, 77), (640, 480)]),                              # two columns per thread
    (520, [(90, 60), (701, 333)]),                               # four columns per thread, 8-row bands
])
def test_ragged_batch_bit_exact(be, dev, size, geoms):
    rng = np.random.default_rng(size * 31 + len(geoms))
    imgs = _images(rng, geoms)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    pipe = preprocess.ValPipeline(size=size, mean=mean, std=std, device=dev, backend=be)
    got = pipe(imgs).cpu().numpy()
    assert got.shape == (len(imgs), 3, size, size)
    for i, a in enumerate(imgs):
        exp = ref.preprocess(a, size, mean, std)
        assert np.array_equal(_bits(got[i]), _bits(exp)), (geoms[i], np.abs(got[i] - exp).max())


def test_status_codes_and_host_errors(be, dev):
    import ctypes as C
    S = 32
    pipe = preprocess.ValPipeline(size=S, device=dev, backend=be)
    with pytest.raises(ValueError, match="height and width must be > 0"):      # PIL's error for a side truncating to 0, before any launch
        pipe([np.zeros((2, 900, 3), dtype=np.uint8)])
    with pytest.raises(TypeError):
        pipe([np.zeros((8, 8), dtype=np.uint8)])
    with pytest.raises(NotImplementedError):
        preprocess.create_AugTransforms([{"random_horizonflip": {"p": 0.5}}, {"to_tensor": "no_params"}], device=dev, backend=be)
    # the raw ABI reports the same condition per image and still writes the normalised zero canvas for it
    good = np.random.default_rng(0).integers(0, 256, (20, 30, 3), dtype=np.uint8)
    bad = np.full((2, 900, 3), 200, dtype=np.uint8)
    flat = np.concatenate([good.reshape(-1), bad.reshape(-1), np.zeros(32, np.uint8)])
    px = torch.from_numpy(flat).to(dev)
    off = torch.tensor([0, good.size], dtype=torch.int64, device=dev)
    wh = torch.tensor([[30, 20], [900, 2]], dtype=torch.int32, device=dev)
    need = C.c_size_t(0)
    be.check(be.lib.vdk_preprocess_workspace_bytes(2, S, 900, C.byref(need)), "ws")
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    out = torch.empty(2, 3, S, S, device=dev)
    st = torch.full((2,), -1, dtype=torch.int32, device=dev)
    m, s = pipe.mean, pipe.std
    be.check(be.lib.vdk_preprocess_resize_pad_normalize(be.ptr(px), be.ptr(off), be.ptr(wh), 2, S, 900, m[0], m[1], m[2], s[0], s[1], s[2], be.ptr(out), be.ptr(st),
                                                        be.ptr(ws), ws.numel(), be.stream()), "run")
    assert st.cpu().tolist() == [0, 1]
    assert np.array_equal(_bits(out[0].cpu().numpy()), _bits(ref.preprocess(good, S, m, s)))
    canvas = ref.to_tensor_normalize(np.zeros((S, S, 3), np.uint8), m, s)
    assert np.array_equal(_bits(out[1].cpu().numpy()), _bits(canvas))
    assert be.lib.vdk_preprocess_workspace_bytes(2, S, 9000, C.byref(need)) < 0       # sides above 8192 are refused, not truncated


def test_soft_targets_vs_reference_staticmethod():
    z = np.load(G / "preprocess.npz")
    slt = preprocess.set_label_transforms
    assert np.array_equal(slt(3, 7, 0.1).numpy(), z["lab_int"])
    assert np.array_equal(slt([0, 1, 0, 0, 1, 0, 1], 7, 0.2).numpy(), z["lab_list"])
    onehot = torch.tensor([1., 0., 0., 1., 0., 0., 0.])
    assert np.array_equal(slt(onehot, 7, 0.1).numpy(), z["lab_onehot"])
    assert np.array_equal(slt(onehot, 7, 0.0).numpy(), z["lab_onehot_nosmooth"])


def test_yaml_list_factory_matches_reference_format(be, dev):
    aug = [{"resize_and_padding": {"size": 48, "training": False}}, {"to_tensor": "no_params"},
           {"normalize": {"mean": [0.5, 0.4, 0.3], "std": [0.2, 0.25, 0.3]}}]                     # the shape of pet.yaml:94-101
    pipe = preprocess.create_AugTransforms(aug, device=dev, backend=be)
    a = np.random.default_rng(3).integers(0, 256, (37, 91, 3), dtype=np.uint8)
    got = pipe(a).cpu().numpy()[0]
    assert np.array_equal(_bits(got), _bits(ref.preprocess(a, 48, (0.5, 0.4, 0.3), (0.2, 0.25, 0.3))))


@pytest.mark.gpu
def test_full_size_batch_bit_exact_gpu(hip):
    """A val batch at the reference's size (pet.yaml val bs 320 at 224): Oxford-pet-like geometries, every image against the oracle."""
    rng = np.random.default_rng(5)
    geoms = [(int(rng.integers(150, 640)), int(rng.integers(150, 640))) for _ in range(320)]
    imgs = _images(rng, geoms)
    pipe = preprocess.ValPipeline(size=224, device="cuda", backend=hip)
    got = pipe(imgs).cpu().numpy()
    for i in range(0, 320, 7):
        assert np.array_equal(_bits(got[i]), _bits(ref.preprocess(imgs[i], 224))), geoms[i]
    # size-independent property over the whole batch: padding columns / rows hold exactly the normalised zero
    canvas = ref.to_tensor_normalize(np.zeros((1, 1, 3), np.uint8), pipe.mean, pipe.std).reshape(3)
    for i, (w, h) in enumerate(geoms):
        nw, nh, pl, pt = ref.output_geometry(w, h, 224)
        if pl > 0:
            assert np.all(got[i][:, :, :pl] == canvas[:, None, None])
        if pt > 0:
            assert np.all(got[i][:, :pt, :] == canvas[:, None, None])
