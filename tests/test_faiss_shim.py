"""The drop-in boundary proven with the reference's OWN code.

* `index()` / `search()` of /root/reference/engine/cbir/evaluation.py (lines 106-200) are read at test time, exec'd into a namespace whose `faiss` is
  visiondk_amd.faiss_shim and whose extractor is visiondk_amd.face.FeatureExtractor, and must return the oracle's scores / indices bit for bit.
* `Trainer.update` of /root/reference/engine/procedure/train.py (lines 203-215) is exec'd as it stands and drives visiondk_amd.vit.VisionTransformer with the stock
  torch.optim.SGD, torch's GradScaler and the reference's own ModelEMA (models/ema.py, loaded by path).

Nothing of the reference is copied into the repository: the functions are cut out of the source files with `ast` when the tests run.  /root/reference does not exist on the
GPU box: those tests skip there, and the committed fixture tests/golden/faiss_shim_reference_run.npz (written by tests/golden/make_faiss_shim_golden.py from a run of the
reference's functions over the shim) pins the same numbers for the `-m gpu` run."""
import ast
import importlib.util
import textwrap
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import cbir as ocbir
from visiondk_amd import cbir, faiss_shim

REF = Path("/root/reference")
GOLD = Path(__file__).resolve().parent / "golden" / "faiss_shim_reference_run.npz"
needs_ref = pytest.mark.skipif(not REF.exists(), reason="/root/reference is not present on this box")


def _cut(path: Path, names, cls=None) -> str:
    """source text of the named top-level functions (or methods of `cls`) of a reference file, as written there"""
    src = path.read_text()
    tree = ast.parse(src)
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    out = []
    for n in body:
        if isinstance(n, ast.FunctionDef) and n.name in names:
            seg = ast.get_source_segment(src, n)
            for d in n.decorator_list:       # (the method is a @staticmethod: keep it a plain function here)
                pass
            out.append(textwrap.dedent(seg))
    assert len(out) == len(names), (names, len(out))
    return "\n\n".join(out)


class _Logger:
    def __init__(self):
        self.lines = []

    def console(self, msg):
        self.lines.append(msg)


class _EmbeddingExtractor:
    """extract_cbir(dataloader, device) -> float32 ndarray, the FeatureExtractor protocol index() / search() use (face_model.py:119-143); the loader yields embeddings"""

    def extract_cbir(self, dataloader, device):
        return np.concatenate([np.asarray(b, dtype=np.float32) for b in dataloader], 0)


class _Passthrough(torch.nn.Module):
    """a `model` for visiondk_amd.face.FeatureExtractor whose forward returns the batch it is given: the extractor then does what it does for a backbone's output --
    eval(), no_grad, L2-normalise on the device (vdk_l2norm_rows), host numpy, concatenate in loader order (face_model.py:119-143)"""

    def __init__(self, be):
        super().__init__()
        self.be = be

    def forward(self, x):
        return x


def reference_namespace():
    from typing import Optional
    from torch.utils.data import DataLoader
    from tqdm import tqdm
    ns = {"faiss": faiss_shim, "np": np, "torch": torch, "tqdm": tqdm, "Optional": Optional, "DataLoader": DataLoader,
          "FeatureExtractor": object, "SmartLogger": object}
    exec(compile(_cut(REF / "engine" / "cbir" / "evaluation.py", ("index", "search")), str(REF / "engine/cbir/evaluation.py"), "exec"), ns)
    return ns


def _data(n=3000, nq=70, d=128, seed=0):
    rng = np.random.default_rng(seed)
    g = ocbir.l2norm_rows(rng.standard_normal((n, d), dtype=np.float32))
    q = ocbir.l2norm_rows(rng.standard_normal((nq, d), dtype=np.float32))
    return g, q


@needs_ref
def test_reference_index_and_search_run_unmodified_over_the_shim(emu, tmp_path):
    """device.type == 'cpu' in the reference's index(): the plain index_factory index (fp32 storage); k > some batch boundaries, a ragged last query batch, the memmap
    save path; scores and indices bit-equal to the oracle"""
    faiss_shim.set_default(backend=emu, device="cpu")
    try:
        from visiondk_amd import face
        ns = reference_namespace()
        g0, q0 = _data()
        rng = np.random.default_rng(5)
        g0 = g0 * rng.uniform(0.5, 3.0, (g0.shape[0], 1)).astype(np.float32)      # un-normalised "backbone outputs": the extractor normalises them
        q0 = q0 * rng.uniform(0.5, 3.0, (q0.shape[0], 1)).astype(np.float32)
        extractor = face.FeatureExtractor(_Passthrough(emu))                         # the library's FeatureExtractor, as index() / search() receive it
        gl = [torch.from_numpy(g0[:1000]), torch.from_numpy(g0[1000:])]
        ql = [torch.from_numpy(q0[:50]), torch.from_numpy(q0[50:])]
        g, q = extractor.extract_cbir(gl, torch.device("cpu")), extractor.extract_cbir(ql, torch.device("cpu"))      # what index() / search() will extract themselves
        assert np.abs(g - ocbir.l2norm_rows(g0)).max() < 1e-6 and g.dtype == np.float32
        log = _Logger()
        mm = tmp_path / "gallery.mm"
        fi = ns["index"](extractor, gl, torch.device("cpu"), log, index_factory="Flat", memmap_save_path=str(mm))
        assert isinstance(fi, cbir.FlatIPIndex) and fi.ntotal == g.shape[0] and fi.storage == "float32"
        assert np.array_equal(np.memmap(str(mm), mode="r", dtype=np.float32).reshape(-1, g.shape[1]), g)
        s, i = ns["search"](extractor, ql, fi, torch.device("cpu"), log, k=100, batch_size=32)
        so, io = ocbir.flat_ip_search(q, g, 100)
        assert s.dtype == np.float32 and i.dtype == np.int64 and s.shape == (q.shape[0], 100)
        assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
        assert "Adding embeddings..." in log.lines and "Searching ..." in log.lines
        # the memmap-load branch of the reference never sets `dim`: it raises NameError there as well -- the shim does not paper over it
        with pytest.raises((NameError, UnboundLocalError)):
            ns["index"](_EmbeddingExtractor(), None, torch.device("cpu"), log, memmap_feat_dim=g.shape[1], memmap_dtype=np.float32, memmap_save_path=str(mm),
                        memmap_load_embedding=True)
    finally:
        faiss_shim.set_default(None, None)


@needs_ref
def test_reference_gpu_branch_clones_with_float16_storage(emu):
    """device.type == 'cuda' in the reference's index(): GpuMultipleClonerOptions().useFloat16 = True -> index_cpu_to_all_gpus (evaluation.py:157-162).  The branch is taken
    by handing the function a cuda torch.device (only its .type is read); the clone stores fp16 rows and answers like the fp32 index fed fp16-rounded vectors."""
    faiss_shim.set_default(backend=emu, device="cpu")
    try:
        ns = reference_namespace()
        g, q = _data(n=2000, nq=40, seed=1)
        log = _Logger()
        fi = ns["index"](_EmbeddingExtractor(), [g], torch.device("cuda"), log)
        assert fi.storage == "float16" and fi.ntotal == g.shape[0]
        s, i = ns["search"](_EmbeddingExtractor(), [q], fi, torch.device("cuda"), log, k=50, batch_size=16)
        so, io = ocbir.flat_ip_search(q.astype(np.float16).astype(np.float32), g.astype(np.float16).astype(np.float32), 50)
        assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
    finally:
        faiss_shim.set_default(None, None)


def test_shim_surface(emu):
    """every faiss name the reference touches (and the commented alternatives beside them) exists with faiss's call shapes"""
    faiss_shim.set_default(backend=emu, device="cpu")
    try:
        assert faiss_shim.METRIC_INNER_PRODUCT == 0 and faiss_shim.Index is cbir.FlatIPIndex
        idx = faiss_shim.index_factory(64, "Flat", faiss_shim.METRIC_INNER_PRODUCT)
        g, q = _data(n=500, nq=9, d=64, seed=2)
        idx.train(g); idx.add(g)
        co = faiss_shim.GpuClonerOptions(); co.useFloat16 = True
        res = faiss_shim.StandardGpuResources()
        gi = faiss_shim.index_cpu_to_gpu(res, 0, idx, co)                 # vectors added before the clone travel with it
        assert gi.ntotal == 500 and gi.storage == "float16"
        s, i = gi.search(q, k=10)
        so, io = ocbir.flat_ip_search(q.astype(np.float16).astype(np.float32), g.astype(np.float16).astype(np.float32), 10)
        assert np.array_equal(i, io) and np.array_equal(s.view(np.uint32), so.view(np.uint32))
        back = faiss_shim.index_gpu_to_cpu(gi)
        assert back.storage == "float32" and back.ntotal == 500
        with pytest.raises(NotImplementedError):
            faiss_shim.index_factory(64, "IVF100,Flat", faiss_shim.METRIC_INNER_PRODUCT)
        mco = faiss_shim.GpuMultipleClonerOptions(); mco.shard = True
        with pytest.raises(NotImplementedError):
            faiss_shim.index_cpu_to_all_gpus(idx, mco)
        x = g.copy() * 3.0
        faiss_shim.normalize_L2(x)
        assert np.allclose(x, g, atol=1e-6)
    finally:
        faiss_shim.set_default(None, None)


def test_committed_fixture_of_the_reference_run(emu):
    """tests/golden/faiss_shim_reference_run.npz = what the reference's index() / search() returned over the shim (make_faiss_shim_golden.py); the shim's own calls
    reproduce it bit for bit wherever it runs"""
    z = np.load(GOLD)
    faiss_shim.set_default(backend=emu, device="cpu")
    try:
        _check_fixture(z)
    finally:
        faiss_shim.set_default(None, None)


@pytest.mark.gpu
def test_committed_fixture_of_the_reference_run_on_the_mi355x(hip):
    _check_fixture(np.load(GOLD))


def _check_fixture(z):
    g, q = z["gallery"], z["queries"]
    for half in (False, True):
        idx = faiss_shim.index_factory(g.shape[1], "Flat", faiss_shim.METRIC_INNER_PRODUCT)
        if half:
            co = faiss_shim.GpuMultipleClonerOptions(); co.useFloat16 = True
            idx = faiss_shim.index_cpu_to_all_gpus(idx, co)
        idx.train(g); idx.add(g)
        s, i = idx.search(q, k=int(z["k"]))
        tag = "f16" if half else "f32"
        assert np.array_equal(i, z[f"indices_{tag}"]) and np.array_equal(s.view(np.uint32), z[f"scores_{tag}"].view(np.uint32))


# ---- Trainer.update, as the reference wrote it, driving the HIP ViT with the stock optimizer / scaler / the reference's ModelEMA -------------------------------------------
def _load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@needs_ref
def test_reference_trainer_update_drives_the_hip_vit(emu):
    """Trainer.update (train.py:203-215) exec'd from the reference's file: scaler.scale(loss).backward() -> unscale_ -> clip_grad_norm_(10) -> scaler.step -> scaler.update ->
    zero_grad -> ema.update, with torch.optim.SGD over model.parameters(), torch's GradScaler and models/ema.py's ModelEMA.  The model is visiondk_amd.vit.VisionTransformer
    (nn.Module protocol: parameters(), deepcopy for the EMA, autograd through one node); the same sequence on the fp32 oracle gives the same weights within bf16's reach, and
    the library's own FusedTrainStep lands on the same weights as the reference's function does on this model."""
    import copy
    from oracle.vit_ref import VisionTransformerRef
    from visiondk_amd import vit
    ns = {"torch": torch}
    src = _cut(REF / "engine" / "procedure" / "train.py", ("update",), cls="Trainer").replace("@staticmethod\n", "")
    exec(compile(src, str(REF / "engine/procedure/train.py"), "exec"), ns)
    update = ns["update"]
    ModelEMA = _load_by_path("ref_ema", "models/ema.py").ModelEMA
    spec = vit.VitSpec(img_size=32, patch_size=8, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256)
    torch.manual_seed(0)
    ref = VisionTransformerRef(32, 8, 3, 10, 128, 2, 2, 256)
    model = vit.VisionTransformer(spec, device="cpu", seed=0, backend=emu)
    model.load_state_dict(ref.state_dict())
    twin = vit.VisionTransformer(spec, device="cpu", seed=0, backend=emu)
    twin.load_state_dict(ref.state_dict())
    lr, mom, wd = 0.05, 0.9, 5e-4
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=mom, weight_decay=wd)
    opt_ref = torch.optim.SGD(ref.parameters(), lr=lr, momentum=mom, weight_decay=wd)
    scaler = torch.amp.GradScaler("cpu", init_scale=1024.0)
    scaler_ref = torch.amp.GradScaler("cpu", init_scale=1024.0)
    ema = ModelEMA(model)
    fused = vit.FusedTrainStep(twin, lr=lr, momentum=mom, weight_decay=wd, label_smoothing=0.05, max_norm=10.0, ema=True)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=0.05)
    start = {n: p.detach().clone() for n, p in ref.named_parameters()}
    torch.manual_seed(1)
    for _ in range(2):
        x = torch.randn(8, 3, 32, 32); y = torch.randint(0, 10, (8,))
        model.train(); ref.train()
        update(model, crit(model(x), y), scaler, opt, ema)
        update(ref, crit(ref(x), y), scaler_ref, opt_ref, None)
        fused.step(x, y)
    got, exp, tw = dict(model.named_parameters()), dict(ref.named_parameters()), dict(twin.named_parameters())
    for n, p in exp.items():
        upd_ref = p.detach() - start[n]
        d = (got[n].detach() - start[n] - upd_ref).norm() / upd_ref.norm().clamp_min(1e-12)
        assert d < 0.1, (n, d.item())                                    # two steps' UPDATE, bf16 operands against the fp32 oracle (tests/test_vit.py holds the fused step to the same)
        d2 = (got[n].detach() - tw[n].detach()).norm() / upd_ref.norm().clamp_min(1e-12)
        assert d2 < 2e-3, (n, d2.item())                                 # the reference's update on the HIP model == the library's fused step on the HIP model
    assert all(g.grad is None or float(g.grad.abs().sum()) == 0.0 for g in model.parameters())      # optimizer.zero_grad() ran
    e = dict(ema.ema.named_parameters())
    n0 = "blocks.0.mlp.fc1.weight"
    assert not torch.equal(e[n0], got[n0].detach()) and (e[n0] - got[n0].detach()).abs().max() < 1e-2      # the deep-copied EMA model moved towards the weights


TRAINER_GOLD = Path(__file__).resolve().parent / "golden" / "trainer_update_reference_run.npz"


@pytest.mark.parametrize("operand", ["fp16", "bf16"])
def test_committed_trainer_update_fixture(be, dev, operand):
    """tests/golden/trainer_update_reference_run.npz (written by make_trainer_update_golden.py from the reference's own Trainer.update on the fp32 PyTorch-CPU path: three
    steps of scaler.scale(loss).backward / unscale_ / clip / step / update / ema.update) against the HIP library's step on the same seeds -- on the emulator and, where
    /root/reference does not exist, on the MI355X: the loss sequence and the final weights of every parameter.  fp16 operands (the reference's autocast dtype) with the
    GradScaler protocol at the fixture's initial scale; bf16 beside it with its 8x coarser bounds."""
    from oracle.vit_ref import VisionTransformerRef
    from visiondk_amd import vit
    if not be.device_only and (operand == "bf16" or REF.exists()):
        pytest.skip("on the emulator the reference's update() itself drives the library (test_reference_trainer_update_drives_the_hip_vit) where /root/reference exists; "
                    "the recorded run is for the GPU box, and for an emulator run without the reference (fp16 arm)")
    z = np.load(TRAINER_GOLD)
    c = {k[4:]: float(z[k]) for k in z.files if k.startswith("cfg_")}
    img, patch, classes, dim, depth, heads, mlp = (int(c[k]) for k in ("img", "patch", "classes", "dim", "depth", "heads", "mlp"))
    torch.manual_seed(int(c["seed_model"]))
    ref = VisionTransformerRef(img, patch, 3, classes, dim, depth, heads, mlp)
    start = {n: p.detach().clone() for n, p in ref.named_parameters()}
    model = vit.VisionTransformer(vit.VitSpec(img_size=img, patch_size=patch, num_classes=classes, dim=dim, depth=depth, heads=heads, mlp_dim=mlp), device=dev, backend=be, seed=0,
                                  operand=operand)
    model.load_state_dict(ref.state_dict())
    step = vit.FusedTrainStep(model, lr=c["lr"], momentum=c["momentum"], weight_decay=c["weight_decay"], label_smoothing=c["label_smoothing"], max_norm=10.0, ema=True,
                              init_scale=c["init_scale"])
    g = torch.Generator(); g.manual_seed(int(c["seed_data"]))
    losses = []
    for _ in range(int(c["steps"])):
        x = torch.randn(int(c["batch"]), 3, img, img, generator=g); y = torch.randint(0, classes, (int(c["batch"]),), generator=g)
        step.step(x.to(dev), y.to(dev))
        losses.append(step.loss_value())
    ltol, utol = (3e-4, 5e-3) if operand == "fp16" else (3e-3, 5e-2)      # measured: losses 1.2e-5 / 1.9e-4, worst update 8.5e-4 / 6.4e-3
    assert np.allclose(losses, z["losses"], rtol=ltol), (losses, z["losses"])
    if operand == "fp16":
        assert step.loss_scale() == float(z["scale_after"]) and step.skipped_steps() == 0
    got = dict(model.named_parameters())
    worst = (0.0, None)
    for n, w0 in start.items():
        want = torch.from_numpy(z["w:" + n])
        upd_ref = want - w0
        d = ((got[n].detach().cpu() - w0 - upd_ref).norm() / upd_ref.norm().clamp_min(1e-12)).item()
        worst = max(worst, (d, n))
        assert d < utol, (n, d)
    print(operand, "losses", losses, "worst update error", worst)
    n0 = "blocks.0.mlp.fc1.weight"
    eng = model.engine
    off, numel, shape = next((o, m, s) for (nm, o, m, s) in eng.entries if nm == n0)
    e_ref = torch.from_numpy(z["ema:" + n0])
    e = step.ema[off:off + numel].view(shape).cpu()
    assert ((e - e_ref).norm() / (e_ref - start[n0]).norm().clamp_min(1e-12)).item() < max(utol, 2e-2)      # the EMA moved the same way (relative to its own movement)
