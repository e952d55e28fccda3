"""N>1 path of hot path A on CPU: 2 ranks over gloo (SIMT-emulated kernels), bucketed flat-gradient all-reduce.
Each rank takes half of the batch; the averaged gradients and the stepped weights must equal a single process run
on the whole batch (mean reduction: grad(full batch) = mean of the per-rank grads)."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VDK_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu.emu_backend import load_emu
    from tests.test_vit import SPEC
    from visiondk_amd import comm, vit
    be = load_emu()
    model = vit.VisionTransformer(SPEC, device="cpu", backend=be, seed=100 + rank)   # ranks start DIFFERENT on purpose
    c = comm.GradAllReduce(bucket_bytes=200_000)                                     # small buckets -> several collectives
    c.broadcast_params(model.engine.params, src=0)
    step = vit.FusedTrainStep(model, lr=0.01, label_smoothing=0.05, ema=False, comm=c)
    torch.manual_seed(7)
    x = torch.randn(4, 3, 32, 32); y = torch.randint(0, 10, (4,))
    lo, hi = rank * 2, rank * 2 + 2
    step.step(x[lo:hi], y[lo:hi])
    torch.save({"params": model.engine.params.clone(), "grads": model.engine.grads.clone()}, f"{out_dir}/rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def _worker_fp16(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VDK_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu.emu_backend import load_emu
    from tests.test_vit import SPEC
    from visiondk_amd import comm, vit
    be = load_emu()
    model = vit.VisionTransformer(SPEC, device="cpu", backend=be, seed=100 + rank, operand="fp16")
    c = comm.GradAllReduce(bucket_bytes=200_000)
    c.broadcast_params(model.engine.params, src=0, engine=model.engine)
    # rank 0 keeps the EMA like the bench does; the first step overflows on purpose (scale 2^30): BOTH ranks must skip it and halve their scale
    step = vit.FusedTrainStep(model, lr=0.01, label_smoothing=0.05, ema=(rank == 0), comm=c, init_scale=2.0 ** 30)
    torch.manual_seed(7)
    x = torch.randn(4, 3, 32, 32); y = torch.randint(0, 10, (4,))
    lo, hi = rank * 2, rank * 2 + 2
    p0 = model.engine.params.clone()
    step.step(x[lo:hi], y[lo:hi])
    skipped_first = bool(torch.equal(model.engine.params, p0)) and step.skipped_steps() == 1
    step.loss_state[0] = 1024.0                                   # (a usable scale for the second step)
    step.step(x[lo:hi], y[lo:hi])
    torch.save({"params": model.engine.params.clone(), "skipped_first": skipped_first, "skipped": step.skipped_steps(), "scale": step.loss_scale()}, f"{out_dir}/rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_fp16_step_skips_and_steps_in_lockstep(tmp_path, emu):
    """fp16 operands under data parallelism: the inf check runs on the ALL-REDUCED gradient inside the optimizer kernel, so every rank takes the same skip / step decision
    (train.py:205-211 under DDP: `scaler.step` sees the averaged gradients) and the replicas stay bit-identical"""
    port = 29500 + ((os.getpid() + 211) % 500)
    mp.start_processes(_worker_fp16, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rank0.pt"); r1 = torch.load(tmp_path / "rank1.pt")
    assert r0["skipped_first"] and r1["skipped_first"]
    assert r0["skipped"] == r1["skipped"] == 1 and r0["scale"] == r1["scale"]
    assert torch.equal(r0["params"], r1["params"])
    assert torch.isfinite(r0["params"]).all()


@pytest.mark.slow
def test_two_rank_step_equals_single_process(tmp_path, emu):
    port = 29500 + (os.getpid() % 500)
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rank0.pt"); r1 = torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["params"], r1["params"])          # replicas stay bit-identical
    assert torch.equal(r0["grads"], r1["grads"])            # both hold the same summed gradient
    # single process, whole batch, same initial weights (rank 0's seed)
    from tests.test_vit import SPEC
    from visiondk_amd import vit
    model = vit.VisionTransformer(SPEC, device="cpu", backend=emu, seed=100)
    p0 = model.engine.params.clone()
    step = vit.FusedTrainStep(model, lr=0.01, label_smoothing=0.05, ema=False)
    torch.manual_seed(7)
    x = torch.randn(4, 3, 32, 32); y = torch.randint(0, 10, (4,))
    step.step(x, y)
    g_single = model.engine.grads
    g_ddp = r0["grads"] / 2                                  # sum over 2 ranks of per-rank means
    rel = ((g_ddp - g_single).norm() / g_single.norm()).item()
    assert rel < 2e-2, rel                                   # bf16 rounding differs with the batch split, math is identical
    d_single = model.engine.params - p0
    d_ddp = r0["params"] - p0
    rel = ((d_ddp - d_single).norm() / d_single.norm()).item()
    assert rel < 2e-2, rel


# ---- face / CBIR task: 2-rank FaceTrainStep (ConvNeXt backbone + BatchNorm neck + ArcFace) and sharded gallery search --------------------------
# (operand bf16: these tests compare gradients bit for bit across ranks; the fp16 default would start at GradScaler's 65 536 and skip the first tiny-batch steps -- the
# skip / back-off in lockstep across ranks is test_two_gloo_ranks_on_fp16_operands...; the class-sharded head is built for bf16)
FACE_CFG = {"task": "cbir", "image_size": 32, "backbone": {"timm-convnext_test": {"pretrained": False, "image_size": 32, "feat_dim": 64, "operand": "bf16"}},
            "head": {"arcface": {"feat_dim": 64, "num_class": 24, "margin_arc": 0.35, "margin_am": 0.0, "scale": 32}}}


def _face_model(be, seed):
    from visiondk_amd import convnext, face
    convnext.TIMM_CONVNEXTS["convnext_test"] = dict(depths=(1, 1, 1, 1), dims=(8, 16, 24, 32))
    torch.manual_seed(seed)
    model = face.get_model(FACE_CFG, None, 0, backend=be, device="cpu").model.train()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("gamma"):
                p.fill_(0.4)
    return model


def _face_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VDK_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu.emu_backend import load_emu
    from visiondk_amd import cbir, comm, face
    be = load_emu()
    model = _face_model(be, 100 + rank)                      # ranks start different; the step object broadcasts rank 0's weights
    c = comm.GradAllReduce(bucket_bytes=20_000)
    step = face.FaceTrainStep(model, lr=0.05, momentum=0.9, weight_decay=5e-4, max_norm=10.0, ema=True, comm=c)
    init = {k: v.clone() for k, v in model.state_dict().items()}      # after the broadcast: rank 0's weights everywhere
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32); y = torch.randint(0, 24, (8,))
    lo, hi = rank * 4, rank * 4 + 4
    step.step(x[lo:hi], y[lo:hi])
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    # sharded gallery search: rank r holds rows [r*150, r*150+150) of a 300-row gallery and its own 5 queries
    g = torch.Generator().manual_seed(3)
    gal = torch.nn.functional.normalize(torch.randn(300, 32, generator=g)); qry = torch.nn.functional.normalize(torch.randn(10, 32, generator=g))
    s, i = cbir.search_sharded(qry[rank * 5:rank * 5 + 5], gal[rank * 150:rank * 150 + 150], k=7, idx_base=rank * 150, backend=be, device="cpu", cap=200)
    torch.save({"sd": sd, "init": init, "grads": step.eng.grads.clone(), "head_grad": model.trainingwrapper["head"].weight.grad.clone(), "s": s, "i": i}, f"{out_dir}/face{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_face_step_and_sharded_search(tmp_path, emu):
    port = 29500 + ((os.getpid() + 137) % 500)
    mp.start_processes(_face_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "face0.pt"); r1 = torch.load(tmp_path / "face1.pt")
    for k in r0["init"]:
        assert torch.equal(r0["init"][k], r1["init"][k]), k  # broadcast at construction
    for k in r0["sd"]:
        if "running" in k or "num_batches" in k:
            continue                                          # BatchNorm statistics are per-rank between the per-forward broadcasts (torch DDP semantics)
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k      # replicas stay bit-identical
    assert torch.equal(r0["grads"], r1["grads"]) and torch.equal(r0["head_grad"], r1["head_grad"])
    # the all-reduced gradient is the sum of the two local gradients (each computed on its half with rank 0's weights)
    from visiondk_amd import face
    total, total_head = None, None
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32); y = torch.randint(0, 24, (8,))
    for r in range(2):
        model = _face_model(emu, 100)
        model.load_state_dict(r0["init"], strict=True)
        st = face.FaceTrainStep(model, lr=0.0, momentum=0.0, weight_decay=0.0, ema=False)
        st.step(x[r * 4:r * 4 + 4], y[r * 4:r * 4 + 4])
        total = st.eng.grads.clone() if total is None else total + st.eng.grads
        hg = model.trainingwrapper["head"].weight.grad
        total_head = hg.clone() if total_head is None else total_head + hg
    assert torch.equal(total, r0["grads"]) and torch.equal(total_head, r0["head_grad"])
    # sharded search == single search over the whole gallery, bit for bit, each rank holding its own queries' results
    import numpy as np
    from oracle import cbir as ocbir
    g = torch.Generator().manual_seed(3)
    gal = torch.nn.functional.normalize(torch.randn(300, 32, generator=g)); qry = torch.nn.functional.normalize(torch.randn(10, 32, generator=g))
    so, io = ocbir.flat_ip_search(qry.numpy(), gal.numpy(), 7)
    got_s = torch.cat([r0["s"], r1["s"]]).numpy(); got_i = torch.cat([r0["i"], r1["i"]]).numpy()
    np.testing.assert_array_equal(got_i, io)
    np.testing.assert_array_equal(got_s.view(np.uint32), so.view(np.uint32))


# ---- BatchNorm CNN: 2-rank ResNetTrainStep ------------------------------------------------------------------------------------------------------------
def _resnet_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VDK_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu.emu_backend import load_emu
    from visiondk_amd import comm, resnet
    be = load_emu()
    spec = resnet.ResNetSpec(img_size=32, widths=(8, 8, 16, 16), depths=(1, 1, 1, 1), num_classes=5)
    model = resnet.ResNet(spec, device="cpu", backend=be, seed=50 + rank)
    step = resnet.ResNetTrainStep(model, lr=0.05, loss="bce", ema=False, comm=comm.GradAllReduce(bucket_bytes=8_000))
    init = model.engine.params.clone()
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32); t = (torch.rand(8, 5) > 0.5).float()
    step.step(x[rank * 4:rank * 4 + 4], t[rank * 4:rank * 4 + 4])
    torch.save({"init": init, "params": model.engine.params.clone(), "grads": model.engine.grads.clone(), "buffers": model.engine.buffers.clone()}, f"{out_dir}/rn{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_resnet_step(tmp_path, emu):
    port = 29500 + ((os.getpid() + 271) % 500)
    mp.start_processes(_resnet_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rn0.pt"); r1 = torch.load(tmp_path / "rn1.pt")
    assert torch.equal(r0["init"], r1["init"]) and torch.equal(r0["params"], r1["params"]) and torch.equal(r0["grads"], r1["grads"])
    # the reduced gradient == the sum of the two local gradients computed from rank 0's initial weights (BatchNorm statistics are per rank)
    from visiondk_amd import resnet
    spec = resnet.ResNetSpec(img_size=32, widths=(8, 8, 16, 16), depths=(1, 1, 1, 1), num_classes=5)
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32); t = (torch.rand(8, 5) > 0.5).float()
    total = None
    for r in range(2):
        model = resnet.ResNet(spec, device="cpu", backend=emu, seed=0)
        with torch.no_grad():
            model.engine.params.copy_(r0["init"])
        st = resnet.ResNetTrainStep(model, lr=0.0, momentum=0.0, weight_decay=0.0, loss="bce", ema=False)
        st.step(x[r * 4:r * 4 + 4], t[r * 4:r * 4 + 4])
        total = model.engine.grads.clone() if total is None else total + model.engine.grads
    assert torch.equal(total, r0["grads"])


# ---- SyncBatchNorm: 2 ranks x half batch == 1 process x whole batch (statistics all-reduced inside forward and backward) -------------------------------
def _syncbn_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VDK_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu.emu_backend import load_emu
    from visiondk_amd import comm, resnet
    be = load_emu()
    spec = resnet.ResNetSpec(img_size=32, widths=(8, 8, 16, 16), depths=(1, 1, 1, 1), num_classes=5)
    model = resnet.ResNet(spec, device="cpu", backend=be, seed=50 + rank)
    step = resnet.ResNetTrainStep(model, lr=0.05, loss="bce", ema=False, comm=comm.GradAllReduce(bucket_bytes=8_000), sync_bn=True)
    init = model.engine.params.clone()
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32); t = (torch.rand(8, 5) > 0.5).float()
    rows = step.step(x[rank * 4:rank * 4 + 4], t[rank * 4:rank * 4 + 4])
    torch.save({"init": init, "params": model.engine.params.clone(), "grads": model.engine.grads.clone(), "buffers": model.engine.buffers.clone(), "loss": rows.clone()},
               f"{out_dir}/sbn{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_syncbn_equals_single_process(tmp_path, emu):
    port = 29500 + ((os.getpid() + 389) % 500)
    mp.start_processes(_syncbn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "sbn0.pt"); r1 = torch.load(tmp_path / "sbn1.pt")
    assert torch.equal(r0["params"], r1["params"]) and torch.equal(r0["grads"], r1["grads"])
    assert torch.equal(r0["buffers"], r1["buffers"])           # running statistics come from the global batch on every rank
    from visiondk_amd import resnet
    spec = resnet.ResNetSpec(img_size=32, widths=(8, 8, 16, 16), depths=(1, 1, 1, 1), num_classes=5)
    model = resnet.ResNet(spec, device="cpu", backend=emu, seed=0)
    with torch.no_grad():
        model.engine.params.copy_(r0["init"])
    st = resnet.ResNetTrainStep(model, lr=0.05, loss="bce", ema=False)
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32); t = (torch.rand(8, 5) > 0.5).float()
    rows = st.step(x, t)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    # per-rank BCE rows use gscale 1/(4*5); the summed gradient over 2 ranks / 2 is the whole-batch mean gradient
    assert rel(r0["grads"] / 2, model.engine.grads) < 2e-3
    assert rel(r0["buffers"], model.engine.buffers) < 1e-5
    assert rel(r0["params"] - r0["init"], model.engine.params - r0["init"]) < 2e-3
    assert rel(torch.cat([r0["loss"], r1["loss"]]), rows) < 1e-4


# ---- ConvNeXt classifier (no BatchNorm): 2 ranks x half batch == 1 process x whole batch, SAM path included ------------------------------------------
def _convnext_cls_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VDK_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu.emu_backend import load_emu
    from visiondk_amd import comm, convnext, resnet
    be = load_emu()
    spec = convnext.ConvNeXtSpec(img_size=32, depths=(1, 1, 1, 1), dims=(8, 16, 24, 32), num_classes=6)
    out = {}
    for sam in (False, True):
        model = convnext.ConvNeXt(spec, device="cpu", backend=be, seed=70 + rank)
        step = resnet.ClassifierTrainStep(model, lr=0.05, loss="ce", label_smoothing=0.05, ema=False, sam=sam, comm=comm.GradAllReduce(bucket_bytes=4_000))
        init = model.engine.params.clone()
        torch.manual_seed(9)
        x = torch.randn(8, 3, 32, 32); t = torch.randint(0, 6, (8,))
        step.step(x[rank * 4:rank * 4 + 4], t[rank * 4:rank * 4 + 4])
        out[sam] = {"init": init, "params": model.engine.params.clone(), "grads": model.engine.grads.clone()}
    torch.save(out, f"{out_dir}/cn{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_convnext_classifier_step(tmp_path, emu):
    port = 29500 + ((os.getpid() + 389) % 500)
    mp.start_processes(_convnext_cls_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "cn0.pt"); r1 = torch.load(tmp_path / "cn1.pt")
    from visiondk_amd import convnext, resnet
    spec = convnext.ConvNeXtSpec(img_size=32, depths=(1, 1, 1, 1), dims=(8, 16, 24, 32), num_classes=6)
    torch.manual_seed(9)
    x = torch.randn(8, 3, 32, 32); t = torch.randint(0, 6, (8,))
    for sam in (False, True):
        a, b = r0[sam], r1[sam]
        assert torch.equal(a["init"], b["init"]) and torch.equal(a["params"], b["params"]) and torch.equal(a["grads"], b["grads"])
    # plain step: no BatchNorm, so two ranks x 4 samples == one process x 8 samples (mean loss over the global batch: the summed gradient is scaled by 1/world)
    model = convnext.ConvNeXt(spec, device="cpu", backend=emu, seed=0)
    with torch.no_grad():
        model.engine.params.copy_(r0[False]["init"])
    st = resnet.ClassifierTrainStep(model, lr=0.05, loss="ce", label_smoothing=0.05, ema=False)
    st.step(x, t)
    rel = ((model.engine.params - r0[False]["params"]).norm() / (model.engine.params - r0[False]["init"]).norm()).item()
    assert rel < 2e-2, rel


# ---- class-sharded margin head: 2 ranks x (half of the classes, half of the batch) == the full head on the whole batch -----------------------------------
def _sharded_head_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VDK_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu.emu_backend import load_emu
    from visiondk_amd import heads
    be = load_emu()
    D, Cn, B = 64, 96, 6
    torch.manual_seed(21)
    W = torch.randn(D, Cn); feats = torch.randn(2 * B, D); labels = torch.randint(0, Cn, (2 * B,))
    labels[0], labels[B] = 3, 90                                  # targets on both shards for both ranks' samples
    out = {}
    for tag in ("arcface", "mv_arc"):
        head = (heads.ArcFace(D, Cn, backend=be, device="cpu") if tag == "arcface" else heads.MV_Softmax(D, Cn, is_am=False, backend=be, device="cpu"))
        c0 = rank * (Cn // 2)
        loss, df, dW = heads.sharded_margin_ce(head, feats[rank * B:(rank + 1) * B].contiguous(), labels[rank * B:(rank + 1) * B].contiguous(),
                                               W[:, c0:c0 + Cn // 2].contiguous(), c0, Cn, label_smoothing=0.1)
        out[tag] = (loss, df, dW)
    torch.save(out, f"{out_dir}/sh{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_class_sharded_head_equals_full_head(tmp_path, emu):
    port = 29500 + ((os.getpid() + 457) % 500)
    mp.start_processes(_sharded_head_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"sh{i}.pt") for i in range(2)]
    from visiondk_amd import heads
    D, Cn, B = 64, 96, 6
    torch.manual_seed(21)
    W = torch.randn(D, Cn); feats = torch.randn(2 * B, D); labels = torch.randint(0, Cn, (2 * B,))
    labels[0], labels[B] = 3, 90
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    for tag in ("arcface", "mv_arc"):
        head = (heads.ArcFace(D, Cn, backend=emu, device="cpu") if tag == "arcface" else heads.MV_Softmax(D, Cn, is_am=False, backend=emu, device="cpu"))
        with torch.no_grad():
            head.weight.copy_(W)
        # the full head on the whole batch with the same per-sample gradient scale (1 / B_local)
        loss, df, dW = head.margin_ce(feats, labels, label_smoothing=0.1, grad_scale=1.0 / B)
        got_loss = torch.cat([r[0][tag][0], r[1][tag][0]]); got_df = torch.cat([r[0][tag][1], r[1][tag][1]]); got_dW = torch.cat([r[0][tag][2], r[1][tag][2]], 1)
        assert rel(got_loss, loss) < 1e-5, (tag, rel(got_loss, loss))
        assert rel(got_df, df) < 5e-3 and rel(got_dW, dW) < 5e-3, (tag, rel(got_df, df), rel(got_dW, dW))     # bf16 dcos planes, different split of the class sum


# ---- FaceTrainStep(shard_head=True) == FaceTrainStep with the replicated head: same weights after a step --------------------------------------------------
def _face_shard_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VDK_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu.emu_backend import load_emu
    from visiondk_amd import comm, face
    be = load_emu()
    out = {}
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32); y = torch.randint(0, 24, (8,))
    lo, hi = rank * 4, rank * 4 + 4
    init = None
    for shard in (False, True):
        model = _face_model(be, 100)
        if init is None:
            init = {k: v.clone() for k, v in model.state_dict().items()}
        model.load_state_dict(init)                            # the same initial weights for both variants (the backbone's own init is unseeded)
        step = face.FaceTrainStep(model, lr=0.05, momentum=0.9, weight_decay=5e-4, max_norm=0.5, ema=True, comm=comm.GradAllReduce(bucket_bytes=20_000),
                                  shard_head=shard, layer_wise=True)
        rows = step.step(x[lo:hi], y[lo:hi])
        head_w = step.gather_head().detach().clone()
        head_ema = step.gather_head(ema=True).detach().clone() if shard else step.ema_small[-1].clone()
        out[shard] = {"rows": rows.clone(), "params": step.eng.params.clone(), "head": head_w, "head_ema": head_ema,
                      "neck": [p.detach().clone() for p in step.bb.output_layer.parameters()]}
    torch.save(out, f"{out_dir}/fs{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_face_step_with_class_sharded_head(tmp_path, emu):
    port = 29500 + ((os.getpid() + 523) % 500)
    mp.start_processes(_face_shard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "fs0.pt"); r1 = torch.load(tmp_path / "fs1.pt")
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    for r in (r0, r1):
        a, b = r[True], r[False]
        assert rel(a["rows"], b["rows"]) < 1e-5                                   # same loss rows
        assert rel(a["params"], b["params"]) < 1e-4 and rel(a["head"], b["head"]) < 1e-4 and rel(a["head_ema"], b["head_ema"]) < 1e-5
        for p, q in zip(a["neck"], b["neck"]):
            assert rel(p, q) < 1e-3
    assert torch.equal(r0[True]["head"], r1[True]["head"]) and torch.equal(r0[True]["params"], r1[True]["params"])   # replicas agree after the gather


# ---- SyncBatchNorm in the embedding neck: 2 ranks x half batch == 1 process x whole batch ----------------------------------------------------------
def _face_syncbn_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VDK_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu.emu_backend import load_emu
    from visiondk_amd import comm, face
    be = load_emu()
    model = _face_model(be, 100)
    c = comm.GradAllReduce(bucket_bytes=20_000)
    step = face.FaceTrainStep(model, lr=0.05, momentum=0.9, weight_decay=5e-4, max_norm=10.0, ema=False, comm=c, sync_bn=True)
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32); y = torch.randint(0, 24, (8,))
    step.step(x[rank * 4:rank * 4 + 4], y[rank * 4:rank * 4 + 4])
    torch.save({k: v.clone() for k, v in model.state_dict().items()}, f"{out_dir}/sbn{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_face_step_with_sync_batchnorm_equals_whole_batch(tmp_path, emu):
    """FaceTrainStep(sync_bn=True): the neck's BatchNorm2d and BatchNorm1d use the statistics of BOTH ranks' samples (forward) and the global gradient sums (backward),
    so two ranks on half the batch each leave the weights -- and the running statistics -- a single process on the whole batch leaves (the ConvNeXt backbone has
    no BatchNorm).  Without sync_bn the statistics are per rank and the running statistics differ at 1e-2."""
    port = 29500 + ((os.getpid() + 271) % 500)
    mp.start_processes(_face_syncbn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "sbn0.pt"); r1 = torch.load(tmp_path / "sbn1.pt")
    from visiondk_amd import face
    model = _face_model(emu, 100)
    init = {k: v.clone() for k, v in model.state_dict().items()}
    step = face.FaceTrainStep(model, lr=0.05, momentum=0.9, weight_decay=5e-4, max_norm=10.0, ema=False)
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32); y = torch.randint(0, 24, (8,))
    step.step(x, y)
    sd = model.state_dict()
    for k in sd:
        if "num_batches" in k or k.endswith("model.head.norm.bias"):
            continue      # head.norm.bias sits right in front of the BatchNorm2d, which removes per-channel shifts: its true gradient is 0, what is computed is rounding noise
        a, b = r0[k].float(), sd[k].float()
        assert torch.equal(r0[k], r1[k]) or "running" in k, k            # replicas identical (running statistics are identical too: they come from the global sums)
        if "running" in k:
            assert torch.allclose(r0[k], r1[k]) and torch.allclose(a, b, rtol=1e-4, atol=1e-5), k
        else:
            da, db = a - init[k].float(), b - init[k].float()
            rel = ((da - db).norm() / db.norm().clamp_min(1e-12)).item()
            assert rel < 3e-2, (k, rel)


# ---- SigLIP-family model (no class token + attention-pool head): MapTrainStep over 2 ranks, plain and SAM -----------------------------------------
def _map_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VDK_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu.emu_backend import load_emu
    from visiondk_amd import comm, vit
    be = load_emu()
    out = {}
    for sam in (False, True):
        torch.manual_seed(100 + rank)                     # ranks start DIFFERENT on purpose: the constructor broadcasts rank 0's weights (trunk and head)
        model = vit.VisionTransformerMap(vit.VitSpec(img_size=32, patch_size=8, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256, class_token=False), device="cpu",
                                         backend=be, seed=100 + rank)
        c = comm.GradAllReduce(bucket_bytes=200_000)
        step = vit.MapTrainStep(model, lr=0.02, momentum=0.9, weight_decay=5e-4, label_smoothing=0.05, ema=False, sam=sam, comm=c)
        torch.manual_seed(7)
        x = torch.randn(4, 3, 32, 32); y = torch.randint(0, 10, (4,))
        step.step(x[rank * 2:rank * 2 + 2], y[rank * 2:rank * 2 + 2])
        out[sam] = {"params": step.big.clone(), "grads": step.gbig.clone(), "collectives": c.collectives}
    torch.save(out, f"{out_dir}/rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_map_step_equals_single_process(tmp_path, emu):
    port = 29500 + (os.getpid() % 500)
    mp.start_processes(_map_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rank0.pt"); r1 = torch.load(tmp_path / "rank1.pt")
    from visiondk_amd import vit
    for sam in (False, True):
        assert torch.equal(r0[sam]["params"], r1[sam]["params"]) and torch.equal(r0[sam]["grads"], r1[sam]["grads"])      # replicas bit-identical
        assert r0[sam]["collectives"] >= 3                   # broadcast + the head bucket + at least one trunk bucket
        torch.manual_seed(100)
        model = vit.VisionTransformerMap(vit.VitSpec(img_size=32, patch_size=8, num_classes=10, dim=128, depth=2, heads=2, mlp_dim=256, class_token=False), device="cpu",
                                         backend=emu, seed=100)
        step = vit.MapTrainStep(model, lr=0.02, momentum=0.9, weight_decay=5e-4, label_smoothing=0.05, ema=False, sam=sam)
        p0 = step.big.clone()
        torch.manual_seed(7)
        x = torch.randn(4, 3, 32, 32); y = torch.randint(0, 10, (4,))
        step.step(x, y)
        d_single = step.big - p0; d_ddp = r0[sam]["params"] - p0
        rel = ((d_ddp - d_single).norm() / d_single.norm()).item()
        assert rel < (3e-2 if not sam else 2e-1), (sam, rel)      # SAM: the first pass is local (per-rank e(w)), as in the reference's no_sync()


# ---- bench.py's own multi-rank code: the two legs the driver's `--gpus N` run executes, on 2 CPU ranks over gloo ---------------------------------------
def _bench_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VDK_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from tests.emu.emu_backend import load_emu
    from tests.test_vit import SPEC
    be = load_emu()
    dev = torch.device("cpu")
    dt, loss, gemm, ncoll = bench.train_leg(be, dev, rank, world, steps=2, warmup=1, batch=2, spec=SPEC, img=32, classes=10, bucket_bytes=200_000)
    cb = bench.cbir_sharded_leg(be, dev, rank, world, nq=10, n=301, d=32, k=7, iters=2, warm=1, cap=200, check_queries=5)
    torch.save({"dt": dt, "loss": loss, "gemm": gemm, "ncoll": ncoll, "cbir": cb}, f"{out_dir}/bench{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_bench_multi_rank_legs_on_two_gloo_ranks(tmp_path, emu):
    port = 29500 + ((os.getpid() + 389) % 500)
    mp.start_processes(_bench_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "bench0.pt", weights_only=False); r1 = torch.load(tmp_path / "bench1.pt", weights_only=False)
    assert r0["dt"] == r1["dt"] > 0                          # MAX over ranks: both report the same time
    assert r0["ncoll"] == r1["ncoll"] and r0["ncoll"] >= 1 + 3 * 2      # the broadcast + at least two buckets in each of the 3 steps
    assert r0["loss"] > 0 and r0["gemm"] is None             # (HIP events only on the GPU)
    c0 = r0["cbir"]
    assert c0["n_gpus"] == 2 and c0["scaling"] == "strong" and c0["value"] > 0 and r1["cbir"]["value"] == c0["value"]
    assert c0["parity_vs_oracle"] == {"queries": 5, "gallery_rows": 301, "indices_equal": True, "scores_bit_equal": True}
    assert "parity_vs_oracle" not in r1["cbir"]


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus 2` with WORLD_SIZE unset re-executes itself under torch.distributed.run (one rank per GPU)"""
    import subprocess
    import bench
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd: seen.setdefault("cmd", cmd) and 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "2", "--steps", "3"] and cmd[-5].endswith("bench.py")


def test_bucket_schedule_tapers_at_the_gradient_tail(monkeypatch):
    """GradAllReduce's bucket rule on ViT-B/16-like ready ranges (head, 12 blocks of 7.09 M floats, embeddings): every block its own collective, and nothing
    large is left for finish_step (the only collective the clipped SGD step waits for with nothing left to hide it)."""
    from visiondk_amd import comm
    sizes = []

    class _W:
        def wait(self): pass

    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda g=None: 8)
    monkeypatch.setattr(dist, "get_rank", lambda g=None: 0)
    monkeypatch.setattr(dist, "all_reduce", lambda t, op=None, group=None, async_op=False: sizes.append(t.numel()) or _W())
    c = comm.GradAllReduce()
    blk, stem, head = 7_087_872, 742_656, 770_536
    total = stem + 12 * blk + head
    c.begin_step(torch.empty(total))
    off = total - head
    c.on_grad_ready(off, head)
    for _ in range(12):
        off -= blk
        c.on_grad_ready(off, blk)
    before_tail = list(sizes)
    c.on_grad_ready(0, stem)
    c.finish_step()
    assert sum(sizes) == total
    assert before_tail == [head + blk] + [blk] * 11          # the last block left as soon as it was ready ...
    assert sizes[len(before_tail):] == [stem]                # ... and only the 3 MB of embedding gradients are exposed
    # small ranges: the tail halves (a bucket closes when no more than its own size is still to come)
    sizes.clear()
    c = comm.GradAllReduce(bucket_bytes=1 << 30)
    c.begin_step(torch.empty(1024))
    for o in range(1023, -1, -1):
        c.on_grad_ready(o, 1)
    c.finish_step()
    assert sizes == [512, 256, 128, 64, 32, 16, 8, 4, 2, 1, 1]


# ---- Swin (the reference's default backbone) under the reference's own wrapping: torch DistributedDataParallel over the module's autograd nodes ------------------------
def _swin_spec():
    from visiondk_amd import swin
    return swin.SwinSpec(img_size=224, num_classes=5, embed_dim=32, depths=(2, 1), heads=(1, 2))


def _swin_worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["VDK_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.emu.emu_backend import load_emu
    from visiondk_amd import swin
    be = load_emu()
    model = swin.SwinTransformer(_swin_spec(), device="cpu", backend=be, seed=200 + rank)      # ranks start DIFFERENT: DDP broadcasts rank 0's weights
    ddp = torch.nn.parallel.DistributedDataParallel(model)                                      # engine/vision_engine.py:313
    opt = torch.optim.SGD(ddp.parameters(), lr=0.05, momentum=0.9)
    torch.manual_seed(9)
    x = torch.randn(2, 3, 224, 224); y = torch.randint(0, 5, (2,))
    loss = torch.nn.functional.cross_entropy(ddp(x[rank:rank + 1]), y[rank:rank + 1])
    loss.backward()
    opt.step()
    torch.save({"params": {n: p.detach().clone() for n, p in model.named_parameters()}, "grads": {n: p.grad.clone() for n, p in model.named_parameters()}},
               f"{out_dir}/swin{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_swin_under_torch_ddp_equals_whole_batch(tmp_path, emu):
    """every parameter of the Swin module gets its gradient through the autograd nodes (DDP would hang or raise on an unused one), the replicas stay identical, and the
    averaged gradient equals the whole-batch gradient of a single process"""
    port = 29500 + (os.getpid() % 500) + 37
    mp.start_processes(_swin_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "swin0.pt"); r1 = torch.load(tmp_path / "swin1.pt")
    for n in r0["params"]:
        assert torch.equal(r0["params"][n], r1["params"][n]), n
        assert torch.equal(r0["grads"][n], r1["grads"][n]), n
    from visiondk_amd import swin
    model = swin.SwinTransformer(_swin_spec(), device="cpu", backend=emu, seed=200)
    torch.manual_seed(9)
    x = torch.randn(2, 3, 224, 224); y = torch.randint(0, 5, (2,))
    torch.nn.functional.cross_entropy(model(x), y).backward()
    gmax = max(p.grad.norm().item() for p in model.parameters())
    for n, p in model.named_parameters():
        if p.grad.norm().item() < 1e-3 * gmax:
            continue
        rel = ((r0["grads"][n] - p.grad).norm() / p.grad.norm()).item()
        assert rel < 3e-2, (n, rel)                          # bf16 rounding differs with the batch split, the mathematics is identical
