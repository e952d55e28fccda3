"""The N > 1 DEVICE path on hardware with the one MI355X a test box has: two processes, BOTH on cuda:0, exchanging over gloo (RCCL refuses two ranks on one device).
What tests/test_ddp_gloo.py proves on the CPU emulation -- replicas bit-identical, the all-reduced gradient = the sum of the local ones, two ranks on half batches = one
process on the whole batch, fp16 skip / back-off in lockstep, the sharded search bit-equal to the whole-gallery oracle -- is repeated here on the product library: the
stream-ordered bucket exchange issued from inside the native backward (events on the launch stream, collectives on their own stream), the device-side optimizer / EMA /
GradScaler kernels behind it, and `cbir.search_sharded`'s all-gather / all-to-all / merge.  (RCCL itself: tests/test_rccl_gpu.py at world size 1.)
Reference semantics: torch DDP at engine/vision_engine.py:313,510; sharded gallery: SURVEY.md 8(e)."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _init(rank, world, port):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from visiondk_amd import _lib
    return _lib.load()


def _vit_worker(rank, world, port, out_dir, operand):
    be = _init(rank, world, port)
    from tests.test_vit import SPEC
    from visiondk_amd import comm, vit
    model = vit.VisionTransformer(SPEC, device=DEV, backend=be, seed=100 + rank, operand=operand)      # ranks start DIFFERENT on purpose
    c = comm.GradAllReduce(bucket_bytes=200_000)                                                         # small buckets -> several collectives inside the backward
    c.broadcast_params(model.engine.params, src=0, engine=model.engine)
    torch.manual_seed(7)
    x = torch.randn(4, 3, 32, 32).to(DEV); y = torch.randint(0, 10, (4,)).to(DEV)
    lo, hi = rank * 2, rank * 2 + 2
    out = {}
    if operand == "fp16":      # the first step overflows on purpose (scale 2^30): BOTH ranks must skip it and halve their scale
        step = vit.FusedTrainStep(model, lr=0.01, label_smoothing=0.05, ema=(rank == 0), comm=c, init_scale=2.0 ** 30)
        p0 = model.engine.params.clone()
        step.step(x[lo:hi], y[lo:hi])
        out["skipped_first"] = bool(torch.equal(model.engine.params, p0)) and step.skipped_steps() == 1
        step.loss_state[0] = 1024.0
        step.step(x[lo:hi], y[lo:hi])
        out.update(skipped=step.skipped_steps(), scale=step.loss_scale())
    else:
        step = vit.FusedTrainStep(model, lr=0.01, label_smoothing=0.05, ema=False, comm=c)
        step.step(x[lo:hi], y[lo:hi])
        out["grads"] = model.engine.grads.cpu()
    out["params"] = model.engine.params.cpu(); out["collectives"] = c.collectives
    torch.save(out, f"{out_dir}/rank{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_gpu_equal_a_single_process_on_the_whole_batch(tmp_path, hip):
    port = _free_port()
    mp.start_processes(_vit_worker, args=(2, port, str(tmp_path), "bf16"), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rank0.pt"); r1 = torch.load(tmp_path / "rank1.pt")
    assert r0["collectives"] >= 3                              # the gradient really left in buckets
    assert torch.equal(r0["params"], r1["params"])            # replicas stay bit-identical
    assert torch.equal(r0["grads"], r1["grads"])              # both hold the same summed gradient
    from tests.test_vit import SPEC
    from visiondk_amd import vit
    model = vit.VisionTransformer(SPEC, device=DEV, backend=hip, seed=100)
    p0 = model.engine.params.cpu()
    step = vit.FusedTrainStep(model, lr=0.01, label_smoothing=0.05, ema=False)
    torch.manual_seed(7)
    x = torch.randn(4, 3, 32, 32).to(DEV); y = torch.randint(0, 10, (4,)).to(DEV)
    step.step(x, y)
    g_single = model.engine.grads.cpu()
    rel = (((r0["grads"] / 2) - g_single).norm() / g_single.norm()).item()      # sum over 2 ranks of per-rank means
    assert rel < 2e-2, rel                                     # bf16 rounding differs with the batch split, the math is identical
    d_single = model.engine.params.cpu() - p0
    rel = (((r0["params"] - p0) - d_single).norm() / d_single.norm()).item()
    assert rel < 2e-2, rel


def test_two_ranks_on_the_gpu_fp16_skip_and_step_in_lockstep(tmp_path, hip):
    """the inf check runs on the ALL-REDUCED gradient inside the optimizer kernel: every rank takes the same skip / step decision (train.py:205-211 under DDP)"""
    port = _free_port()
    mp.start_processes(_vit_worker, args=(2, port, str(tmp_path), "fp16"), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rank0.pt"); r1 = torch.load(tmp_path / "rank1.pt")
    assert r0["skipped_first"] and r1["skipped_first"]
    assert r0["skipped"] == r1["skipped"] == 1 and r0["scale"] == r1["scale"]
    assert torch.equal(r0["params"], r1["params"]) and torch.isfinite(r0["params"]).all()


def _face_worker(rank, world, port, out_dir):
    be = _init(rank, world, port)
    from tests.test_ddp_gloo import FACE_CFG
    from visiondk_amd import cbir, comm, convnext, face
    convnext.TIMM_CONVNEXTS["convnext_test"] = dict(depths=(1, 1, 1, 1), dims=(8, 16, 24, 32))
    torch.manual_seed(100 + rank)                              # ranks start different; the step object broadcasts rank 0's weights
    model = face.get_model(FACE_CFG, None, 0, backend=be, device=DEV).model.train()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("gamma"):
                p.fill_(0.4)
    c = comm.GradAllReduce(bucket_bytes=20_000)
    step = face.FaceTrainStep(model, lr=0.05, momentum=0.9, weight_decay=5e-4, max_norm=10.0, ema=True, comm=c)
    init = {k: v.cpu().clone() for k, v in model.state_dict().items()}
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32).to(DEV); y = torch.randint(0, 24, (8,)).to(DEV)
    lo, hi = rank * 4, rank * 4 + 4
    step.step(x[lo:hi], y[lo:hi])
    sd = {k: v.cpu().clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    gal = torch.nn.functional.normalize(torch.randn(300, 32, generator=g)); qry = torch.nn.functional.normalize(torch.randn(10, 32, generator=g))
    s, i = cbir.search_sharded(qry[rank * 5:rank * 5 + 5].to(DEV), gal[rank * 150:rank * 150 + 150].to(DEV), k=7, idx_base=rank * 150, backend=be, device=DEV, cap=200)
    torch.save({"sd": sd, "init": init, "grads": step.eng.grads.cpu(), "head_grad": model.trainingwrapper["head"].weight.grad.cpu(), "s": s.cpu(), "i": i.cpu()},
               f"{out_dir}/face{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_gpu_face_step_and_sharded_search(tmp_path, hip):
    port = _free_port()
    mp.start_processes(_face_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "face0.pt"); r1 = torch.load(tmp_path / "face1.pt")
    for k in r0["init"]:
        assert torch.equal(r0["init"][k], r1["init"][k]), k    # broadcast at construction
    for k in r0["sd"]:
        if "running" in k or "num_batches" in k:
            continue                                            # BatchNorm statistics are per-rank between the per-forward broadcasts (torch DDP semantics)
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k        # replicas stay bit-identical
    assert torch.equal(r0["grads"], r1["grads"]) and torch.equal(r0["head_grad"], r1["head_grad"])
    # the all-reduced gradient is the sum of the two local gradients (each computed on its half with rank 0's weights)
    from tests.test_ddp_gloo import FACE_CFG
    from visiondk_amd import convnext, face
    convnext.TIMM_CONVNEXTS["convnext_test"] = dict(depths=(1, 1, 1, 1), dims=(8, 16, 24, 32))
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32).to(DEV); y = torch.randint(0, 24, (8,)).to(DEV)
    total, total_head = None, None
    for r in range(2):
        torch.manual_seed(100)
        model = face.get_model(FACE_CFG, None, 0, backend=hip, device=DEV).model.train()
        model.load_state_dict({k: v.to(DEV) for k, v in r0["init"].items()}, strict=True)
        st = face.FaceTrainStep(model, lr=0.0, momentum=0.0, weight_decay=0.0, ema=False)
        st.step(x[r * 4:r * 4 + 4], y[r * 4:r * 4 + 4])
        total = st.eng.grads.cpu().clone() if total is None else total + st.eng.grads.cpu()
        hg = model.trainingwrapper["head"].weight.grad.cpu()
        total_head = hg.clone() if total_head is None else total_head + hg
    assert torch.equal(total, r0["grads"]) and torch.equal(total_head, r0["head_grad"])
    # sharded search == the oracle's single search over the whole gallery, bit for bit, each rank holding its own queries' results
    import numpy as np
    from oracle import cbir as ocbir
    g = torch.Generator().manual_seed(3)
    gal = torch.nn.functional.normalize(torch.randn(300, 32, generator=g)); qry = torch.nn.functional.normalize(torch.randn(10, 32, generator=g))
    so, io = ocbir.flat_ip_search(qry.numpy(), gal.numpy(), 7)
    got_s = torch.cat([r0["s"], r1["s"]]).numpy(); got_i = torch.cat([r0["i"], r1["i"]]).numpy()
    np.testing.assert_array_equal(got_i, io)
    np.testing.assert_array_equal(got_s.view(np.uint32), so.view(np.uint32))


# ---- BatchNorm CNN: ResNetTrainStep, per-rank statistics and SyncBatchNorm -------------------------------------------------------------------------------------------
def _resnet_worker(rank, world, port, out_dir, sync_bn):
    be = _init(rank, world, port)
    from visiondk_amd import comm, resnet
    spec = resnet.ResNetSpec(img_size=32, widths=(8, 8, 16, 16), depths=(1, 1, 1, 1), num_classes=5)
    model = resnet.ResNet(spec, device=DEV, backend=be, seed=50 + rank)
    step = resnet.ResNetTrainStep(model, lr=0.05, loss="bce", ema=False, comm=comm.GradAllReduce(bucket_bytes=8_000), sync_bn=sync_bn)
    init = model.engine.params.cpu()
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32).to(DEV); t = (torch.rand(8, 5) > 0.5).float().to(DEV)
    rows = step.step(x[rank * 4:rank * 4 + 4], t[rank * 4:rank * 4 + 4])
    torch.save({"init": init, "params": model.engine.params.cpu(), "grads": model.engine.grads.cpu(), "buffers": model.engine.buffers.cpu(), "loss": rows.cpu()},
               f"{out_dir}/rn{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def _resnet_single(hip, init, lr):
    from visiondk_amd import resnet
    spec = resnet.ResNetSpec(img_size=32, widths=(8, 8, 16, 16), depths=(1, 1, 1, 1), num_classes=5)
    model = resnet.ResNet(spec, device=DEV, backend=hip, seed=0)
    with torch.no_grad():
        model.engine.params.copy_(init.to(DEV))
    st = resnet.ResNetTrainStep(model, lr=lr, momentum=0.0 if lr == 0.0 else 0.9, weight_decay=0.0 if lr == 0.0 else 5e-4, loss="bce", ema=False) if lr == 0.0 else \
        resnet.ResNetTrainStep(model, lr=lr, loss="bce", ema=False)
    return model, st


def test_two_ranks_on_the_gpu_resnet_step(tmp_path, hip):
    port = _free_port()
    mp.start_processes(_resnet_worker, args=(2, port, str(tmp_path), False), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rn0.pt"); r1 = torch.load(tmp_path / "rn1.pt")
    assert torch.equal(r0["init"], r1["init"]) and torch.equal(r0["params"], r1["params"]) and torch.equal(r0["grads"], r1["grads"])
    # the reduced gradient == the sum of the two local gradients computed from rank 0's initial weights (BatchNorm statistics are per rank)
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32).to(DEV); t = (torch.rand(8, 5) > 0.5).float().to(DEV)
    total = None
    for r in range(2):
        model, st = _resnet_single(hip, r0["init"], 0.0)
        st.step(x[r * 4:r * 4 + 4], t[r * 4:r * 4 + 4])
        total = model.engine.grads.cpu().clone() if total is None else total + model.engine.grads.cpu()
    assert torch.equal(total, r0["grads"])


def test_two_ranks_on_the_gpu_syncbn_equals_single_process(tmp_path, hip):
    port = _free_port()
    mp.start_processes(_resnet_worker, args=(2, port, str(tmp_path), True), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "rn0.pt"); r1 = torch.load(tmp_path / "rn1.pt")
    assert torch.equal(r0["params"], r1["params"]) and torch.equal(r0["grads"], r1["grads"])
    assert torch.equal(r0["buffers"], r1["buffers"])           # running statistics come from the global batch on every rank
    model, st = _resnet_single(hip, r0["init"], 0.05)
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32).to(DEV); t = (torch.rand(8, 5) > 0.5).float().to(DEV)
    rows = st.step(x, t).cpu()
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    assert rel(r0["grads"] / 2, model.engine.grads.cpu()) < 2e-3
    assert rel(r0["buffers"], model.engine.buffers.cpu()) < 1e-5
    assert rel(r0["params"] - r0["init"], model.engine.params.cpu() - r0["init"]) < 2e-3
    assert rel(torch.cat([r0["loss"], r1["loss"]]), rows) < 1e-4


# ---- the class-sharded margin head (the [D, C] gradient is never all-reduced) -----------------------------------------------------------------------------
def _sharded_head_worker(rank, world, port, out_dir):
    be = _init(rank, world, port)
    from visiondk_amd import heads
    D, Cn, B = 64, 96, 6
    torch.manual_seed(21)
    W = torch.randn(D, Cn); feats = torch.randn(2 * B, D); labels = torch.randint(0, Cn, (2 * B,))
    labels[0], labels[B] = 3, 90                                  # targets on both shards for both ranks' samples
    out = {}
    for tag in ("arcface", "mv_arc"):
        head = (heads.ArcFace(D, Cn, backend=be, device=DEV) if tag == "arcface" else heads.MV_Softmax(D, Cn, is_am=False, backend=be, device=DEV))
        c0 = rank * (Cn // 2)
        loss, df, dW = heads.sharded_margin_ce(head, feats[rank * B:(rank + 1) * B].contiguous().to(DEV), labels[rank * B:(rank + 1) * B].contiguous().to(DEV),
                                               W[:, c0:c0 + Cn // 2].contiguous().to(DEV), c0, Cn, label_smoothing=0.1)
        out[tag] = (loss.cpu(), df.cpu(), dW.cpu())
    torch.save(out, f"{out_dir}/sh{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_gpu_class_sharded_head_equals_full_head(tmp_path, hip):
    port = _free_port()
    mp.start_processes(_sharded_head_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"sh{i}.pt") for i in range(2)]
    from visiondk_amd import heads
    D, Cn, B = 64, 96, 6
    torch.manual_seed(21)
    W = torch.randn(D, Cn); feats = torch.randn(2 * B, D); labels = torch.randint(0, Cn, (2 * B,))
    labels[0], labels[B] = 3, 90
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    for tag in ("arcface", "mv_arc"):
        head = (heads.ArcFace(D, Cn, backend=hip, device=DEV) if tag == "arcface" else heads.MV_Softmax(D, Cn, is_am=False, backend=hip, device=DEV))
        with torch.no_grad():
            head.weight.copy_(W.to(DEV))
        loss, df, dW = head.margin_ce(feats.to(DEV), labels.to(DEV), label_smoothing=0.1, grad_scale=1.0 / B)      # the full head, same per-sample gradient scale (1 / B_local)
        got_loss = torch.cat([r[0][tag][0], r[1][tag][0]]); got_df = torch.cat([r[0][tag][1], r[1][tag][1]]); got_dW = torch.cat([r[0][tag][2], r[1][tag][2]], 1)
        assert rel(got_loss, loss.cpu()) < 1e-5, (tag, rel(got_loss, loss.cpu()))
        assert rel(got_df, df.cpu()) < 5e-3 and rel(got_dW, dW.cpu()) < 5e-3, (tag, rel(got_df, df.cpu()), rel(got_dW, dW.cpu()))


# ---- ConvNeXt classifier (no BatchNorm): 2 ranks x half batch == 1 process x whole batch, SAM path included --------------------------------------------------------------
def _convnext_cls_worker(rank, world, port, out_dir):
    be = _init(rank, world, port)
    from visiondk_amd import comm, convnext, resnet
    spec = convnext.ConvNeXtSpec(img_size=32, depths=(1, 1, 1, 1), dims=(8, 16, 24, 32), num_classes=6)
    out = {}
    for sam in (False, True):
        model = convnext.ConvNeXt(spec, device=DEV, backend=be, seed=70 + rank)
        step = resnet.ClassifierTrainStep(model, lr=0.05, loss="ce", label_smoothing=0.05, ema=False, sam=sam, comm=comm.GradAllReduce(bucket_bytes=4_000))
        init = model.engine.params.cpu()
        torch.manual_seed(9)
        x = torch.randn(8, 3, 32, 32).to(DEV); t = torch.randint(0, 6, (8,)).to(DEV)
        step.step(x[rank * 4:rank * 4 + 4], t[rank * 4:rank * 4 + 4])
        out[sam] = {"init": init, "params": model.engine.params.cpu(), "grads": model.engine.grads.cpu()}
    torch.save(out, f"{out_dir}/cn{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_gpu_convnext_classifier_step(tmp_path, hip):
    port = _free_port()
    mp.start_processes(_convnext_cls_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "cn0.pt"); r1 = torch.load(tmp_path / "cn1.pt")
    from visiondk_amd import convnext, resnet
    spec = convnext.ConvNeXtSpec(img_size=32, depths=(1, 1, 1, 1), dims=(8, 16, 24, 32), num_classes=6)
    torch.manual_seed(9)
    x = torch.randn(8, 3, 32, 32).to(DEV); t = torch.randint(0, 6, (8,)).to(DEV)
    for sam in (False, True):
        a, b = r0[sam], r1[sam]
        assert torch.equal(a["init"], b["init"]) and torch.equal(a["params"], b["params"]) and torch.equal(a["grads"], b["grads"])
    model = convnext.ConvNeXt(spec, device=DEV, backend=hip, seed=0)
    with torch.no_grad():
        model.engine.params.copy_(r0[False]["init"].to(DEV))
    st = resnet.ClassifierTrainStep(model, lr=0.05, loss="ce", label_smoothing=0.05, ema=False)
    st.step(x, t)
    p = model.engine.params.cpu()
    rel = ((p - r0[False]["params"]).norm() / (p - r0[False]["init"]).norm()).item()
    assert rel < 2e-2, rel


# ---- Swin (the default backbone of both shipped YAMLs): the native engine under the fused step, buckets leaving from inside vdk_swin_backward ------------------------------
def _swin_spec():
    from visiondk_amd import swin
    return swin.SwinSpec(img_size=224, num_classes=5, embed_dim=32, depths=(2, 1), heads=(1, 2))


def _swin_worker(rank, world, port, out_dir):
    be = _init(rank, world, port)
    from visiondk_amd import comm, swin, vit
    model = swin.SwinTransformer(_swin_spec(), device=DEV, backend=be, seed=200 + rank, drop_path_rate=0.0)      # ranks start DIFFERENT: the step broadcasts rank 0's weights
    c = comm.GradAllReduce(bucket_bytes=50_000)
    step = vit.FusedTrainStep(model, lr=0.05, momentum=0.9, weight_decay=0.0, label_smoothing=0.0, max_norm=10.0, ema=False, comm=c)
    init = model.engine.params.cpu()
    torch.manual_seed(9)
    x = torch.randn(4, 3, 224, 224).to(DEV); y = torch.randint(0, 5, (4,)).to(DEV)
    step.step(x[rank * 2:rank * 2 + 2], y[rank * 2:rank * 2 + 2])
    torch.save({"init": init, "params": model.engine.params.cpu(), "grads": model.engine.grads.cpu(), "collectives": c.collectives}, f"{out_dir}/swin{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_gpu_native_swin_step_equals_whole_batch(tmp_path, hip):
    port = _free_port()
    mp.start_processes(_swin_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "swin0.pt"); r1 = torch.load(tmp_path / "swin1.pt")
    assert r0["collectives"] >= 2
    assert torch.equal(r0["init"], r1["init"]) and torch.equal(r0["params"], r1["params"]) and torch.equal(r0["grads"], r1["grads"])
    from visiondk_amd import swin, vit
    model = swin.SwinTransformer(_swin_spec(), device=DEV, backend=hip, seed=200, drop_path_rate=0.0)
    with torch.no_grad():
        model.engine.params.copy_(r0["init"].to(DEV))
    model.engine.refresh_weights()
    step = vit.FusedTrainStep(model, lr=0.05, momentum=0.9, weight_decay=0.0, label_smoothing=0.0, max_norm=10.0, ema=False)
    torch.manual_seed(9)
    x = torch.randn(4, 3, 224, 224).to(DEV); y = torch.randint(0, 5, (4,)).to(DEV)
    step.step(x, y)
    g = model.engine.grads.cpu()
    rel = (((r0["grads"] / 2) - g).norm() / g.norm()).item()
    assert rel < 3e-2, rel                                     # bf16 rounding differs with the batch split, the mathematics is identical
    d1 = model.engine.params.cpu() - r0["init"]; d2 = r0["params"] - r0["init"]
    rel = ((d2 - d1).norm() / d1.norm()).item()
    assert rel < 3e-2, rel


# ---- FaceTrainStep(shard_head=True) == FaceTrainStep with the replicated head; SyncBatchNorm in the embedding neck ---------------------------------------------------------
def _face_model(be, seed):
    from tests.test_ddp_gloo import FACE_CFG
    from visiondk_amd import convnext, face
    convnext.TIMM_CONVNEXTS["convnext_test"] = dict(depths=(1, 1, 1, 1), dims=(8, 16, 24, 32))
    torch.manual_seed(seed)
    model = face.get_model(FACE_CFG, None, 0, backend=be, device=DEV).model.train()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("gamma"):
                p.fill_(0.4)
    return model


def _face_shard_worker(rank, world, port, out_dir):
    be = _init(rank, world, port)
    from visiondk_amd import comm, face
    out = {}
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32).to(DEV); y = torch.randint(0, 24, (8,)).to(DEV)
    lo, hi = rank * 4, rank * 4 + 4
    init = None
    for shard in (False, True):
        model = _face_model(be, 100)
        if init is None:
            init = {k: v.clone() for k, v in model.state_dict().items()}
        model.load_state_dict(init)                            # the same initial weights for both variants
        step = face.FaceTrainStep(model, lr=0.05, momentum=0.9, weight_decay=5e-4, max_norm=0.5, ema=True, comm=comm.GradAllReduce(bucket_bytes=20_000),
                                  shard_head=shard, layer_wise=True)
        rows = step.step(x[lo:hi], y[lo:hi])
        head_w = step.gather_head().detach().cpu()
        head_ema = step.gather_head(ema=True).detach().cpu() if shard else step.ema_small[-1].cpu()
        out[shard] = {"rows": rows.cpu(), "params": step.eng.params.cpu(), "head": head_w, "head_ema": head_ema,
                      "neck": [p.detach().cpu() for p in step.bb.output_layer.parameters()]}
    torch.save(out, f"{out_dir}/fs{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_gpu_face_step_with_class_sharded_head(tmp_path, hip):
    port = _free_port()
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


def _face_syncbn_worker(rank, world, port, out_dir):
    be = _init(rank, world, port)
    from visiondk_amd import comm, face
    model = _face_model(be, 100)
    step = face.FaceTrainStep(model, lr=0.05, momentum=0.9, weight_decay=5e-4, max_norm=10.0, ema=False, comm=comm.GradAllReduce(bucket_bytes=20_000), sync_bn=True)
    init = {k: v.cpu().clone() for k, v in model.state_dict().items()}      # (after the constructor's broadcast: rank 0's weights)
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32).to(DEV); y = torch.randint(0, 24, (8,)).to(DEV)
    step.step(x[rank * 4:rank * 4 + 4], y[rank * 4:rank * 4 + 4])
    torch.save({"init": init, "sd": {k: v.cpu().clone() for k, v in model.state_dict().items()}}, f"{out_dir}/sbn{rank}.pt")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_gpu_face_step_with_sync_batchnorm_equals_whole_batch(tmp_path, hip):
    port = _free_port()
    mp.start_processes(_face_syncbn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(tmp_path / "sbn0.pt"); r1 = torch.load(tmp_path / "sbn1.pt")
    from visiondk_amd import face
    model = _face_model(hip, 100)
    model.load_state_dict({k: v.to(DEV) for k, v in r0["init"].items()}, strict=True)
    step = face.FaceTrainStep(model, lr=0.05, momentum=0.9, weight_decay=5e-4, max_norm=10.0, ema=False)
    torch.manual_seed(7)
    x = torch.randn(8, 3, 32, 32).to(DEV); y = torch.randint(0, 24, (8,)).to(DEV)
    step.step(x, y)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    for k in sd:
        if "num_batches" in k or k.endswith("model.head.norm.bias"):
            continue      # head.norm.bias sits right in front of the BatchNorm2d, which removes per-channel shifts: its true gradient is 0, what is computed is rounding noise
        a, b = r0["sd"][k].float(), sd[k].float()
        assert torch.equal(r0["sd"][k], r1["sd"][k]) or "running" in k, k
        if "running" in k:
            assert torch.allclose(r0["sd"][k], r1["sd"][k]) and torch.allclose(a, b, rtol=1e-4, atol=1e-5), k
        else:
            da, db = a - r0["init"][k].float(), b - r0["init"][k].float()
            rel = ((da - db).norm() / db.norm().clamp_min(1e-12)).item()
            assert rel < 3e-2, (k, rel)
