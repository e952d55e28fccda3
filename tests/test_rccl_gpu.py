"""The data-parallel exchange on the REAL stack: torch.distributed backend "nccl" (= RCCL on ROCm) on the one MI355X a test box has, world_size 1.
The reference's only parallelism is NCCL DDP (main.py:39-40, engine/vision_engine.py:313,510, train.py:157-159); the 2-rank gloo tests
(tests/test_ddp_gloo.py) prove the arithmetic of the exchange on CPU, these prove that the stream-ordered GPU path executes: the gradient buckets are
all-reduced by RCCL kernels on RCCL's stream, ordered against the backward kernels of the launch stream, while backward is still being enqueued.
`GradAllReduce(always_communicate=True)` issues every collective although a one-rank group makes each an identity, so results must equal the
communication-free step BIT FOR BIT."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rccl():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    t = torch.arange(8, dtype=torch.float32, device="cuda:0")
    dist.all_reduce(t)                                   # the communicator is created lazily: make RCCL really come up here
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
    yield
    dist.destroy_process_group()


def test_vit_step_buckets_through_rccl_equal_the_plain_step(hip, rccl):
    from tests.test_vit import SPEC
    from visiondk_amd import comm, vit
    res = []
    for use_comm in (False, True):
        model = vit.VisionTransformer(SPEC, device="cuda:0", backend=hip, seed=11)
        c = comm.GradAllReduce(bucket_bytes=100_000, always_communicate=True) if use_comm else None      # small buckets: several collectives per backward
        step = vit.FusedTrainStep(model, lr=0.01, label_smoothing=0.05, ema=True, comm=c)
        torch.manual_seed(7)
        for _ in range(3):
            x = torch.randn(8, 3, 32, 32).cuda(); y = torch.randint(0, 10, (8,)).cuda()
            step.step(x, y)
        torch.cuda.synchronize()
        res.append((model.engine.params.clone(), model.engine.grads.clone(), step.ema.clone(), step.loss_value()))
        if use_comm:
            assert c.collectives >= 1 + 3 * 3            # the parameter broadcast + at least 3 buckets per step
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2]) and res[0][3] == res[1][3]


def test_vit_base_step_with_32mb_buckets_through_rccl(hip, rccl):
    """the bench's geometry (ViT-B/16, 86.6 M parameters, 32 MB buckets) at batch 8"""
    from visiondk_amd import comm, vit
    res = []
    for use_comm in (False, True):
        model = vit.VisionTransformer(vit.spec_from_timm_name("vit_base_patch16_224", 1000), device="cuda:0", backend=hip, seed=3)
        c = comm.GradAllReduce(always_communicate=True) if use_comm else None
        step = vit.FusedTrainStep(model, lr=0.006, label_smoothing=0.05, ema=False, comm=c)
        torch.manual_seed(9)
        x = torch.randn(8, 3, 224, 224).cuda(); y = torch.randint(0, 1000, (8,)).cuda()
        step.step(x, y); step.step(x, y)
        torch.cuda.synchronize()
        res.append(model.engine.params.clone())
        if use_comm:
            assert c.collectives >= 1 + 2 * 5             # one bucket closes every two layers (28 MB of gradients per layer) + the embeddings
    assert torch.equal(res[0], res[1])


def _face_model(be, seed, classes=24):
    from visiondk_amd import convnext, face
    convnext.TIMM_CONVNEXTS["convnext_test"] = dict(depths=(1, 1, 1, 1), dims=(8, 16, 24, 32))
    cfg = {"task": "cbir", "image_size": 32, "backbone": {"timm-convnext_test": {"pretrained": False, "image_size": 32, "feat_dim": 64, "operand": "bf16"}},
           "head": {"arcface": {"feat_dim": 64, "num_class": classes, "margin_arc": 0.35, "margin_am": 0.0, "scale": 32}}}
    torch.manual_seed(seed)
    model = face.get_model(cfg, None, 0, backend=be, device="cuda:0").model.train()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("gamma"):
                p.fill_(0.4)
    return model


@pytest.mark.parametrize("shard_head,sync_bn", [(False, False), (True, False), (False, True)])
def test_face_step_through_rccl(hip, rccl, shard_head, sync_bn):
    """FaceTrainStep(comm=...): backbone buckets + neck / head gradients + BatchNorm buffer broadcasts through RCCL; with shard_head the class-sharded margin
    head (all-gathered features, three small all-reduces for the softmax across the shards, one for the feature gradient) on one rank holding all the classes."""
    from visiondk_amd import comm, face
    res = []
    for use_comm in (False, True):
        model = _face_model(hip, 21)
        c = comm.GradAllReduce(bucket_bytes=20_000, always_communicate=True) if use_comm else None
        step = face.FaceTrainStep(model, lr=0.05, momentum=0.9, weight_decay=5e-4, max_norm=10.0, ema=True, comm=c, shard_head=shard_head and use_comm,
                                  sync_bn=sync_bn and use_comm)
        torch.manual_seed(5)
        for _ in range(2):
            x = torch.randn(8, 3, 32, 32).cuda(); y = torch.randint(0, 24, (8,)).cuda()
            rows = step.step(x, y)
        if shard_head and use_comm:
            step.gather_head()
        torch.cuda.synchronize()
        res.append(({k: v.clone() for k, v in model.state_dict().items()}, rows.clone()))
    for k in res[0][0]:
        a, b = res[0][0][k].float(), res[1][0][k].float()
        if shard_head or sync_bn:      # the sharded head / the three-kernel SyncBatchNorm are different kernel sequences: same math, fp32 rounding differs
            if k.endswith("model.head.norm.bias") or k.endswith("output_layer.0.bias"):
                continue                   # analytically zero gradients (a per-channel shift in front of a train-mode BatchNorm): pure rounding noise
            # two equivalent kernel sequences: fp32 rounding differs, a bf16 rounding that flips moves a gradient by ~1e-3.  Weights of O(1) norm: 2e-3; tensors that
            # start at zero (biases, whose value after two steps IS the accumulated update, ~1e-3 per element) or sit furthest from the loss (the stem): 2e-2
            tol = 2e-3 if (a.norm().item() > 1.0 and "stem" not in k) else 2e-2
            assert ((a - b).norm() / a.norm().clamp_min(1e-30)).item() < tol, (k, ((a - b).norm() / a.norm().clamp_min(1e-30)).item())
        else:
            assert torch.equal(a, b), k
    rerr = ((res[0][1] - res[1][1]).abs() / res[0][1].abs().clamp_min(1e-6)).max().item()
    assert rerr <= (5e-3 if (shard_head or sync_bn) else 0.0), rerr      # loss rows of the second step (the first step's weights already differ by rounding)


def test_resnet_step_with_sync_batchnorm_through_rccl(hip, rccl):
    """ResNetTrainStep(sync_bn=True): the (sum, sum of squares, count) / (sum g, sum g x^, count) vectors of every BatchNorm go through RCCL between the statistics kernel
    and its consumer, in forward and backward (the reference's sync_bn flag, vision_engine.py:224-225).  One rank: the all-reduce is an identity."""
    from visiondk_amd import comm, resnet
    res = []
    for use_comm in (False, True):
        torch.manual_seed(31)
        model = resnet.ResNet(resnet.ResNetSpec(img_size=32, num_classes=5, widths=(8, 8, 16, 16), depths=(1, 1, 1, 1)), device="cuda:0", backend=hip)
        c = comm.GradAllReduce(bucket_bytes=10_000, always_communicate=True) if use_comm else None
        step = resnet.ResNetTrainStep(model, lr=0.05, momentum=0.9, weight_decay=5e-4, loss="bce", ema=True, comm=c, sync_bn=use_comm)
        torch.manual_seed(6)
        for _ in range(2):
            x = torch.randn(8, 3, 32, 32).cuda(); t = (torch.rand(8, 5) > 0.5).float().cuda()
            step.step(x, t)
        torch.cuda.synchronize()
        res.append((model.engine.params.clone(), model.engine.buffers.clone(), step.ema_buffers.clone()))
    for a, b in zip(res[0], res[1]):
        assert ((a - b).norm() / a.norm()).item() < 1e-5


def test_sharded_gallery_search_through_rccl(hip, rccl):
    import numpy as np
    from oracle import cbir as ocbir
    from visiondk_amd import cbir
    g = torch.Generator().manual_seed(3)
    gal = torch.nn.functional.normalize(torch.randn(5000, 128, generator=g)); qry = torch.nn.functional.normalize(torch.randn(70, 128, generator=g))
    s, i = cbir.search_sharded(qry.cuda(), gal.cuda(), k=100, idx_base=0, backend=hip, device="cuda:0")
    so, io = ocbir.flat_ip_search(qry.numpy(), gal.numpy(), 100)
    np.testing.assert_array_equal(i.cpu().numpy(), io)
    np.testing.assert_array_equal(s.cpu().numpy().view("uint32"), so.view("uint32"))


def test_c_abi_collectives_drive_the_backward_without_c10d(hip):
    """SURVEY 8(b)'s vdk_comm_init / vdk_allreduce_bucket / vdk_allgather (csrc/comm.hip: RCCL by dlopen, its own stream, event-ordered against the launch stream), the way a
    C host would use them: the ready-range callback of vdk_vit_backward issues the bucket's all-reduce, vdk_comm_finish joins before the optimizer.  One rank (a one-GPU
    box): every collective is an identity, so the step must equal the communication-free one bit for bit -- what is proven is that the path executes in stream order."""
    import ctypes as C
    from tests.test_vit import SPEC
    from visiondk_amd import _abi, vit
    lib = hip.lib
    uid = (C.c_ubyte * 128)()
    hip.check(lib.vdk_comm_unique_id(uid), "vdk_comm_unique_id")
    comm = C.c_void_p()
    hip.check(lib.vdk_comm_init(uid, 0, 1, C.byref(comm)), "vdk_comm_init")
    assert lib.vdk_comm_rank(comm) == 0 and lib.vdk_comm_world(comm) == 1
    try:
        # all-gather (the sharded search's query exchange)
        q = torch.randn(40, 128, device="cuda:0"); out = torch.empty_like(q)
        hip.check(lib.vdk_allgather(comm, q.data_ptr(), out.data_ptr(), q.numel() * 4, hip.stream()), "vdk_allgather")
        torch.cuda.synchronize()
        assert torch.equal(q, out)
        res = []
        for use_comm in (False, True):
            model = vit.VisionTransformer(SPEC, device="cuda:0", backend=hip, seed=11)
            eng = model.engine
            torch.manual_seed(7)
            x = torch.randn(8, 3, 32, 32).cuda(); y = torch.randint(0, 10, (8,)).cuda()
            logits = eng.forward(x)
            from visiondk_amd import ops
            _, dl, _ = ops.softmax_ce(logits[:, :10].contiguous(), y, grad_scale=1.0 / 8, pad_to=eng.cp, backend=hip)
            calls = []

            def on_ready(off, n):
                calls.append((off, n))
                hip.check(lib.vdk_allreduce_bucket(comm, eng.grads.data_ptr(), off, n, hip.stream()), "vdk_allreduce_bucket")

            eng.backward(dl, on_ready=on_ready if use_comm else None)
            if use_comm:
                hip.check(lib.vdk_comm_finish(comm, hip.stream()), "vdk_comm_finish")
                assert len(calls) >= 3 and sum(n for _, n in calls) == eng.n_floats      # every slice of the flat gradient went through a collective exactly once
            g2 = ops.sumsq(eng.grads, backend=hip)
            torch.cuda.synchronize()
            res.append((eng.grads.clone(), g2.item()))
        assert torch.equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]
    finally:
        hip.check(lib.vdk_comm_destroy(comm), "vdk_comm_destroy")


def test_cu_reserve_is_scoped_to_the_backward_window(hip, rccl):
    """GradAllReduce holds CUs back from the persistent GEMM grids only while a collective can be in flight: from the step's first bucket to finish_step (round 3 set the
    reserve process-wide in the constructor).  Outside a step -- forward, optimizer, evaluation, CBIR -- vdk_gemm_reserved_cus() is what it was before."""
    from tests.test_vit import SPEC
    from visiondk_amd import comm, vit
    lib = hip.lib
    before = lib.vdk_gemm_reserved_cus()
    c = comm.GradAllReduce(bucket_bytes=100_000, always_communicate=True, reserve_cus=32)
    assert lib.vdk_gemm_reserved_cus() == before                       # constructing it reserves nothing
    model = vit.VisionTransformer(SPEC, device="cuda:0", backend=hip, seed=11)
    step = vit.FusedTrainStep(model, lr=0.01, comm=c)
    seen = []
    orig = c._flush

    def spy():
        orig(); seen.append(lib.vdk_gemm_reserved_cus())

    c._flush = spy
    x = torch.randn(8, 3, 32, 32).cuda(); y = torch.randint(0, 10, (8,)).cuda()
    step.step(x, y)
    torch.cuda.synchronize()
    assert seen and all(v == max(32, before) for v in seen)             # held while buckets are being issued ...
    assert lib.vdk_gemm_reserved_cus() == before                        # ... and given back when the step's collectives have been joined


def _overlap_run(hip, batch, steps, route_kw):
    """ViT-B/16 fused steps; returns (ms per step, params, the last step's trace or None)"""
    from visiondk_amd import comm, vit
    model = vit.VisionTransformer(vit.spec_from_timm_name("vit_base_patch16_224", 1000), device="cuda:0", backend=hip, seed=3, operand="fp16")
    c = comm.GradAllReduce(always_communicate=True, **route_kw) if route_kw is not None else None
    step = vit.FusedTrainStep(model, lr=0.006, label_smoothing=0.05, ema=True, comm=c)
    torch.manual_seed(9)
    x = torch.randn(batch, 3, 224, 224).cuda(); y = torch.randint(0, 1000, (batch,)).cuda()
    for _ in range(2):
        step.step(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step.step(x, y)
    e1.record(); torch.cuda.synchronize()
    tr = c.trace_read() if (c is not None and route_kw.get("trace")) else None
    out = (e0.elapsed_time(e1) / steps, model.engine.params.clone(), tr)
    if c is not None:
        c.close()
    return out


def test_c_abi_route_equals_c10d_route_and_the_plain_step(hip, rccl):
    """GradAllReduce(route="abi"): the same buckets through vdk_comm_init / vdk_allreduce_bucket / vdk_comm_finish (csrc/comm.hip: RCCL on the library's own stream) instead of
    c10d -- bit-identical weights to the c10d route and to the communication-free step"""
    plain = _overlap_run(hip, 8, 1, None)
    c10d = _overlap_run(hip, 8, 1, {"route": "c10d"})
    abi = _overlap_run(hip, 8, 1, {"route": "abi"})
    assert torch.equal(plain[1], c10d[1]) and torch.equal(plain[1], abi[1])


def test_allreduce_buckets_run_while_the_backward_is_still_executing(hip):
    """The one-GPU evidence of torch DDP's overlap (engine/vision_engine.py:313,510): tools/overlap_probe.py in a FRESH process (ViT-B/16, batch 64, fp16 fused step; buckets
    through the C-ABI route with the timing trace on).  A one-rank all-reduce moves nothing, so behind every one a STAND-IN kernel holds 32 CUs on the collectives' stream for
    the time an 8-GPU ring all-reduce of that bucket takes at a pessimistic 150 GB/s bus bandwidth (2 * 7/8 * bytes / bandwidth = 11.7 us per MB: 4 ms for the 346 MB gradient).
      * the first bucket's all-reduce STARTS in the first half of the backward, and all but the gradient's tail END before the backward's last kernel;
      * what the optimizer waits for after the backward is the tail, not the sum;
      * the step gets slower by a fraction of the collectives' total duration: they ran under the backward.
    (A fresh process because the measurement is about hardware-queue concurrency: after the ~400 tests of a whole-suite run the process owns so many HIP streams that the
    driver time-slices its queues and every cross-stream overlap degrades -- observed in round 5; a training process has the two streams this probe has.)"""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "overlap_probe.py"), "64", "150"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    st = d["standin"]
    ar, total, bwd_end = st["allreduce_intervals_ms"], st["collective_ms_total"], st["backward_end_ms"]
    print({k: st[k] for k in ("ms_step", "collectives_per_step", "collective_ms_total", "backward_end_ms", "ended_before_backward_end", "exposed_tail_ms",
                              "step_cost_of_the_exchange_ms", "hidden_fraction")}, "plain", d["ms_plain"])
    assert len(ar) >= 4 and total > 2.0                           # the stand-ins really ran (about 4 ms per step)
    assert ar[0][0] < 0.5 * bwd_end                               # the first bucket leaves in the first half of the backward ...
    assert st["ended_before_backward_end"] >= len(ar) - 2         # ... and only the gradient's tail is still in flight when the backward ends
    assert st["exposed_tail_ms"] < 0.5 * total                    # what the optimizer waits for is the tail, not the sum
    assert st["step_cost_of_the_exchange_ms"] < 0.5 * total       # the step pays a fraction of the collectives' duration
    assert abs(d["ms_abi_route_empty_collectives"] - d["ms_plain"]) < 0.05 * d["ms_plain"]      # the plumbing itself (callbacks, events, empty collectives) is in the noise


def test_bench_two_ranks_code_path_on_one_gpu(hip):
    """The driver launches `bench.py --gpus N` under torch.distributed.run with one rank per GPU; this box has one.  VDK_BENCH_SHARE_GPU=1 (a diagnostic switch of bench.py,
    flagged in its line) puts both ranks on cuda:0 and exchanges over gloo, so the whole N > 1 path of the file runs end to end here: weights broadcast, the bucketed
    gradient exchange issued from inside the backward, barrier + max-over-ranks timing, the sharded search (all-gather of the queries, all-to-all of the per-shard top-k
    lists, merge) checked bit for bit against the oracle over the WHOLE gallery, and the one JSON line from rank 0."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, VDK_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "32"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(root))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, "rank 0 prints ONE line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 64 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak" and "diagnostic" in d
    assert d["exchange"]["collectives_per_step"] >= 4                                  # the flat gradient left in buckets, every step
    assert d["roofline"]["frac"] > 0 and d["value"] > 0
    c = d["cbir"]
    assert c["n_gpus"] == 2 and c["scaling"] == "strong" and c["parity_vs_oracle"]["indices_equal"] and c["parity_vs_oracle"]["scores_bit_equal"]
