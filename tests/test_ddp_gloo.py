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
