"""Writes tests/golden/trainer_update_reference_run.npz: what the reference's OWN Trainer.update (engine/procedure/train.py:203-215, cut out of /root/reference at run time,
nothing copied) does to a ViT in three steps -- scaler.scale(loss).backward() -> unscale_ -> clip_grad_norm_(10) -> scaler.step -> scaler.update -> zero_grad -> ema.update
with torch.optim.SGD, torch's GradScaler and the reference's models/ema.py ModelEMA -- on the fp32 PyTorch-CPU path: the model is oracle/vit_ref.VisionTransformerRef (timm's
VisionTransformer restated and pinned against transformers, tests/test_oracle_vit.py).  Recorded: the three losses, the final weights of every parameter, the EMA of one tensor.
Inputs and initial weights are regenerated from the seeds stored beside them.  The GPU box, where /root/reference does not exist, runs the same three steps through the HIP
library (tests/test_faiss_shim.py::test_committed_trainer_update_fixture_on_the_mi355x).

    python tests/golden/make_trainer_update_golden.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from oracle.vit_ref import VisionTransformerRef                                   # noqa: E402
from tests.test_faiss_shim import REF, _cut, _load_by_path                         # noqa: E402

GEOM = dict(img=32, patch=8, classes=10, dim=64, depth=2, heads=1, mlp=128)
HYPER = dict(lr=0.05, momentum=0.9, weight_decay=5e-4, label_smoothing=0.05, init_scale=1024.0, steps=3, batch=8, seed_model=0, seed_data=1)


def build_ref():
    torch.manual_seed(HYPER["seed_model"])
    return VisionTransformerRef(GEOM["img"], GEOM["patch"], 3, GEOM["classes"], GEOM["dim"], GEOM["depth"], GEOM["heads"], GEOM["mlp"])


def batches():
    g = torch.Generator(); g.manual_seed(HYPER["seed_data"])
    for _ in range(HYPER["steps"]):
        yield torch.randn(HYPER["batch"], 3, GEOM["img"], GEOM["img"], generator=g), torch.randint(0, GEOM["classes"], (HYPER["batch"],), generator=g)


def main():
    ns = {"torch": torch}
    src = _cut(REF / "engine" / "procedure" / "train.py", ("update",), cls="Trainer").replace("@staticmethod\n", "")
    exec(compile(src, str(REF / "engine/procedure/train.py"), "exec"), ns)
    update = ns["update"]
    ModelEMA = _load_by_path("ref_ema", "models/ema.py").ModelEMA
    ref = build_ref()
    opt = torch.optim.SGD(ref.parameters(), lr=HYPER["lr"], momentum=HYPER["momentum"], weight_decay=HYPER["weight_decay"])
    scaler = torch.amp.GradScaler("cpu", init_scale=HYPER["init_scale"])
    ema = ModelEMA(ref)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=HYPER["label_smoothing"])
    losses = []
    ref.train()
    for x, y in batches():
        loss = crit(ref(x), y)
        losses.append(float(loss.item()))
        update(ref, loss, scaler, opt, ema)
    out = {"losses": np.asarray(losses, np.float64), "scale_after": np.float64(scaler.get_scale())}
    for k, v in {**GEOM, **HYPER}.items():
        out["cfg_" + k] = np.float64(v)
    for n, p in ref.named_parameters():
        out["w:" + n] = p.detach().numpy().copy()
    out["ema:blocks.0.mlp.fc1.weight"] = dict(ema.ema.named_parameters())["blocks.0.mlp.fc1.weight"].detach().numpy().copy()
    np.savez_compressed(ROOT / "tests" / "golden" / "trainer_update_reference_run.npz", **out)
    print("written", losses, sum(v.size for k, v in out.items() if k.startswith("w:")), "weights")


if __name__ == "__main__":
    main()
