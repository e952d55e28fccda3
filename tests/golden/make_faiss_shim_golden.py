"""Writes tests/golden/faiss_shim_reference_run.npz: what the reference's OWN index() / search() (engine/cbir/evaluation.py:106-200, cut out of /root/reference at run time,
nothing copied) return when their `faiss` is visiondk_amd.faiss_shim on the CPU SIMT emulation -- once through the plain index (device.type 'cpu' branch, fp32 storage) and
once through the GPU branch (GpuMultipleClonerOptions().useFloat16 -> index_cpu_to_all_gpus).  The GPU box, where /root/reference does not exist, checks the shim against
these arrays (tests/test_faiss_shim.py::test_committed_fixture_of_the_reference_run_on_the_mi355x).

    python tests/golden/make_faiss_shim_golden.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from tests.emu.emu_backend import load_emu                     # noqa: E402
from tests.test_faiss_shim import _EmbeddingExtractor, _Logger, _data, reference_namespace      # noqa: E402
from visiondk_amd import faiss_shim                            # noqa: E402


def main():
    faiss_shim.set_default(backend=load_emu(), device="cpu")
    ns = reference_namespace()
    g, q = _data(n=4000, nq=48, d=128, seed=7)
    k = 100
    out = {"gallery": g, "queries": q, "k": np.int64(k)}
    for tag, dev in (("f32", torch.device("cpu")), ("f16", torch.device("cuda"))):      # only device.type is read by the reference's index()
        fi = ns["index"](_EmbeddingExtractor(), [g], dev, _Logger())
        s, i = ns["search"](_EmbeddingExtractor(), [q], fi, dev, _Logger(), k=k, batch_size=16)
        out[f"scores_{tag}"], out[f"indices_{tag}"] = s, i
    np.savez_compressed(ROOT / "tests" / "golden" / "faiss_shim_reference_run.npz", **out)
    print("written", {k_: getattr(v, "shape", v) for k_, v in out.items()})


if __name__ == "__main__":
    main()
