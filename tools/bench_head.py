"""Margin head (ArcFace) fused forward + backward at BASELINE.json configs[2] size: B = 512 embeddings of 512 dims, C = 10^6 identities.
One JSON object: ms per head step in both cosine modes (cos_planes 3 = split-bf16 planes, 1 = single bf16 operands as under the reference's autocast)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from visiondk_amd import _lib, heads  # noqa: E402


def main():
    be = _lib.load()
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    B, D = 512, 512
    torch.manual_seed(0)
    h = heads.ArcFace(D, C, margin_arc=0.35, margin_am=0.0, scale=32, backend=be, device="cuda")
    f = torch.randn(B, D, device="cuda"); y = torch.randint(0, C, (B,), device="cuda")
    out = {"B": B, "D": D, "C": C}
    for planes, fused in ((3, False), (1, False), (3, True), (1, True)):
        key = f"{planes}_{'epilogue_fused' if fused else 'materialised_cos'}"
        for _ in range(2):
            h.margin_ce(f, y, cos_planes=planes, fused=fused)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        n = 5
        for _ in range(n):
            loss, df, dW = h.margin_ce(f, y, cos_planes=planes, fused=fused)
        e1.record(); torch.cuda.synchronize()
        out[f"ms_cos_planes_{key}"] = e0.elapsed_time(e1) / n
        out[f"loss_{key}"] = loss.mean().item()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
