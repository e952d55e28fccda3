"""cfg4: CBIR with ConvNeXt-Base embeddings (feat_dim 128, configs/faceX/cbir.yaml) — the two halves of the reference's `valuate_cbir`:
(1) FeatureExtractor.extract_cbir (eval forward of backbone + neck, L2-normalise; bf16 path and the fp32-MFMA precise path) in images/s, and
(2) index + search + metrics on those embeddings through the faiss-like index (the 10k x 1M search itself is bench.py's "cbir" leg).
usage: python tools/bench_cfg4.py [n_gallery_images] [batch]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from visiondk_amd import cbir, face, metrics

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
operand = sys.argv[3] if len(sys.argv) > 3 else "fp16"      # the face / CBIR path's default operand format (BackboneFactory); "bf16" = rounds 1-4
tw = face.TimmWrapper("convnext_base", feat_dim=128, image_size=224, device=dev, operand=operand).eval()
g = torch.Generator(device="cpu"); g.manual_seed(0)
# synthetic "identities": 64 prototype images + noise, so that retrieval has a ground truth
protos = torch.randn(64, 3, 224, 224, generator=g)
ident = torch.randint(0, 64, (n_img,), generator=g)
imgs = (protos[ident] + 0.3 * torch.randn(n_img, 3, 224, 224, generator=g))
loader = [imgs[i:i + bs] for i in range(0, n_img, bs)]
out = {"workload": f"cfg4 ConvNeXt-B + neck(128) embeddings of {n_img} synthetic 224x224 images, batch {bs}"}
for name, precise in ((operand, False), ("precise_fp32_mfma", True)):
    ex = face.FeatureExtractor(tw, precise=precise)
    ex.extract_cbir(loader[:1], dev); torch.cuda.synchronize()
    t0 = time.time(); emb = ex.extract_cbir(loader, dev); torch.cuda.synchronize(); dt = time.time() - t0
    out[f"extract_images_per_sec_{name}"] = n_img / dt          # includes the per-batch H2D copy and the final D2H, like the reference's loop
    out[f"emb_{name}"] = emb
rel = float(np.linalg.norm(out[f"emb_{operand}"] - out["emb_precise_fp32_mfma"]) / np.linalg.norm(out["emb_precise_fp32_mfma"]))
emb = out.pop("emb_precise_fp32_mfma"); out.pop(f"emb_{operand}")
out[f"{operand}_vs_precise_embedding_rel"] = rel
nq = n_img // 8
idx = cbir.index(None, None, dev, None, "Flat", gallery_embeddings=emb[nq:])
t0 = time.time(); s, i = idx.search(emb[:nq], k=10); dt = time.time() - t0
labels = ident.numpy()
preds = [[str(labels[nq + j]) for j in row] for row in i]
truth = [[str(labels[q])] for q in range(nq)]
m = metrics.CBIRMetrics(cutoffs=[1, 10])
m.compute_mrr(preds, truth); m.compute_recall(preds, truth)
out["search_pairs_per_sec_small"] = nq * (n_img - nq) / dt
out["metrics"] = {k: float(v) for k, v in m.metrics.items()}
print(json.dumps(out))
