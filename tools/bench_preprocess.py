"""Validation-time input pipeline: resize_and_padding(224) -> to_tensor -> normalize for a pet.yaml-sized val batch (bs 320) of
images with Oxford-pet-like geometries (sides 150..640), device kernel vs the chain the reference runs per image on the CPU (PIL resize + ImageOps.expand
+ the two torchvision float32 expressions).  Prints one JSON line: images/s with inputs resident in HBM, algorithmic HBM bytes and GB/s, the
host-inclusive rate (pack + H2D), the CPU chain on 1 core, bit-equality of a sample.
usage: python tools/bench_preprocess.py [batch] [iters]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image, ImageOps
from visiondk_amd import preprocess

B = int(sys.argv[1]) if len(sys.argv) > 1 else 320
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
S = 224
rng = np.random.default_rng(0)
geoms = [(int(rng.integers(150, 640)), int(rng.integers(150, 640))) for _ in range(B)]
imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for w, h in geoms]
pipe = preprocess.ValPipeline(size=S, device="cuda:0")
mean, std = pipe.mean, pipe.std


def cpu_chain(a):
    image = Image.fromarray(a)
    width, height = image.size
    sf = S / max(width, height)
    nw, nh = int(width * sf), int(height * sf)
    image = image.resize((nw, nh), Image.BILINEAR)
    pw, ph = (S - nw) // 2, (S - nh) // 2
    image = ImageOps.expand(image, (pw, ph, S - nw - pw, S - nh - ph), fill=(0, 0, 0))
    t = torch.from_numpy(np.asarray(image).copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    return t.sub_(torch.as_tensor(mean, dtype=torch.float32)[:, None, None]).div_(torch.as_tensor(std, dtype=torch.float32)[:, None, None])


px, off, wh, max_side = pipe.pack(imgs)
dpx, doff, dwh = px.cuda(), off.cuda(), wh.cuda()
out = torch.empty(B, 3, S, S, device="cuda:0")
for _ in range(3):
    pipe.run_packed(dpx, doff, dwh, max_side, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    pipe.run_packed(dpx, doff, dwh, max_side, out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
in_bytes = sum(a.size for a in imgs)
out_bytes = B * 3 * S * S * 4
t0 = time.time()
for _ in range(5):
    res = pipe(imgs)
torch.cuda.synchronize()
host_ms = (time.time() - t0) / 5 * 1e3
torch.set_num_threads(1)
t0 = time.time(); n_cpu = 0
while time.time() - t0 < 10.0:
    exp = cpu_chain(imgs[n_cpu % B]); n_cpu += 1
cpu_rate = n_cpu / (time.time() - t0)
got = res.cpu().numpy()
same = all(np.array_equal(got[i].view(np.uint32), cpu_chain(imgs[i]).numpy().view(np.uint32)) for i in range(0, B, 16))
print(json.dumps({
    "workload": f"val batch of {B} RGB images, sides 150..640 -> resize_and_padding({S}) + to_tensor + normalize, f32 NCHW out",
    "images_per_sec": B / ms * 1e3, "ms_per_batch": ms,
    "algorithmic_bytes_per_batch": in_bytes + out_bytes, "hbm_GBps": (in_bytes + out_bytes) / ms / 1e6, "hbm_frac_of_8TBps": (in_bytes + out_bytes) / ms / 1e6 / 8000,
    "images_per_sec_incl_host_pack_and_h2d": B / host_ms * 1e3,
    "cpu_baseline": {"value": cpu_rate, "unit": "images/s", "cores": 1, "kind": "reference", "sample": f"{n_cpu} images through PIL.resize + ImageOps.expand + float32 normalise, 10 s"},
    "bit_equal_to_cpu_chain": bool(same)}))
