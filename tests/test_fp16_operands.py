"""The fp16-operand mode of hot path A: the arithmetic of the reference's own GPU path.

`torch.autocast(device_type=self.device.type, enabled=...)` (engine/procedure/train.py:118) names no dtype, so on a GPU the reference's matmuls read **float16** operands,
and `Trainer.update` wraps the backward in a GradScaler (train.py:203-211, built at engine/vision_engine.py:232).  `VdkVitConfig.operand = VDK_F16` / `VdkGemmDesc.ab_dtype =
VDK_F16` run the same kernels on v_mfma_f32_32x32x16_f16 with fp16 copies of weights / activations / gradients; the loss scale lives on the device.

What is asserted here, literally, is north_star's tolerance: **logits within 1e-3, every parameter gradient within 5e-3 (Frobenius-relative) of the fp32 oracle** -- the
reference's PyTorch-CPU path restated (oracle/vit_ref.py) -- which the bf16 mode cannot meet (tests/test_parity_bf16.py) and this one does.  Per kernel, on identical
fp16 operands, the bar is the bf16 kernels' bar: 1e-5 against torch fp32 for the GEMMs, the oracle's fp16-operand mode for attention."""
import math

import pytest
import torch

from visiondk_amd import ops

NORTH_STAR_LOGITS, NORTH_STAR_GRAD = 1e-3, 5e-3      # BASELINE.json north_star: "logits/embeddings within 1e-3 rel of reference"; gradients: VERDICT r3 item 1


def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


# ---------------------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("kern", [1, 5, 6])
@pytest.mark.parametrize("M,N,K,splitk", [(300, 264, 192, 1), (512, 256, 256, 1), (264, 256, 768, 3), (130, 520, 128, 1)])
def test_gemm_fp16_operands_every_kernel(be, dev, M, N, K, splitk, kern):
    """128x128 (1), four waves 256x256 (5) and 256x128 two-workgroups-per-CU (6) on fp16 operands vs torch fp32 on the same values: fp32 / fp16 outputs, bias + residual,
    GELU with the saved pre-activation, dGELU, split-K"""
    torch.manual_seed(5)
    a = torch.randn(M, K).half().to(dev); b = torch.randn(N, K).half().to(dev)
    b[:, 3] += 2.0
    bias = torch.randn(N).to(dev); res = torch.randn(M, N).to(dev)
    ref = a.float() @ b.float().T
    be.lib.vdk_gemm_force_kernel(kern)
    try:
        if splitk > 1:
            out = ops.gemm_nt(a, b, out_dtype=torch.float32, splitk=splitk, backend=be)
            assert _rel(out, ref) < 1e-5
            return
        out = ops.gemm_nt(a, b, out_dtype=torch.float32, bias=bias, residual=res, backend=be)
        assert _rel(out, ref + bias + res) < 1e-5
        if kern != 1:
            assert be.lib.vdk_gemm_last_kernel() in (5, 6)      # an fp16 problem never lands on the eight-wave bf16 kernel
        aux = torch.empty(M, N, dtype=torch.float16, device=dev)
        g = ops.gemm_nt(a, b * 0.1, bias=bias, act=ops.ACT_GELU, aux=aux, backend=be)
        pre = ref * 0.1 + bias if False else (a.float() @ (b * 0.1).float().T + bias)
        assert g.dtype == torch.float16
        assert _rel(aux.float(), pre.half().float()) < 2e-5 and _rel(g.float(), torch.nn.functional.gelu(pre).half().float()) < 2e-5
        u = torch.randn(M, N).half().to(dev)
        d = ops.gemm_nt(a, b, out_dtype=torch.float16, act=ops.ACT_DGELU, aux=u, backend=be)
        uu = u.float().requires_grad_(True)
        torch.nn.functional.gelu(uu).sum().backward()
        assert _rel(d.float(), (ref * uu.grad).half().float()) < 2e-5
    finally:
        be.lib.vdk_gemm_force_kernel(0)


def test_gemm_fp16_tn_weight_gradient_and_colsum(be, dev):
    """dW = dY^T X straight from row-major fp16 dY / X (TN kernel, split-K) and the c_colsum by-product of the dGELU epilogue on fp16 values"""
    torch.manual_seed(6)
    T, NO, NI = 512, 256, 384
    dy = torch.randn(T, NO).half().to(dev); x = torch.randn(T, NI).half().to(dev)
    dw = ops.gemm_nt(dy, x, out_dtype=torch.float32, trans=True, splitk=2, backend=be)
    assert _rel(dw, dy.float().T @ x.float()) < 1e-5
    w = (torch.randn(NI, NO) * 0.1).half().to(dev); u = torch.randn(T, NI).half().to(dev)
    for kern in (5, 6):          # (small problems reach the tiled kernels only when forced)
        be.lib.vdk_gemm_force_kernel(kern)
        try:
            rows = be.lib.vdk_gemm_c_colsum_rows(T, NI, NO)
            assert rows == 2 * ((T + 255) // 256)
            part = torch.full((rows, NI), float("nan"), dtype=torch.float32, device=dev)
            du = ops.gemm_nt(dy, w, act=ops.ACT_DGELU, aux=u, c_colsum=part, backend=be)
            assert _rel(part.sum(0), du.float().sum(0)) < 1e-5
        finally:
            be.lib.vdk_gemm_force_kernel(0)


def test_gemm_rejects_mixed_formats(be, dev):
    a = torch.randn(64, 64).half().to(dev); b = torch.randn(64, 64).bfloat16().to(dev)
    with pytest.raises(AssertionError):
        ops.gemm_nt(a, b, backend=be)
    with pytest.raises(AssertionError):
        ops.gemm_nt(a, a, out_dtype=torch.bfloat16, backend=be)


# ---------------------------------------------------------------------------------------------------------------- attention / LayerNorm
@pytest.mark.parametrize("B,N,H,long_min", [(3, 17, 2, None), (1, 197, 1, None), (2, 100, 2, 64)])
def test_attention_fp16_vs_fp16_operand_oracle(be, dev, B, N, H, long_min, monkeypatch):
    """short-sequence kernels (and, with VDK_ATTN_LONG_MIN, the streaming kernels) on fp16 q / k / v: P, O, dS rounded to fp16 where the oracle's fp16-operand mode rounds"""
    from oracle import bf16ops
    if long_min is not None:
        monkeypatch.setenv("VDK_ATTN_LONG_MIN", str(long_min))
    torch.manual_seed(3)
    D = H * 64
    qkv = (torch.randn(B, N, 3 * D) * 1.3).half()
    qkv[0, N // 3, D:D + 64] *= 5.0
    o, lse = ops.attention_fwd(qkv.to(dev), H, backend=be)
    assert o.dtype == torch.float16
    x = qkv.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    q, k, v = (t.clone().requires_grad_(True) for t in (x[0], x[1], x[2]))
    with bf16ops.precision("fp16_operands"):
        oo = bf16ops.attention(q, k, v, 0.125)
        exp = oo.detach().transpose(1, 2).reshape(B, N, D)
        tol = 2e-4 if long_min is None else 2e-3       # (the streaming forward rounds P unnormalised, as its bf16 form does)
        assert _rel(o.float(), exp) < tol
        dout = torch.randn(B, N, D).half()
        oo.backward(dout.float().reshape(B, N, H, 64).transpose(1, 2))
        dqkv = ops.attention_bwd(qkv.to(dev), o, dout.to(dev), lse, H, backend=be).float().cpu().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
        for got, ref, name in zip(dqkv, (q.grad, k.grad, v.grad), "qkv"):
            assert _rel(got, bf16ops.rb(ref)) < (2e-4 if long_min is None else 2e-3), name      # (on the device v_exp_f32 / v_rcp_f32 stand where the emulator has exp2f / division: 6e-5 measured)


def test_layernorm_fp16_in_and_out(be, dev):
    torch.manual_seed(1)
    T, Cn = 70, 256
    x = torch.randn(T, Cn).to(dev); gm = torch.randn(Cn).to(dev); bt = torch.randn(Cn).to(dev)
    y, mean, rstd = ops.layernorm_fwd(x, gm, bt, out_dtype=torch.float16, backend=be)
    assert y.dtype == torch.float16 and _rel(y.float(), torch.nn.functional.layer_norm(x, (Cn,), gm, bt, 1e-6).half().float()) < 2e-5
    dy = torch.randn(T, Cn).half().to(dev)
    dx, dxh, dg, db = ops.layernorm_bwd(dy, x, mean, rstd, gm, backend=be)
    xr = x.clone().requires_grad_(True); gr = gm.clone().requires_grad_(True); br = bt.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (Cn,), gr, br, 1e-6).backward(dy.float())
    assert dxh.dtype == torch.float16 and _rel(dx, xr.grad) < 2e-5 and _rel(dxh.float(), xr.grad.half().float()) < 2e-5
    assert _rel(dg, gr.grad) < 2e-5 and _rel(db, br.grad) < 2e-5


# ---------------------------------------------------------------------------------------------------------------- the engine
def assert_north_star(r):
    """the engine's logits and EVERY parameter gradient against the fp32 oracle (the reference's PyTorch-CPU path): north_star's numbers, not multiples of a measurement"""
    assert r["operand"] == "fp16"
    assert r["vs_fp32"]["logits"] <= NORTH_STAR_LOGITS, r["vs_fp32"]
    assert r["vs_fp32"]["worst_grad"] <= NORTH_STAR_GRAD, r["vs_fp32"]
    assert r["vs_fp32"]["loss"] <= 1e-4, r["vs_fp32"]
    # and it IS the fp16-operand evaluation: as close to the oracle's fp16-operand mode as two evaluations of that mode (fp32 / fp64 accumulation) are to each other,
    # up to the factor a flip-driven quantity scatters by (x3), and no further from fp32 than PyTorch's own fp16 autocast of the same module (x1.25) where that arm exists
    fl = r["floor_o32_vs_o64"]
    for side in ("vs_o32", "vs_o64"):
        assert r[side]["logits"] <= 3.0 * fl["logits"] + 1e-5 and r[side]["worst_grad"] <= 3.0 * fl["worst_grad"] + 1e-5, (side, r[side], fl)
    if "torch_autocast_vs_fp32" in r:
        ac = r["torch_autocast_vs_fp32"]
        assert r["vs_fp32"]["logits"] <= 1.25 * ac["logits"] + 1e-5 and r["vs_fp32"]["worst_grad"] <= 1.25 * ac["worst_grad"] + 1e-5, (r["vs_fp32"], ac)


@pytest.mark.parametrize("img,patch,B", [(32, 8, 3), (112, 8, 2)])       # 17 and 197 tokens
def test_vit_fp16_operands_meets_north_star_tolerance(be, dev, img, patch, B):
    from oracle.parity import vit_fwd_bwd_vs_oracle, vit_pair
    ref, model = vit_pair(be, dev, img, patch, 128, 2, 2, 256, 10, operand="fp16")
    torch.manual_seed(5)
    x = torch.randn(B, 3, img, img); y = torch.randint(0, 10, (B,))
    r = vit_fwd_bwd_vs_oracle(ref, model, x, y, dev)
    print(r)
    assert_north_star(r)


def test_vit_fp16_16bit_residual_gradient_stream_matches_the_fp32_stream(be, dev, monkeypatch):
    """widths 512 < D <= 1024 in fp16 mode keep the residual-gradient stream in 16 bits (csrc/vit_engine.hip `g16`, ln_bwd_kernel<..., DR16>): a 576-wide two-block model --
    the narrowest shape that takes the path, small enough for the emulator -- meets the stated tolerance, and its gradients sit within a few fp16 roundings of the
    fp32-stream form (VDK_VIT_G16=0 is read per process: the comparison arm is the oracle, the switch is exercised on the GPU by bench A/B runs)."""
    from oracle.parity import vit_fwd_bwd_vs_oracle, vit_pair
    ref, model = vit_pair(be, dev, 32, 16, 576, 2, 9, 1152, 10, operand="fp16")
    torch.manual_seed(7)
    x = torch.randn(3, 3, 32, 32); y = torch.randint(0, 10, (3,))
    r = vit_fwd_bwd_vs_oracle(ref, model, x, y, dev)
    print(r)
    assert_north_star(r)


@pytest.mark.gpu
def test_vit_base_patch16_full_size_fp16_operands_within_1e3_of_the_fp32_reference(hip):
    """BASELINE.json configs[1]'s model at full width and depth (ViT-B/16, 224, 1000 classes), batch 8, every parameter gradient: logits <= 1e-3, gradients <= 5e-3 of the
    reference's fp32 CPU path.  (bf16 operands on the same inputs: 6.5e-3 / 1.0e-2.)"""
    from oracle.parity import vit_fwd_bwd_vs_oracle, vit_pair
    ref, model = vit_pair(hip, "cuda:0", 224, 16, 768, 12, 12, 3072, 1000, seed=2, operand="fp16")
    torch.manual_seed(6)
    x = torch.randn(8, 3, 224, 224); y = torch.randint(0, 1000, (8,))
    r = vit_fwd_bwd_vs_oracle(ref, model, x, y, "cuda:0")
    print(r)
    assert_north_star(r)


def _oracle_grads(ref, x, y, smoothing, S):
    from oracle import bf16ops
    for p in ref.parameters():
        p.grad = None
    with bf16ops.precision("fp16_operands"):
        loss = torch.nn.functional.cross_entropy(ref(x), y, label_smoothing=smoothing)
        (loss * S).backward()
    return loss.item(), {n: p.grad.detach().clone() / S for n, p in ref.named_parameters()}


def test_fused_step_fp16_is_gradscaler_plus_clip_plus_sgd(be, dev):
    """FusedTrainStep on fp16 operands = the reference's Trainer.update under GradScaler (train.py:203-215): scaled backward, unscale, clip_grad_norm_(10), SGD(momentum,
    weight decay), EMA.  The oracle's fp16-operand gradients, un-scaled, clipped and applied by torch's own SGD, must leave the same weights."""
    from oracle.parity import vit_pair
    from visiondk_amd import vit
    ref, model = vit_pair(be, dev, 32, 8, 128, 2, 2, 256, 10, operand="fp16")
    torch.manual_seed(7)
    x = torch.randn(4, 3, 32, 32); y = torch.randint(0, 10, (4,))
    step = vit.FusedTrainStep(model, lr=0.05, momentum=0.9, weight_decay=1e-3, label_smoothing=0.05, max_norm=0.5, init_scale=1024.0)     # max_norm small enough to clip
    opt = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3)
    for it in range(2):
        loss_ref, g = _oracle_grads(ref, x, y, 0.05, 1024.0)
        for n, p in ref.named_parameters():
            p.grad = g[n]
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        opt.step()
        step.step(x.to(dev), y.to(dev))
        assert abs(step.loss_value() - loss_ref) < 2e-4 * abs(loss_ref)
    sd = model.state_dict()
    for n, p in ref.named_parameters():
        assert _rel(sd[n], p.detach()) < 2e-4, n
    assert step.loss_scale() == 1024.0 and step.skipped_steps() == 0
    # the operand copy the next forward reads is fp16 and equals the rounded master weights
    eng = model.engine
    assert eng.wb16.dtype == torch.float16
    name, off, numel, shape = next(e for e in eng.entries if e[0] == "blocks.0.attn.qkv.weight")
    assert torch.equal(eng.wb16[off:off + numel].view(shape).cpu(), sd[name].half().cpu())


def test_fused_step_fp16_overflow_skips_the_step_and_backs_the_scale_off(be, dev):
    """an inf in the scaled gradient: GradScaler.step skips optimizer.step() and update() halves the scale (train.py:210-211); growth after `growth_interval` clean steps"""
    from oracle.parity import vit_pair
    from visiondk_amd import vit
    ref, model = vit_pair(be, dev, 32, 8, 128, 2, 2, 256, 10, operand="fp16")
    torch.manual_seed(8)
    x = torch.randn(4, 3, 32, 32); y = torch.randint(0, 10, (4,))
    step = vit.FusedTrainStep(model, lr=0.05, label_smoothing=0.0, init_scale=2.0 ** 30)      # dlogits * 2^30 overflows fp16 at once
    step.growth_interval = 2
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    step.step(x.to(dev), y.to(dev))
    assert step.skipped_steps() == 1 and step.loss_scale() == 2.0 ** 29
    for n, p in model.named_parameters():
        assert torch.equal(p.detach(), before[n]), n                 # nothing moved, nothing turned NaN
    assert math.isfinite(step.loss_value())
    step.loss_state[0] = 256.0                                       # a usable scale: the step goes through, and the second clean step doubles the scale
    step.step(x.to(dev), y.to(dev))
    assert step.skipped_steps() == 1 and step.loss_scale() == 256.0
    moved = sum(int(not torch.equal(p.detach(), before[n])) for n, p in model.named_parameters())
    assert moved == len(before)
    step.step(x.to(dev), y.to(dev))
    assert step.loss_scale() == 512.0
    for n, p in model.named_parameters():
        assert torch.isfinite(p).all(), n


def test_set_operand_switches_an_existing_model(be, dev):
    """the same weights evaluated in both formats: fp16 is the closer one to the fp32 oracle; switching back reproduces the bf16 logits bit for bit"""
    from oracle.parity import vit_pair
    ref, model = vit_pair(be, dev, 32, 8, 128, 2, 2, 256, 10)
    torch.manual_seed(9)
    x = torch.randn(3, 3, 32, 32)
    with torch.no_grad():
        want = ref(x)
        lb = model(x.to(dev)).clone()
        model.engine.set_operand("fp16")
        lh = model(x.to(dev)).clone()
        model.engine.set_operand("bf16")
        lb2 = model(x.to(dev)).clone()
    assert torch.equal(lb, lb2)
    assert _rel(lh, want) < 0.35 * _rel(lb, want) and _rel(lh, want) <= NORTH_STAR_LOGITS


@pytest.mark.gpu
@pytest.mark.parametrize("name,patch,dim,depth,heads,mlp", [("vit_small_patch16_224", 16, 384, 12, 6, 1536), ("vit_large_patch16_224", 16, 1024, 24, 16, 4096),
                                                            ("vit_base_patch8_224", 8, 768, 12, 12, 3072), ("vit_base_patch16_clip_224", 16, 768, 12, 12, 3072)])
def test_vit_family_full_size_fp16_operands_within_the_stated_tolerance(hip, name, patch, dim, depth, heads, mlp):
    """the narrower and the wider / deeper members of the family (384 and 1024 channels, 24 blocks) the 785-token one (patch 8: the streaming attention kernels) and a CLIP one (pre_norm) at full
    size on fp16 operands against ONE fp32 evaluation of the oracle (the base model's test above also runs the operand-rounded and float64 arms): the same literal bounds"""
    from oracle.parity import vit_pair
    from visiondk_amd import vit
    tv = vit.TIMM_VITS[name]
    assert (tv["dim"], tv["depth"], tv["heads"], tv["mlp_dim"], tv.get("patch_size", 16)) == (dim, depth, heads, mlp, patch)      # the id table is what is tested
    ref, model = vit_pair(hip, "cuda:0", 224, patch, dim, depth, heads, mlp, 1000, seed=2, operand="fp16", pre_norm=tv.get("pre_norm", False), eps=tv.get("ln_eps", 1e-6))
    torch.manual_seed(6)
    x = torch.randn(2, 3, 224, 224); y = torch.randint(0, 1000, (2,))
    S = 1024.0
    lo = model(x.cuda()); lr = ref(x)
    (torch.nn.functional.cross_entropy(lo, y.cuda(), label_smoothing=0.05) * S).backward()
    torch.nn.functional.cross_entropy(lr, y, label_smoothing=0.05).backward()
    errs = sorted((_rel(p.grad / S, pr.grad), n) for (n, p), (_, pr) in zip(model.named_parameters(), ref.named_parameters()))
    print(name, _rel(lo, lr), errs[-1])
    assert _rel(lo, lr) <= NORTH_STAR_LOGITS and errs[-1][0] <= NORTH_STAR_GRAD, (_rel(lo, lr), errs[-1])


@pytest.mark.gpu
def test_siglip_vit_large_336_full_size_fp16_operands_within_the_stated_tolerance(hip):
    """BASELINE.json configs[4]'s model -- vit_large_patch14_siglip_336: 576 tokens (the streaming attention kernels), 24 blocks of width 1024, patch-14 stem, the
    attention-pool head -- at full size on fp16 operands against the fp32 oracle (oracle/vit_ref.SiglipVisionTransformerRef): the literal north_star bounds on the
    logits and on EVERY parameter gradient.  This is the mode bench.py's cfg5 block times (bf16 operands measure 4.2e-3 / 8.0e-3 on the same model)."""
    from oracle.vit_ref import SiglipVisionTransformerRef
    from visiondk_amd import vit
    torch.manual_seed(0)
    ref = SiglipVisionTransformerRef(336, 14, 3, 1000, 1024, 24, 16, 4096)
    model = vit.create_model("vit_large_patch14_siglip_336", num_classes=1000, device="cuda:0", backend=hip, operand="fp16")
    model.load_state_dict({k: v.cuda() for k, v in ref.state_dict().items()}, strict=True)
    x = torch.randn(2, 3, 336, 336); y = torch.randint(0, 1000, (2,))
    S = 1024.0
    lr = ref(x); torch.nn.functional.cross_entropy(lr, y).backward()
    lo = model(x.cuda()); (torch.nn.functional.cross_entropy(lo, y.cuda()) * S).backward()
    got = dict(model.named_parameters())
    errs = sorted(((_rel(got[n].grad / S, p.grad), n) for n, p in ref.named_parameters()), reverse=True)
    print({"logits_rel": _rel(lo.detach(), lr.detach()), "worst_grads": errs[:4], "median_grad": errs[len(errs) // 2]})
    assert _rel(lo.detach(), lr.detach()) <= NORTH_STAR_LOGITS and errs[0][0] <= NORTH_STAR_GRAD, (_rel(lo.detach(), lr.detach()), errs[:4])
