"""Full-size architectures on the MI355X against the CPU oracles (the emulated tests use toy widths; these exercise the real tile shapes: 256x256 GEMM tiles, the
row-streaming depthwise kernel at 56/28/14/7, ResNet-18's implicit-GEMM convolutions at 112 .. 7).  GPU only: the emulator would take minutes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm()).item()


def test_convnext_base_forward_vs_oracle(hip):
    from oracle.convnext_ref import ConvNeXtRef
    from visiondk_amd import convnext
    torch.manual_seed(0)
    model = convnext.create_model("convnext_base", device="cuda:0", backend=hip)
    ref = ConvNeXtRef()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith("gamma"):
                p.fill_(0.1)
    model.load_state_dict(ref.state_dict(), strict=True)
    x = torch.randn(2, 3, 224, 224)
    with torch.no_grad():
        exp = ref(x)
    assert _rel(model.forward_precise(x.cuda()), exp) < 1e-4
    assert _rel(model(x.cuda()).detach(), exp) < 3e-2


def test_resnet18_eval_forward_and_train_grads_vs_oracle(hip):
    from oracle.resnet_ref import ResNetRef
    from visiondk_amd import resnet
    torch.manual_seed(1)
    model = resnet.create_model("resnet18", num_classes=5, device="cuda:0", backend=hip)
    ref = ResNetRef(5)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.weight.normal_(0, (2.0 / m.weight[0].numel()) ** 0.5)
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    model.load_state_dict(ref.state_dict(), strict=True)
    x = torch.randn(4, 3, 224, 224)
    model.eval(); ref.eval()
    with torch.no_grad():
        assert _rel(model(x.cuda()), ref(x)) < 3e-2
    # training mode at bs 16: fc gradient (clean signal, before the ReLU/BatchNorm mask sensitivity accumulates) and loss
    model.train(); ref.train()
    x = torch.randn(16, 3, 224, 224); t = (torch.rand(16, 5) > 0.5).float()
    lr = torch.nn.functional.binary_cross_entropy_with_logits(ref(x), t); lr.backward()
    lo = torch.nn.functional.binary_cross_entropy_with_logits(model(x.cuda()), t.cuda()); lo.backward()
    assert abs(lo.item() - lr.item()) < 2e-2 * abs(lr.item())
    got = dict(model.named_parameters())
    assert _rel(got["fc.weight"].grad, ref.fc.weight.grad) < 5e-2
    assert _rel(got["fc.bias"].grad, ref.fc.bias.grad) < 5e-2


def test_vit_base_logits_precise_vs_oracle(hip):
    from oracle.vit_ref import VisionTransformerRef
    from visiondk_amd import vit
    torch.manual_seed(2)
    model = vit.create_model("vit_base_patch16_224", num_classes=1000, device="cuda:0", backend=hip)
    ref = VisionTransformerRef()
    model.load_state_dict(ref.state_dict(), strict=True)
    x = torch.randn(2, 3, 224, 224)
    with torch.no_grad():
        exp = ref(x)
    assert _rel(model.forward_precise(x.cuda()), exp) < 1e-4
    assert _rel(model(x.cuda()).detach(), exp) < 3e-2


@pytest.mark.parametrize("B,N,H", [(8, 576, 16), (4, 577, 16), (8, 257, 16)])
def test_attention_at_vit_large_sequence_lengths(hip, B, N, H):
    """The streaming attention kernels (csrc/attention_long.hip) at the sequence lengths they serve -- ViT-L/14 at 336 (576 tokens, SigLIP: no class token; 577 with one)
    and at 224 (257) -- with enough (batch, head) items that workgroups walk over several units: forward against torch fp32 on the same bf16 inputs, backward against
    torch autograd, and both against the flash-style kernels of csrc/attention.hip (a second implementation with different tiling and summation order)."""
    from visiondk_amd import ops
    torch.manual_seed(7)
    D = H * 64
    qkv = (torch.randn(B, N, 3 * D, device="cuda") * 1.3).bfloat16()
    dout = torch.randn(B, N, D, device="cuda").bfloat16()
    o, lse = ops.attention_fwd(qkv, H, backend=hip)
    d = ops.attention_bwd(qkv, o, dout, lse, H, backend=hip)
    hip.lib.vdk_attention_force_legacy(1)
    try:
        o2, lse2 = ops.attention_fwd(qkv, H, backend=hip)
        d2 = ops.attention_bwd(qkv, o, dout, lse, H, backend=hip)
    finally:
        hip.lib.vdk_attention_force_legacy(-1)
    x = qkv.float().requires_grad_(True)
    q, k, v = (t.reshape(B, N, H, 64).transpose(1, 2) for t in x.split(D, dim=2))
    att = (q * 0.125) @ k.transpose(-2, -1)
    ref = (att.softmax(-1) @ v).transpose(1, 2).reshape(B, N, D)
    ref.backward(dout.float())
    assert _rel(lse, torch.logsumexp(att, -1).detach()) < 1e-5 and _rel(lse, lse2) < 1e-6
    assert _rel(o.float(), ref.detach()) < 6e-3 and _rel(o.float(), o2.float()) < 6e-3
    assert _rel(d.float(), x.grad) < 1.5e-2 and _rel(d.float(), d2.float()) < 6e-3
