"""K3 attention fwd/bwd on the CPU SIMT emulation vs plain torch fp32 on the same bf16-rounded inputs."""
import pytest
import torch

from visiondk_amd import ops


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _ref(qkv, heads):
    B, N, _ = qkv.shape
    D = qkv.shape[2] // 3
    x = qkv.float().reshape(B, N, 3, heads, D // heads).permute(2, 0, 3, 1, 4)  # timm Attention.forward
    q, k, v = x[0], x[1], x[2]
    att = (q * (D // heads) ** -0.5) @ k.transpose(-2, -1)
    lse = torch.logsumexp(att, -1)
    o = att.softmax(-1) @ v
    return o.transpose(1, 2).reshape(B, N, D), lse


@pytest.mark.parametrize("B,N,H", [(2, 17, 2), (1, 50, 1), (1, 197, 2), (1, 300, 1)])
def test_attention_fwd_bwd(be, dev, B, N, H):
    torch.manual_seed(0)
    D = H * 64
    qkv = (torch.randn(B, N, 3 * D) * 1.5).bfloat16()
    qkv[0, N // 2, :D] *= 4.0
    qkv = qkv.to(dev)   # a peaky row: exercises the online-softmax rescale
    o, lse = ops.attention_fwd(qkv, H, backend=be)
    qr = qkv.float().requires_grad_(True)
    oref, lseref = _ref(qr, H)
    assert _rel(lse, lseref) < 1e-5
    assert _rel(o.float(), oref) < 6e-3        # P and O are rounded to bf16 on the way
    dout = torch.randn(B, N, D).bfloat16().to(dev)
    oref.backward(dout.float())
    # backward consumes the forward's own (bf16) o
    dqkv = ops.attention_bwd(qkv, o, dout, lse, H, backend=be)
    for i, name in enumerate("qkv"):
        got = dqkv[..., i * D:(i + 1) * D].float(); ref = qr.grad[..., i * D:(i + 1) * D]
        assert _rel(got, ref) < 1.5e-2, name


@pytest.mark.parametrize("grid", [1, 2, 3])
def test_attention_fwd_persistent_workgroups(be, dev, grid, monkeypatch):
    """For N <= 256 the forward kernel is persistent over (batch, head) items with the next item's K / V prefetched under the current one's q-tile rounds;
    real launches give a workgroup several items only for B * H > 512, so the tests force it (1 = one workgroup walks over everything)."""
    monkeypatch.setenv("VDK_ATTN_GRID", str(grid))
    torch.manual_seed(1)
    B, N, H = 3, 45, 2
    D = H * 64
    qkv = (torch.randn(B, N, 3 * D) * 1.2).bfloat16().to(dev)
    o, lse = ops.attention_fwd(qkv, H, backend=be)
    oref, lseref = _ref(qkv.float(), H)
    assert _rel(lse, lseref) < 1e-5 and _rel(o.float(), oref) < 6e-3
    monkeypatch.delenv("VDK_ATTN_GRID")
    o2, lse2 = ops.attention_fwd(qkv, H, backend=be)            # one item per workgroup: bit-identical
    assert torch.equal(o, o2) and torch.equal(lse, lse2)
