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


@pytest.fixture(params=[False, True], ids=["routed", "long_sequence_kernels"])
def legacy(request, be):
    """N <= 256 is served by csrc/attention_small.hip (exact softmax, fused backward); the flash-style kernels of csrc/attention.hip are forced for the second pass"""
    be.lib.vdk_attention_force_legacy(1 if request.param else 0)
    yield request.param
    be.lib.vdk_attention_force_legacy(-1)


@pytest.mark.parametrize("B,N,H", [(2, 17, 2), (1, 50, 1), (1, 197, 2), (2, 224, 1), (1, 256, 1), (1, 300, 1), (2, 257, 2), (1, 577, 1), (9, 384, 1), (3, 240, 2)])
def test_attention_fwd_bwd(be, dev, B, N, H, legacy):
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


@pytest.mark.parametrize("B,N,H", [(3, 17, 2), (2, 65, 2), (1, 197, 1), (1, 224, 2)])
def test_attention_small_vs_bf16_operand_oracle(be, dev, B, N, H):
    """The short-sequence kernels round exactly where autocast does (oracle/bf16ops.py): normalised P once, O once, dS once.  What is left against the
    oracle's bf16-operand mode is fp32 summation order and the rare element whose rounding flips: <= 2e-4 (the fp32 oracle is 2e-3 ... 1e-2 away)."""
    from oracle import bf16ops
    torch.manual_seed(3)
    D = H * 64
    qkv = (torch.randn(B, N, 3 * D) * 1.3).bfloat16()
    qkv[0, N // 3, D:D + 64] *= 5.0                    # one dominant key: a near one-hot softmax row next to flat ones
    o, lse = ops.attention_fwd(qkv.to(dev), H, backend=be)
    x = qkv.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    q, k, v = (t.clone().requires_grad_(True) for t in (x[0], x[1], x[2]))
    with bf16ops.precision("bf16_operands"):
        oo = bf16ops.attention(q, k, v, 0.125)
    exp = oo.detach().transpose(1, 2).reshape(B, N, D)
    assert _rel(o.float().cpu(), exp) < 2e-4
    dout = torch.randn(B, N, D).bfloat16()
    oo.backward(dout.float().reshape(B, N, H, 64).transpose(1, 2))
    dqkv = ops.attention_bwd(qkv.to(dev), o, dout.to(dev), lse, H, backend=be).float().cpu().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    for got, ref, name in zip(dqkv, (q.grad, k.grad, v.grad), "qkv"):
        assert _rel(got, bf16ops.rb(ref)) < 2e-4, name


@pytest.mark.parametrize("grid", [1, 2, 3])
def test_attention_persistent_workgroups(be, dev, grid, monkeypatch):
    """The short-sequence kernels are persistent over (batch, head) items; real launches give a workgroup several items only for B * H > 256 / 512, so the
    tests force it (1 = one workgroup walks over everything, reusing its LDS arrays and zero rows item after item)."""
    torch.manual_seed(1)
    B, N, H = 3, 45, 2
    D = H * 64
    qkv = (torch.randn(B, N, 3 * D) * 1.2).bfloat16().to(dev)
    dout = torch.randn(B, N, D).bfloat16().to(dev)
    o2, lse2 = ops.attention_fwd(qkv, H, backend=be)            # one item per workgroup
    d2 = ops.attention_bwd(qkv, o2, dout, lse2, H, backend=be)
    monkeypatch.setenv("VDK_ATTN_GRID", str(grid))
    o, lse = ops.attention_fwd(qkv, H, backend=be)
    d = ops.attention_bwd(qkv, o, dout, lse, H, backend=be)
    monkeypatch.delenv("VDK_ATTN_GRID")
    assert torch.equal(o, o2) and torch.equal(lse, lse2) and torch.equal(d, d2)


@pytest.mark.parametrize("grid", [1, 2, 3])
def test_attention_fwd_persistent_workgroups(be, dev, grid, monkeypatch, legacy):
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


@pytest.mark.parametrize("grid", [1, 8, 24])
def test_attention_long_fwd_units_per_workgroup(be, dev, grid, monkeypatch):
    """N > 256 (csrc/attention_long.hip): a work unit is (batch, head, group of 4 query tiles), dealt to XCDs by item mod 8, and a workgroup walks over its units with the
    K / V chunk buffers running on across unit boundaries (the next unit's first chunk is requested under the current unit's last).  B * H = 10 items (two residues mod 8 hold
    two items, six hold one), N = 290 -> 10 query tiles = 3 groups (the last with two idle waves), 4 chunks of 96 keys (the last ragged).  Any grid gives the same bits."""
    monkeypatch.setenv("VDK_ATTN_GRID", str(grid))
    torch.manual_seed(2)
    B, N, H = 5, 290, 2
    D = H * 64
    qkv = (torch.randn(B, N, 3 * D) * 1.2).bfloat16().to(dev)
    o, lse = ops.attention_fwd(qkv, H, backend=be)
    oref, lseref = _ref(qkv.float(), H)
    assert _rel(lse, lseref) < 1e-5 and _rel(o.float(), oref) < 6e-3
    monkeypatch.delenv("VDK_ATTN_GRID")
    o2, lse2 = ops.attention_fwd(qkv, H, backend=be)
    assert torch.equal(o, o2) and torch.equal(lse, lse2)


@pytest.mark.parametrize("grid", [1, 8, 24])
def test_attention_long_bwd_units_per_workgroup(be, dev, grid, monkeypatch):
    """N > 224 backward (csrc/attention_long.hip): the q kernel (dQ, D) and the kv kernel (dK, dV) walk the same unit numbering as the forward, the chunk buffers (and
    the kv kernel's staged lse / D values) run on across unit boundaries.  Any grid gives the same bits; the flash-style kernels agree within the bf16 rounding of dS."""
    torch.manual_seed(3)
    B, N, H = 5, 290, 2
    D = H * 64
    qkv = (torch.randn(B, N, 3 * D) * 1.2).bfloat16().to(dev)
    dout = torch.randn(B, N, D).bfloat16().to(dev)
    o, lse = ops.attention_fwd(qkv, H, backend=be)
    monkeypatch.setenv("VDK_ATTN_GRID", str(grid))
    d = ops.attention_bwd(qkv, o, dout, lse, H, backend=be)
    monkeypatch.delenv("VDK_ATTN_GRID")
    d2 = ops.attention_bwd(qkv, o, dout, lse, H, backend=be)
    assert torch.equal(d, d2)
    be.lib.vdk_attention_force_legacy(1)
    try:
        d3 = ops.attention_bwd(qkv, o, dout, lse, H, backend=be)
    finally:
        be.lib.vdk_attention_force_legacy(-1)
    assert _rel(d.float(), d3.float()) < 6e-3


@pytest.mark.parametrize("B,N,H", [(2, 17, 2), (1, 50, 1), (2, 197, 1), (1, 224, 1)])
def test_attention_streaming_kernels_at_short_n(be, dev, B, N, H, monkeypatch):
    """The streaming kernels of csrc/attention_long.hip forced at short N (VDK_ATTN_LONG_MIN): single-chunk sequences, chunks with whole tiles beyond N (skipped), groups
    with idle waves -- against torch fp32 and, for the backward, against the LDS-resident kernels (same recompute form, same rounding points: dS rounded to bf16)."""
    torch.manual_seed(4)
    D = H * 64
    qkv = (torch.randn(B, N, 3 * D) * 1.3).bfloat16().to(dev)
    dout = torch.randn(B, N, D).bfloat16().to(dev)
    o_s, lse_s = ops.attention_fwd(qkv, H, backend=be)
    d_s = ops.attention_bwd(qkv, o_s, dout, lse_s, H, backend=be)
    monkeypatch.setenv("VDK_ATTN_LONG_MIN", "1")
    o, lse = ops.attention_fwd(qkv, H, backend=be)
    d = ops.attention_bwd(qkv, o_s, dout, lse_s, H, backend=be)
    monkeypatch.delenv("VDK_ATTN_LONG_MIN")
    oref, lseref = _ref(qkv.float(), H)
    assert _rel(lse, lseref) < 1e-5 and _rel(o.float(), oref) < 6e-3
    assert _rel(o.float(), o_s.float()) < 6e-3
    assert _rel(d.float(), d_s.float()) < 2e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,N,H", [(2, 17, 2), (1, 50, 1), (2, 64, 1), (1, 197, 2), (3, 100, 1), (1, 224, 1), (5, 197, 1)])
@pytest.mark.parametrize("grid", [0, 2])
def test_attention_bwd_one_pass_ds_exchange(be, dev, B, N, H, grid, dtype, monkeypatch):
    """The LDS-resident backward (csrc/attention_small.hip, N <= 224): one pass, the waves exchange dS ([key][query] rows in LDS) and split dQ_j by output block (16 x 16
    blocks on v_mfma_f32_16x16x32, each contracted over ALL keys).  Reference: the streaming two-kernel backward of csrc/attention_long.hip forced at the same N (same
    rounding points: P and dS rounded to 16 bits once) and torch fp32.  Both operand formats; grid = 2: a workgroup walks over several (batch, head) items -- the next
    item's first tiles, K tile and K / V fragments are requested behind the current item's last barrier, the store staging lives in the dS buffers."""
    torch.manual_seed(5)
    D = H * 64
    qkv = (torch.randn(B, N, 3 * D) * 1.4).to(dtype)
    qkv[0, N // 2, :D] *= 3.0
    qkv = qkv.to(dev)
    dout = torch.randn(B, N, D).to(dtype).to(dev)
    o, lse = ops.attention_fwd(qkv, H, backend=be)
    monkeypatch.setenv("VDK_ATTN_LONG_MIN", "1")
    ref = ops.attention_bwd(qkv, o, dout, lse, H, backend=be)          # the streaming two-kernel form
    monkeypatch.delenv("VDK_ATTN_LONG_MIN")
    if grid:
        monkeypatch.setenv("VDK_ATTN_GRID", str(grid))
    got = ops.attention_bwd(qkv, o, dout, lse, H, backend=be)
    got2 = ops.attention_bwd(qkv, o, dout, lse, H, backend=be)
    assert torch.equal(got, got2)                                       # deterministic (no atomics, fixed summation order)
    tol = 3e-3 if dtype == torch.bfloat16 else 4e-4
    assert _rel(got[..., 2 * D:].float(), ref[..., 2 * D:].float()) < tol
    assert _rel(got[..., D:2 * D].float(), ref[..., D:2 * D].float()) < tol
    assert _rel(got[..., :D].float(), ref[..., :D].float()) < tol       # dQ: 16-bit outputs of two summation orders
    qr = qkv.float().requires_grad_(True)
    oref, _ = _ref(qr, H)
    oref.backward(dout.float())
    for lo, hi_ in ((0, D), (D, 2 * D), (2 * D, 3 * D)):
        assert _rel(got[..., lo:hi_].float(), qr.grad[..., lo:hi_]) < (1.5e-2 if dtype == torch.bfloat16 else 2e-3)
