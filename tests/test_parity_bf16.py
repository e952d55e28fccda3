"""Parity of hot path A against the oracle's bf16-operand mode (oracle/bf16ops.py): the SAME network with a bf16 rounding at exactly the tensors the
reference's autocast (engine/procedure/train.py:118) holds in bf16.  north_star asks for logits / embeddings within 1e-3 rel of the reference; against the
fp32 oracle a bf16-operand engine sits at 5e-3 ... 1e-2 whatever its quality, against this mode what is left is fp32 summation order and the rare element
whose rounding flips, so a kernel that drops a term, mis-scales a gradient or rounds in an extra place shows up two orders of magnitude above the bound.

How small can "what is left" be?  A bf16 rounding is a step function, and a chain of them is a noise AMPLIFIER: a relative perturbation d in the input of a
rounding stage flips a fraction ~d/ulp of its outputs by one ulp (2^-8), i.e. leaves sqrt(d * ulp) behind, so after three or four stages ANY two
evaluations that started 1e-7 apart (different fp32 summation order is enough) sit at the bf16 quantisation noise itself, ~1e-3 per tensor, and the deviations
compound over the depth of the network.  The test measures that floor directly -- the oracle accumulating in float32 (`o32`) against the same oracle
accumulating in float64 (`o64`), two valid evaluations of the SAME bf16-operand arithmetic -- and requires the engine to be no further from either than
1.5x their own distance (+1e-5).  Measured (MI355X, round 2): ViT-B/16 at full depth, batch 8: floor 5.1e-3 (logits) / 9.3e-3 (worst gradient,
patch_embed.proj.weight, the one furthest from the loss), engine 5.2e-3 / 9.2e-3 from o32 and 5.2e-3 / 9.3e-3 from o64; the fp32 oracle is 6.5e-3 / 1.0e-2 away.
north_star's "within 1e-3 rel" is therefore not attainable by ANY bf16-operand evaluation of a 12-block network, the reference's own autocast path
included (it would miss itself by 5e-3 under a different summation order); it IS met, with margin, by `forward_precise` (fp32 MFMA, tests/test_precise.py),
and the bf16 engine meets it per kernel on identical operands (tests/test_gemm.py, test_attention.py: 1e-5 ... 2e-4).

One full-width block (batch 8): floor 2.6e-3 / 5.0e-3, engine 2.6e-3 / 5.1e-3 and 2.4e-3 / 5.1e-3.
Absolute bounds, asserted next to the floor criterion (Frobenius-relative), 1.5x the measured engine values: toy widths (2 blocks): logits 2e-3, gradients 6e-3;
one full-width block: 3.9e-3 / 7.7e-3; full depth: 7.8e-3 / 1.4e-2.

Both arms of that floor are oracle/bf16ops.py, the builder's own restatement of autocast.  The independent arm (round 3): the oracle's plain fp32 module under
`torch.autocast("cpu", dtype=torch.bfloat16)` -- PyTorch's own choice of which ops read bf16 operands, the mechanism engine/procedure/train.py:118 switches on for the
reference.  Measured on the 2-block toys (CPU): PyTorch's autocast lands 3.7e-3 ... 5.7e-3 (logits) / 6.5e-3 ... 7.0e-3 (worst gradient) from fp32, the engine
2.0e-3 ... 3.4e-3 / 5.8e-3 ... 6.4e-3: the engine is CLOSER to the fp32 path than the reference's own mixed-precision mechanism is.  Asserted: the engine's distance
from fp32 is at most 1.25x torch.autocast's, for logits and for the worst gradient."""
import pytest
import torch

LOGITS_TOL_TOY, LOSS_TOL, GRAD_TOL = 2e-3, 1e-3, 6e-3
LOGITS_TOL_FULL, GRAD_TOL_FULL = 7.8e-3, 1.4e-2


from oracle.parity import vit_fwd_bwd_vs_oracle as check_fwd_bwd, vit_pair as _pair  # noqa: E402


def assert_within_floor(r, logits_tol, grad_tol=GRAD_TOL):
    fl = r["floor_o32_vs_o64"]
    for side in ("vs_o32", "vs_o64"):
        got = r[side]
        assert got["logits"] <= 1.5 * fl["logits"] + 1e-5, (side, got, fl)
        assert got["worst_grad"] <= 1.5 * fl["worst_grad"] + 1e-5, (side, got, fl)
        assert got["logits"] < logits_tol and got["loss"] < LOSS_TOL and got["worst_grad"] < grad_tol, (side, got)
    assert r["vs_fp32"]["logits"] < 2e-2          # and the engine is a bf16-operand engine, not something else
    ac = r["torch_autocast_vs_fp32"]              # independent arm: no further from fp32 than PyTorch's own bf16 autocast of the same module
    assert r["vs_fp32"]["logits"] <= 1.25 * ac["logits"] + 1e-5 and r["vs_fp32"]["worst_grad"] <= 1.25 * ac["worst_grad"] + 1e-5, (r["vs_fp32"], ac)


@pytest.mark.parametrize("img,patch,B", [(32, 8, 3), (64, 8, 4), (112, 8, 2)])       # 17, 65 and 197 tokens
def test_vit_forward_backward_vs_bf16_operand_oracle(be, dev, img, patch, B):
    ref, model = _pair(be, dev, img, patch, 128, 2, 2, 256, 10)
    torch.manual_seed(5)
    x = torch.randn(B, 3, img, img)
    y = torch.randint(0, 10, (B,))
    r = check_fwd_bwd(ref, model, x, y, dev)
    print(r)
    assert_within_floor(r, LOGITS_TOL_TOY)


@pytest.mark.gpu
def test_vit_base_patch16_full_size_vs_bf16_operand_oracle(hip):
    """BASELINE.json configs[1] at full width and depth (ViT-B/16, 224, 1000 classes), batch 8: the 256x256 GEMM tiles, the TN weight-gradient kernel with its
    split-K, the fused bias-gradient column sums, the dGELU epilogue, the short-sequence attention kernels at N = 197 — every parameter gradient."""
    ref, model = _pair(hip, "cuda:0", 224, 16, 768, 12, 12, 3072, 1000, seed=2)
    torch.manual_seed(6)
    x = torch.randn(8, 3, 224, 224)
    y = torch.randint(0, 1000, (8,))
    r = check_fwd_bwd(ref, model, x, y, "cuda:0")
    print(r)
    assert_within_floor(r, LOGITS_TOL_FULL, GRAD_TOL_FULL)


@pytest.mark.gpu
def test_vit_base_width_single_block_vs_bf16_operand_oracle(hip):
    """One full-width block (dim 768, 12 heads, mlp 3072, 197 tokens, batch 8) between the patch embedding and the head: the same kernels and tile shapes as the
    12-block network, before the rounding noise has compounded over the depth."""
    ref, model = _pair(hip, "cuda:0", 224, 16, 768, 1, 12, 3072, 1000, seed=4)
    torch.manual_seed(8)
    x = torch.randn(8, 3, 224, 224)
    y = torch.randint(0, 1000, (8,))
    r = check_fwd_bwd(ref, model, x, y, "cuda:0")
    print(r)
    assert_within_floor(r, 3.9e-3, 7.7e-3)
