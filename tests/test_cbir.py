"""Hot path B (csrc/cbir.hip) vs the oracle: bit-exact indices AND scores, on the CPU SIMT emulation and (-m gpu)
through the same C ABI on the MI355X."""
import numpy as np
import pytest
import torch

from oracle import cbir as ocbir
from visiondk_amd import cbir


def _data(nq, n, d, seed=0, normalize=True):
    rng = np.random.default_rng(seed)
    g = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((nq, d), dtype=np.float32)
    if normalize:
        g, q = ocbir.l2norm_rows(g), ocbir.l2norm_rows(q)
    return q, g


def _search(be, q, g, k, cap=4096, idx_base=0):
    idx = cbir.FlatIPIndex(q.shape[1], backend=be, device="cuda" if be.device_only else "cpu", cap=cap, idx_base=idx_base)
    idx.add(g)
    return idx.search(q, k)


@pytest.mark.parametrize("nq,n,d,k", [(33, 1000, 128, 100), (5, 700, 64, 10), (130, 300, 128, 7), (3, 50, 128, 100),
                                      (4, 0, 128, 5), (9, 1300, 260, 20), (2, 129, 12, 129)])
def test_search_bit_exact(be, dev, nq, n, d, k):
    q, g = _data(nq, n, d)
    s, i = _search(be, q, g, k)
    so, io = ocbir.flat_ip_search(q, g, k)
    np.testing.assert_array_equal(i, io)
    np.testing.assert_array_equal(s.view(np.uint32), so.view(np.uint32))


def test_ties_lower_index_first(be, dev):
    q, g = _data(4, 400, 128, seed=3)
    g[37] = g[5]; g[399] = g[5]; g[200] = g[5]          # exact duplicates -> exact score ties
    q[0] = g[5]
    s, i = _search(be, q, g, 10)
    so, io = ocbir.flat_ip_search(q, g, 10)
    np.testing.assert_array_equal(i, io)
    assert list(i[0, :4]) == [5, 37, 200, 399]


def test_multi_stage_small_cap(be, dev):
    # cap small -> many stages with rising thresholds; result must not depend on staging
    q, g = _data(7, 3000, 128, seed=5)
    s, i = _search(be, q, g, 16, cap=200)
    so, io = ocbir.flat_ip_search(q, g, 16)
    np.testing.assert_array_equal(i, io)
    np.testing.assert_array_equal(s.view(np.uint32), so.view(np.uint32))


def test_sorted_gallery_worst_case(be, dev):
    # adversarial order: similarity increases with the index, every row beats the threshold
    q, g = _data(2, 1500, 128, seed=7)
    order = np.argsort(g @ q[0])
    g = np.ascontiguousarray(g[order])
    s, i = _search(be, q, g, 32, cap=300)
    so, io = ocbir.flat_ip_search(q, g, 32)
    np.testing.assert_array_equal(i, io)


def test_idx_base_and_merge(be, dev):
    q, g = _data(6, 900, 128, seed=9)
    k = 12
    parts_s, parts_i = [], []
    for lo, hi in [(0, 300), (300, 600), (600, 900)]:
        s, i = _search(be, q, g[lo:hi], k, idx_base=lo)
        parts_s.append(torch.from_numpy(s)); parts_i.append(torch.from_numpy(i))
    ms, mi = cbir.merge_topk(torch.stack(parts_s).to(dev), torch.stack(parts_i).to(dev), backend=be)
    ms, mi = ms.cpu(), mi.cpu()
    so, io = ocbir.flat_ip_search(q, g, k)
    np.testing.assert_array_equal(mi.numpy(), io)
    np.testing.assert_array_equal(ms.numpy().view(np.uint32), so.view(np.uint32))


def test_l2_normalize(be, dev):
    x = torch.randn(37, 130)
    x[3] = 0
    x = x.to(dev)
    y = cbir.l2_normalize(x, backend=be)
    ref = torch.nn.functional.normalize(x, p=2, dim=1, eps=1e-12)
    torch.testing.assert_close(y, ref, rtol=1e-6, atol=1e-7)
