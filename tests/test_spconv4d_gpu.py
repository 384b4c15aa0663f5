"""GPU parity of 4-D sparse convolution (`get_indice_pairs_4d`; SparseConv4d / SubMConv4d, reference conv.py:285-309,
:457-480): rulebook bit-exact against oracle/rulebook_nd.py, features and gradients within 1e-4 of the gather-mm-scatter
restatement (oracle.indice_conv), and — independently of any rulebook — the densified output equal to a dense 4-D
cross-correlation composed from torch conv3d calls in float64."""
import numpy as np
import pytest
import torch

import oracle
from oracle import rulebook_nd
import pcdet.ops.spconv as spconv
from pcdet.ops.spconv import ops
from sparse_util import canon_pairs
from test_rulebook_nd_oracle import CASES4, active, conv4d_dense

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", CASES4)
def test_rulebook_4d_matches_the_restatement(gpu, case):
    batch, shape, n, k, s, p, d, subm = case
    ind = active(sum(shape) + n, batch, shape, n)
    o_out, o_pairs, o_num = rulebook_nd.indice_pairs(ind, batch, shape, k, s, p, d, subm=subm)
    outids, pairs, num = ops.get_indice_pairs(torch.from_numpy(ind).to(gpu), batch, shape, k, s, p, d, 0, subm, False)
    assert outids.dtype == torch.int32 and pairs.dtype == torch.int32 and pairs.shape == (int(np.prod(k)), 2, n)
    assert np.array_equal(outids.cpu().numpy(), o_out)
    from pcdet.ops.spconv import sparse_conv_ext as ext       # the pybind name of the reference (all.cc:28-29)
    e_out, e_pairs, e_num = ext.get_indice_pairs_4d(torch.from_numpy(ind).to(gpu), batch, list(o_out.max(0)[1:] + 1) if len(o_out) else shape, shape,
                                                    k, s, p, d, [0] * 4, int(subm), 0)
    assert torch.equal(e_out, outids) and torch.equal(e_pairs, pairs) and torch.equal(e_num, num)
    num, pairs = num.cpu().numpy(), pairs.cpu().numpy()
    assert np.array_equal(num, o_num)
    for kk, (a, b) in enumerate(zip(canon_pairs(pairs, num), canon_pairs(o_pairs, o_num))):
        assert np.array_equal(a, b), f"offset {kk}"
        assert (pairs[kk, :, int(num[kk]):] == -1).all()


@pytest.mark.parametrize("k,s,p", [(3, 1, 1), (3, 2, 1), (2, 2, 0), ([3, 1, 3, 3], [2, 1, 1, 2], [0, 0, 1, 1])])
def test_sparse_conv4d_equals_a_dense_4d_convolution(gpu, k, s, p):
    batch, shape, cin, cout = 2, [4, 5, 8, 7], 8, 16
    ind = active(9, batch, shape, 400)
    torch.manual_seed(1)
    feats = torch.randn(ind.shape[0], cin, device=gpu, requires_grad=True)
    conv = spconv.SparseConv4d(cin, cout, k, stride=s, padding=p, bias=False).to(gpu)
    x = spconv.SparseConvTensor(feats, torch.from_numpy(ind).to(gpu), shape, batch)
    y = conv(x)
    ks, ss, ps = conv.kernel_size, conv.stride, conv.padding
    dense_in = x.dense().detach().double().cpu()                                   # [B, C, T, Z, Y, X]
    want = conv4d_dense(dense_in, conv.weight.detach().double().cpu(), ss, ps)
    got = y.dense().detach().double().cpu()
    assert got.shape == want.shape
    assert float((got - want).abs().max()) < 1e-4 * float(want.abs().max())
    # gradients of sum(y * g): features and weight against autograd through the dense composition
    g = torch.randn_like(y.features)
    (y.features * g).sum().backward()
    xd = dense_in.clone().requires_grad_(True)
    wd = conv.weight.detach().double().cpu().requires_grad_(True)
    gd = spconv.SparseConvTensor(g, y.indices, y.spatial_shape, batch).dense().double().cpu()
    (conv4d_dense(xd, wd, ss, ps) * gd).sum().backward()
    i = torch.from_numpy(ind).long()
    want_df = xd.grad[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3], i[:, 4]]
    assert float((feats.grad.double().cpu() - want_df).abs().max()) < 1e-4 * float(want_df.abs().max())
    assert float((conv.weight.grad.double().cpu() - wd.grad).abs().max()) < 1e-4 * float(wd.grad.abs().max())


def test_subm_conv4d_chain_matches_the_gather_mm_restatement(gpu):
    """SubMConv4d -> BatchNorm1d -> ReLU -> SparseConv4d(stride 2) in a SparseSequential, with a shared indice_key reused by a second
    SubMConv4d: features and all gradients against oracle.indice_conv / indice_conv_backward on the restated rulebooks."""
    batch, shape = 2, [4, 5, 8, 7]
    ind = active(3, batch, shape, 350)
    torch.manual_seed(2)
    a = spconv.SubMConv4d(8, 16, 3, bias=True, indice_key="s4").to(gpu)
    b = spconv.SubMConv4d(16, 16, 3, bias=False, indice_key="s4").to(gpu)
    c = spconv.SparseConv4d(16, 32, 3, stride=2, padding=1, bias=False).to(gpu)
    feats = torch.randn(ind.shape[0], 8, device=gpu, requires_grad=True)
    x = spconv.SparseConvTensor(feats, torch.from_numpy(ind).to(gpu), shape, batch)
    h1 = a(x)
    h2 = b(h1)
    assert h2.find_indice_pair("s4") is h1.find_indice_pair("s4")
    y = c(h2)
    g = torch.randn_like(y.features)
    (y.features * g).sum().backward()
    one = [1, 1, 1, 1]
    _, ps, ns = rulebook_nd.indice_pairs(ind, batch, shape, [3] * 4, one, one, one, subm=True)
    oc, pc, nc = rulebook_nd.indice_pairs(ind, batch, shape, [3] * 4, [2] * 4, one, one)
    assert np.array_equal(y.indices.cpu().numpy(), oc)
    f0 = feats.detach().cpu()
    wa, wb, wc = (m.weight.detach().cpu() for m in (a, b, c))
    r1 = oracle.indice_conv(f0, wa, ps, ns, ind.shape[0], subm=True) + a.bias.detach().cpu()
    r2 = oracle.indice_conv(r1, wb, ps, ns, ind.shape[0], subm=True)
    r3 = oracle.indice_conv(r2, wc, pc, nc, oc.shape[0])
    rel = lambda u, v: float((u.detach().cpu().double() - v.double()).abs().max() / v.double().abs().max())
    assert rel(h1.features, r1) < 1e-4 and rel(h2.features, r2) < 1e-4 and rel(y.features, r3) < 1e-4
    d2, dwc = oracle.indice_conv_backward(r2, wc, g.cpu(), pc, nc)
    d1, dwb = oracle.indice_conv_backward(r1, wb, d2, ps, ns, subm=True)
    d0, dwa = oracle.indice_conv_backward(f0, wa, d1, ps, ns, subm=True)
    assert rel(c.weight.grad, dwc) < 1e-4 and rel(b.weight.grad, dwb) < 1e-4 and rel(a.weight.grad, dwa) < 1e-4
    assert rel(feats.grad, d0) < 1e-4 and rel(a.bias.grad, d1.sum(0)) < 1e-4


@pytest.mark.parametrize("case", CASES4)
def test_hashed_4d_builder_equals_the_sorted_key_formulation(gpu, case):
    """fv2p_rulebook4d_begin / _finish (hash set + radix sort, rulebook.hip) against `ops.nd_tables` (torch sort / unique / searchsorted)
    on the same rows: output rows, both neighbour tables, bit for bit; plus an empty tensor."""
    batch, shape, n, k, s, p, d, subm = case
    ind = torch.from_numpy(active(sum(shape) + n + 1, batch, shape, n)).to(gpu)
    out_shape = shape if subm else ops.get_conv_output_size(shape, k, s, p, d)
    want = ops.nd_tables(ind, batch, shape, out_shape, k, s, p, d, subm)
    got = ops._native_tables_4d(ind, batch, shape, out_shape, k, s, p, d, subm)
    for name, a, b in zip(("outids", "tab_in", "tab_out"), got, want):
        assert a.dtype == torch.int32 and torch.equal(a, b), name
    empty = ops._native_tables_4d(ind[:0], batch, shape, out_shape, k, s, p, d, subm)
    assert empty[0].shape[0] == 0 and empty[1].shape == (int(np.prod(k)), 0)
