"""CPU checks of the N-D rulebook (`get_indice_pairs_4d`): the any-dimension restatement
(oracle/rulebook_nd.py) pinned against the 3-D C restatement (oracle/rulebook_oracle.c) the HIP rulebook is tested against, and
the product's table builder (`pcdet.ops.spconv.ops.nd_tables`, torch sort / unique / searchsorted — it runs on CPU tensors too)
against the restatement at 2, 3 and 4 dimensions.  Bar: bit-exact (output cells, pair counts, canonical pair lists)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from oracle import rulebook_nd
from pcdet.ops.spconv import ops
from sparse_util import canon_pairs


def active(seed, batch, shape, n):
    rng = np.random.default_rng(seed)
    vol = int(np.prod(shape))
    flat = rng.choice(batch * vol, size=min(n, batch * vol), replace=False)
    cols = [flat // vol]
    cell = flat % vol
    for i in range(len(shape)):
        cols.append((cell // int(np.prod(shape[i + 1:]))) % shape[i])
    return np.stack(cols, 1).astype(np.int32)


CASES3 = [  # batch, shape, n, ksize, stride, padding, dilation, subm
    (2, [9, 20, 18], 500, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], True),
    (1, [8, 16, 16], 300, [3, 3, 3], [1, 1, 1], [1, 1, 1], [2, 2, 2], True),
    (1, [8, 10, 10], 250, [2, 2, 2], [1, 1, 1], [1, 1, 1], [1, 1, 1], True),
    (2, [9, 20, 18], 500, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], False),
    (2, [11, 20, 18], 400, [3, 3, 3], [2, 2, 2], [0, 1, 1], [1, 1, 1], False),
    (2, [5, 20, 18], 400, [3, 1, 1], [2, 1, 1], [0, 0, 0], [1, 1, 1], False),
    (1, [9, 12, 12], 300, [3, 3, 3], [1, 1, 1], [0, 0, 0], [2, 2, 2], False),
    (1, [9, 12, 12], 300, [3, 3, 3], [3, 3, 3], [0, 0, 0], [1, 1, 1], False),
    (1, [4, 5, 6], 120, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], False),
]


def same_rulebook(a, b):
    (oa, pa, na), (ob, pb, nb) = a, b
    assert np.array_equal(np.asarray(oa), np.asarray(ob)), "output cells"
    assert np.array_equal(np.asarray(na), np.asarray(nb)), "pair counts"
    for k, (x, y) in enumerate(zip(canon_pairs(np.asarray(pa), np.asarray(na)), canon_pairs(np.asarray(pb), np.asarray(nb)))):
        assert np.array_equal(x, y), f"offset {k}"


def tables_to_pairs(outids, tab_in):
    """reference pair-list format from a tab_in table: ascending input row within an offset, -1 padded."""
    tab = tab_in.numpy()
    kvol, n = tab.shape
    pairs, num = np.full((kvol, 2, n), -1, np.int32), np.zeros((kvol,), np.int32)
    for k in range(kvol):
        rows = np.nonzero(tab[k] >= 0)[0]
        num[k] = len(rows)
        pairs[k, 0, :len(rows)], pairs[k, 1, :len(rows)] = rows, tab[k, rows]
    return outids.numpy(), pairs, num


def product_rulebook(ind, batch, shape, k, s, p, d, subm):
    out_shape = shape if subm else ops.get_conv_output_size(shape, k, s, p, d)
    outids, tab_in, tab_out = ops.nd_tables(torch.from_numpy(ind), batch, shape, out_shape, k, s, p, d, subm)
    # the two tables are each other's inverse
    ti, to = tab_in.numpy(), tab_out.numpy()
    kk, ii = np.nonzero(ti >= 0)
    assert np.array_equal(to[kk, ti[kk, ii]], ii) and (to >= 0).sum() == len(kk)
    return tables_to_pairs(outids, tab_in)


@pytest.mark.parametrize("case", CASES3)
def test_nd_restatement_equals_the_3d_c_restatement(case):
    batch, shape, n, k, s, p, d, subm = case
    ind = active(sum(shape) + n, batch, shape, n)
    want = oracle.indice_pairs(ind, batch, shape, k, s, p, d, subm=subm)
    same_rulebook(rulebook_nd.indice_pairs(ind, batch, shape, k, s, p, d, subm=subm), want)
    same_rulebook(product_rulebook(ind, batch, shape, k, s, p, d, subm), want)


def test_2d_through_the_embedding():
    ind = active(5, 2, [14, 15], 150)
    emb = np.concatenate([ind[:, :1], np.zeros_like(ind[:, :1]), ind[:, 1:]], 1)
    for subm, k, s, p in ((True, [3, 3], [1, 1], [1, 1]), (False, [3, 3], [2, 2], [1, 1]), (False, [2, 2], [2, 2], [0, 0])):
        o3, p3, n3 = oracle.indice_pairs(emb, 2, [1, 14, 15], [1] + k, [1] + s, [0] + p, [1, 1, 1], subm=subm)
        got = rulebook_nd.indice_pairs(ind, 2, [14, 15], k, s, p, [1, 1], subm=subm)
        same_rulebook(got, (o3[:, [0, 2, 3]], p3, n3))
        same_rulebook(product_rulebook(ind, 2, [14, 15], k, s, p, [1, 1], subm), got)


CASES4 = [  # batch, shape (t, z, y, x), n, ksize, stride, padding, dilation, subm
    (2, [4, 5, 8, 7], 300, [3, 3, 3, 3], [1, 1, 1, 1], [1, 1, 1, 1], [1, 1, 1, 1], True),
    (1, [3, 4, 6, 6], 150, [1, 3, 3, 3], [1, 1, 1, 1], [0, 1, 1, 1], [1, 1, 1, 1], True),
    (1, [5, 5, 6, 6], 200, [3, 3, 3, 3], [1, 1, 1, 1], [1, 1, 1, 1], [1, 2, 2, 1], True),
    (2, [4, 5, 8, 7], 300, [3, 3, 3, 3], [2, 2, 2, 2], [1, 1, 1, 1], [1, 1, 1, 1], False),
    (2, [4, 5, 8, 7], 200, [2, 2, 2, 2], [2, 2, 2, 2], [0, 0, 0, 0], [1, 1, 1, 1], False),
    (1, [6, 3, 6, 5], 150, [3, 1, 3, 3], [2, 1, 1, 2], [0, 0, 1, 1], [1, 1, 1, 1], False),
    (1, [5, 4, 5, 5], 120, [3, 3, 3, 3], [1, 1, 1, 1], [0, 0, 0, 0], [2, 1, 1, 2], False),
]


@pytest.mark.parametrize("case", CASES4)
def test_4d_tables_equal_the_restatement(case):
    batch, shape, n, k, s, p, d, subm = case
    ind = active(sum(shape) + n, batch, shape, n)
    want = rulebook_nd.indice_pairs(ind, batch, shape, k, s, p, d, subm=subm)
    assert want[2].sum() > n // 2
    same_rulebook(product_rulebook(ind, batch, shape, k, s, p, d, subm), want)


def test_4d_edge_cases():
    k, one = [3, 3, 3, 3], [1, 1, 1, 1]
    empty = np.zeros((0, 5), np.int32)
    for subm in (True, False):
        outids, pairs, num = product_rulebook(empty, 1, [3, 3, 3, 3], k, one, one, one, subm)
        assert outids.shape == (0, 5) and num.sum() == 0
    lone = np.array([[0, 1, 1, 1, 1]], np.int32)                         # one cell in the middle of a 3^4 grid
    outids, pairs, num = product_rulebook(lone, 1, [3, 3, 3, 3], k, one, one, one, False)
    assert outids.shape == (81, 5) and np.all(num == 1)                  # it feeds all 81 outputs, one per offset
    same_rulebook((outids, pairs, num), rulebook_nd.indice_pairs(lone, 1, [3, 3, 3, 3], k, one, one, one))
    outids, pairs, num = product_rulebook(lone, 1, [3, 3, 3, 3], k, one, one, one, True)
    assert num[40] == 1 and num.sum() == 1                               # submanifold: only the centre tap


def conv4d_dense(x, w, stride, padding):
    """x [B, C, T, Z, Y, X], w [kt, kz, ky, kx, cin, cout] -> [B, cout, To, Zo, Yo, Xo]: cross-correlation, one conv3d per time tap."""
    kt = w.shape[0]
    st, pt = stride[0], padding[0]
    xp = F.pad(x, (0, 0, 0, 0, 0, 0, pt, pt))
    t_out = (x.shape[2] + 2 * pt - kt) // st + 1
    outs = []
    for to in range(t_out):
        acc = 0
        for a in range(kt):
            acc = acc + F.conv3d(xp[:, :, to * st + a], w[a].permute(4, 3, 0, 1, 2), stride=stride[1:], padding=padding[1:])
        outs.append(acc)
    return torch.stack(outs, 2)


@pytest.mark.parametrize("k,s,p", [(3, 1, 1), (3, 2, 1), (2, 2, 0)])
def test_4d_restatement_is_a_4d_convolution(k, s, p):
    """Independent of any rulebook code: gather-mm-scatter over the restated 4-D pairs == a dense 4-D cross-correlation (composed from
    float64 conv3d calls) at the active outputs, and the dense result is zero at every cell the rulebook leaves inactive."""
    batch, shape, cin, cout = 2, [4, 5, 8, 7], 4, 6
    ind = active(9, batch, shape, 400)
    torch.manual_seed(0)
    f, w = torch.randn(ind.shape[0], cin), torch.randn(k, k, k, k, cin, cout)
    o, pr, nm = rulebook_nd.indice_pairs(ind, batch, shape, [k] * 4, [s] * 4, [p] * 4, [1] * 4)
    y = oracle.indice_conv(f, w, pr, nm, o.shape[0])
    dense = torch.zeros(batch, cin, *shape, dtype=torch.float64)
    i = torch.from_numpy(ind).long()
    dense[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3], i[:, 4]] = f.double()
    want = conv4d_dense(dense, w.double(), [s] * 4, [p] * 4)
    oi = torch.from_numpy(o).long()
    assert float((want[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3], oi[:, 4]] - y.double()).abs().max()) < 1e-4
    inactive = torch.ones_like(want[:, 0], dtype=torch.bool)
    inactive[oi[:, 0], oi[:, 1], oi[:, 2], oi[:, 3], oi[:, 4]] = False
    assert float(want.permute(0, 2, 3, 4, 5, 1)[inactive].abs().sum()) == 0
