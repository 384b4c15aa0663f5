"""ORACLE — test infrastructure only.

getIndicePair<NDim> for any NDim (the reference instantiates 2, 3 and 4: spconv/src/all.cc:22-29), restated with plain Python
integers, one input cell and one candidate output at a time, in the order of the reference's CPU functors:
  candidate outputs + kernel offset of an input cell   geometry.h:25-84 (getValidOutPos; C division truncates toward zero)
  regular conv: first-touch output numbering             geometry.h:144-194 (getIndicePairsConv)
  submanifold: outputs = active inputs, last duplicate wins   geometry.h:247-297 (getIndicePairsSubM), padding = ksize // 2 spconv_ops.h:76-80
  canonical=True: outputs renumbered in ascending flat (batch, cell) order = the GPU path's torch::_unique order (spconv_ops.h:128-137)
Parity pin: tests/test_rulebook_nd_oracle.py holds it against oracle/rulebook_oracle.c (the 3-D restatement the HIP rulebook is
tested against, itself checked on the committed rulebook goldens) at NDim = 3 on random and edge cases; NDim = 2 against the same
code through the (b, 0, y, x) embedding.  Small inputs only (pure Python loops)."""
import itertools

import numpy as np


def _tdiv(a, b):
    return a // b if a >= 0 else -((-a) // b)


def valid_out_pos(pos, ksize, stride, padding, dilation, out_shape):
    """-> list of (output cell tuple, kernel offset) in the reference's enumeration order (invalid cells dropped)."""
    nd = len(pos)
    lowers = [_tdiv(pos[i] - (ksize[i] - 1) * dilation[i] - 1 + stride[i] + padding[i], stride[i]) for i in range(nd)]
    uppers = [_tdiv(pos[i] + padding[i], stride[i]) for i in range(nd)]
    sizes = [_tdiv(uppers[i] - lowers[i], dilation[i]) + 1 for i in range(nd)]
    out = []
    if any(s <= 0 for s in sizes):
        return out
    for counter in itertools.product(*[range(s) for s in sizes]):      # last dimension fastest, as the carry loop of :73-79
        valid, m, offset, cell = True, 1, 0, [0] * nd
        for j in range(nd - 1, -1, -1):
            val = uppers[j] - counter[j] * dilation[j]
            cell[j] = val
            if val < 0 or val > out_shape[j] - 1:
                valid = False
            offset += _tdiv(m * (pos[j] - val * stride[j] + padding[j]), dilation[j])
            m *= ksize[j]
        if valid:
            out.append((tuple(cell), offset))
    return out


def conv_out_shape(shape, k, s, p, d):
    return [(shape[i] + 2 * p[i] - d[i] * (k[i] - 1) - 1) // s[i] + 1 for i in range(len(shape))]


def indice_pairs(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm=False, canonical=True):
    """-> (outids [M, 1 + NDim] i32, pairs [K, 2, n] i32 (-1 padded), num [K] i32)."""
    ind = np.asarray(indices, np.int64)
    n, nd = ind.shape[0], ind.shape[1] - 1
    ksize, dilation = [int(v) for v in ksize], [int(v) for v in dilation]
    kvol = int(np.prod(ksize))
    pairs = np.full((kvol, 2, n), -1, np.int32)
    num = np.zeros((kvol,), np.int32)
    if subm:
        stride, padding, out_shape = [1] * nd, [k // 2 for k in ksize], [int(v) for v in spatial_shape]
        grid = {}
        for j in range(n):
            grid[tuple(ind[j])] = j
        for j in range(n):
            for cell, off in valid_out_pos([int(v) for v in ind[j, 1:]], ksize, stride, padding, dilation, out_shape):
                o = grid.get((int(ind[j, 0]),) + cell, -1)
                if o > -1:
                    pairs[off, 0, num[off]] = j
                    pairs[off, 1, num[off]] = o
                    num[off] += 1
        return ind.astype(np.int32), pairs, num
    stride, padding = [int(v) for v in stride], [int(v) for v in padding]
    out_shape = conv_out_shape([int(v) for v in spatial_shape], ksize, stride, padding, dilation)
    grid, outs = {}, []
    for j in range(n):
        for cell, off in valid_out_pos([int(v) for v in ind[j, 1:]], ksize, stride, padding, dilation, out_shape):
            key = (int(ind[j, 0]),) + cell
            o = grid.get(key)
            if o is None:
                o = grid[key] = len(outs)
                outs.append(key)
            pairs[off, 0, num[off]] = j
            pairs[off, 1, num[off]] = o
            num[off] += 1
    outids = np.asarray(outs, np.int32).reshape(len(outs), nd + 1)
    if canonical and len(outs):
        order = sorted(range(len(outs)), key=lambda o: outs[o])          # tuple order == flat (batch, cell) order
        rank = np.empty(len(outs), np.int32)
        rank[order] = np.arange(len(outs), dtype=np.int32)
        outids = outids[order]
        live = pairs[:, 1] >= 0
        pairs[:, 1][live] = rank[pairs[:, 1][live]]
    return outids, pairs, num
