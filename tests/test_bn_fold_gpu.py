"""GPU parity of the round-6 conv / BatchNorm arrangement (train-time half):

  * fv2p_sparse_conv_rows_bnfin: the conv's LAST workgroup finalises the BatchNorm statistics (mean / invstd / running statistics);
  * the same call with `pre_*`: BatchNorm (+ReLU) of the SOURCE rows applied on the gather (bn1 -> relu -> conv2 of a residual block,
    spconv_backbone.py:47-68, without materialising relu(bn1(y1))) — forward, backward data (sums finalised by the launch) and the
    weight gradient with the same normalisation of its gathered operand;
  * fv2p_batchnorm_apply_res / _backward_res: bn2 + identity + ReLU as one launch forward, its backward with the mask read from the output.

Oracles: oracle.indice_conv / indice_conv_backward on the oracle rulebook (the reference's gather -> mm -> scatter loop) composed with
torch's own BatchNorm / ReLU on the host in float64 — tolerance 1e-4 relative (north_star).  Bit-identity where it is claimed:
the conv over rows normalised on the gather against the same conv over materialised rows (same statistics), and run to run."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle
import pcdet.ops.spconv as spconv
from sparse_util import random_active

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def rel_l2(a, b):
    """Relative L2 distance.  For gradients against a float64 run: a float32 pre-activation within rounding of zero can take the other
    side of its ReLU than the float64 run does (tests/f64_calibration.py); one such element changes the gradient of ~700 neighbouring
    rows two convs upstream by ~1e-3 of the largest entry, whatever the implementation - an element-wise maximum then measures the
    event, not the kernels.  The folded run is held element-wise (1e-5) to the unfolded one, which takes the same decisions."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def ext():
    import fv2p_native
    e = fv2p_native.torch_ext()
    assert e is not None, "lib/fv2p_torch.so is missing"
    return e


def make_input(seed, batch, shape, n, cin, gpu):
    ind = random_active(seed, batch, shape, n)
    rng = np.random.default_rng(seed + 1)
    feats = rng.standard_normal((ind.shape[0], cin)).astype(np.float32)
    return ind, feats, spconv.SparseConvTensor(torch.from_numpy(feats).to(gpu), torch.from_numpy(ind).to(gpu), shape, batch)


def subm_tables(x, conv, planned=False):
    """(rulebook, forward table, flip, backward table, flip) of a submanifold conv on x, as the Python layer passes them.  planned: the
    table carries a tiling plan, as from the third conv of a residual stage on (ops.Rulebook._plan) - a K-split launch then runs on the
    plan's tile count (>= 256 whatever the row count)."""
    from pcdet.ops.spconv import ops
    rb = ops.build_rulebook(x.indices, x.batch_size, x.spatial_shape, conv.kernel_size, conv.stride, conv.padding, conv.dilation,
                            conv.output_padding, conv.subm, conv.transposed)
    for _ in range(3 if planned else 0):
        rb.out_table(conv.in_channels)
    (tab_f, flip_f), (tab_b, flip_b) = rb.out_table(conv.in_channels), rb.in_table(conv.out_channels)
    if planned and conv.in_channels >= 64 and x.indices.shape[0] >= 1024:
        assert flip_f & 2, "the table should carry its plan by now (FV2P_TAB_PLANNED)"
    return rb, tab_f, flip_f, tab_b, flip_b


@pytest.mark.parametrize("c,n,planned", [(16, 1500, False), (32, 1500, False), (64, 1500, False), (128, 1500, False), (64, 5000, False), (128, 70, False),
                                         (64, 1100, True), (128, 1500, True), (128, 2600, True), (64, 9000, True)])
def test_statistics_finalised_by_the_conv_launch(gpu, c, n, planned):
    """(planned: a small table with a tiling plan - 256 tiles for ~1 500 rows - once overran the row buffer of the finalisation.)
    conv_fin leaves mean / invstd of ITS output and the running statistics as BatchNorm1d would compute them: against float64 sums of the
    conv's own output at 1e-6, running statistics like torch's (momentum 0.01, unbiased variance), num_batches_tracked + 1; a second call
    (the slots were cleared by the last workgroup, the counter reset) gives the same statistics again, and so does a third on another layer
    size in between (one slot buffer per stream serves every layer)."""
    e = ext()
    batch, shape = 2, [9, 20, 18]
    ind, feats, x = make_input(c * 3 + n, batch, shape, n, c, gpu)
    conv = spconv.SubMConv3d(c, c, 3, padding=1, bias=True, indice_key="k").to(gpu)
    bn = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(gpu)
    rb, tab_f, flip_f, tab_b, flip_b = subm_tables(x, conv, planned)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()

    def run():
        return e.conv_fin(x.features, conv.weight, tab_f, flip_f, tab_b, flip_b, x.features.shape[0], rb.kvol // 2, None, None, 0, None, conv.bias, True,
                          bn.running_mean, bn.running_var, bn.num_batches_tracked, True, 0.01, 1e-3, None, None, None, False, False)
    y, saved = run()
    y64 = y.double()
    mean, var = y64.mean(0), y64.var(0, unbiased=False)
    assert rel(saved[0], mean) < 1e-6 and rel(saved[1], 1.0 / torch.sqrt(var + 1e-3)) < 1e-6
    nrows = y.shape[0]
    assert rel(bn.running_mean, 0.99 * rm0.double() + 0.01 * mean) < 1e-6
    assert rel(bn.running_var, 0.99 * rv0.double() + 0.01 * var * nrows / (nrows - 1)) < 1e-6
    assert int(bn.num_batches_tracked) == 1
    # conv output itself: oracle
    _, pairs, num = oracle.indice_pairs(ind, batch, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], subm=True)
    ref = oracle.indice_conv(feats, conv.weight.detach().cpu().numpy(), pairs, num, ind.shape[0], subm=True) + conv.bias.detach().cpu()
    assert rel(y, ref) < RTOL
    # another layer size on the same stream in between, then the first again: same statistics (fp64 slot sums of the same fp32 values)
    _, _, x2 = make_input(99, batch, shape, 400, 32, gpu)
    conv2 = spconv.SubMConv3d(32, 32, 3, padding=1, bias=False, indice_key="k2").to(gpu)
    bn2 = nn.BatchNorm1d(32).to(gpu)
    rb2, t2f, f2f, t2b, f2b = subm_tables(x2, conv2)
    y2, s2 = e.conv_fin(x2.features, conv2.weight, t2f, f2f, t2b, f2b, x2.features.shape[0], rb2.kvol // 2, None, None, 0, None, None, True,
                        bn2.running_mean, bn2.running_var, bn2.num_batches_tracked, True, 0.1, 1e-5, None, None, None, False, False)
    assert rel(s2[0], y2.double().mean(0)) < 1e-6
    y_again, saved_again = run()
    assert torch.equal(y, y_again)
    assert rel(saved_again, saved) < 1e-7
    assert int(bn.num_batches_tracked) == 2


@pytest.mark.parametrize("cin,cout,n", [(16, 16, 1500), (32, 32, 1500), (32, 16, 900), (64, 64, 1500), (128, 128, 1500), (64, 128, 1200), (128, 64, 70), (64, 64, 20000)])
@pytest.mark.parametrize("relu", [True, False])
def test_source_batchnorm_on_the_gather(gpu, cin, cout, n, relu):
    """conv over rows normalised (+ rectified) on the gather: bit-identical to the same conv over the materialised rows (same operations in
    the same order), 1e-4 against the oracle conv of torch's own float64 BatchNorm + ReLU; backward: input gradient (through the BatchNorm's
    batch statistics), gamma / beta gradients and the weight gradient against torch autograd of the float64 composition."""
    e = ext()
    batch, shape = 2, [9, 24, 22] if n < 10000 else [9, 60, 60]
    ind, feats, x = make_input(cin * 5 + cout + n, batch, shape, n, cin, gpu)
    if cin == 32:   # conv_rows_res has no such form (it spilled: 66.8 against 47.1 us): 32-channel consumers read materialised rows
        assert not e.prenorm_supported(cin, cout, 27, ind.shape[0], 1)
        return
    assert e.prenorm_supported(cin, cout, 27, ind.shape[0], 1), "this shape is one the residual blocks of the backbones use"
    conv = spconv.SubMConv3d(cin, cout, 3, padding=1, bias=False, indice_key="k").to(gpu)
    rb, tab_f, flip_f, tab_b, flip_b = subm_tables(x, conv)
    torch.manual_seed(cin + cout)
    gamma = (torch.rand(cin, device=gpu) + 0.5).requires_grad_(True)
    beta = (torch.randn(cin, device=gpu) * 0.3).requires_grad_(True)
    src = (x.features * 1.7 + 0.4).detach().requires_grad_(True)     # raw rows with a non-trivial mean / variance
    with torch.no_grad():
        s64 = src.double()
        saved = torch.stack([s64.mean(0), 1.0 / torch.sqrt(s64.var(0, unbiased=False) + 1e-3)]).float().contiguous()
    n_rows = src.shape[0]
    args = (tab_f, flip_f, tab_b, flip_b, n_rows, rb.kvol // 2, None, None, 0, None, None, False, None, None, None, True, 0.01, 1e-3)
    y, _ = e.conv_fin(src, conv.weight, *args, saved, gamma, beta, relu, True)
    mat = e.bn_apply(src.detach(), saved, gamma.detach(), beta.detach(), relu, None, True)
    y_mat, _ = e.conv_fin(mat, conv.weight, *args, None, None, None, False, False)
    assert torch.equal(y, y_mat), "normalising on the gather must give the bits of the conv over materialised rows"
    # float64 composition on the host, conv by the oracle (gather -> mm -> scatter on the oracle's rulebook)
    _, pairs, num = oracle.indice_pairs(ind, batch, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], subm=True)
    s_h = src.detach().cpu().double().requires_grad_(True)
    g_h, b_h = gamma.detach().cpu().double().requires_grad_(True), beta.detach().cpu().double().requires_grad_(True)
    a_h = torch.nn.functional.batch_norm(s_h, None, None, g_h, b_h, True, 0.0, 1e-3)
    a_h = torch.relu(a_h) if relu else a_h
    w = conv.weight.detach().cpu().numpy()
    ref = oracle.indice_conv(a_h.detach().float().numpy(), w, pairs, num, ind.shape[0], subm=True)
    assert rel(y, ref) < RTOL
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
    y.backward(g.to(gpu))
    da, dw = oracle.indice_conv_backward(a_h.detach().float().numpy(), w, g.numpy(), pairs, num, subm=True)
    a_h.backward(da.double())
    assert rel(conv.weight.grad, dw) < RTOL
    assert rel(src.grad, s_h.grad) < RTOL
    assert rel(gamma.grad, g_h.grad) < RTOL and rel(beta.grad, b_h.grad) < RTOL


def _block(c, gpu, seed):
    from fv2p_harness.backbone import _BasicBlock
    from functools import partial
    torch.manual_seed(seed)
    blk = _BasicBlock(c, partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01), "res").to(gpu)
    with torch.no_grad():
        for bn in (blk.bn1, blk.bn2):
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.2)
    return blk


@pytest.mark.parametrize("c,n", [(16, 2500), (32, 2500), (64, 2500), (128, 2500), (128, 9000)])
def test_residual_block_with_and_without_the_fold(gpu, c, n):
    """A SparseBasicBlock (spconv_backbone.py:32-68) through the round-6 arrangement (3 launches forward) and through the round-5 one
    (modules one by one): same output, input gradient and parameter gradients at 1e-5 of each other (same arithmetic per element; only
    the BatchNorm sums are taken in a different grouping), both 1e-4 from the float64 host composition with the oracle's convs; the
    conv biases - they feed a train-mode BatchNorm - get exact zeros where torch returns column sums of rounding noise (< 1e-5 of the
    weight gradient's scale); running statistics move alike; the folded run is bit-identical from call to call."""
    batch, shape = 2, [9, 30, 30]
    ind, feats, _ = make_input(c * 11 + n, batch, shape, n, c, gpu)
    blk = _block(c, gpu, c)
    state = {k: v.clone() for k, v in blk.state_dict().items()}

    def run(fold):
        spconv.set_bn_fold(fold)
        blk.load_state_dict(state)
        blk.zero_grad(set_to_none=True)
        blk.train()
        x = spconv.SparseConvTensor(torch.from_numpy(feats).to(gpu).requires_grad_(True), torch.from_numpy(ind).to(gpu), shape, batch)
        out = blk(x)
        gw = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(3)).to(gpu)
        out.features.backward(gw)
        grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in blk.named_parameters()}
        return out.features.detach().clone(), x.features.grad.clone(), grads, {k: v.clone() for k, v in blk.state_dict().items() if "running" in k or "tracked" in k}

    try:
        on, on2, off = run(True), run(True), run(False)
    finally:
        spconv.set_bn_fold(True)
    assert torch.equal(on[0], on2[0]) and torch.equal(on[1], on2[1]), "the folded block must be reproducible from call to call"
    assert rel(on[0], off[0]) < 1e-5 and rel(on[1], off[1]) < 1e-5
    wscale = float(off[2]["conv1.weight"].abs().max())
    for k in on[2]:
        if k.endswith("conv1.bias") or k.endswith("conv2.bias"):
            assert float(on[2][k].abs().max()) == 0.0 and float(off[2][k].abs().max()) < 1e-5 * wscale, k
        else:
            assert rel(on[2][k], off[2][k]) < 2e-5, k
    for k in on[3]:
        assert rel(on[3][k].double(), off[3][k].double()) < 1e-6, k
    # float64 composition on the host with the oracle's convs
    _, pairs, num = oracle.indice_pairs(ind, batch, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], subm=True)
    f = torch.from_numpy(feats).double().requires_grad_(True)
    p = {k: v.detach().cpu().double().requires_grad_(True) for k, v in state.items() if v.is_floating_point() and "running" not in k}

    class Conv(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a, w):
            ctx.save_for_backward(a, w)
            return oracle.indice_conv(a.detach().numpy(), w.detach().numpy(), pairs, num, ind.shape[0], subm=True).double()

        @staticmethod
        def backward(ctx, g):
            a, w = ctx.saved_tensors
            da, dw = oracle.indice_conv_backward(a.detach().numpy(), w.detach().numpy(), g.numpy(), pairs, num, subm=True)
            return da.double(), dw.double().view_as(w)
    bnf = lambda t, w, b: torch.nn.functional.batch_norm(t, None, None, w, b, True, 0.0, 1e-3)
    h = torch.relu(bnf(Conv.apply(f, p["conv1.weight"]) + p["conv1.bias"], p["bn1.weight"], p["bn1.bias"]))
    o = torch.relu(bnf(Conv.apply(h, p["conv2.weight"]) + p["conv2.bias"], p["bn2.weight"], p["bn2.bias"]) + f)
    o.backward(torch.randn(o.shape, generator=torch.Generator().manual_seed(3)).double())
    assert rel(on[0], o) < RTOL
    FLIPS = 4e-3   # one flipped element of |g| ~ 4 in a gradient of norm ~ 1e3 is 4e-3 by itself (tests/f64_calibration.py: FLIPS)
    assert rel_l2(on[1], f.grad) < FLIPS
    for k in ("conv1.weight", "conv2.weight", "bn1.weight", "bn1.bias", "bn2.weight", "bn2.bias"):
        assert rel_l2(on[2][k], p[k].grad) < FLIPS, k


def test_post_act_block_and_eval_mode(gpu):
    """conv -> BatchNorm1d -> ReLU inside a SparseSequential (post_act_block, spconv_backbone.py:8-27) on the new arrangement: training
    mode against the round-5 arrangement (1e-5) including the gradient that reaches the input through TWO such blocks (the second block's
    backward-data conv takes the first BatchNorm's backward sums and finalises them); eval mode (running statistics) equal too."""
    batch, shape, c = 2, [9, 30, 30], 32
    ind, feats, _ = make_input(5, batch, shape, 3000, 16, gpu)
    torch.manual_seed(0)
    net = spconv.SparseSequential(spconv.SubMConv3d(16, c, 3, padding=1, bias=False, indice_key="a"), nn.BatchNorm1d(c, eps=1e-3, momentum=0.01), nn.ReLU(),
                                  spconv.SparseConv3d(c, 64, 3, stride=2, padding=1, bias=False, indice_key="b"), nn.BatchNorm1d(64, eps=1e-3, momentum=0.01), nn.ReLU(),
                                  spconv.SubMConv3d(64, 64, 3, padding=1, bias=False, indice_key="c"), nn.BatchNorm1d(64, eps=1e-3, momentum=0.01), nn.ReLU()).to(gpu)
    state = {k: v.clone() for k, v in net.state_dict().items()}

    from pcdet.ops.spconv import conv as conv_mod

    def run(fold, train):
        spconv.set_bn_fold(fold)
        conv_mod.FOLD_SEQUENTIAL = fold      # (off by default: only the residual blocks use the arrangement, pcdet/ops/spconv/conv.py)
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        net.train(train)
        x = spconv.SparseConvTensor(torch.from_numpy(feats).to(gpu).requires_grad_(True), torch.from_numpy(ind).to(gpu), shape, batch)
        out = net(x)
        out.features.square().mean().backward()
        return out.features.detach().clone(), x.features.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()}
    try:
        for train in (True, False):
            on, off = run(True, train), run(False, train)
            assert rel(on[0], off[0]) < 1e-5 and rel(on[1], off[1]) < 2e-5, train
            for k in on[2]:
                assert rel(on[2][k], off[2][k]) < 2e-5, (train, k)
    finally:
        spconv.set_bn_fold(True)
        conv_mod.FOLD_SEQUENTIAL = False


@pytest.mark.parametrize("c,n", [(16, 3000), (64, 777), (128, 5000), (20, 100)])
def test_residual_tail_forward_backward(gpu, c, n):
    """bn_apply with a residual: out = relu(bn(x) + identity) in one launch and its backward (mask read from `out`, dz written beside dx)
    against torch autograd in float64 at 1e-5; also without the ReLU."""
    e = ext()
    torch.manual_seed(c + n)
    x = torch.randn(n, c, device=gpu) * 2 + 0.5
    res = torch.randn(n, c, device=gpu)
    gamma, beta = torch.rand(c, device=gpu) + 0.5, torch.randn(c, device=gpu) * 0.2
    x64 = x.double()
    saved = torch.stack([x64.mean(0), 1.0 / torch.sqrt(x64.var(0, unbiased=False) + 1e-3)]).float().contiguous()
    for relu in (True, False):
        xs, rs, gs, bs = (t.clone().requires_grad_(True) for t in (x, res, gamma, beta))
        out = e.bn_apply(xs, saved, gs, bs, relu, rs, True)
        g = torch.randn(n, c, device=gpu)
        out.backward(g)
        xh, rh, gh, bh = (t.detach().cpu().double().requires_grad_(True) for t in (x, res, gamma, beta))
        o = torch.nn.functional.batch_norm(xh, None, None, gh, bh, True, 0.0, 1e-3) + rh
        o = torch.relu(o) if relu else o
        o.backward(g.cpu().double())
        assert rel(out, o) < 1e-5
        for a, b in ((xs, xh), (rs, rh), (gs, gh), (bs, bh)):
            assert rel(a.grad, b.grad) < 1e-5, relu


@pytest.mark.parametrize("n,c", [(49152, 128), (49152, 64), (35000, 16), (777, 32), (2, 5), (100000, 20), (300, 256)])
@pytest.mark.parametrize("relu", [True, False])
def test_one_launch_batchnorm_passes(gpu, n, c, relu):
    """BatchNorm1d (+ReLU) whose sums no conv epilogue takes (the decoder's Linear -> BatchNorm1d -> ReLU rows): reduce, grid barrier and
    apply in ONE launch per direction (fv2p_batchnorm_forward_one / _backward_one) against the two-launch passes (1e-6 forward, 1e-5
    gradients: the same arithmetic per element, the sums taken over <= 128 instead of <= 64 partials) and against torch in float64
    (1e-5); running statistics alike; twenty back-to-back calls give the same bits (the barrier's counters are reset by the launch)."""
    e = ext()
    from pcdet.ops.spconv.norm import batch_norm_relu
    torch.manual_seed(n + c)
    x0 = torch.randn(n, c, device=gpu) * 1.5 + 0.3
    g = torch.randn(n, c, device=gpu)

    def run(one, wide=True):
        e.set_bn_one(one)
        e.set_bn_wide(wide)
        torch.manual_seed(1)
        bn = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(gpu)
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.2)
        x = x0.clone().requires_grad_(True)
        y = batch_norm_relu(bn, x, nn.ReLU() if relu else None)
        assert y is not None
        y.backward(g)
        return y.detach(), x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone(), bn.weight.detach().clone(), bn.bias.detach().clone()
    try:
        one, two = run(True), run(False, False)
        for _ in range(20):
            again = run(True)
        assert all(torch.equal(a, b) for a, b in zip(one, again))
        # the large-tensor form: reduce on up to 512 workgroups finalised by that launch + apply (what run(True) uses above ~2 M elements)
        wide = run(False, True)
        for _ in range(5):
            wide_again = run(False, True)
        assert all(torch.equal(a, b) for a, b in zip(wide, wide_again))
        assert rel(wide[0], two[0]) < 1e-6 and rel(wide[4], two[4]) < 1e-6 and rel(wide[5], two[5]) < 1e-6
        for a, b in zip(wide[1:4], two[1:4]):
            assert rel(a, b) < 1e-5
    finally:
        e.set_bn_one(True)
        e.set_bn_wide(True)
    assert rel(one[0], two[0]) < 1e-6 and rel(one[4], two[4]) < 1e-6 and rel(one[5], two[5]) < 1e-6
    for a, b in zip(one[1:4], two[1:4]):
        assert rel(a, b) < 1e-5
    xh = x0.cpu().double().requires_grad_(True)
    bnh = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).double()
    with torch.no_grad():
        bnh.weight.copy_(one[6].cpu().double())
        bnh.bias.copy_(one[7].cpu().double())
    yh = bnh(xh)
    yh = torch.relu(yh) if relu else yh
    yh.backward(g.cpu().double())
    assert rel(one[0], yh) < 1e-5 and rel(one[1], xh.grad) < 1e-5 and rel(one[2], bnh.weight.grad) < 1e-5 and rel(one[3], bnh.bias.grad) < 1e-5
