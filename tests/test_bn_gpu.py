"""GPU parity of the fused BatchNorm1d(+ReLU) op through the C ABI (pcdet.ops.spconv.norm).

Checked against (1) torch.nn.BatchNorm1d + nn.ReLU themselves on the same GPU tensors — the modules the reference
instantiates (spconv_backbone.py:8-27) — and (2) the float64 restatement oracle/bn_oracle.py.
Tolerance: 1e-4 relative for features and gradients (north_star's float tolerance), 1e-5 for the running statistics."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

import pcdet.ops.spconv as spconv
from oracle import bn_oracle
from pcdet.ops.spconv import norm
from sparse_util import random_active

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def make(n, c, seed, gpu):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((n, c)) * rng.uniform(0.2, 3.0, c) + rng.uniform(-4, 4, c)).astype(np.float32)
    dy = rng.standard_normal((n, c)).astype(np.float32)
    bn = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)))
        bn.bias.copy_(torch.from_numpy(rng.uniform(-0.5, 0.5, c).astype(np.float32)))
        bn.running_mean.copy_(torch.from_numpy(rng.uniform(-1, 1, c).astype(np.float32)))
        bn.running_var.copy_(torch.from_numpy(rng.uniform(0.5, 2, c).astype(np.float32)))
    return x, dy, bn.to(gpu)


@pytest.mark.parametrize("n,c", [(2, 16), (1000, 16), (50783, 32), (29446, 64), (4097, 128), (777, 40), (513, 7), (300, 300), (64, 1024)])
@pytest.mark.parametrize("relu", [True, False])
def test_training_matches_torch_and_oracle(gpu, front_end, n, c, relu):
    x, dy, bn = make(n, c, n + c, gpu)
    ref_bn = copy.deepcopy(bn)
    xt = torch.from_numpy(x).to(gpu).requires_grad_(True)
    xr = torch.from_numpy(x).to(gpu).requires_grad_(True)
    g = torch.from_numpy(dy).to(gpu)
    act = nn.ReLU() if relu else None
    y = norm.batch_norm_relu(bn, xt, act)
    assert y is not None
    yr = ref_bn(xr)
    if relu:
        yr = torch.relu(yr)
    y.backward(g)
    yr.backward(g)
    # (1) torch's own modules
    assert rel(y.detach().cpu(), yr.detach().cpu()) < RTOL
    assert rel(xt.grad.cpu(), xr.grad.cpu()) < 5 * RTOL   # torch's own fp32 reductions are the looser side here
    assert rel(bn.weight.grad.cpu(), ref_bn.weight.grad.cpu()) < 5 * RTOL
    assert rel(bn.bias.grad.cpu(), ref_bn.bias.grad.cpu()) < 5 * RTOL
    assert rel(bn.running_mean.cpu(), ref_bn.running_mean.cpu()) < 1e-5
    assert rel(bn.running_var.cpu(), ref_bn.running_var.cpu()) < 1e-5
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1
    # (2) float64 oracle
    w, b = ref_bn.weight.detach().cpu().numpy().astype(np.float64), ref_bn.bias.detach().cpu().numpy().astype(np.float64)
    _, _, bn0 = make(n, c, n + c, "cpu")
    yo, saved, rm, rv, _ = bn_oracle.bn_relu_forward(x, w, b, bn0.running_mean.numpy().astype(np.float64), bn0.running_var.numpy().astype(np.float64),
                                                     0, True, 0.01, 1e-3, relu)
    dxo, dgo, dbo = bn_oracle.bn_relu_backward(dy, yo, saved, w, relu, True)
    assert rel(y.detach().cpu(), yo) < RTOL
    assert rel(xt.grad.cpu(), dxo) < RTOL
    assert rel(bn.weight.grad.cpu(), dgo) < RTOL and rel(bn.bias.grad.cpu(), dbo) < RTOL
    assert rel(bn.running_mean.cpu(), rm) < 1e-5 and rel(bn.running_var.cpu(), rv) < 1e-5


def test_eval_mode_cumulative_momentum_and_determinism(gpu, front_end):
    x, dy, bn = make(3000, 64, 5, gpu)
    ref = copy.deepcopy(bn)
    bn.eval(), ref.eval()
    xt = torch.from_numpy(x).to(gpu).requires_grad_(True)
    xr = torch.from_numpy(x).to(gpu).requires_grad_(True)
    y = norm.batch_norm_relu(bn, xt, nn.ReLU())
    yr = torch.relu(ref(xr))
    g = torch.from_numpy(dy).to(gpu)
    y.backward(g), yr.backward(g)
    assert rel(y.detach().cpu(), yr.detach().cpu()) < RTOL and rel(xt.grad.cpu(), xr.grad.cpu()) < RTOL
    assert rel(bn.weight.grad.cpu(), ref.weight.grad.cpu()) < 5 * RTOL
    assert torch.equal(bn.running_mean, ref.running_mean) and int(bn.num_batches_tracked) == 0
    # momentum=None: cumulative moving average over three batches
    bn2 = nn.BatchNorm1d(32, eps=1e-3, momentum=None).to(gpu)
    ref2 = copy.deepcopy(bn2)
    for s in range(3):
        xs = torch.from_numpy(np.random.default_rng(s).standard_normal((500, 32)).astype(np.float32) * (s + 1)).to(gpu)
        norm.batch_norm_relu(bn2, xs, None)
        ref2(xs)
    assert int(bn2.num_batches_tracked) == 3
    assert rel(bn2.running_mean.cpu(), ref2.running_mean.cpu()) < 1e-5 and rel(bn2.running_var.cpu(), ref2.running_var.cpu()) < 1e-5
    # fixed-order fp64 folding: bit-identical across repeats
    x3 = torch.from_numpy(x).to(gpu)
    bn.train()
    outs = [norm.batch_norm_relu(copy.deepcopy(bn), x3, nn.ReLU()) for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


def test_sparse_sequential_uses_fused_pair_and_falls_back(gpu, front_end, monkeypatch):
    batch, shape = 2, [9, 20, 18]
    ind = random_active(3, batch, shape, 900)
    feats = np.random.default_rng(4).standard_normal((ind.shape[0], 16)).astype(np.float32)

    def build():
        torch.manual_seed(0)
        return spconv.SparseSequential(spconv.SubMConv3d(16, 32, 3, padding=1, bias=False, indice_key="s"),
                                       nn.BatchNorm1d(32, eps=1e-3, momentum=0.01), nn.ReLU()).to(gpu)

    def run(net):
        x = spconv.SparseConvTensor(torch.from_numpy(feats).to(gpu), torch.from_numpy(ind).to(gpu), shape, batch)
        out = net(x).features
        out.square().sum().backward()
        return out.detach().cpu().numpy(), [p.grad.cpu().numpy() for p in net.parameters()], net[1].running_var.cpu().numpy()

    calls = []
    from pcdet.ops.spconv import modules
    from pcdet.ops.spconv.conv import SparseConvolution
    orig = modules.batch_norm_relu
    orig_block = SparseConvolution._conv_bn_relu

    def counting(bn, x, relu_module=None):          # BN(+ReLU) fused on its own (ctypes front end)
        out = orig(bn, x, relu_module)
        calls.append(out is not None and relu_module is not None)
        return out

    def counting_block(self, features, rb, n_out, post, weight=None):   # conv -> BN -> ReLU in one call (compiled front end)
        out = orig_block(self, features, rb, n_out, post, weight)
        if out is not None:
            calls.append(post[1] is not None)
        return out

    from pcdet.ops.spconv import conv as conv_mod
    orig_fold = conv_mod.conv_bn_fold

    def counting_fold(conv, x, bn, relu_module, defer=False, residual=None):   # round 6: conv (statistics finalised by its launch) + one apply launch
        out = orig_fold(conv, x, bn, relu_module, defer, residual)
        if out is not None:
            calls.append(relu_module is not None)
        return out

    monkeypatch.setattr(modules, "batch_norm_relu", counting)
    monkeypatch.setattr(SparseConvolution, "_conv_bn_relu", counting_block)
    monkeypatch.setattr(conv_mod, "conv_bn_fold", counting_fold)
    y1, g1, rv1 = run(build())
    assert calls == [True]                       # the BN+ReLU pair went through the fused op
    net = build()
    net[2].register_forward_hook(lambda m, i, o: None)   # a hook on the ReLU: stay on the module-by-module path
    y2, g2, rv2 = run(net)
    assert calls == [True, False]
    assert rel(y1, y2) < RTOL and rel(rv1, rv2) < 1e-5
    for a, b in zip(g1, g2):
        assert rel(a, b) < 5 * RTOL
