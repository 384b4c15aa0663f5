"""CPU: pins oracle/bn_oracle.py to torch.nn.BatchNorm1d + ReLU (the modules the reference instantiates,
spconv_backbone.py:8-27) — forward, running statistics and all three gradients."""
import numpy as np
import torch
from torch import nn

from oracle import bn_oracle


def test_bn_oracle_equals_torch_batchnorm1d_relu():
    rng = np.random.default_rng(0)
    for n, c, relu, momentum in [(257, 16, True, 0.01), (64, 7, False, 0.1), (1000, 64, True, None)]:
        x = (rng.standard_normal((n, c)) * 2 + 1).astype(np.float64)
        dy = rng.standard_normal((n, c))
        bn = nn.BatchNorm1d(c, eps=1e-3, momentum=momentum).double()
        with torch.no_grad():
            bn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, c)))
            bn.bias.copy_(torch.from_numpy(rng.uniform(-0.5, 0.5, c)))
        w, b = bn.weight.detach().numpy().copy(), bn.bias.detach().numpy().copy()
        rm0, rv0 = bn.running_mean.numpy().copy(), bn.running_var.numpy().copy()
        xt = torch.from_numpy(x).requires_grad_(True)
        y = bn(xt)
        if relu:
            y = torch.relu(y)
        y.backward(torch.from_numpy(dy))
        yo, saved, rm, rv, nbt = bn_oracle.bn_relu_forward(x, w, b, rm0, rv0, 0, True, momentum, 1e-3, relu)
        dx, dg, db = bn_oracle.bn_relu_backward(dy, yo, saved, w, relu, True)
        assert np.allclose(yo, y.detach().numpy(), rtol=1e-10, atol=1e-12)
        assert np.allclose(rm, bn.running_mean.numpy(), rtol=1e-10) and np.allclose(rv, bn.running_var.numpy(), rtol=1e-10)
        assert nbt == int(bn.num_batches_tracked) == 1
        assert np.allclose(dx, xt.grad.numpy(), rtol=1e-8, atol=1e-11)
        assert np.allclose(dg, bn.weight.grad.numpy(), rtol=1e-8) and np.allclose(db, bn.bias.grad.numpy(), rtol=1e-8)
        # eval mode
        bn.eval()
        xe = torch.from_numpy(x).requires_grad_(True)
        ye = torch.relu(bn(xe)) if relu else bn(xe)
        ye.backward(torch.from_numpy(dy))
        yo2, saved2, *_ = bn_oracle.bn_relu_forward(x, w, b, bn.running_mean.numpy(), bn.running_var.numpy(), 1, False, momentum, 1e-3, relu)
        dx2, _, _ = bn_oracle.bn_relu_backward(dy, yo2, saved2, w, relu, False)
        assert np.allclose(yo2, ye.detach().numpy(), rtol=1e-10, atol=1e-12) and np.allclose(dx2, xe.grad.numpy(), rtol=1e-8, atol=1e-11)
