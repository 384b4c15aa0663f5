"""TEST INFRASTRUCTURE ONLY (oracle/): float64 restatement of torch.nn.BatchNorm1d (+ReLU) as the reference backbones use it
— `norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)` followed by nn.ReLU in every block
(pcdet/models/backbones_3d/spconv_backbone.py:8-27, :75).  Pinned: tests/test_bn_gpu.py checks this restatement against
torch's own BatchNorm1d (the actual reference implementation, importable here) on the same inputs."""
import numpy as np


def bn_relu_forward(x, gamma, beta, running_mean, running_var, num_batches_tracked, training, momentum, eps, relu):
    x = np.asarray(x, np.float64)
    n = x.shape[0]
    if training or running_mean is None:
        mean, var = x.mean(0), x.var(0)
        if training and running_mean is not None:
            num_batches_tracked += 1
            f = 1.0 / num_batches_tracked if momentum is None else momentum
            running_mean = (1 - f) * running_mean + f * mean
            running_var = (1 - f) * running_var + f * var * n / (n - 1)
    else:
        mean, var = running_mean, running_var
    invstd = 1.0 / np.sqrt(var + eps)
    xhat = (x - mean) * invstd
    y = xhat * gamma + beta
    if relu:
        y = np.maximum(y, 0.0)
    return y, (mean, invstd, xhat), running_mean, running_var, num_batches_tracked


def bn_relu_backward(dy, y, saved, gamma, relu, batch_stats):
    mean, invstd, xhat = saved
    dz = np.asarray(dy, np.float64) * ((y > 0) if relu else 1.0)
    dbeta, dgamma = dz.sum(0), (dz * xhat).sum(0)
    if batch_stats:
        dx = gamma * invstd * (dz - dz.mean(0) - xhat * (dz * xhat).mean(0))
    else:
        dx = gamma * invstd * dz
    return dx, dgamma, dbeta
