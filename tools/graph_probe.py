"""Does hipGraph capture of a MIOpen conv + BatchNorm + ReLU trunk (forward and backward) work on this stack, and what does a replay
cost on the host?  python tools/graph_probe.py"""
import os
import sys
import time

import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "from-voxel-to-point_amd"))
from fv2p_harness.fv2p_model import BEVBackbone, FV2PConfig  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = BEVBackbone(FV2PConfig, 256).to(dev)
x = torch.randn(3, 256, 200, 176, device=dev, requires_grad=True)


def run(f, inp):
    y = f(inp)
    g, = torch.autograd.grad(y.sum(), inp, retain_graph=False, allow_unused=True)
    return y, g


for _ in range(3):   # MIOpen's solver search happens here, in line
    y0, g0 = run(net, x)
    for p in net.parameters():
        p.grad = None
torch.cuda.synchronize()


def host_ms(f, n=10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c0 = time.thread_time()
    for _ in range(n):
        y = f(x)
        y.sum().backward()
    c1 = time.thread_time()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, t_issue / n * 1e3, (c1 - c0) / n * 1e3


print("eager   : %.2f ms per fwd+bwd, host issue %.2f ms, main-thread cpu %.2f ms" % host_ms(net))
try:
    g = torch.cuda.make_graphed_callables(net, (x.detach().clone().requires_grad_(True),))
    y1 = g(x)
    print("graphed output equals eager:", float((y1 - net(x)).abs().max()))
    print("graphed : %.2f ms per fwd+bwd, host issue %.2f ms, main-thread cpu %.2f ms" % host_ms(g))
except Exception as e:   # noqa: BLE001
    print("graph capture failed:", type(e).__name__, str(e)[:500])
