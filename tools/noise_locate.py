"""Where along the reduced MGAF step does the HIP run depart from the host float64 run more than the host float32 run does?
Forward: every module's output; backward: every module's output gradient.  Relative L2 distance to the float64 run, HIP and host32.
    python tools/noise_locate.py [weights_seed] [cloud_seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "from-voxel-to-point_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from conftest import deterministic_libraries  # noqa: E402
from oracle.backend import oracle_backend  # noqa: E402
from oracle.spconv_cpu import cpu_mirror  # noqa: E402
from fv2p_harness import mgaf_model as mm  # noqa: E402
from test_mgaf_head import SmallMGAF, small_inputs  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cloud = int(sys.argv[2]) if len(sys.argv) > 2 else 90 + 10 * seed
gpu = torch.device("cuda:0")


def tens(o):
    if isinstance(o, torch.Tensor):
        return o
    if hasattr(o, "features"):
        return o.features
    if isinstance(o, (tuple, list)) and o:
        return tens(o[0])
    if isinstance(o, dict) and o:
        return tens(next(iter(o.values())))
    return None


def instrument(net):
    fwd, bwd, order = {}, {}, []
    for name, m in net.named_modules():
        if not name:
            continue

        def hook(mod, inp, out, name=name):
            t = tens(out)
            if t is None or not t.is_floating_point():
                return
            if name not in fwd:
                order.append(name)
            fwd[name] = t.detach().double().cpu()
            if t.requires_grad:
                t.register_hook(lambda g, name=name: bwd.__setitem__(name, g.detach().double().cpu()))
        m.register_forward_hook(hook)
    return fwd, bwd, order


torch.manual_seed(seed)
model = mm.MGAFDetector(SmallMGAF)
feats, coords, gt = small_inputs(cloud)
ref, ref64 = cpu_mirror(model), cpu_mirror(model).double()
rec = {}
with oracle_backend():
    for tag, net, cast in (("h32", ref, lambda t: t), ("h64", ref64, lambda t: t.double())):
        rec[tag] = instrument(net)
        net.taps = {}
        net(cast(feats), coords, 2, cast(gt)).backward()
net = model.to(gpu)
rec["hip"] = instrument(net)
net.taps = {}
with deterministic_libraries():
    net(feats.to(gpu), coords.to(gpu), 2, gt.to(gpu)).backward()
torch.cuda.synchronize()
d = lambda a, t: float((a - t).norm() / t.norm().clamp_min(1e-300))
print(f"seed {seed} clouds {cloud}+   columns: forward hip, host32, ratio | backward (gradient of the module output) hip, host32, ratio")
for name in rec["h64"][2]:
    f64, b64 = rec["h64"][0][name], rec["h64"][1].get(name)
    if name not in rec["hip"][0] or name not in rec["h32"][0] or rec["hip"][0][name].shape != f64.shape:
        continue
    fh, f3 = d(rec["hip"][0][name], f64), d(rec["h32"][0][name], f64)
    line = f"{name:58s} {fh:9.2e} {f3:9.2e} {fh / max(f3, 1e-300):6.2f}"
    if b64 is not None and name in rec["hip"][1] and name in rec["h32"][1] and float(b64.norm()) > 0:
        bh, b3 = d(rec["hip"][1][name], b64), d(rec["h32"][1][name], b64)
        line += f" | {bh:9.2e} {b3:9.2e} {bh / max(b3, 1e-300):6.2f}"
    print(line)
