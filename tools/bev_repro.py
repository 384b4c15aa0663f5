"""Which module makes two identical forward passes of the FV2P step differ?  (DESIGN 8.7 / tests/arrangement_check.py: the BEV map was
seen to differ by ~1e-5 relative between identical passes in some call sequences.)  Hooks every leaf module, repeats the same step under
the stream arrangements of the check and prints, per pair of runs, the first modules in execution order whose outputs differ bitwise.
python tools/bev_repro.py [runs per arrangement]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "from-voxel-to-point_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from fv2p_harness.fv2p_model import FV2PDetector  # noqa: E402
from test_fv2p_step_gpu import SmallFV2P, make_inputs  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if len(sys.argv) > 2 and sys.argv[2] == "deterministic":
    # MIOpen: deterministic solvers only; rocBLAS: no atomics (split-K GEMMs); torch ops with a deterministic variant take it
    torch.backends.cudnn.deterministic = True
    torch.use_deterministic_algorithms(True, warn_only=True)
    print("deterministic library settings ON")
gpu = torch.device("cuda:0")
torch.manual_seed(3)
model = FV2PDetector(SmallFV2P).to(gpu)
clouds, feats, coords, gt, u = make_inputs(SmallFV2P, 2, 4096)
args = ([c.to(gpu) for c in clouds], feats.to(gpu), coords.to(gpu), gt.to(gpu), u.to(gpu))
model.cfg = type("Cfg", (SmallFV2P,), {"dense_branch_stream": False, "point_branch_stream": False})
model(*args).backward()
torch.cuda.synchronize()

log = []


def tensor_of(out):
    if torch.is_tensor(out):
        return out
    if hasattr(out, "features"):
        return out.features
    if isinstance(out, (tuple, list)) and out and torch.is_tensor(out[0]):
        return out[0]
    return None


def hook(name):
    def f(mod, inp, out):
        t = tensor_of(out)
        if t is not None and t.is_floating_point():
            torch.cuda.synchronize()
            log.append((name, t.detach().clone()))
    return f


for name, mod in model.named_modules():
    if not list(mod.children()):
        mod.register_forward_hook(hook(name))

runs = []
for dense, point in ((False, False), (True, True)):
    for r in range(reps):
        model.cfg = type("Cfg", (SmallFV2P,), {"dense_branch_stream": dense, "point_branch_stream": point})
        model.taps = {}
        model.zero_grad(set_to_none=True)
        torch.manual_seed(11)
        log.clear()
        loss = model(*args)
        fw = list(log)
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        runs.append((f"{'dense+point streams' if dense else 'one stream'} #{r}", fw, grads, float(loss)))

base = runs[0]
for name, fw, grads, loss in runs[1:]:
    diffs = []
    for (n0, t0), (n1, t1) in zip(base[1], fw):
        assert n0 == n1, (n0, n1)
        if t0.shape != t1.shape or not torch.equal(t0, t1):
            d = float((t0 - t1).abs().max()) if t0.shape == t1.shape else float("nan")
            diffs.append((n0, d, float(t0.abs().max())))
    gd = [(k, float((g - base[2][k]).norm() / base[2][k].norm().clamp_min(1e-30))) for k, g in grads.items() if not torch.equal(g, base[2][k])]
    gd = [t for t in gd if not (t[0].startswith("backbone_3d.") and t[0].endswith((".conv1.bias", ".conv2.bias")))]   # zero gradients: noise over noise
    gd.sort(key=lambda t: -t[1])
    print(f"== {name} vs {base[0]}: loss {loss!r} vs {base[3]!r}; {len(diffs)} of {len(fw)} module outputs differ; {len(gd)} of {len(grads)} gradients differ")
    for n, d, m in diffs[:40]:
        print(f"     forward  {n}: max abs diff {d:.3e} (max |value| {m:.3e})")
    for k, e in gd[:12]:
        print(f"     gradient {k}: rel L2 {e:.3e}")
