"""Generates tests/golden/bev_interp_*.npz with the reference's own `bilinear_interpolate_torch`
(/root/reference/pcdet/models/backbones_3d/pfe/bev_grid_pooling.py:11-45; build container only).

The reference module cannot be imported as a whole — it imports the CUDA extensions of pcdet.ops at module load — so
this script parses the file, takes the one function definition out of the syntax tree and compiles it against torch
(CPU).  Nothing of the reference is copied into the repository: the fixtures hold inputs and outputs only.  Run:
    python oracle/gen_golden_bev.py
"""
import ast
import os

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/pcdet/models/backbones_3d/pfe/bev_grid_pooling.py"


def load_reference_function():
    tree = ast.parse(open(REF).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "bilinear_interpolate_torch"]
    assert len(fn) == 1
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fn, type_ignores=[]), REF, "exec"), ns)
    return ns["bilinear_interpolate_torch"]


def main():
    ref = load_reference_function()
    out_dir = os.path.join(REPO, "tests", "golden")
    cases = {"small": (7, 9, 5, 300, 1), "odd_channels": (13, 6, 7, 257, 2), "bev128": (20, 18, 128, 600, 3)}
    for name, (h, w, c, n, seed) in cases.items():
        rng = np.random.default_rng(seed)
        im = rng.standard_normal((h, w, c)).astype(np.float32)
        # inside the map, on integer positions, on and beyond every border (clamped corners change the weights there)
        x = rng.uniform(-2.5, w + 1.5, n).astype(np.float32)
        y = rng.uniform(-2.5, h + 1.5, n).astype(np.float32)
        x[:20] = np.round(x[:20])
        y[10:30] = np.round(y[10:30])
        x[30:34] = [0.0, w - 1.0, -0.0, float(w)]
        y[30:34] = [float(h), h - 1.0, 0.0, -1.0]
        out = ref(torch.from_numpy(im), torch.from_numpy(x), torch.from_numpy(y)).numpy()
        np.savez_compressed(os.path.join(out_dir, f"bev_interp_{name}.npz"), im=im, x=x, y=y, out=out)
        print(name, out.shape, float(np.abs(out).max()))


if __name__ == "__main__":
    main()
