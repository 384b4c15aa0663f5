"""What the BatchNorm statistics cost a conv launch: plain conv / sums by fp64 atomics (fire and forget, rounds 2-5) / rows + folds by the
launch's last workgroups (round 6), on a submanifold layer shaped like the KITTI residual backbone's levels."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "from-voxel-to-point_amd"), os.path.join(REPO, "tests")]
import fv2p_native as nat
from pcdet.ops.spconv import ops
import pcdet.ops.spconv as spconv
from sparse_util import random_active
dev = torch.device("cuda")
lib = nat.lib()
for c, n, shape in [(128, 9919, [5, 100, 88]), (64, 22331, [11, 200, 176]), (32, 39000, [21, 400, 352]), (16, 35000, [41, 800, 704])]:
    ind = random_active(c, 3, shape, n)
    x = torch.randn(ind.shape[0], c, device=dev)
    idx = torch.from_numpy(ind).to(dev)
    rb = ops.build_rulebook(idx, 3, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], [0, 0, 0], True, False)
    for _ in range(3):
        tab, flip = rb.out_table(c)
    w = torch.randn(27, c, c, device=dev) * 0.05
    y = torch.empty(ind.shape[0], c, device=dev)
    slots = torch.zeros(lib.fv2p_sparse_conv_stat_slots() * 2 * c, dtype=torch.float64, device=dev)
    ws = torch.empty(lib.fv2p_sparse_conv_fin_ws_bytes(ind.shape[0], c), dtype=torch.uint8, device=dev)
    counter = torch.zeros(lib.fv2p_sparse_conv_fin_counter_words(), dtype=torch.int32, device=dev)
    mean, invstd = torch.empty(c, device=dev), torch.empty(c, device=dev)
    st = nat.stream()
    nrows = ind.shape[0]
    def plain(): nat.call("fv2p_sparse_conv_rows", x, nrows, c, w, 27, tab, nrows, c, flip, 0, None, y, st)
    def atom(): nat.call("fv2p_sparse_conv_rows_stats", x, nrows, c, w, 27, tab, nrows, c, flip, 0, None, y, slots, st)
    def fin(): nat.call("fv2p_sparse_conv_rows_bnfin", x, nrows, c, w, 27, tab, nrows, c, flip, 0, None, y, ws, counter, 1e-3, 0.01, None, None, None, mean, invstd, None, None, None, None, 0, st)
    def pre(): nat.call("fv2p_sparse_conv_rows_bnfin", x, nrows, c, w, 27, tab, nrows, c, flip, 0, None, y, ws, counter, 1e-3, 0.01, None, None, None, mean, invstd, mean, invstd, None, None, 1, st)
    line = f"{c:3d} ch {nrows:6d} rows {int(rb.indice_pair_num.sum()):8d} pairs:"
    for name, fn in (("plain", plain), ("atomics", atom), ("rows+fin", fin)) + ((("rows+fin+pre", pre),) if lib.fv2p_sparse_conv_prenorm_supported(c, c, 27, nrows, flip, 0) else ()):
        for _ in range(50): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); torch.cuda.synchronize()
        line += f"  {name} {e0.elapsed_time(e1) / 200 * 1e3:6.1f} us"
    print(line)
