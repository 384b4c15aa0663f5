#!/usr/bin/env python3
"""Per-kernel timings on one GPU (events on the launch stream). Not part of the product or of bench.py; used to
iterate on kernel variants:  python tools/microbench.py [conv|dcn|fps|nms|all]"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "from-voxel-to-point_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def conv(only=None):
    from fv2p_harness import synth
    from fv2p_harness.backbone import VoxelBackBone8x, mean_vfe
    from pcdet.datasets.processor.voxel_generator import points_to_voxel_gpu
    from pcdet.ops.spconv import ops
    from pcdet.ops.spconv.conv import SparseConvolution
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    waymo = bool(os.environ.get("FV2P_WAYMO"))   # BASELINE configs[4]: ~180 k points per cloud, 0.1 m voxels, [41, 1504, 1504] grid
    res = bool(os.environ.get("FV2P_RES"))       # BASELINE configs[2]: the FV2P step's VoxelResBackBone8x at batch 3
    if res:
        from fv2p_harness.backbone import VoxelResBackBone8x
    nbatch = 3 if res else 4
    model = (VoxelResBackBone8x if res else VoxelBackBone8x)(4, [1504, 1504, 40] if waymo else [1408, 1600, 40]).to(dev)
    feats, coords = [], []
    for b in range(nbatch):
        if waymo:
            v, c, n = points_to_voxel_gpu(torch.from_numpy(synth.waymo_like_cloud(b, 180000)).to(dev), synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, True, 150000)
        else:
            v, c, n = points_to_voxel_gpu(torch.from_numpy(synth.lidar_cloud(b, 16384)).to(dev), synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 16000)
        feats.append(mean_vfe(v, n))
        coords.append(torch.nn.functional.pad(c, (1, 0), value=b))
    recs = []
    bench_seeds = True

    def hook(mod, inp, out):
        x = inp[0]
        rb = x.indice_dict[mod.indice_key]
        if bench_seeds and mod.indice_key is not None:
            recs.append((mod, x.features.detach(), rb, out.features.shape[0]))
    hs = [m.register_forward_hook(hook) for m in model.modules() if isinstance(m, SparseConvolution)]
    with torch.no_grad():
        model(torch.cat(feats), torch.cat(coords), nbatch)
    for h in hs:
        h.remove()
    if only is not None:  # just the roofline kernel (bench.py's probe): forward conv of the layer with most flops
        mod, f, rb, n_out = max(recs, key=lambda r: int(r[2].indice_pair_num.sum().item()) * r[0].in_channels * r[0].out_channels)
        w = mod.weight.detach()
        t = timeit(lambda: ops.indice_conv(f, w, rb, None, n_out, False, mod.subm), reps=only, warm=3)
        print(f"roofline kernel: {'subm' if mod.subm else 'conv'} {mod.in_channels}->{mod.out_channels} key={mod.indice_key} "
              f"n={f.shape[0]} pairs={int(rb.indice_pair_num.sum().item())}: {t:.1f} us")
        if os.environ.get("FV2P_SORT_EXPERIMENT"):
            # what would tiles sorted by neighbour pattern buy?  Renumber the rows (sorted by the 27-bit offset mask, by its
            # 9-bit (dz,dy)-line summary, or shuffled) and rebuild the rulebook: tiles then follow that order.
            tabn = rb.tab_in.cpu().numpy() >= 0
            K = tabn.shape[0]
            mask = np.zeros(tabn.shape[1], np.int64)
            for k in range(K):
                mask |= tabn[k].astype(np.int64) << k
            line = np.zeros_like(mask)
            for q in range(K // 3):
                line |= tabn[3 * q:3 * q + 3].any(0).astype(np.int64) << q
            pc = np.array([bin(v).count("1") for v in mask])
            orders = {"as is": np.arange(mask.size), "mask sort": np.argsort(mask, kind="stable"), "line9 sort": np.argsort(line, kind="stable"),
                      "popcount sort": np.argsort(pc, kind="stable"), "shuffle": np.random.default_rng(0).permutation(mask.size)}
            ind = rb.indices
            inp = ind.cpu().numpy().astype(np.int64)            # (b, z, y, x): what would XCD-major tiles that are y bands buy?
            orders["(y, b, z, x) sort"] = np.lexsort((inp[:, 3], inp[:, 1], inp[:, 0], inp[:, 2]))
            orders["(b, y, z, x) sort"] = np.lexsort((inp[:, 3], inp[:, 1], inp[:, 2], inp[:, 0]))
            # balanced variant: rows sorted by (popcount, mask); 64-row tiles handed out so that the two workgroups a CU
            # receives (taken from a placement trace of this very launch shape) are one heavy and one light tile
            import fv2p_native
            nblk = (n_out + 63) // 64
            tr = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
            fv2p_native.call("fv2p_sparse_conv_set_trace", tr)
            ops.indice_conv(f, w, rb, None, n_out, False, mod.subm)
            torch.cuda.synchronize()
            fv2p_native.call("fv2p_sparse_conv_set_trace", None)
            trn = tr.cpu().numpy().reshape(nblk, 8)
            cu_key = (trn[:, 1] & 0xF) * 4096 + ((trn[:, 0] >> 8) & 0xFFF)
            order_w = np.argsort(pc * (1 << 27) + mask, kind="stable")           # light -> heavy rows
            tiles_sorted = [order_w[i * 64:(i + 1) * 64] for i in range(nblk)]   # tile i: i-th lightest
            by_cu = {}
            for b in range(nblk):
                by_cu.setdefault(int(cu_key[b]), []).append(b)
            singles = [v[0] for v in by_cu.values() if len(v) == 1]
            pairs_b = [v for v in by_cu.values() if len(v) >= 2]
            assign = {}
            lo, hi = 0, nblk - 1
            for v in pairs_b:                      # heavy + light on the same CU
                assign[v[0]] = hi; hi -= 1
                assign[v[1]] = lo; lo += 1
                for extra in v[2:]:
                    assign[extra] = lo; lo += 1
            for b in singles:                      # CUs with one workgroup take the middle
                assign[b] = lo; lo += 1
            bb = np.arange(nblk); xx = bb & 7
            tile_of_block = xx * (nblk >> 3) + np.minimum(xx, nblk & 7) + (bb >> 3)
            row_order = np.zeros(nblk * 64, np.int64) - 1
            for b in range(nblk):
                rows = tiles_sorted[assign[b]]
                row_order[tile_of_block[b] * 64: tile_of_block[b] * 64 + len(rows)] = rows
            row_order = row_order[row_order >= 0]
            if row_order.size == mask.size and len(set(row_order.tolist())) == mask.size:
                orders["(popcount,mask) sort + heavy/light per CU"] = row_order
            orders["(popcount,mask) sort"] = order_w
            for name, perm in orders.items():
                pt = torch.from_numpy(perm).to(ind.device)
                ind_p, f_p = ind[pt].contiguous(), f[pt].contiguous()
                rb_p = ops.build_rulebook(ind_p, 4, rb.spatial_shape, mod.kernel_size, mod.stride, mod.padding, mod.dilation, 0, True)
                act = (rb_p.tab_in.cpu().numpy() >= 0)
                padn = (-act.shape[1]) % 16
                tiles = np.concatenate([act, np.zeros((K, padn), bool)], 1).reshape(K, -1, 16).any(2).sum(0)
                tt = timeit(lambda: ops.indice_conv(f_p, w, rb_p, None, n_out, False, True), reps=only, warm=3)
                print(f"  rows {name:44s}: {tt:6.1f} us   active offsets per 16-row tile {tiles.mean():5.2f}")
        if os.environ.get("FV2P_TRACE"):
            import collections
            import fv2p_native
            nblk = (n_out + 63) // 64
            tr = torch.zeros(4 * nblk * 8, dtype=torch.int64, device="cuda")   # the K-split tile records up to 4 workgroups per 64 rows
            fv2p_native.call("fv2p_sparse_conv_set_trace", tr)
            ops.indice_conv(f, w, rb, None, n_out, False, mod.subm)
            torch.cuda.synchronize()
            fv2p_native.call("fv2p_sparse_conv_set_trace", None)
            full = tr.cpu().numpy().reshape(4 * nblk, 8)
            live = full[full[:, 2] > 0]
            if len(live) > nblk:   # every workgroup of a split launch: placement and timeline
                key = list(zip((live[:, 1] & 0xF).tolist(), ((live[:, 0] >> 13) & 7).tolist(), ((live[:, 0] >> 12) & 1).tolist(), ((live[:, 0] >> 8) & 0xF).tolist()))
                per_cu = collections.Counter(key)
                d = (live[:, 3] - live[:, 2]).astype(float)
                t_first, t_last = live[:, 2].min(), live[:, 3].max()
                busy = collections.defaultdict(float)
                for kk, dd in zip(key, d):
                    busy[kk] += dd
                print(f"all workgroups: {len(live)} on {len(per_cu)} CUs, workgroups/CU histogram {sorted(collections.Counter(per_cu.values()).items())}; "
                      f"span {t_last - t_first} clocks; workgroup clocks: median {np.median(d):.0f} max {d.max():.0f} sum/256 CUs {d.sum() / 256:.0f}; "
                      f"last start {live[:, 2].max() - t_first}; wait clocks median {np.median(live[:, 5]):.0f}")
                # one time base for all XCDs: the 100 MHz SoC clock at every workgroup's start and its lifetime in those ticks
                w0 = live[:, 6].astype(np.int64); life = (live[:, 1] >> 32).astype(np.int64)
                ok = life > 0
                if ok.any():
                    mhz = 100.0 * np.median(d[ok] / life[ok])
                    print(f"  SoC clock: shader clock {mhz:.0f} MHz; workgroup starts spread over {(w0.max() - w0.min()) / 100:.2f} us; lifetime median {np.median(life) / 100:.2f} "
                          f"max {life.max() / 100:.2f} us; first start -> last end of the main loop {((w0 + life).max() - w0.min()) / 100:.2f} us; "
                          f"prologue median {np.median(live[:, 4] - live[:, 2]) / mhz:.2f} us")
                    share_n = np.array([per_cu[kk] for kk in key])
                    for sn in sorted(set(share_n.tolist())):
                        sel = (share_n == sn) & ok
                        print(f"  workgroups on CUs holding {sn}: {int(sel.sum())}, lifetime median {np.median(life[sel]) / 100:.2f} us, main-loop MFMA clocks (wave 0) median {np.median(live[sel, 7]):.0f}, "
                              f"wait clocks median {np.median(live[sel, 5]):.0f}")
                    starts = np.sort(w0 - w0.min()) / 100
                    print("  start time of workgroup number 0/64/128/256/384/last (us):", [round(float(starts[min(i, len(starts) - 1)]), 2) for i in (0, 64, 128, 256, 384, len(starts) - 1)])
                ends = np.sort(live[:, 3] - t_first)
                print("  workgroups still running at 25/50/75/90 % of the span:", [int((ends > q * (t_last - t_first)).sum()) for q in (0.25, 0.5, 0.75, 0.9)])
            tr = full[:nblk]
            hw, xcc, t0, t1 = tr[:, 0], tr[:, 1] & 0xF, tr[:, 2], tr[:, 3]
            cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
            place = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
            print("distinct CUs used:", len(place), "blocks/CU histogram:", sorted(collections.Counter(place.values()).items()))
            dur = (t1 - t0).astype(float)
            pro, wait = (tr[:, 4] - t0).astype(float), tr[:, 5].astype(float)
            print(f"prologue clocks: median {np.median(pro):.0f} max {pro.max():.0f}; barrier-wait clocks per block: median {np.median(wait):.0f} "
                  f"({100 * np.median(wait / dur):.0f}% of the loop+prologue time)")
            # per-block activity from the rulebook: U = offsets with any active row in the 64-row tile, A = per-wave max
            tab = (rb.tab_in if getattr(rb, "tab_out", None) is None else rb.tab_out).cpu().numpy() >= 0   # [K, n]
            K = tab.shape[0]
            padn = nblk * 64 - tab.shape[1]
            act = np.concatenate([tab, np.zeros((K, padn), bool)], 1).reshape(K, nblk, 4, 16).any(3)      # [K, tile, wave]
            U_t = act.any(2).sum(0); A_t = act.sum(0).max(1)
            b = np.arange(nblk); x = b & 7; tile = x * (nblk >> 3) + np.minimum(x, nblk & 7) + (b >> 3)
            U, A = U_t[tile], A_t[tile]
            keys = list(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
            share = np.array([place[k] for k in keys])
            for sh_n in (1, 2):
                m = share == sh_n
                if m.any():
                    Xm = np.stack([U[m], np.ones(m.sum())], 1)
                    coef = np.linalg.lstsq(Xm, dur[m], rcond=None)[0]
                    print(f"blocks/CU={sh_n}: n={m.sum()} mean dur {dur[m].mean():.0f} mean U {U[m].mean():.1f} mean maxA {A[m].mean():.1f}; fit dur = {coef[0]:.0f}*U + {coef[1]:.0f}")
            order = np.argsort(dur)
            for i in list(order[:3]) + list(order[-3:]):
                print(f"  block {i} share={share[i]} U={U[i]} maxA={A[i]} dur={dur[i]:.0f} start={t0[i]-t0.min()}")
            print(f"block duration (shader clocks): min {dur.min():.0f} median {np.median(dur):.0f} max {dur.max():.0f}; "
                  f"span first start -> last end {t1.max() - t0.min()} clocks; starts spread {t0.max() - t0.min()}")
        return
    print(f"{'layer':34s} {'n_in':>6s} {'n_out':>6s} {'pairs':>8s} {'fwd us':>8s} {'TF/s':>6s} {'dX us':>8s} {'dW us':>8s} {'TF/s':>6s} {'dWpair':>8s} {'TF/s':>6s}")
    for mod, f, rb, n_out in recs:
        w = mod.weight.detach()
        p = int(rb.indice_pair_num.sum().item())
        g = torch.randn((n_out, mod.out_channels), device=dev)
        t_f = timeit(lambda: ops.indice_conv(f, w, rb, None, n_out, False, mod.subm))
        cin, cout = mod.in_channels, mod.out_channels
        w3 = w.reshape(-1, cin, cout)
        tab_b, flip_b = rb.in_table(cout)
        if cout in (64, 128) and cin % 64 == 0 and cin <= 128:   # as ops.indice_conv_backward: W_k^T materialised (the copy is inside the clock)
            t_dx = timeit(lambda: ops._conv_rows(g, w3.transpose(1, 2).contiguous(), tab_b, flip_b, f.shape[0], cin, False))
        else:
            t_dx = timeit(lambda: ops._conv_rows(g, w3, tab_b, flip_b, f.shape[0], cin, True))
        extra = ""
        if not mod.subm and not mod.transposed:   # strided conv: backward-data with the rows grouped by parity class
            import ctypes
            import fv2p_native
            arr = lambda v: (ctypes.c_int * 3)(*v)
            perm = torch.empty(f.shape[0], dtype=torch.int32, device=dev)
            ind = rb.indices
            pws = fv2p_native.workspace(int(fv2p_native.lib().fv2p_rulebook_class_perm_ws_bytes(ind.shape[0])), ind.device)
            fv2p_native.call("fv2p_rulebook_class_perm", ind, ind.shape[0], arr(mod.stride), arr(mod.padding), perm, pws, pws.numel(), fv2p_native.stream())
            din = torch.empty((f.shape[0], cin), device=dev)
            t_dxp = timeit(lambda: fv2p_native.call("fv2p_sparse_conv_rows_perm", g, g.shape[0], cout, w3, w3.shape[0], tab_b, f.shape[0], cin,
                                                    int(flip_b), 1, None, din, perm, fv2p_native.stream()))
            t_perm = timeit(lambda: fv2p_native.call("fv2p_rulebook_class_perm", ind, ind.shape[0], arr(mod.stride), arr(mod.padding), perm, pws,
                                                     pws.numel(), fv2p_native.stream()))
            extra = f"   dX parity-ordered {t_dxp:6.1f} us (perm build {t_perm:5.1f} us)"
        pairs_saved, rb._wpairs = rb._wpairs, None     # table-based weight gradient
        t_all = timeit(lambda: ops.indice_conv_backward(f, w, g, rb, None, False, mod.subm))
        rb._wpairs = pairs_saved
        rb.wgrad_pairs()                                 # pair-list weight gradient
        t_allp = timeit(lambda: ops.indice_conv_backward(f, w, g, rb, None, False, mod.subm))
        fl = 2.0 * p * cin * cout
        name = f"{'subm' if mod.subm else 'conv'} {cin}->{cout} {mod.indice_key}"
        print(f"{name:34s} {f.shape[0]:6d} {n_out:6d} {p:8d} {t_f:8.1f} {fl / t_f / 1e6:6.1f} {t_dx:8.1f} {t_all - t_dx:8.1f} {fl / max(t_all - t_dx, 1e-3) / 1e6:6.1f} {t_allp - t_dx:8.1f} {fl / max(t_allp - t_dx, 1e-3) / 1e6:6.1f}{extra}")
    # tiling-plan build per table (fv2p_conv_plan_build: what the first 64 / 128-channel conv on a rulebook pays once)
    import fv2p_native
    seen = set()
    for mod, f, rb, n_out in recs:
        if mod.indice_key in seen or mod.in_channels not in (64, 128):
            continue
        seen.add(mod.indice_key)
        for name, tab in (("tab_in", rb.tab_in), ("tab_out", rb.tab_out)):
            if tab is None:
                continue
            kv, nn_ = tab.shape
            ws = fv2p_native.workspace(int(fv2p_native.lib().fv2p_conv_plan_ws_bytes(nn_)), tab.device)
            t = timeit(lambda: fv2p_native.call("fv2p_conv_plan_build", tab, kv, nn_, ws, ws.numel(), fv2p_native.stream()), reps=20, warm=3)
            print(f"plan build {mod.indice_key} {name} [{kv} x {nn_}]: {t:.1f} us")
    # rulebook build timings
    x = torch.cat(coords)
    for subm, k, s, p in [(True, 3, 1, 1), (False, 3, 2, 1)]:
        t = timeit(lambda: ops.build_rulebook(x, 4, [41, 1600, 1408], k, s, p, 1, 0, subm), reps=20)
        print(f"rulebook {'subm' if subm else 'conv s2'} n={x.shape[0]}: {t:.1f} us")
    pts = torch.from_numpy(synth.lidar_cloud(0, 16384)).to(dev)
    t = timeit(lambda: points_to_voxel_gpu(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, True, 16000), reps=20)
    print(f"points_to_voxel 16384 pts: {t:.1f} us")


def dcn():
    from pcdet.ops.DeformableConvolutionV2PyTorch.modules.mdeformable_conv_block import MdeformConvBlock
    dev = torch.device("cuda:0")
    for (b, c, h, w, dg) in [(4, 128, 200, 176, 1), (4, 256, 100, 88, 1), (4, 256, 50, 44, 1), (4, 256, 200, 176, 4)]:
        blk = MdeformConvBlock(c, c, deformable_groups=dg).to(dev)
        with torch.no_grad():
            blk.conv_offset_mask.weight.normal_(0, 0.01)
        x = torch.randn(b, c, h, w, device=dev, requires_grad=True)
        off = torch.randn(b, dg * 18, h, w, device=dev) * 0.5
        msk = torch.sigmoid(torch.randn(b, dg * 9, h, w, device=dev))
        t_f = timeit(lambda: blk.conv_adaption(x.detach(), off, msk), reps=10, warm=2)
        y = blk.conv_adaption(x, off, msk)
        g = torch.randn_like(y)
        t_fb = timeit(lambda: torch.autograd.grad(blk.conv_adaption(x, off, msk), [x, blk.conv_adaption.weight], g), reps=5, warm=1)
        fl = 2.0 * b * h * w * c * c * 9
        print(f"DCNv2 [{b},{c},{h},{w}] dg={dg}: fwd {t_f:9.1f} us ({fl / t_f / 1e6:6.1f} TF/s)   fwd+bwd {t_fb:9.1f} us")


def fps():
    from fv2p_harness import synth
    from pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as bu
    dev = torch.device("cuda:0")
    for n, m, b in [(16384, 16384, 1), (16384, 16384, 3), (16384, 4096, 4), (20000, 16384, 1), (24576, 16384, 2), (40000, 16384, 2), (180000, 16384, 2)]:
        gen = synth.waymo_like_cloud if n > 30000 else synth.lidar_cloud    # n > 24576: the streaming kernel
        pts = torch.from_numpy(np.stack([gen(i, n)[:, :3] for i in range(b)])).to(dev)
        t = timeit(lambda: bu.furthest_point_sample(pts, m), reps=3, warm=1)
        print(f"FPS B={b} N={n} M={m}: {t / 1e3:8.2f} ms  ({t / m:6.3f} us/round)")
    kp = torch.from_numpy(synth.lidar_cloud(1, 16384)[None, :, :3]).to(dev)
    for v in (60000, 15000, 4000):
        known = torch.rand(1, v, 3, device=dev) * 70
        t = timeit(lambda: bu.three_nn(kp, known), reps=5, warm=1)
        print(f"three_nn 16384 x {v}: {t:8.1f} us ({8.0 * 16384 * v / t / 1e6:6.2f} TFLOP/s)")


def fpstrace():
    """Where a round of the streaming sampler goes: clocks per phase and wave (fv2p_fps_set_trace)."""
    import fv2p_native
    from fv2p_harness import synth
    from pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as bu
    dev = torch.device("cuda:0")
    for n in (16384, 40000, 180000):
        pts = torch.from_numpy((synth.lidar_cloud(0, n) if n <= 24576 else synth.waymo_like_cloud(0, n))[None, :, :3]).to(dev)
        bu.furthest_point_sample(pts, 2048)
        tr = torch.zeros(16 * 8, dtype=torch.int64, device=dev)
        fv2p_native.call("fv2p_fps_set_trace", tr)
        bu.furthest_point_sample(pts, 16384)
        torch.cuda.synchronize()
        fv2p_native.call("fv2p_fps_set_trace", None)
        t = tr.cpu().numpy().reshape(16, 8).astype(np.float64)
        rounds = 16383.0
        print(f"n = {n}: clocks per round and wave: box test, load issue, first bucket, other buckets, wave arg-max, exchange + barrier, pick | touched buckets per round")
        t = t[t[:, 6] > 0]   # the waves that ran (8 of the 16 trace rows)
        for w in range(t.shape[0]):
            print("  wave %2d: %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f | %5.2f" % ((w,) + tuple(t[w, :7] / rounds) + (t[w, 7] / rounds,)))
        print("  mean   : %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f | %5.2f   sum %7.1f" % (tuple(t[:, :7].mean(0) / rounds) + (t[:, 7].mean() / rounds, t[:, :7].mean(0).sum() / rounds)))


def nn():
    """Decoder-shaped 3-NN: 3 x 16384 key points (the clouds' own points) against the voxel centres of the four backbone levels,
    scan (fv2p_three_nn_stack) vs grid (fv2p_three_nn_stack_grid) with the lattice hint and with the library's own spacing estimate."""
    from fv2p_harness import synth
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as su
    dev = torch.device("cuda:0")
    waymo = bool(os.environ.get("FV2P_WAYMO"))
    nb = 2 if waymo else 3
    clouds = [synth.waymo_like_cloud(b, 180000) if waymo else synth.lidar_cloud(b, 16384) for b in range(nb)]
    vsz = np.array([0.1, 0.1, 0.15] if waymo else [0.05, 0.05, 0.1], np.float32)
    lo = np.array([-75.2, -75.2, -2.0] if waymo else [0.0, -40.0, -3.0], np.float32)
    rng = np.random.default_rng(0)
    keys = [c[rng.permutation(c.shape[0])[:16384], :3] for c in clouds]
    key = torch.from_numpy(np.concatenate(keys)).to(dev).contiguous()
    kc = torch.full((nb,), 16384, dtype=torch.int32, device=dev)
    for stride in (1, 2, 4, 8):
        cen, cnt = [], []
        for c in clouds:
            cell = np.unique(np.floor((c[:, :3] - lo) / (vsz * stride)).astype(np.int64), axis=0)
            cen.append(((cell + 0.5) * (vsz * stride) + lo).astype(np.float32))
            cnt.append(cell.shape[0])
        known = torch.from_numpy(np.concatenate(cen)).to(dev).contiguous()
        kcnt = torch.tensor(cnt, dtype=torch.int32, device=dev)
        os.environ["FV2P_NN_GRID"] = "0"
        t_scan = timeit(lambda: su.three_nn(key, kc, known, kcnt), reps=5, warm=1)
        d0, i0 = su.three_nn(key, kc, known, kcnt)
        os.environ["FV2P_NN_GRID"] = "1"
        su.GRID_MIN_KNOWN = 0    # the grid whatever the size: the threshold of the Python layer is what this table is for
        pitch = float(vsz[0]) * stride
        t_hint = {f: timeit(lambda: su.three_nn(key, kc, known, kcnt, f * pitch), reps=10, warm=2) for f in (1.0, 2.0, 3.0, 4.0)}
        t_auto = timeit(lambda: su.three_nn(key, kc, known, kcnt), reps=10, warm=2)
        d1, i1 = su.three_nn(key, kc, known, kcnt, 2.0 * pitch)
        same = bool(torch.equal(i0, i1) and torch.equal(d0, d1))
        print(f"three_nn {key.shape[0]} queries x {known.shape[0]} voxel centres (stride {stride}): scan {t_scan:8.1f} us   grid, cell = 1 / 2 / 3 / 4 voxel pitches "
              + " / ".join(f"{t_hint[f]:6.1f}" for f in (1.0, 2.0, 3.0, 4.0)) + f" us   estimated spacing {t_auto:7.1f} us   identical {same}")


def nms():
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from boxes_util import random_boxes
    from pcdet.ops.iou3d_nms import iou3d_nms_utils
    dev = torch.device("cuda:0")
    for n, th in [(9000, 0.8), (4096, 0.1)]:
        boxes = torch.from_numpy(random_boxes(n, n)).to(dev)
        scores = torch.rand(n, device=dev)
        t = timeit(lambda: iou3d_nms_utils.nms_gpu(boxes, scores, th), reps=5, warm=1)
        print(f"nms_gpu N={n} thr={th}: {t:8.1f} us")
    # proposal layer of the FV2P step: three samples, 9000 sorted boxes each, first 512 survivors (fv2p_nms_batch)
    from pcdet.ops.iou3d_nms import iou3d_nms_cuda
    batch = torch.from_numpy(np.stack([random_boxes(7 + i, 9000) for i in range(3)])).to(dev)
    for keep in (512, 0):
        t = timeit(lambda: iou3d_nms_cuda.nms_batch_device(batch, 0.8, keep), reps=5, warm=1)
        print(f"nms_batch B=3 N=9000 thr=0.8 max_keep={keep}: {t:8.1f} us")


def sa():
    from pcdet.ops.pointnet2.pointnet2_batch import fused, pointnet2_utils as bu
    dev = torch.device("cuda:0")
    r, n, m = 384, 512, 216
    g = torch.Generator().manual_seed(0)
    xyz = (torch.rand(r, n, 3, generator=g) * torch.tensor([7.0, 5.0, 4.0]) - torch.tensor([3.5, 2.5, 2.0])).to(dev)
    ctr = (torch.rand(r, m, 3, generator=g) * torch.tensor([4.0, 1.6, 1.5]) - torch.tensor([2.0, 0.8, 0.75])).to(dev)
    pp = torch.randn(r, n, 64, generator=g).to(dev).requires_grad_(True)
    pc = torch.randn(r, m, 64, generator=g).to(dev).requires_grad_(True)
    w2 = (torch.randn(64, 64, generator=g) * 0.1).to(dev).requires_grad_(True)
    for radius, s in ((0.8, 16), (1.6, 32)):
        idx = bu.ball_query(radius, s, xyz, ctr)
        tf = timeit(lambda: fused.sa_grid_max(pp, pc, idx, w2), reps=10, warm=2)
        out = fused.sa_grid_max(pp, pc, idx, w2)
        go = torch.randn_like(out)
        tb = timeit(lambda: out.backward(go, retain_graph=True), reps=10, warm=2)
        flops = 2.0 * r * m * s * 64 * 64
        print(f"fused grid SA R={r} M={m} S={s}: fwd {tf:7.1f} us ({flops / tf / 1e6:5.1f} TF/s)  bwd {tb:7.1f} us")

        def grouped():
            gpt = bu.grouping_operation(pp.transpose(1, 2).contiguous(), idx)
            h2 = torch.relu(torch.einsum("oc,rcms->roms", w2, torch.relu(gpt - pc.transpose(1, 2).unsqueeze(-1))))
            return h2.amax(dim=-1)
        tg = timeit(lambda: grouped(), reps=3, warm=1)
        ref = grouped()
        gr = torch.randn_like(ref)
        tgb = timeit(lambda: ref.backward(gr, retain_graph=True), reps=3, warm=1)
        print(f"   grouped formulation (group_points + einsum + relu + amax): fwd {tg:7.1f} us  bwd {tgb:7.1f} us")


def bn():
    import copy
    from torch import nn
    from pcdet.ops.spconv import norm
    dev = torch.device("cuda:0")
    for n, c in [(45868, 16), (50783, 32), (29446, 64), (13425, 64), (11446, 128), (49152, 64), (49152, 128), (49152, 256)]:   # backbone levels, decoder / point-head Linear blocks
        x = torch.randn(n, c, device=dev, requires_grad=True)
        g = torch.randn(n, c, device=dev)
        m = nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).to(dev)
        act = nn.ReLU()
        tf = timeit(lambda: norm.batch_norm_relu(m, x, act))
        y = norm.batch_norm_relu(m, x, act)
        tb = timeit(lambda: y.backward(g, retain_graph=True))
        m2 = copy.deepcopy(m)
        rf = timeit(lambda: act(m2(x)))
        y2 = act(m2(x))
        rb = timeit(lambda: y2.backward(g, retain_graph=True))
        print(f"bn+relu n={n} c={c}: fused fwd {tf:6.1f} us bwd {tb:6.1f} us | torch fwd {rf:6.1f} us bwd {rb:6.1f} us | "
              f"fwd {3 * 4 * n * c / tf / 1e6:5.2f} TB/s")


def bev():
    """Bilinear BEV gather at the FV2P size vs the reference's torch composition (permute + 4 index gathers + weights)."""
    from pcdet.models.backbones_3d.pfe import bev_grid_pooling as bgp
    dev = torch.device("cuda:0")
    b, c, h, w, n = 4, 128, 200, 176, 27648
    bev_f = torch.randn(b, c, h, w, device=dev, requires_grad=True)
    kp = torch.rand(b, n, 3, device=dev) * torch.tensor([70.4, 80.0, 4.0], device=dev) + torch.tensor([0.0, -40.0, -3.0], device=dev)
    rng, vox = [0.0, -40.0, -3.0, 70.4, 40.0, 1.0], [0.05, 0.05, 0.1]

    def torch_ref():
        xs = ((kp[:, :, 0] - rng[0]) / vox[0]) / 8
        ys = ((kp[:, :, 1] - rng[1]) / vox[1]) / 8
        outs = []
        for k in range(b):
            im = bev_f[k].permute(1, 2, 0).contiguous()
            x, y = xs[k], ys[k]
            x0, y0 = torch.floor(x).long(), torch.floor(y).long()
            x1, y1 = x0 + 1, y0 + 1
            x0, x1 = x0.clamp(0, w - 1), x1.clamp(0, w - 1)
            y0, y1 = y0.clamp(0, h - 1), y1.clamp(0, h - 1)
            wa, wb = (x1.float() - x) * (y1.float() - y), (x1.float() - x) * (y - y0.float())
            wc, wd = (x - x0.float()) * (y1.float() - y), (x - x0.float()) * (y - y0.float())
            outs.append((im[y0, x0] * wa[:, None] + im[y1, x0] * wb[:, None] + im[y0, x1] * wc[:, None] + im[y1, x1] * wd[:, None]).unsqueeze(0))
        return torch.cat(outs)

    ours = lambda: bgp.interpolate_from_bev_features(kp, bev_f, b, 8, rng, vox)
    g = torch.randn(b, n, c, device=dev)
    t_f, t_rf = timeit(lambda: ours().detach(), reps=20), timeit(lambda: torch_ref().detach(), reps=20)
    o, r = ours(), torch_ref()
    t_b = timeit(lambda: o.backward(g, retain_graph=True), reps=20)
    t_rb = timeit(lambda: r.backward(g, retain_graph=True), reps=20)
    byt = 4.0 * b * (c * h * w * 2 + n * c * 5)
    print(f"bev gather [{b},{c},{h},{w}] x {n} pts: fwd {t_f:7.1f} us ({byt / t_f / 1e6:5.2f} TB/s)  bwd {t_b:7.1f} us | torch composition fwd {t_rf:7.1f} us bwd {t_rb:7.1f} us")


def oproof():
    """Per-op roofline fractions with SURVEY 8(d)'s algorithmic bytes / flops at the FV2P step's shapes.
    HBM ops are priced against 8 TB/s, the brute-force searches against the 157.3 TFLOP/s fp32 vector peak as well."""
    from fv2p_harness import synth
    from pcdet.datasets.processor.voxel_generator import points_to_voxel_batch
    from pcdet.ops.iou3d_nms import iou3d_nms_cuda, iou3d_nms_utils
    from pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as bu
    from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as su
    from pcdet.ops.roiaware_pool3d import roiaware_pool3d_utils
    from pcdet.ops.roipoint_pool3d import roipoint_pool3d_utils
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from boxes_util import random_boxes
    dev = torch.device("cuda:0")
    HBM, VEC = 8000.0, 157.3   # GB/s, TFLOP/s
    rows = []

    def add(name, t_us, nbytes=None, flops=None, note=""):
        gbs = nbytes / t_us / 1e3 if nbytes else None
        tf = flops / t_us / 1e6 if flops else None
        rows.append((name, t_us, gbs, gbs / HBM if gbs else None, tf, tf / VEC if tf else None, note))

    # voxeliser: 16 N + M (max_pts * 16 + 12 + 4)
    pts = [torch.from_numpy(synth.lidar_cloud(i, 16384)).to(dev) for i in range(3)]
    vs, rng = np.array([0.05, 0.05, 0.1], np.float32), np.array([0, -40, -3, 70.4, 40, 1], np.float32)
    f, c = points_to_voxel_batch(pts, vs, rng, 5, 16000, mean_vfe=True, cloud_streams=False)
    m = f.shape[0]
    t = timeit(lambda: points_to_voxel_batch(pts, vs, rng, 5, 16000, mean_vfe=True, cloud_streams=False), reps=10, warm=2)
    add("points_to_voxel + MeanVFE, 3 x 16384 pts", t, 16.0 * 3 * 16384 + m * (5 * 16 + 12 + 4), note=f"{m} voxels; includes the host wait for the counts")
    # stacked three_nn / interpolate of the decoder: 3 x 16384 key points against a level's voxel centres
    key = torch.cat([p[:, :3] for p in pts]).contiguous()
    kc = torch.full((3,), 16384, dtype=torch.int32, device=dev)
    for nk, ch in ((35146, 16), (22331, 64), (9919, 128)):
        known = (torch.rand(nk, 3, device=dev) * torch.tensor([70.4, 80.0, 4.0], device=dev) + torch.tensor([0.0, -40.0, -3.0], device=dev)).contiguous()
        cnt = torch.tensor([nk // 3, nk // 3, nk - 2 * (nk // 3)], dtype=torch.int32, device=dev)
        os.environ["FV2P_NN_GRID"] = "0"   # the scan itself (uniform random points here; the decoder's lattices go through the hashed grid: microbench nn)
        t = timeit(lambda: su.three_nn(key, kc, known, cnt), reps=5, warm=1)
        os.environ.pop("FV2P_NN_GRID")
        nu = key.shape[0]
        add(f"three_nn (stack) {nu} x {nk // 3} per sample", t, 12.0 * (nu + nk) + 24.0 * nu, 8.0 * nu * (nk / 3), "brute force, LDS tiled")
        dist, idx = su.three_nn(key, kc, known, cnt)
        w = torch.rand(nu, 3, device=dev)
        feats = torch.randn(nk, ch, device=dev, requires_grad=True)
        t = timeit(lambda: su.three_interpolate(feats, idx, w), reps=10, warm=2)
        add(f"three_interpolate (stack) {nu} x C={ch}", t, 4.0 * ch * (3 * nu + nu) + 24.0 * nu)
        out = su.three_interpolate(feats, idx, w)
        g = torch.randn_like(out)
        os.environ["FV2P_INTERP_GATHER"] = "0"
        t = timeit(lambda: out.backward(g, retain_graph=True), reps=10, warm=2)
        add(f"three_interpolate grad {nu} x C={ch}", t, 4.0 * ch * (3 * nu + nu) + 24.0 * nu, note="scatter form: zero fill + float atomics")
        os.environ["FV2P_INTERP_GATHER"] = "1"
        t = timeit(lambda: out.backward(g, retain_graph=True), reps=10, warm=2)
        os.environ.pop("FV2P_INTERP_GATHER")
        add(f"three_interpolate grad {nu} x C={ch}, gather form", t, 4.0 * ch * (3 * nu + nu) + 24.0 * nu, note="sorted (row, entry) keys summed in 32-entry segments: no float atomics, fixed order")
    # batch grouping at the RoI head's shape: 384 RoIs x 512 points, 216 centres, 16 / 32 samples, C = 64 + 3
    r, n, mc, ch = 384, 512, 216, 67
    xyz = torch.rand(r, n, 3, device=dev)
    ctr = torch.rand(r, mc, 3, device=dev)
    featb = torch.randn(r, ch, n, device=dev, requires_grad=True)
    for ns, rad in ((16, 0.2), (32, 0.4)):
        t = timeit(lambda: bu.ball_query(rad, ns, xyz, ctr), reps=10, warm=2)
        add(f"ball_query R={r} N={n} M={mc} ns={ns}", t, 12.0 * r * (n + mc) + 4.0 * r * mc * ns, 8.0 * r * mc * n)
        idx = bu.ball_query(rad, ns, xyz, ctr)
        t = timeit(lambda: bu.grouping_operation(featb, idx), reps=10, warm=2)
        add(f"group_points R={r} C={ch} M={mc} ns={ns}", t, 4.0 * ch * r * mc * ns * 2 + 4.0 * r * mc * ns)
        out = bu.grouping_operation(featb, idx)
        g = torch.randn_like(out)
        t = timeit(lambda: out.backward(g, retain_graph=True), reps=10, warm=2)
        add(f"group_points grad R={r} C={ch} M={mc} ns={ns}", t, 4.0 * ch * r * mc * ns * 2 + 4.0 * r * mc * ns, note="LDS accumulator per (RoI, channel) row")
    # NMS of the proposal layer and the RoI-head IoU / pools
    batch = torch.from_numpy(np.stack([random_boxes(7 + i, 9000) for i in range(3)])).to(dev)
    t = timeit(lambda: iou3d_nms_cuda.nms_batch_device(batch, 0.8, 512), reps=5, warm=1)
    # priced by the tiles the launch sequence EXECUTES (a sample stops at the end of the chunk of row blocks that holds its 512-th survivor;
    # a row block evaluates the column blocks at and right of it): 64 x 64 overlaps of ~300 flop and 64 mask words per tile
    keep, cnt = iou3d_nms_cuda.nms_batch_device(batch, 0.8, 512)
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    cb, first, tiles = (9000 + 63) // 64, (2 * 512 + 63) // 64, 0
    for smp in range(3):
        last_row = int(keep[smp, 511]) if cnt[smp] >= 512 else 9000 - 1
        end, span = first, first
        while end < min(cb, last_row // 64 + 1):
            span *= 4
            end += span
        end = min(end, cb)
        tiles += sum(cb - rb for rb in range(end))
    add("nms_batch 3 x 9000 -> 512 (truncated)", t, 3 * 28.0 * 9000 + tiles * 64 * 8.0, tiles * 4096.0 * 300,
        f"priced by the {tiles} executed 64 x 64 tiles ({tiles / (3 * cb * (cb + 1) / 2):.2f} of the full triangle)")
    t = timeit(lambda: iou3d_nms_cuda.nms_batch_device(batch, 0.8, 0), reps=3, warm=1)
    add("nms_batch 3 x 9000, all survivors", t, 3 * (28.0 * 9000 + 9000.0 * 9000 / 8), 3 * 9000.0 * 9000 / 2 * 300)
    a, b2 = batch[0, :512].contiguous(), batch[1, :40].contiguous()
    t = timeit(lambda: iou3d_nms_utils.boxes_iou3d_gpu(a, b2), reps=10, warm=2)
    add("boxes_iou3d 512 x 40", t, 28.0 * 552 + 4.0 * 512 * 40, 512 * 40 * 300.0, "launch latency")
    kp = torch.stack([p[:, :3] for p in pts]).contiguous()
    gt = batch[:, :40].contiguous()
    t = timeit(lambda: roiaware_pool3d_utils.points_in_boxes_gpu(kp, gt), reps=10, warm=2)
    add("points_in_boxes 3 x 16384 pts x 40 boxes", t, 3 * (12.0 * 16384 + 28.0 * 40 + 4.0 * 16384))
    pool = roipoint_pool3d_utils.RoIPointPool3d(num_sampled_points=512, pool_extra_width=(1.0, 1.0, 1.0))
    fe = torch.randn(3, 16384, 130, device=dev)
    rois = batch[:, :128].contiguous()
    t = timeit(lambda: pool(kp, fe, rois), reps=5, warm=1)
    add("roipoint_pool3d 3 x 128 RoIs x 512 pts x (3+130)", t, 3 * 128 * 512 * 133 * 4.0, note="bytes written (8(d))")
    print(f"{'op':58s} {'us':>9s} {'GB/s':>8s} {'of HBM':>7s} {'TFLOP/s':>8s} {'of vec':>7s}  note")
    for name, t_us, gbs, fh, tf, fv, note in rows:
        fmt = lambda v, w, p: (f"{v:{w}.{p}f}" if v is not None else " " * w)
        print(f"{name:58s} {t_us:9.1f} {fmt(gbs, 8, 1)} {fmt(fh, 7, 3)} {fmt(tf, 8, 2)} {fmt(fv, 7, 3)}  {note}")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "convone":
        conv(only=20)
        sys.exit(0)
    for name, fn in (("conv", conv), ("dcn", dcn), ("fps", fps), ("fpstrace", fpstrace), ("nn", nn), ("nms", nms), ("sa", sa), ("bn", bn), ("bev", bev), ("oproof", oproof)):
        if which in (name, "all"):
            print(f"==== {name}")
            fn()
