"""GPU parity of the hashed rulebook (A3/A4) against the geometry.h restatement (oracle/rulebook_oracle.c).

Bar: bit-exact — output coordinates (sorted flat order, the GPU reference's torch::_unique order), per-offset
pair counts, and the pair lists in canonical order (ascending input row within an offset)."""
import numpy as np
import pytest
import torch

import oracle
from pcdet.ops.spconv import ops
from sparse_util import canon_pairs, random_active, voxel_indices_from_clouds

pytestmark = pytest.mark.gpu

CASES = [
    # name, batch, shape, n, ksize, stride, padding, dilation, subm, transpose
    ("subm3", 2, [9, 20, 18], 700, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], True, False),
    ("subm3_dense", 1, [4, 5, 6], 120, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], True, False),
    ("subm_k133", 2, [6, 12, 12], 300, [1, 3, 3], [1, 1, 1], [0, 1, 1], [1, 1, 1], True, False),
    ("subm_dil2", 1, [8, 16, 16], 400, [3, 3, 3], [1, 1, 1], [1, 1, 1], [2, 2, 2], True, False),
    ("subm_k2", 1, [8, 10, 10], 300, [2, 2, 2], [1, 1, 1], [1, 1, 1], [1, 1, 1], True, False),
    ("conv_k3s2p1", 2, [9, 20, 18], 700, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], False, False),
    ("conv_k3s2p011", 2, [11, 20, 18], 500, [3, 3, 3], [2, 2, 2], [0, 1, 1], [1, 1, 1], False, False),
    ("conv_k311s211", 2, [5, 20, 18], 500, [3, 1, 1], [2, 1, 1], [0, 0, 0], [1, 1, 1], False, False),
    ("conv_k2s2", 1, [8, 16, 16], 600, [2, 2, 2], [2, 2, 2], [0, 0, 0], [1, 1, 1], False, False),
    ("conv_k3s1p1", 1, [6, 10, 10], 200, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], False, False),
    ("conv_k3s1p0_dil2", 1, [9, 12, 12], 300, [3, 3, 3], [1, 1, 1], [0, 0, 0], [2, 2, 2], False, False),
    ("conv_k3s3p0", 1, [9, 12, 12], 300, [3, 3, 3], [3, 3, 3], [0, 0, 0], [1, 1, 1], False, False),
    ("deconv_k3s2p1", 2, [5, 10, 9], 300, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], False, True),
    ("deconv_k2s2", 1, [4, 8, 8], 100, [2, 2, 2], [2, 2, 2], [0, 0, 0], [1, 1, 1], False, True),
]


def _check(gpu, ind, batch, shape, k, s, p, d, subm, transpose):
    o_out, o_pairs, o_num = oracle.indice_pairs(ind, batch, shape, k, s, p, d, subm=subm, transpose=transpose,
                                                force_sparse=np.prod(shape) * batch > 5e7)
    outids, pairs, num = ops.get_indice_pairs(torch.from_numpy(ind).to(gpu), batch, shape, k, s, p, d, 0, subm, transpose)
    assert outids.dtype == torch.int32 and pairs.dtype == torch.int32 and num.dtype == torch.int32
    assert pairs.shape == (int(np.prod(k)), 2, ind.shape[0])
    assert np.array_equal(outids.cpu().numpy(), o_out)
    num = num.cpu().numpy()
    assert np.array_equal(num, o_num)
    pairs = pairs.cpu().numpy()
    for kk, (a, b) in enumerate(zip(canon_pairs(pairs, num), canon_pairs(o_pairs, o_num))):
        assert np.array_equal(a, b), f"offset {kk}"
        assert (pairs[kk, :, int(num[kk]):] == -1).all()  # -1 padding as spconv_ops.h:55-57
    # our pair order is already canonical (ascending input row)
    for kk in range(pairs.shape[0]):
        assert (np.diff(pairs[kk, 0, : int(num[kk])]) > 0).all()
    return outids, num


@pytest.fixture(params=["bitmap", "hashed"])
def rb_path(request):
    """Strided / transposed rulebooks rank their output cells through a bitmap of the grid (default) or through the hash
    set + radix sort (grids above 2^28 cells): both must equal the oracle."""
    import fv2p_native
    fv2p_native.lib().fv2p_rulebook_set_path(1 if request.param == "hashed" else 0)
    yield request.param
    fv2p_native.lib().fv2p_rulebook_set_path(0)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_rulebook_matches_oracle(gpu, case, rb_path):
    _, batch, shape, n, k, s, p, d, subm, transpose = case
    ind = random_active(sum(map(ord, case[0])), batch, shape, n)
    _check(gpu, ind, batch, shape, k, s, p, d, subm, transpose)


def test_rulebook_2d(gpu):
    ind3 = random_active(5, 2, [1, 30, 28], 300)
    ind2 = np.ascontiguousarray(ind3[:, [0, 2, 3]])
    for subm, k, s, p in [(True, [3, 3], [1, 1], [1, 1]), (False, [3, 3], [2, 2], [1, 1])]:
        o_out, o_pairs, o_num = oracle.indice_pairs(ind3, 2, [1, 30, 28], [1] + k, [1] + s, [0] + p, [1, 1, 1], subm=subm)
        outids, pairs, num = ops.get_indice_pairs(torch.from_numpy(ind2).to(gpu), 2, [30, 28], k, s, p, 1, 0, subm)
        assert outids.shape[1] == 3
        assert np.array_equal(outids.cpu().numpy(), o_out[:, [0, 2, 3]])
        assert np.array_equal(num.cpu().numpy(), o_num)
        for a, b in zip(canon_pairs(pairs.cpu().numpy(), o_num), canon_pairs(o_pairs, o_num)):
            assert np.array_equal(a, b)


def test_rulebook_kitti_backbone_chain(gpu, rb_path):
    """Full-size KITTI grid, batch 2: the eight rulebooks of VoxelBackBone8x chained level to level."""
    ind = voxel_indices_from_clouds([0, 1])
    shape = [41, 1600, 1408]
    chain = [
        ([3, 3, 3], [1, 1, 1], [1, 1, 1], True), ([3, 3, 3], [2, 2, 2], [1, 1, 1], False),
        ([3, 3, 3], [1, 1, 1], [1, 1, 1], True), ([3, 3, 3], [2, 2, 2], [1, 1, 1], False),
        ([3, 3, 3], [1, 1, 1], [1, 1, 1], True), ([3, 3, 3], [2, 2, 2], [0, 1, 1], False),
        ([3, 3, 3], [1, 1, 1], [1, 1, 1], True), ([3, 1, 1], [2, 1, 1], [0, 0, 0], False),
    ]
    for k, s, p, subm in chain:
        outids, num = _check(gpu, ind, 2, shape, k, s, p, [1, 1, 1], subm, False)
        if subm:
            assert num[13] == ind.shape[0] and num.argmax() == 13  # centre offset = identity (spconv_ops.h:272-277)
        else:
            shape = ops.get_conv_output_size(shape, k, s, p, [1, 1, 1])
            ind = outids.cpu().numpy()
    assert shape == [2, 200, 176]


def test_rulebook_edge_cases(gpu, rb_path):
    # empty active set
    e = torch.zeros((0, 4), dtype=torch.int32, device=gpu)
    outids, pairs, num = ops.get_indice_pairs(e, 1, [4, 4, 4], 3, 1, 1, 1, 0, True)
    assert outids.shape[0] == 0 and pairs.shape == (27, 2, 0) and int(num.sum()) == 0
    outids, pairs, num = ops.get_indice_pairs(e, 1, [4, 4, 4], 3, 2, 1, 1, 0, False)
    assert outids.shape == (0, 4) and int(num.sum()) == 0
    # single voxel in a corner
    one = torch.tensor([[0, 0, 0, 0]], dtype=torch.int32, device=gpu)
    outids, pairs, num = ops.get_indice_pairs(one, 1, [4, 4, 4], 3, 1, 1, 1, 0, True)
    assert int(num.sum()) == 1 and int(num[13]) == 1
    # fully dense grid: subm has every neighbour inside the volume
    full = random_active(0, 1, [3, 4, 5], 60, sort=True)
    _check(gpu, full, 1, [3, 4, 5], [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], True, False)
    # duplicate coordinates: the highest row wins the slot (geometry.h:275-280)
    dup = np.array([[0, 1, 1, 1], [0, 1, 1, 2], [0, 1, 1, 1]], np.int32)
    _check(gpu, dup, 1, [3, 3, 4], [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], True, False)
    # CPU tensors are rejected, not silently handled
    with pytest.raises(Exception):
        ops.get_indice_pairs(torch.zeros((3, 4), dtype=torch.int32), 1, [4, 4, 4], 3, 1, 1, 1, 0, True)


def test_foreign_pair_lists_round_trip(gpu):
    """A rulebook handed over in the reference format is converted to tables and reproduces the same pairs."""
    ind = random_active(3, 2, [9, 20, 18], 500)
    o_out, o_pairs, o_num = oracle.indice_pairs(ind, 2, [9, 20, 18], [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1])
    rb = ops._rulebook_of(torch.from_numpy(o_pairs).to(gpu), torch.from_numpy(o_num).to(gpu), ind.shape[0], o_out.shape[0], False)
    rb._pairs = None
    pairs = rb.indice_pairs.cpu().numpy()
    for a, b in zip(canon_pairs(pairs, o_num), canon_pairs(o_pairs, o_num)):
        assert np.array_equal(a, b)


def test_waymo_scale_rulebooks_and_conv_against_oracle(gpu):
    """BASELINE configs[4] shape: one ~180 k-point Waymo-like cloud on the [41, 1504, 1504] grid (0.1 m voxels): the
    hashed rulebooks (submanifold and strided) equal the oracle's bit for bit at ~1e5 rows, and the fused conv on
    them equals the oracle's gather -> mm -> scatter loop within 1e-4."""
    import torch
    import pcdet.ops.spconv as spconv
    from fv2p_harness import synth
    from pcdet.datasets.processor.voxel_generator import points_to_voxel_gpu
    from pcdet.ops.spconv import ops
    pts = torch.from_numpy(synth.waymo_like_cloud(3, 180000)).to(gpu)
    v, c, n = points_to_voxel_gpu(pts, synth.WAYMO_VOXEL, synth.WAYMO_RANGE, 5, True, 150000)
    assert v.shape[0] > 40000
    ind = torch.nn.functional.pad(c, (1, 0), value=0).contiguous()
    shape = [41, 1504, 1504]
    ind_np = ind.cpu().numpy()
    rb_s = ops.build_rulebook(ind, 1, shape, 3, 1, 1, 1, 0, True)
    _, p_s, n_s = oracle.indice_pairs(ind_np, 1, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], subm=True)
    assert np.array_equal(rb_s.indice_pair_num.cpu().numpy(), n_s)
    assert np.array_equal(rb_s.indice_pairs.cpu().numpy(), p_s)
    rb_d = ops.build_rulebook(ind, 1, shape, 3, 2, 1, 1, 0, False)
    o_d, p_d, n_d = oracle.indice_pairs(ind_np, 1, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1])
    assert np.array_equal(rb_d.outids.cpu().numpy(), o_d) and np.array_equal(rb_d.indice_pair_num.cpu().numpy(), n_d)
    assert np.array_equal(rb_d.indice_pairs.cpu().numpy(), p_d)
    feats = np.random.default_rng(0).standard_normal((ind_np.shape[0], 16)).astype(np.float32)
    torch.manual_seed(0)
    conv = spconv.SparseConv3d(16, 32, 3, stride=2, padding=1, bias=False).to(gpu)
    y = conv(spconv.SparseConvTensor(torch.from_numpy(feats).to(gpu), ind, shape, 1))
    ref = oracle.indice_conv(feats, conv.weight.detach().cpu().numpy(), p_d, n_d, o_d.shape[0]).numpy()
    got = y.features.detach().cpu().numpy()
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-4
    # backward-data of the strided conv at this size on parity-ordered tiles (fv2p_rulebook_class_perm): the permutation is
    # the stable grouping by (coordinate + padding) mod stride and the rows equal the plain launch bit for bit
    import ctypes
    import fv2p_native
    n_in, n_out = ind.shape[0], o_d.shape[0]
    arr = lambda v: (ctypes.c_int * 3)(*v)
    perm = torch.empty(n_in, dtype=torch.int32, device=gpu)
    ws = fv2p_native.workspace(int(fv2p_native.lib().fv2p_rulebook_class_perm_ws_bytes(n_in)), ind.device)
    fv2p_native.call("fv2p_rulebook_class_perm", ind, n_in, arr((2, 2, 2)), arr((1, 1, 1)), perm, ws, ws.numel(), fv2p_native.stream())
    cls = ((ind_np[:, 1:] + 1) % 2) @ np.array([4, 2, 1])
    assert np.array_equal(perm.cpu().numpy(), np.argsort(cls, kind="stable"))
    g = torch.randn((n_out, 32), device=gpu)
    w3 = conv.weight.detach().reshape(27, 16, 32).contiguous()
    tab, flip = rb_d.in_table()
    plain, ordered = torch.empty((n_in, 16), device=gpu), torch.empty((n_in, 16), device=gpu)
    fv2p_native.call("fv2p_sparse_conv_rows", g, n_out, 32, w3, 27, tab, n_in, 16, int(flip), 1, None, plain, fv2p_native.stream())
    fv2p_native.call("fv2p_sparse_conv_rows_perm", g, n_out, 32, w3, 27, tab, n_in, 16, int(flip), 1, None, ordered, perm, fv2p_native.stream())
    assert torch.equal(plain, ordered)
