"""GPU parity of DCNv2 / DCNv1 (A13/A14) — the reference's own self-checks (DeformableConvolutionV2PyTorch/test.py)
restated, plus random-offset parity against the torch-CPU restatement (oracle/dcn_oracle.py).

Tolerances: forward and all five gradients 1e-4 relative to the largest entry (north_star; the reference's own checks use 1e-5
absolute on tiny tensors for the forward and atol 1e-3 / rtol 1e-2 in its gradcheck, test.py:351-435).  The backward holds no float
atomics: two runs give the same bits (test_backward_is_bit_identical_run_to_run)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import dcn_oracle
from pcdet.ops.DeformableConvolutionV2PyTorch.modules.deform_conv import DeformConv
from pcdet.ops.DeformableConvolutionV2PyTorch.modules.mdeformable_conv_block import MdeformConvBlock
from pcdet.ops.DeformableConvolutionV2PyTorch.modules.modulated_deform_conv import ModulatedDeformConv, ModulatedDeformConvPack

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("cin,cout,dg,hw", [(16, 16, 1, (9, 11)), (64, 32, 4, (12, 10)), (128, 128, 1, (20, 16)), (32, 200, 2, (7, 7))])
def test_zero_offset_equals_conv2d(gpu, cin, cout, dg, hw):
    """test.py:69-110 (check_mdconv_zero_offset): zero offsets, mask = 1 -> plain convolution."""
    torch.manual_seed(3)
    B, (H, W) = 2, hw
    x = torch.randn(B, cin, H, W, device=gpu)
    m = ModulatedDeformConv(cin, cout, 3, stride=1, padding=1, deformable_groups=dg, bias=True).to(gpu)
    offset = torch.zeros(B, dg * 18, H, W, device=gpu)
    mask = torch.ones(B, dg * 9, H, W, device=gpu)
    y = m(x, offset, mask)
    ref = F.conv2d(x.cpu(), m.weight.detach().cpu(), m.bias.detach().cpu(), padding=1)
    assert y.shape == ref.shape and y.is_contiguous()
    assert rel(y, ref) < 1e-4
    # DCNv1 with zero offsets as well (test.py:36-67)
    d = DeformConv(cin, cout, 3, stride=1, padding=1, deformable_groups=dg, bias=False).to(gpu)
    assert rel(d(x, offset), F.conv2d(x.cpu(), d.weight.detach().cpu(), d.bias.detach().cpu(), padding=1)) < 1e-4


def test_identity_kernel_returns_input(gpu):
    """test.py:142-181: centre-tap identity weights, zero offset, unit mask -> output == input."""
    x = torch.randn(2, 32, 10, 12, device=gpu)
    m = ModulatedDeformConv(32, 32, 3, stride=1, padding=1, bias=False).to(gpu)
    with torch.no_grad():
        m.weight.zero_()
        m.weight[torch.arange(32), torch.arange(32), 1, 1] = 1.0
        m.bias.zero_()
    y = m(x, torch.zeros(2, 18, 10, 12, device=gpu), torch.ones(2, 9, 10, 12, device=gpu))
    assert rel(y, x) < 1e-6


@pytest.mark.parametrize("cin,cout,dg,stride,dil", [(32, 48, 2, 1, 1), (16, 16, 1, 2, 1), (64, 64, 4, 1, 2), (32, 50, 1, 1, 1)])
def test_random_offsets_forward_backward_vs_oracle(gpu, cin, cout, dg, stride, dil):
    torch.manual_seed(0)
    B, H, W = 2, 11, 13
    pad = dil
    Ho, Wo = (H + 2 * pad - (dil * 2 + 1)) // stride + 1, (W + 2 * pad - (dil * 2 + 1)) // stride + 1
    x = torch.randn(B, cin, H, W)
    offset = torch.randn(B, dg * 18, Ho, Wo) * 1.5     # frequently leaves the image: exercises the border rule
    mask = torch.sigmoid(torch.randn(B, dg * 9, Ho, Wo))
    m = ModulatedDeformConv(cin, cout, 3, stride=stride, padding=pad, dilation=dil, deformable_groups=dg, bias=True).to(gpu)
    gx, go, gm = (t.clone().to(gpu).requires_grad_(True) for t in (x, offset, mask))
    y = m(gx, go, gm)
    cx, co, cm = (t.clone().double().requires_grad_(True) for t in (x, offset, mask))
    w, b = m.weight.detach().cpu().double().requires_grad_(True), m.bias.detach().cpu().double().requires_grad_(True)
    ref = dcn_oracle.modulated_deform_conv(cx, co, cm, w, b, (stride, stride), (pad, pad), (dil, dil), dg)
    assert rel(y, ref) < 1e-4
    g = torch.randn(ref.shape)
    y.backward(g.to(gpu))
    ref.backward(g.double())
    assert rel(gx.grad, cx.grad) < 1e-4
    assert rel(go.grad, co.grad) < 1e-4
    assert rel(gm.grad, cm.grad) < 1e-4
    assert rel(m.weight.grad, w.grad) < 1e-4
    assert rel(m.bias.grad, b.grad) < 1e-4


@pytest.mark.parametrize("order", [0, 1], ids=["tap-outer", "tap-inner"])
@pytest.mark.parametrize("cin,cout,dg,stride,dil,hw", [(32, 48, 2, 1, 1, (11, 13)), (64, 64, 4, 1, 2, (11, 13)), (16, 16, 1, 2, 1, (12, 9)), (256, 256, 4, 1, 1, (20, 18)),
                                                       (128, 200, 1, 1, 1, (15, 17))])
def test_forward_step_orders_vs_oracle(gpu, cin, cout, dg, stride, dil, hw, order):
    """fv2p_dcn_set_forward_order: the forward kernel in (tap, group, chunk) and in (group, chunk, tap) step order, both against the float64
    oracle at 1e-4 on offsets that leave the image, several groups / chunks per group, stride and dilation, ragged Cout; and against each
    other to 1e-5 (same terms, another order of the sum)."""
    import fv2p_native
    torch.manual_seed(5)
    B, (H, W) = 2, hw
    pad = dil
    Ho, Wo = (H + 2 * pad - (dil * 2 + 1)) // stride + 1, (W + 2 * pad - (dil * 2 + 1)) // stride + 1
    x = torch.randn(B, cin, H, W)
    offset = torch.randn(B, dg * 18, Ho, Wo) * 1.5
    mask = torch.sigmoid(torch.randn(B, dg * 9, Ho, Wo))
    m = ModulatedDeformConv(cin, cout, 3, stride=stride, padding=pad, dilation=dil, deformable_groups=dg, bias=True).to(gpu)
    ref = dcn_oracle.modulated_deform_conv(x.double(), offset.double(), mask.double(), m.weight.detach().cpu().double(), m.bias.detach().cpu().double(),
                                           (stride, stride), (pad, pad), (dil, dil), dg)
    try:
        fv2p_native.call("fv2p_dcn_set_forward_order", order)
        with torch.no_grad():
            y = m(x.to(gpu), offset.to(gpu), mask.to(gpu))
        fv2p_native.call("fv2p_dcn_set_forward_order", 1 - order)
        with torch.no_grad():
            y_other = m(x.to(gpu), offset.to(gpu), mask.to(gpu))
    finally:
        fv2p_native.call("fv2p_dcn_set_forward_order", -1)
    assert rel(y, ref) < 1e-4
    assert rel(y, y_other) < 1e-5
    with pytest.raises(fv2p_native.Fv2pError):
        fv2p_native.call("fv2p_dcn_set_forward_order", 2)


@pytest.mark.parametrize("cin,cout,dg,per_chunk", [(32, 48, 2, 2), (16, 16, 1, 1), (64, 64, 4, 3)])
def test_backward_in_batch_chunks_vs_oracle(gpu, cin, cout, dg, per_chunk):
    """Both entry points cut a call into chunks of whole samples (32-bit addressing inside the kernels; at most 1.5 GiB of column
    gradients in the workspace - the reference's im2col_step loop, modulated_deform_conv_cuda.cu:85-118, 217-262).  With the cap
    lowered to `per_chunk` samples a batch of 5 runs as 2 + 2 + 1 / five single samples / 3 + 2: forward and all five gradients
    against the float64 oracle at 1e-4; input, offset and mask gradients BIT-IDENTICAL to the unchunked call (per-sample work), the
    weight gradient within 1e-6 of it (the chunks are added in ascending order) and bit-identical from run to run."""
    import fv2p_native
    torch.manual_seed(per_chunk)
    B, H, W = 5, 11, 13
    x = torch.randn(B, cin, H, W)
    offset = torch.randn(B, dg * 18, H, W) * 1.5
    mask = torch.sigmoid(torch.randn(B, dg * 9, H, W))
    m = ModulatedDeformConv(cin, cout, 3, stride=1, padding=1, deformable_groups=dg, bias=True).to(gpu)
    g = torch.randn(B, cout, H, W)

    def run():
        m.zero_grad(set_to_none=True)
        gx, go, gm = (t.clone().to(gpu).requires_grad_(True) for t in (x, offset, mask))
        y = m(gx, go, gm)
        y.backward(g.to(gpu))
        return [t.detach().clone() for t in (y, gx.grad, go.grad, gm.grad, m.weight.grad, m.bias.grad)]

    whole = run()
    colg_per_sample = H * W * 9 * cin * 4
    ws_whole = int(fv2p_native.lib().fv2p_dcn_backward_ws_bytes(B, H, W, H, W, cin, cout, 3, 3, dg))
    try:
        fv2p_native.call("fv2p_dcn_set_colg_cap", per_chunk * colg_per_sample)
        assert int(fv2p_native.lib().fv2p_dcn_backward_ws_bytes(B, H, W, H, W, cin, cout, 3, 3, dg)) < ws_whole   # one chunk's workspace
        parts, again = run(), run()
    finally:
        fv2p_native.call("fv2p_dcn_set_colg_cap", 0)
    cx, co, cm = (t.clone().double().requires_grad_(True) for t in (x, offset, mask))
    w, b = m.weight.detach().cpu().double().requires_grad_(True), m.bias.detach().cpu().double().requires_grad_(True)
    ref = dcn_oracle.modulated_deform_conv(cx, co, cm, w, b, (1, 1), (1, 1), (1, 1), dg)
    ref.backward(g.double())
    for got, want in zip(parts, (ref, cx.grad, co.grad, cm.grad, w.grad, b.grad)):
        assert rel(got, want) < 1e-4
    for k in range(4):
        assert torch.equal(parts[k], whole[k]), k
    assert rel(parts[4], whole[4]) < 1e-6
    for a, c in zip(parts, again):
        assert torch.equal(a, c)
    with pytest.raises(fv2p_native.Fv2pError):
        fv2p_native.call("fv2p_dcn_set_colg_cap", -1)


def test_mgaf_block_shapes_and_pack(gpu):
    """MdeformConvBlock as DCNBEVBackbone builds it (dcn_bev_backbone.py:56-62), small spatial size."""
    blk = MdeformConvBlock(128, 128, deformable_groups=1).to(gpu)
    x = torch.randn(2, 128, 25, 22, device=gpu, requires_grad=True)
    y = blk(x)
    # offsets are zero-initialised and mask = sigmoid(0) = 0.5 -> 0.5 * conv2d (bias is frozen but still added)
    ref = 0.5 * F.conv2d(x.detach().cpu(), blk.conv_adaption.weight.detach().cpu(), None, padding=1) \
        + blk.conv_adaption.bias.detach().cpu().view(1, -1, 1, 1)
    assert rel(y, ref) < 1e-4
    y.sum().backward()
    assert x.grad is not None and blk.conv_offset_mask.weight.grad is not None and blk.conv_adaption.bias.grad is None
    pack = ModulatedDeformConvPack(16, 16, 3, stride=1, padding=1, deformable_groups=1).to(gpu)
    assert pack(torch.randn(1, 16, 8, 8, device=gpu)).shape == (1, 16, 8, 8)
    with pytest.raises(Exception):
        blk.cpu()(torch.randn(1, 128, 8, 8))  # "Not implemented on the CPU", as the reference dispatcher


def test_backward_ragged_tail_chunk_on_a_large_map(gpu):
    """A SMALLER tail chunk can need MORE weight-gradient splits than the full chunk the workspace is carved for (pix_per_block is
    rounded up to 16: 100 x 88 px, three samples -> 83 splits, the two-sample tail -> 85; round-5 ADVICE).  The 11 x 13 px maps of
    test_backward_in_batch_chunks_vs_oracle cannot show it (their split count is capped by npix / 64).  Batch 5 as 3 + 2 on a
    100 x 88 map: input / offset / mask gradients bit-identical to the unchunked call, the weight gradient within 1e-6 of it and
    against the float64 oracle at 1e-4, the same bits on a second run, and a canary behind the workspace stays untouched."""
    import fv2p_native
    torch.manual_seed(11)
    B, H, W, cin, cout, dg = 5, 100, 88, 32, 32, 1
    x = torch.randn(B, cin, H, W)
    offset = torch.randn(B, dg * 18, H, W) * 1.5
    near = (offset - offset.round()).abs() < 1e-3
    offset = torch.where(near, offset + 4e-3, offset)
    mask = torch.sigmoid(torch.randn(B, dg * 9, H, W))
    m = ModulatedDeformConv(cin, cout, 3, stride=1, padding=1, deformable_groups=dg, bias=False).to(gpu)
    g = torch.randn(B, cout, H, W)

    def run():
        m.zero_grad(set_to_none=True)
        gx, go, gm = (t.clone().to(gpu).requires_grad_(True) for t in (x, offset, mask))
        y = m(gx, go, gm)
        y.backward(g.to(gpu))
        return [t.detach().clone() for t in (y, gx.grad, go.grad, gm.grad, m.weight.grad)]

    whole = run()
    try:
        fv2p_native.call("fv2p_dcn_set_colg_cap", 3 * H * W * 9 * cin * 4)
        parts, again = run(), run()
        # the C-ABI call itself on a workspace of exactly the advertised size with a canary behind it
        need = int(fv2p_native.lib().fv2p_dcn_backward_ws_bytes(B, H, W, H, W, cin, cout, 3, 3, dg))
        ws = torch.zeros(need + 4096, dtype=torch.uint8, device=gpu)
        ws[need:] = 0xA5
        xh = x.to(gpu).permute(0, 2, 3, 1).contiguous()
        dyh = g.to(gpu).permute(0, 2, 3, 1).contiguous()
        wt = m.weight.detach().permute(2, 3, 1, 0).reshape(9, cin, cout).contiguous()
        dx, doff, dmask, dwt = torch.empty_like(xh), torch.empty_like(offset, device=gpu), torch.empty_like(mask, device=gpu), torch.empty_like(wt)
        fv2p_native.call("fv2p_dcn_backward", xh, wt, offset.to(gpu), mask.to(gpu), dyh, B, H, W, cin, cout, H, W, 3, 3, 1, 1, 1, 1, 1, 1, dg,
                         dx, doff, dmask, dwt, ws, need, fv2p_native.stream())
        torch.cuda.synchronize()
        assert bool((ws[need:] == 0xA5).all()), "dcn_backward wrote behind its workspace"
        assert torch.equal(dx.permute(0, 3, 1, 2), parts[1])
    finally:
        fv2p_native.call("fv2p_dcn_set_colg_cap", 0)
    for k in range(4):
        assert torch.equal(parts[k], whole[k]), k
    assert rel(parts[4], whole[4]) < 1e-6
    for a, c in zip(parts, again):
        assert torch.equal(a, c)
    cx, co, cm = (t.clone().double().requires_grad_(True) for t in (x, offset, mask))
    w = m.weight.detach().cpu().double().requires_grad_(True)
    ref = dcn_oracle.modulated_deform_conv(cx, co, cm, w, m.bias.detach().cpu().double(), (1, 1), (1, 1), (1, 1), dg)   # (bias=False freezes the bias, it does not remove it)
    ref.backward(g.double())
    assert rel(parts[0], ref) < 1e-4 and rel(parts[4], w.grad) < 1e-4 and rel(parts[1], cx.grad) < 1e-4


@pytest.mark.parametrize("cin,dg", [(256, 4), (128, 1)])
def test_mgaf_full_size_maps_forward_backward_vs_oracle(gpu, cin, dg):
    """BASELINE configs[3] shapes: the head's feature adaption [B, 256, 200, 176] with four deformable groups
    (center_af_head_single.py:44-49) and the first backbone block [B, 128, 200, 176] with one (dcn_bev_backbone.py:56-62).
    One sample against the float64 oracle (forward and all gradients 1e-4), then batch 4: every sample of the batched
    call must equal its single-sample result bit for bit (pixels of different samples never share a tile's arithmetic)."""
    torch.manual_seed(cin)
    H, W = 200, 176
    x = torch.randn(1, cin, H, W)
    offset = torch.randn(1, dg * 18, H, W) * 1.2
    # Bilinear sampling has a kink at integer positions.  Of the 2.5 M offsets here a handful fall within 1e-5 of an integer, where
    # fp32 (the reference's and this kernel's arithmetic: 136 - 4e-6 == 136) and the float64 oracle take different one-sided
    # derivatives; move those off the kink so that both sides differentiate the same piece.
    near = (offset - offset.round()).abs() < 1e-3
    offset = torch.where(near, offset + 4e-3, offset)
    mask = torch.sigmoid(torch.randn(1, dg * 9, H, W))
    m = ModulatedDeformConv(cin, cin, 3, stride=1, padding=1, deformable_groups=dg, bias=False).to(gpu)
    gx, go, gm = (t.clone().to(gpu).requires_grad_(True) for t in (x, offset, mask))
    y = m(gx, go, gm)
    cx, co, cm = (t.clone().double().requires_grad_(True) for t in (x, offset, mask))
    w, b = m.weight.detach().cpu().double().requires_grad_(True), m.bias.detach().cpu().double()
    ref = dcn_oracle.modulated_deform_conv(cx, co, cm, w, b, (1, 1), (1, 1), (1, 1), dg)
    assert rel(y, ref) < 1e-4
    g = torch.randn(ref.shape)
    y.backward(g.to(gpu))
    ref.backward(g.double())
    assert rel(gx.grad, cx.grad) < 1e-4
    assert rel(go.grad, co.grad) < 1e-4
    assert rel(gm.grad, cm.grad) < 1e-4
    assert rel(m.weight.grad, w.grad) < 1e-4
    del ref, cx, co, cm
    with torch.no_grad():
        xs = torch.cat([x, x.flip(3), x * 0.5, x.roll(7, 2)]).to(gpu)
        os_ = torch.cat([offset, offset * 0.5, -offset, offset.roll(3, 3)]).to(gpu)
        ms = torch.cat([mask, mask.flip(2), mask, mask * 0.9]).to(gpu)
        y4 = m(xs, os_, ms)
        for i in range(4):
            assert torch.equal(y4[i:i + 1], m(xs[i:i + 1].contiguous(), os_[i:i + 1].contiguous(), ms[i:i + 1].contiguous())), i


@pytest.mark.parametrize("cin,cout,dg,hw,scale", [(128, 128, 1, (100, 88), 1.2), (64, 64, 4, (40, 36), 6.0), (16, 32, 1, (24, 20), 0.0)])
def test_backward_is_bit_identical_run_to_run(gpu, cin, cout, dg, hw, scale):
    """The reference scatters grad_input with atomicAdd (modulated_deform_im2col_cuda.cuh:196-254), so its sums change with the
    scheduling; here grad_input is a gather over per-pixel sample lists in ascending sample order and the weight gradient a fixed-order
    sum of partial tiles.  Offsets of scale 6 pile up to ~100 samples on border-adjacent pixels (lists longer than one wave); scale 0
    puts every sample exactly on a pixel (all four corner weights degenerate to 1, 0, 0, 0)."""
    torch.manual_seed(11)
    H, W = hw
    x = torch.randn(2, cin, H, W, device=gpu)
    offset = torch.randn(2, dg * 18, H, W, device=gpu) * scale
    mask = torch.sigmoid(torch.randn(2, dg * 9, H, W, device=gpu))
    m = ModulatedDeformConv(cin, cout, 3, stride=1, padding=1, deformable_groups=dg, bias=True).to(gpu)
    g = torch.randn(2, cout, H, W, device=gpu)
    runs = []
    for _ in range(3):
        gx, go, gm = (t.clone().requires_grad_(True) for t in (x, offset, mask))
        m.zero_grad()
        m(gx, go, gm).backward(g)
        runs.append([t.clone() for t in (gx.grad, go.grad, gm.grad, m.weight.grad, m.bias.grad)])
        torch.randn(1 << 22, device=gpu).sum()   # other work in between: different scheduling
    for r in runs[1:]:
        for a, b in zip(runs[0], r):
            assert torch.equal(a, b)
    # and the sums are the right ones
    cx, co, cm = (t.detach().cpu().double().requires_grad_(True) for t in (x, offset, mask))
    w = m.weight.detach().cpu().double().requires_grad_(True)
    ref = dcn_oracle.modulated_deform_conv(cx, co, cm, w, m.bias.detach().cpu().double(), (1, 1), (1, 1), (1, 1), dg)
    ref.backward(g.cpu().double())
    assert rel(runs[0][0], cx.grad) < 1e-4 and rel(runs[0][3], w.grad) < 1e-4
    if scale > 0:      # at integer positions the one-sided derivative w.r.t. the offset is implementation-defined (test above)
        assert rel(runs[0][1], co.grad) < 1e-4
    assert rel(runs[0][2], cm.grad) < 1e-4


@pytest.mark.parametrize("cin,cout,dg", [(64, 64, 4), (32, 48, 1)])
def test_reference_structure_baseline_is_the_same_op(gpu, cin, cout, dg):
    """fv2p_harness/refstyle.py:dcn_im2col_gemm - the `columns` + GEMM restatement bench.py --workload mgaf times as `baseline` -
    computes the same layer as the fused kernels (forward and all gradients; 1e-3: its sampling positions pass through
    grid_sample's normalised coordinates)."""
    from fv2p_harness import refstyle
    torch.manual_seed(5)
    B, H, W = 2, 20, 24
    x = torch.randn(B, cin, H, W, device=gpu)
    offset = torch.randn(B, dg * 18, H, W, device=gpu) * 1.5
    offset = torch.where((offset - offset.round()).abs() < 1e-2, offset + 3e-2, offset)   # keep off the kinks of the bilinear kernel
    mask = torch.sigmoid(torch.randn(B, dg * 9, H, W, device=gpu))
    m = ModulatedDeformConv(cin, cout, 3, stride=1, padding=1, deformable_groups=dg, bias=True).to(gpu)
    g = torch.randn(B, cout, H, W, device=gpu)
    res = []
    for structure in (False, True):
        gx, go, gm = (t.clone().requires_grad_(True) for t in (x, offset, mask))
        m.zero_grad()
        if structure:
            with refstyle.reference_dcn_structure():
                y = m(gx, go, gm)
        else:
            y = m(gx, go, gm)
        y.backward(g)
        res.append([y.detach(), gx.grad, go.grad, gm.grad, m.weight.grad.clone(), m.bias.grad.clone()])
    for a, b in zip(*res):
        assert rel(b, a) < 1e-3


def test_channels_last_input_keeps_its_format(gpu):
    """A channels-last activation is the kernels' own layout: the layer reads it as it is, returns a channels-last output and a
    channels-last grad_input (as torch's own convolutions do), with the bits of the contiguous call."""
    torch.manual_seed(11)
    B, cin, cout, dg, H, W = 2, 32, 48, 2, 14, 10
    m = ModulatedDeformConv(cin, cout, 3, stride=1, padding=1, deformable_groups=dg, bias=True).to(gpu)
    x = torch.randn(B, cin, H, W, device=gpu)
    offset = torch.randn(B, dg * 18, H, W, device=gpu) * 0.7
    mask = torch.sigmoid(torch.randn(B, dg * 9, H, W, device=gpu))
    g = torch.randn(B, cout, H, W, device=gpu)
    xa = x.clone().requires_grad_(True)
    ya = m(xa, offset, mask)
    ya.backward(g)
    xb = x.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yb = m(xb, offset, mask)
    assert yb.is_contiguous(memory_format=torch.channels_last) and not yb.is_contiguous()
    yb.backward(g.contiguous(memory_format=torch.channels_last))
    assert torch.equal(ya, yb) and torch.equal(xa.grad, xb.grad)
    assert ya.is_contiguous() and xa.grad.is_contiguous()
