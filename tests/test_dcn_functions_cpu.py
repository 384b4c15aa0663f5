"""CPU check of the deformable-convolution autograd Functions (no kernel runs): the positional arguments they hand to the `DCN`
entry points are the ones the reference's bindings take (DeformableConvolutionV2PyTorch/src/vision.cpp:6-12 ->
src/modulated_deform_conv.h:10-86, src/deform_conv.h), for ints and pairs, and the gradients come back in input order."""
import pytest
import torch

from pcdet.ops.DeformableConvolutionV2PyTorch import DCN
from pcdet.ops.DeformableConvolutionV2PyTorch.functions import DeformConvFunction, DeformRoIPoolingFunction, ModulatedDeformConvFunction
from pcdet.ops.DeformableConvolutionV2PyTorch.modules import DeformConvPack, ModulatedDeformConvPack


@pytest.fixture
def calls(monkeypatch):
    seen = []

    def fake(name, n_out):
        def f(*args):
            seen.append((name, args))
            x = args[0] if name.endswith("forward") else None
            if name == "modulated_deform_conv_forward" or name == "deform_conv_forward":
                return x.new_ones(x.shape[0], args[1].shape[0], x.shape[2], x.shape[3])
            if name == "modulated_deform_conv_backward":      # input, weight, bias, offset, mask, grad_output, ...
                return [torch.full_like(args[0], 1.), torch.full_like(args[3], 2.), torch.full_like(args[4], 3.), torch.full_like(args[1], 4.), torch.full_like(args[2], 5.)]
            if name == "deform_conv_backward":                # input, weight, bias, offset, grad_output, ...
                return [torch.full_like(args[0], 1.), torch.full_like(args[3], 2.), torch.full_like(args[1], 4.), torch.full_like(args[2], 5.)]
            if name == "deform_psroi_pooling_forward":
                out = x.new_ones(args[1].shape[0], args[5], args[7], args[7])
                return out, out.clone()
            if name == "deform_psroi_pooling_backward":       # out_grad, input, bbox, trans, top_count, ...
                return torch.full_like(args[1], 1.), torch.full_like(args[3], 2.)
        return f
    for n in ("modulated_deform_conv_forward", "modulated_deform_conv_backward", "deform_conv_forward", "deform_conv_backward",
              "deform_psroi_pooling_forward", "deform_psroi_pooling_backward"):
        monkeypatch.setattr(DCN, n, fake(n, 0))
    return seen


def test_modulated_function_arguments_and_gradient_order(calls):
    x = torch.randn(2, 4, 6, 5, requires_grad=True)
    w, b = torch.randn(8, 4, 3, 3, requires_grad=True), torch.randn(8, requires_grad=True)
    off, mask = torch.randn(2, 18, 6, 5, requires_grad=True), torch.rand(2, 9, 6, 5, requires_grad=True)
    y = ModulatedDeformConvFunction.apply(x, off, mask, w, b, (1, 2), 1, (2, 1), 1, 1, 64)
    name, args = calls[-1]
    assert name == "modulated_deform_conv_forward" and args[5:] == (3, 3, 1, 2, 1, 1, 2, 1, 1, 1, 64)
    assert all(a is t for a, t in zip(args[:5], (x, w, b, off, mask)))
    y.sum().backward()
    name, args = calls[-1]
    assert name == "modulated_deform_conv_backward" and args[6:] == (3, 3, 1, 2, 1, 1, 2, 1, 1, 1, 64) and args[5].shape == y.shape
    assert [float(t.grad.flatten()[0]) for t in (x, off, mask, w, b)] == [1., 2., 3., 4., 5.]


def test_dcnv1_function_arguments_and_gradient_order(calls):
    x = torch.randn(2, 4, 6, 5, requires_grad=True)
    w, b = torch.randn(8, 4, 3, 3, requires_grad=True), torch.randn(8, requires_grad=True)
    off = torch.randn(2, 18, 6, 5, requires_grad=True)
    y = DeformConvFunction.apply(x, off, w, b, 1, (1, 1), 1, 1, 2, 32)
    name, args = calls[-1]
    assert name == "deform_conv_forward" and args[4:] == (3, 3, 1, 1, 1, 1, 1, 1, 1, 2, 32) and all(a is t for a, t in zip(args[:4], (x, w, b, off)))
    y.sum().backward()
    name, args = calls[-1]
    assert name == "deform_conv_backward" and args[5:] == (3, 3, 1, 1, 1, 1, 1, 1, 1, 2, 32)
    assert [float(t.grad.flatten()[0]) for t in (x, off, w, b)] == [1., 2., 4., 5.]


def test_pack_modules_reach_the_functions(calls):
    """The *Pack modules predict their own offsets (and mask): zero-initialised conv_offset(_mask) -> zero offsets, mask 0.5."""
    x = torch.randn(1, 4, 5, 5)
    m2 = ModulatedDeformConvPack(4, 6, 3, stride=1, padding=1, deformable_groups=2)
    m2(x)
    name, args = calls[-1]
    assert name == "modulated_deform_conv_forward" and args[3].shape == (1, 36, 5, 5) and float(args[3].detach().abs().sum()) == 0
    assert args[4].shape == (1, 18, 5, 5) and torch.allclose(args[4], torch.full_like(args[4], 0.5)) and args[-2] == 2
    m1 = DeformConvPack(4, 6, 3, stride=1, padding=1, deformable_groups=1, bias=False)     # DCNv1 asserts bias == False (deform_conv.py:20)
    m1(x)
    name, args = calls[-1]
    assert name == "deform_conv_forward" and args[3].shape == (1, 18, 5, 5) and float(args[3].detach().abs().sum()) == 0


def test_psroi_function_arguments(calls):
    x = torch.randn(2, 4, 8, 8, requires_grad=True)
    rois = torch.tensor([[0., 0, 0, 8, 8]])
    trans = torch.zeros(1, 2, 3, 3, requires_grad=True)
    y = DeformRoIPoolingFunction.apply(x, rois, trans, 0.25, 3, 4, False, 1, None, 2, 0.1)
    name, args = calls[-1]      # (input, bbox, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size, sample_per_part, trans_std)
    assert name == "deform_psroi_pooling_forward" and args[3:] == (0, 0.25, 4, 1, 3, 3, 2, 0.1)
    y.sum().backward()
    name, args = calls[-1]
    assert name == "deform_psroi_pooling_backward" and args[5:] == (0, 0.25, 4, 1, 3, 3, 2, 0.1)
    assert float(x.grad.flatten()[0]) == 1. and float(trans.grad.flatten()[0]) == 2.
