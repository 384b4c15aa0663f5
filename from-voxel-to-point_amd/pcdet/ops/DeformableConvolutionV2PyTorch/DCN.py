"""`DCN` — the extension module of the reference's DeformableConvolutionV2PyTorch (src/vision.cpp:6-12) as a Python
shim over libfv2p_ops: same six function names and positional arguments."""
import torch

import fv2p_native as _nat


def _geom(input, weight, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group):
    _nat.require_cuda(input)  # raises on a CPU tensor, as the reference dispatcher does ("Not implemented on the CPU", modulated_deform_conv.h:43)
    if group != 1:
        raise NotImplementedError("fv2p DCN: groups > 1 is not supported (no reference config uses it)")
    B, C, H, W = input.shape
    Cout = weight.shape[0]
    assert weight.shape[1] == C and weight.shape[2] == kernel_h and weight.shape[3] == kernel_w, "input / kernel shape mismatch"
    Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) // stride_h + 1
    Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) // stride_w + 1
    return (B, H, W, C, Cout, Ho, Wo, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, deformable_group)


def _channels_last(t):
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()


def _to_nhwc(t):
    """[B, C, H, W] (any strides) -> contiguous fp32 [B, H, W, C]: a view for a channels-last tensor, the library's tiled transpose for a
    contiguous one (torch's generic strided copy moves these 70 - 150 MB per DCN layer at a fraction of the bandwidth)."""
    t = t.float()
    v = t.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v
    if t.is_contiguous() and t.is_cuda and t.numel() > 0:
        B, C, H, W = t.shape
        out = torch.empty((B, H, W, C), dtype=torch.float32, device=t.device)
        with _nat.device_guard(t.device):
            _nat.call("fv2p_transpose_batched", t, B, C, H * W, out, _nat.stream())
        return out
    return v.contiguous()


def _to_nchw(t_nhwc):
    """contiguous fp32 [B, H, W, C] -> contiguous [B, C, H, W] (what the reference returns, modulated_deform_conv_cuda.cu:118)."""
    B, H, W, C = t_nhwc.shape
    if not t_nhwc.is_cuda or t_nhwc.numel() == 0:
        return t_nhwc.permute(0, 3, 1, 2).contiguous()
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=t_nhwc.device)
    with _nat.device_guard(t_nhwc.device):
        _nat.call("fv2p_transpose_batched", t_nhwc, B, H * W, C, out, _nat.stream())
    return out


def _wt(weight):
    """[Cout, Cin, kh, kw] -> [kh*kw][Cin][Cout]: the backward kernels' layout (output channels contiguous)."""
    cout, cin, kh, kw = weight.shape
    return weight.permute(2, 3, 1, 0).reshape(kh * kw, cin, cout).contiguous()


def _wt_oc(weight):
    """[Cout, Cin, kh, kw] -> [kh*kw][Cout][Cin]: the forward kernel's layout (input channels contiguous)."""
    cout, cin, kh, kw = weight.shape
    return weight.permute(2, 3, 0, 1).reshape(kh * kw, cout, cin).contiguous()


def _forward_nhwc(x, weight, bias, offset, mask, g):
    """x NHWC contiguous fp32 -> y [B*Ho*Wo, Cout] (the C-ABI call)."""
    B, H, W, C, Cout, Ho, Wo = g[:7]
    y = torch.empty((B * Ho * Wo, Cout), dtype=torch.float32, device=x.device)
    with _nat.device_guard(x.device):
        _nat.call("fv2p_dcn_forward", x, _wt_oc(weight.float()), bias.float().contiguous() if bias is not None else None,
                  offset.float().contiguous(), mask.float().contiguous(), *g, y, _nat.stream())
    return y


def modulated_deform_conv_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                                  dilation_h, dilation_w, group, deformable_group, im2col_step):
    """-> output [B, Cout, Ho, Wo] (contiguous NCHW, as modulated_deform_conv_cuda.cu:118). im2col_step is accepted and
    irrelevant: the forward has no columns buffer to chunk.  A channels-last `input` is taken as it is and the output is channels-last
    too (the kernels' own layout: no copy on either side); grad_input follows the input's format likewise."""
    g = _geom(input, weight, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group)
    B, H, W, C, Cout, Ho, Wo = g[:7]
    x = _to_nhwc(input)
    y = _forward_nhwc(x, weight, bias, offset, mask, g).view(B, Ho, Wo, Cout)
    if _channels_last(input):   # as torch's own ops: a channels-last input gets a channels-last output (here: the kernel's layout, no copy)
        return y.permute(0, 3, 1, 2).to(input.dtype)
    return _to_nchw(y).to(input.dtype)


def modulated_deform_conv_backward(input, weight, bias, offset, mask, grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h,
                                   pad_w, dilation_h, dilation_w, group, deformable_group, im2col_step):
    """-> [grad_input, grad_offset, grad_mask, grad_weight, grad_bias] (modulated_deform_conv_cuda.cu:127-280).
    No float atomics anywhere: the five gradients are bit-identical from run to run (include/fv2p_ops.h, A13)."""
    g = _geom(input, weight, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group)
    B, H, W, C, Cout, Ho, Wo = g[:7]
    dev = input.device
    x = _to_nhwc(input)
    dy = _to_nhwc(grad_output).view(B * Ho * Wo, Cout)
    wt = _wt(weight.float())
    pad = (-Cout) % 4            # the kernels read output channels four at a time: pad with zero columns
    if pad:
        dy = torch.nn.functional.pad(dy, (0, pad))
        wt = torch.nn.functional.pad(wt, (0, pad))
        g = g[:4] + (Cout + pad,) + g[5:]
    dx = torch.empty_like(x)
    doff = torch.empty_like(offset, dtype=torch.float32).contiguous()
    dmask = torch.empty_like(mask, dtype=torch.float32).contiguous()
    dwt = torch.empty((kernel_h * kernel_w, C, Cout + pad), dtype=torch.float32, device=dev)
    with _nat.device_guard(dev):
        nb = _nat.lib().fv2p_dcn_backward_ws_bytes(B, H, W, Ho, Wo, C, Cout + pad, kernel_h, kernel_w, deformable_group)
        ws = _nat.workspace(nb, dev)
        _nat.call("fv2p_dcn_backward", x, wt, offset.float().contiguous(), mask.float().contiguous(), dy, *g, dx, doff,
                  dmask, dwt, ws, ws.numel(), _nat.stream())
    grad_input = dx.permute(0, 3, 1, 2) if _channels_last(input) else _to_nchw(dx)
    grad_weight = dwt[:, :, :Cout].reshape(kernel_h, kernel_w, C, Cout).permute(3, 2, 0, 1).contiguous()
    grad_bias = dy[:, :Cout].sum(dim=0)
    return [grad_input, doff, dmask, grad_weight, grad_bias]


def deform_conv_forward(input, weight, bias, offset, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                        group, deformable_group, im2col_step):
    """DCNv1 = DCNv2 with an all-ones modulation mask (deform_im2col_cuda.cuh:127-190 vs modulated_*:127-194)."""
    B, _, Ho, Wo = offset.shape
    mask = torch.ones((B, deformable_group * kernel_h * kernel_w, Ho, Wo), dtype=torch.float32, device=input.device)
    return modulated_deform_conv_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                                         dilation_h, dilation_w, group, deformable_group, im2col_step)


def deform_conv_backward(input, weight, bias, offset, grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h,
                         dilation_w, group, deformable_group, im2col_step):
    """-> [grad_input, grad_offset, grad_weight, grad_bias]."""
    B, _, Ho, Wo = offset.shape
    mask = torch.ones((B, deformable_group * kernel_h * kernel_w, Ho, Wo), dtype=torch.float32, device=input.device)
    gi, go, _, gw, gb = modulated_deform_conv_backward(input, weight, bias, offset, mask, grad_output, kernel_h, kernel_w, stride_h,
                                                       stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group,
                                                       im2col_step)
    return [gi, go, gw, gb]


def _ps_geom(input, bbox, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size, sample_per_part, trans_std):
    _nat.require_cuda(input, bbox, *(() if no_trans else (trans,)))  # as the reference dispatcher (deform_psroi_pooling.h: "Not implemented on the CPU")
    B, C, H, W = input.shape
    assert C == output_dim, "input channels and output channels must equal"   # deform_psroi_pooling_cuda.cu:291
    classes = 1 if no_trans else trans.shape[1] // 2
    return (B, C, H, W, bbox.shape[0], int(no_trans), float(spatial_scale), output_dim, group_size, pooled_size, part_size,
            sample_per_part, float(trans_std), classes)


def deform_psroi_pooling_forward(input, bbox, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size,
                                 sample_per_part, trans_std):
    """-> (output, top_count), both [num_rois, output_dim, pooled, pooled] (deform_psroi_pooling_cuda.cu:264-341).
    bbox [num_rois, 5] = (batch index, x1, y1, x2, y2); trans [>= num_rois, 2 * num_classes, part, part] (ignored with no_trans)."""
    g = _ps_geom(input, bbox, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size, sample_per_part, trans_std)
    x = input.float().contiguous()
    out = torch.empty((bbox.shape[0], output_dim, pooled_size, pooled_size), dtype=torch.float32, device=input.device)
    top_count = torch.zeros_like(out)
    with _nat.device_guard(input.device):
        _nat.call("fv2p_deform_psroi_pool_forward", x, bbox.float().contiguous(), None if no_trans else trans.float().contiguous(), *g,
                  out, top_count, _nat.stream())
    return out.to(input.dtype), top_count.to(input.dtype)


def deform_psroi_pooling_backward(out_grad, input, bbox, trans, top_count, no_trans, spatial_scale, output_dim, group_size, pooled_size,
                                  part_size, sample_per_part, trans_std):
    """-> (input_grad, trans_grad) (deform_psroi_pooling_cuda.cu:343-418); trans_grad has trans's shape."""
    g = _ps_geom(input, bbox, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size, sample_per_part, trans_std)
    x = input.float().contiguous()
    dx = torch.zeros_like(x)
    dtrans = torch.zeros_like(trans, dtype=torch.float32).contiguous()
    with _nat.device_guard(input.device):
        _nat.call("fv2p_deform_psroi_pool_backward", out_grad.float().contiguous(), x, bbox.float().contiguous(),
                  None if no_trans else trans.float().contiguous(), top_count.float().contiguous(), *g, dx,
                  None if no_trans else dtrans, _nat.stream())
    return dx.to(input.dtype), dtrans.to(trans.dtype)
