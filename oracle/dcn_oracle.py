"""ORACLE — test infrastructure only.

DCNv2 restated with explicit 4-tap bilinear sampling in torch CPU (autograd provides every gradient):
  sample validity + taps   modulated_deform_im2col_cuda.cuh:24-54 (mdmcn_im2col_bilinear), :170-182
  column layout / GEMM     modulated_deform_conv_cuda.cu:90-118
Parity pin: the reference's own self-checks (DeformableConvolutionV2PyTorch/test.py) are restated in
tests/test_dcn_gpu.py — zero offset + unit mask == nn.Conv2d (:69-110), identity kernel == input (:142-181),
gradcheck-style agreement of all five gradients (:351-435) — with torch.nn.functional.conv2d as the independent oracle."""
import torch


def modulated_deform_conv(x, offset, mask, weight, bias, stride, padding, dilation, deformable_groups):
    """x [B,C,H,W], offset [B,dg*2*K,Ho,Wo], mask [B,dg*K,Ho,Wo], weight [Cout,C,kh,kw] -> [B,Cout,Ho,Wo] (float64 ok)."""
    B, C, H, W = x.shape
    Cout, _, kh, kw = weight.shape
    sh, sw = stride
    ph, pw = padding
    dh, dw = dilation
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    K, dg, cpg = kh * kw, deformable_groups, C // deformable_groups
    ho = torch.arange(Ho, dtype=x.dtype).view(1, Ho, 1)
    wo = torch.arange(Wo, dtype=x.dtype).view(1, 1, Wo)
    out = x.new_zeros((B, Cout, Ho, Wo))
    for g in range(dg):
        xg = x[:, g * cpg:(g + 1) * cpg].reshape(B, cpg, H * W)
        for k in range(K):
            i, j = k // kw, k % kw
            h_im = ho * sh - ph + i * dh + offset[:, g * 2 * K + 2 * k]
            w_im = wo * sw - pw + j * dw + offset[:, g * 2 * K + 2 * k + 1]
            valid = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
            h_low, w_low = torch.floor(h_im), torch.floor(w_im)
            lh, lw = h_im - h_low, w_im - w_low
            hh, hw = 1 - lh, 1 - lw
            val = 0
            for (hq, wq, wt) in ((h_low, w_low, hh * hw), (h_low, w_low + 1, hh * lw), (h_low + 1, w_low, lh * hw),
                                 (h_low + 1, w_low + 1, lh * lw)):
                ok = valid & (hq >= 0) & (hq <= H - 1) & (wq >= 0) & (wq <= W - 1)
                idx = (hq.clamp(0, H - 1) * W + wq.clamp(0, W - 1)).long().view(B, 1, Ho * Wo).expand(B, cpg, Ho * Wo)
                v = torch.gather(xg, 2, idx).view(B, cpg, Ho, Wo)
                val = val + (wt * ok.to(x.dtype)).unsqueeze(1) * v
            col = val * mask[:, g * K + k].unsqueeze(1)
            out = out + torch.einsum("bchw,oc->bohw", col, weight[:, g * cpg:(g + 1) * cpg, i, j])
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out
