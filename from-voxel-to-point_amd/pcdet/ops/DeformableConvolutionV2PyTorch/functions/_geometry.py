"""The trailing positional arguments every convolution entry point of `DCN` takes, from a Function's loose arguments."""
from torch.nn.modules.utils import _pair


def conv_geometry(weight, stride, padding, dilation, groups, deformable_groups, im2col_step):
    """-> (kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, im2col_step),
    the order of src/vision.cpp's bindings; ints or pairs are accepted for stride / padding / dilation."""
    kh, kw = (int(v) for v in weight.shape[2:4])
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    return (kh, kw, sh, sw, ph, pw, dh, dw, groups, deformable_groups, im2col_step)
