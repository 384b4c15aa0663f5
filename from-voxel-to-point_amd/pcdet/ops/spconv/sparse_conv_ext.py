"""`pcdet.ops.spconv.sparse_conv_ext` — the 17 functions the reference binds with pybind11
(pcdet/ops/spconv/src/all.cc:22-71), same names, argument order and return conventions, implemented
over the C ABI of libfv2p_ops so third-party code written against the extension keeps working."""
import torch

from . import ops as _ops


def _pairs(indices, batchSize, outSpatialShape, spatialShape, ksize, stride, padding, dilation, outPadding, subM, transpose):
    outids, pairs, num = _ops.get_indice_pairs(indices, batchSize, list(spatialShape), list(ksize), list(stride), list(padding),
                                               list(dilation), list(outPadding), bool(subM), bool(transpose))
    return [outids, pairs, num]


def get_indice_pairs_2d(indices, batchSize, outSpatialShape, spatialShape, ksize, stride, padding, dilation, outPadding,
                        subM, transpose):
    return _pairs(indices, batchSize, outSpatialShape, spatialShape, ksize, stride, padding, dilation, outPadding, subM, transpose)


def get_indice_pairs_3d(indices, batchSize, outSpatialShape, spatialShape, ksize, stride, padding, dilation, outPadding,
                        subM, transpose):
    return _pairs(indices, batchSize, outSpatialShape, spatialShape, ksize, stride, padding, dilation, outPadding, subM, transpose)


def get_indice_pairs_4d(indices, batchSize, outSpatialShape, spatialShape, ksize, stride, padding, dilation, outPadding,
                        subM, transpose):
    return _pairs(indices, batchSize, outSpatialShape, spatialShape, ksize, stride, padding, dilation, outPadding, subM, transpose)


def get_indice_pairs_grid_2d(indices, gridOut, batchSize, outSpatialShape, spatialShape, ksize, stride, padding, dilation,
                             outPadding, subM, transpose):
    """The pre-allocated dense grid (spconv_ops.h:143-258) is unnecessary with the hashed builder and ignored."""
    return _pairs(indices, batchSize, outSpatialShape, spatialShape, ksize, stride, padding, dilation, outPadding, subM, transpose)


def get_indice_pairs_grid_3d(indices, gridOut, batchSize, outSpatialShape, spatialShape, ksize, stride, padding, dilation,
                             outPadding, subM, transpose):
    return _pairs(indices, batchSize, outSpatialShape, spatialShape, ksize, stride, padding, dilation, outPadding, subM, transpose)


def indice_conv_fp32(features, filters, indicePairs, indiceNum, numActOut, inverse, subM):
    return _ops.indice_conv(features, filters, indicePairs, indiceNum, numActOut, bool(inverse), bool(subM))


def indice_conv_backward_fp32(features, filters, outGrad, indicePairs, indiceNum, inverse, subM):
    return _ops.indice_conv_backward(features, filters, outGrad, indicePairs, indiceNum, bool(inverse), bool(subM))


def fused_indice_conv_fp32(features, filters, bias, indicePairs, indiceNum, numActOut, inverse, subM):
    return _ops.fused_indice_conv(features, filters, bias, indicePairs, indiceNum, numActOut, bool(inverse), bool(subM))


indice_conv_half = indice_conv_fp32
indice_conv_backward_half = indice_conv_backward_fp32
fused_indice_conv_half = fused_indice_conv_fp32


def indice_maxpool_fp32(features, indicePairs, indiceNum, numAct):
    return _ops.indice_maxpool(features, indicePairs, indiceNum, numAct)


def indice_maxpool_backward_fp32(features, outFeatures, outGrad, indicePairs, indiceNum):
    return _ops.indice_maxpool_backward(features, outFeatures, outGrad, indicePairs, indiceNum)


indice_maxpool_half = indice_maxpool_fp32
indice_maxpool_backward_half = indice_maxpool_backward_fp32


def indice_group_fp32(features, indicePairs, indiceNum, numActOut, inverse, subM):
    return _ops.indice_group(features, indicePairs, indiceNum, numActOut, bool(inverse), bool(subM))


def indice_group_backward_fp32(features, outGrad, indicePairs, indiceNum, inverse, subM):
    return _ops.indice_group_backward(features, outGrad, indicePairs, indiceNum, bool(inverse), bool(subM))
