"""The FV2P step in the REFERENCE'S CALL STRUCTURE on this GPU — the baseline `north_star`'s ">= 2x the reference spconv+ops
forward+backward throughput" is measured against (bench.py --impl refstyle, and `vs_baseline` of the default run).

The reference's CUDA cannot be built here, so its structure is restated with the same kernels where the reference has a kernel of
its own and with torch ops where it composes them:

  * sparse conv forward / backward: one host read of indiceNum, then per kernel offset gather -> mm -> scatter-add (index_select,
    torch.mm, index_add_), centre offset of a submanifold conv as a plain GEMM — spconv_ops.h:260-457, functional.py:20-77;
    BatchNorm1d and ReLU as separate modules (spconv_backbone.py:23-27), no statistics in conv epilogues, weight gradients in line;
  * SparseConvTensor.dense(): zeros -> index scatter -> permute -> contiguous (structure.py:5-18, 57-66);
  * key-point sampling: the one-workgroup-per-sample kernel that re-reads its running distances every round
    (sampling_gpu.cu:100-216) — fv2p_furthest_point_sampling without the bucketed kernel's workspace — called in line;
  * grid set abstraction of the RoI head: QueryAndGroup -> grouped tensor (R, 3 + C, M, S) -> shared 1x1 convs -> max
    (pointnet2_modules.py:30-62), no fused kernel;
  * proposal layer: one full rotated NMS per sample, the NMS_POST_MAXSIZE cut afterwards (roi_head_template.py:60-85);
  * target assignment, RoI sampling and the first-stage losses: the tensor formulations (the reference's Python does this work
    op by op and per sample: axis_aligned_target_assigner.py:36-215, proposal_target_layer.py:92-217, anchor_head_template.py:98-206);
  * one stream, nothing prepared ahead.

What stays as in the default run, because the reference has the same thing or nothing comparable: the voxeliser (the reference
voxelises in DataLoader workers on the host, outside the step), the rulebook build (hash tables; the reference's dense-grid build is
not restated), 3-NN / interpolation / ball query / RoI pooling / point-in-box kernels (same algorithm per call), the dense 2-D
layers (MIOpen for both).  The figure is therefore a LOWER bound on the reference's step time in its own structure."""
import contextlib
import os

import torch

from pcdet.ops.spconv import functional as _fsp
from pcdet.ops.spconv import norm as _norm
from pcdet.ops.spconv import structure as _structure
from pcdet.ops.pointnet2.pointnet2_batch import fused as _pn2_fused
from pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as _pn2_stack
from pcdet.ops import _glue as G

from . import fv2p_model as _model


def _gather_mm_scatter(features, filters, rb, n_out, inverse, subm):
    """spconv_ops.h:260-362 in torch ops (autograd gives the gather / mm / scatter-add backward of :364-457)."""
    kvol = rb.kvol
    w = filters.reshape(kvol, filters.shape[-2], filters.shape[-1])
    pairs = rb.indice_pairs                                   # [K, 2, n_in], -1 padded
    num = rb.indice_pair_num.cpu().tolist()                   # the reference's indiceNum.to(CPU) (:271)
    src_side, dst_side = (1, 0) if inverse else (0, 1)
    centre = kvol // 2 if (subm and kvol % 2 == 1) else -1
    if centre >= 0:
        out = torch.mm(features, w[centre])                   # :300-303
    else:
        out = features.new_zeros((n_out, w.shape[-1]))
    for k in range(kvol):
        n = num[k]
        if n <= 0 or k == centre:
            continue
        src = pairs[k, src_side, :n].long()
        dst = pairs[k, dst_side, :n].long()
        out = out.index_add(0, dst, torch.mm(features.index_select(0, src), w[k]))
    return out


def _make_conv(inverse, subm):
    def call(features, filters, indice_pairs, indice_pair_num, num_activate_out):
        from pcdet.ops.spconv import ops
        rb = ops._rulebook_of(indice_pairs, indice_pair_num, features.shape[0], num_activate_out, inverse)
        return _gather_mm_scatter(features, filters, rb, num_activate_out, inverse, subm)
    return call


def _dense_torch(self, channels_first=True):
    out_shape = [self.batch_size] + list(self.spatial_shape) + [self.features.shape[1]]
    res = _structure.scatter_nd(self.indices.long(), self.features, out_shape)
    if not channels_first:
        return res
    ndim = len(self.spatial_shape)
    perm = list(range(0, ndim + 1))
    perm.insert(1, ndim + 1)
    return res.permute(*perm).contiguous()


def _fps_plain(saved, xyz, npoint):
    b, n, _ = xyz.shape
    idx = G.new(xyz, (b, npoint), torch.int32)
    running = G.new(xyz, (b, n), fill=1e10)
    G.run("fv2p_furthest_point_sampling", b, n, npoint, xyz, running, idx, None, 0)
    return idx


@contextlib.contextmanager
def reference_call_structure():
    saved = [(_fsp, "indice_conv", _fsp.indice_conv), (_fsp, "indice_subm_conv", _fsp.indice_subm_conv),
             (_fsp, "indice_inverse_conv", _fsp.indice_inverse_conv),
             (_norm, "fusable", _norm.fusable),
             (_structure.SparseConvTensor, "dense", _structure.SparseConvTensor.dense),
             (_pn2_fused, "supported", _pn2_fused.supported),
             (_pn2_stack, "furthest_point_sample", _pn2_stack.furthest_point_sample),
             (_model, "KERNEL_GLUE", _model.KERNEL_GLUE)]
    env = os.environ.get("FV2P_DEFER_WGRAD")
    plain_fps = G.autograd_op("FurthestPointSamplingPlain", _fps_plain)
    try:
        _fsp.indice_conv = _make_conv(False, False)
        _fsp.indice_subm_conv = _make_conv(False, True)
        _fsp.indice_inverse_conv = _make_conv(True, False)
        _norm.fusable = lambda *a, **k: False        # every BatchNorm1d / ReLU as its own torch module, no conv-epilogue statistics
        _structure.SparseConvTensor.dense = _dense_torch
        _pn2_fused.supported = lambda *a, **k: False
        _pn2_stack.furthest_point_sample = plain_fps.apply
        _model.KERNEL_GLUE = False
        os.environ["FV2P_DEFER_WGRAD"] = "0"
        yield
    finally:
        for obj, name, val in saved:
            setattr(obj, name, val)
        if env is None:
            os.environ.pop("FV2P_DEFER_WGRAD", None)
        else:
            os.environ["FV2P_DEFER_WGRAD"] = env


def inline_config(cfg):
    """cfg with every stream arrangement off: one stream, key points sampled where the decoder asks for them."""
    return type("Cfg", (cfg,), {"dense_branch_stream": False, "point_branch_stream": False, "key_stream": False, "dense_wgrad_stream": False})
