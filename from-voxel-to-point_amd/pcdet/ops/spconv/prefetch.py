"""Rulebook prefetch: build every rulebook of a network pass ahead of time, off the training thread.

A rulebook depends on the voxel coordinates only, never on features, and a strided rulebook ends in a host
synchronisation (its output row count is a tensor shape; the reference synchronises at the same place,
spconv_ops.h:131-139).  Inside the forward pass those waits stall kernel submission; an input pipeline that already
knows the next batch's coordinates can do the whole chain earlier, on its own stream:

    recipe = spconv.rulebook_recipe(x.indice_dict, coords)        # once, after any forward pass of the model
    ...
    spconv.attach_rulebooks(next_coords, spconv.build_rulebooks(recipe, next_coords, batch_size))   # pipeline thread
    model(next_feats, next_coords)      # SparseConvTensor adopts the attached dict: every indice_key hits the cache

The cache lookup in SparseConvolution.forward stays what it is in the reference (conv.py:178-196): a hit on
`indice_key` reuses the rulebook.  A prefetched entry whose geometry differs from the module asking for it, or whose
input rows are not the tensor's, is ignored and rebuilt, so a stale recipe costs time, never correctness.
"""
import os

import torch

import fv2p_native as _nat

from . import ops


def rulebook_recipe(indice_dict, root_indices):
    """[(indice_key, source key or None, input spatial shape, geom)] in build order, from the dict of a finished pass."""
    recipe = []
    by_out = {}
    root = root_indices.data_ptr()
    for key, rb in indice_dict.items():
        if key is None or not hasattr(rb, "geom"):
            continue
        src_ptr = rb.indices.data_ptr()
        if src_ptr == root:
            src = None
        elif src_ptr in by_out:
            src = by_out[src_ptr]
        else:
            continue  # built from indices this pass did not produce: cannot be replayed from the root alone
        # eighth entry: the recorded pass gave this (submanifold) table a tiling plan -> replays build it with the rulebook, off the
        # training stream (ops.Rulebook.build_plan)
        recipe.append((key, src, [int(v) for v in rb.spatial_shape], tuple(rb.geom[:7]) + (bool(rb.__dict__.get("_planned", {}).get("tab_in")),)))
        if not rb.subm:
            by_out[rb.outids.data_ptr()] = key
    return recipe


def build_rulebooks(recipe, indices, batch_size, pair_lists=True):
    """Replays a recipe on new root indices; returns {indice_key: Rulebook}."""
    if indices.dtype != torch.int32:
        indices = indices.int()
    indices = indices.contiguous()
    # the rulebooks keep an alias of the root coordinates, not the tensor object they get attached to: attaching would
    # otherwise close a reference cycle (tensor -> dict -> Rulebook.indices -> tensor) and every batch's device memory
    # would wait for Python's cyclic collector (seen as one 90 ms step every ~300 steps)
    root = indices.detach()
    ext = _nat.torch_ext()
    if ext is not None and root.is_cuda and root.dim() == 2 and root.shape[1] == 4 and all(len(r[2]) == 3 for r in recipe):
        return _build_chain(ext, recipe, root, batch_size, pair_lists)
    out = {}
    for key, src, shape, geom in recipe:
        ksize, stride, padding, dilation, out_padding, subm, transpose = geom[:7]
        ind = root if src is None else out[src].outids
        out[key] = ops.build_rulebook(ind, batch_size, shape, list(ksize), list(stride), list(padding), list(dilation),
                                      list(out_padding), subm, transpose)
        out[key].prefetched = True
        if len(geom) > 7 and geom[7]:
            out[key].build_plan()
        if pair_lists and (subm or _few_offsets(ksize)):  # compacted pair lists: the weight gradient splits its work by them
            out[key].wgrad_pairs()   # (measured: a win for subm rulebooks and for kernels of < 8 offsets, a loss for other strided ones)
    return out


def _few_offsets(ksize):
    """Kernels like (3, 1, 1): the table-driven weight gradient launches chunks x offsets workgroups and starves with 3
    offsets (45 us for 15 k pairs at 64 -> 128); split by pair lists it takes 21 us."""
    k = 1
    for v in ksize:
        k *= int(v)
    return k < 8


def _build_chain(ext, recipe, root, batch_size, pair_lists):
    """The same chain through one call of the compiled binding, which runs it without the interpreter lock (a pipeline
    thread issuing ~60 small calls from Python trades the lock with the training thread ~60 times per batch)."""
    keys = [r[0] for r in recipe]
    specs, shapes = [], []
    for key, src, shape, geom in recipe:
        ksize, stride, padding, dilation, out_padding, subm, transpose = (list(g) if isinstance(g, tuple) else g for g in geom[:7])
        if subm:
            out_shape = list(shape)
        elif transpose:
            out_shape = ops.get_deconv_output_size(shape, ksize, stride, padding, dilation, out_padding)
        else:
            out_shape = ops.get_conv_output_size(shape, ksize, stride, padding, dilation)
        symmetric = bool(subm) and all(k % 2 == 1 for k in ksize) and all(d == 1 for d in dilation)
        specs.append((-1 if src is None else keys.index(src), [int(v) for v in shape], [int(v) for v in out_shape], ksize, stride, padding,
                      dilation, bool(subm), bool(transpose), symmetric, bool(pair_lists and (subm or _few_offsets(ksize)))))
        shapes.append(out_shape)
    with _nat.device_guard(root.device):
        res = ext.build_rulebook_chain(root, int(batch_size), specs)
    out = {}
    for (key, src, shape, geom), (outids, tab_in, tab_out, pairs, num, perm), out_shape in zip(recipe, res, shapes):
        ind = root if src is None else out[src].outids
        subm = geom[5]
        rb = ops.Rulebook(ind if subm else outids, ind, tab_in, tab_out, num, shape, int(tab_in.shape[0]), bool(subm))
        rb.out_spatial_shape = out_shape
        rb.geom, rb.batch_size, rb.prefetched = tuple(geom[:7]), int(batch_size), True
        rb._wpairs = pairs
        rb._perm_in = perm if os.environ.get("FV2P_DX_PERM", "1") != "0" else None   # strided conv: input rows by parity class = tile order of its backward-data conv
        if len(geom) > 7 and geom[7]:
            with _nat.device_guard(root.device):
                rb.build_plan()
        out[key] = rb
    return out


def attach_rulebooks(indices, rulebooks):
    """Marks `indices` (the int32 coordinate tensor a SparseConvTensor will be built from) with prefetched rulebooks."""
    indices._fv2p_indice_dict = rulebooks
    return indices


def geometry_matches(rb, module, indices):
    """False for a prefetched rulebook that was built for another conv geometry or for other input rows (a rebuilt
    upstream conv invalidates everything prefetched downstream of it).  Rulebooks cached by a pass itself keep the
    reference semantics: the key is trusted."""
    g = getattr(rb, "geom", None)
    if g is None or module.inverse or not getattr(rb, "prefetched", False):
        return True
    if rb.indices is not indices and (rb.indices.data_ptr() != indices.data_ptr() or rb.indices.shape != indices.shape):
        return False
    return (g[0] == tuple(module.kernel_size) and g[1] == tuple(module.stride) and g[2] == tuple(module.padding) and
            g[3] == tuple(module.dilation) and g[5] == bool(module.subm) and g[6] == bool(module.transposed))
