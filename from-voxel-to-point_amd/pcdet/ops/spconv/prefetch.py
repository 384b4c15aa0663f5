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
import torch

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
        recipe.append((key, src, [int(v) for v in rb.spatial_shape], rb.geom))
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
    out = {}
    for key, src, shape, geom in recipe:
        ksize, stride, padding, dilation, out_padding, subm, transpose = geom
        ind = root if src is None else out[src].outids
        out[key] = ops.build_rulebook(ind, batch_size, shape, list(ksize), list(stride), list(padding), list(dilation),
                                      list(out_padding), subm, transpose)
        out[key].prefetched = True
        if pair_lists and subm:  # compacted pair lists: the weight gradient of submanifold convs splits its work by them
            out[key].wgrad_pairs()   # (measured: a win for subm rulebooks, a loss for the sparser strided ones)
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
