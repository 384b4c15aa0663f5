"""Host side of the sparse-conv hot path: same function names / argument meaning as the reference's
spconv/ops.py:20-260, implemented on the C ABI of libfv2p_ops (include/fv2p_ops.h).

A rulebook is held as a `Rulebook`: two dense neighbour tables (tab_in [K,n_in], tab_out [K,n_out])
that the fused HIP conv kernels consume directly, plus — only when somebody asks — the reference's
pair-list format (`indice_pairs [K,2,n_in]`, -1 padded; canonical order = ascending input row).
"""
import numpy as np
import torch

import fv2p_native as _nat


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    out = []
    for i in range(len(input_size)):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
        out.append(1 if kernel_size[i] == -1 else size)
    return out


def get_deconv_output_size(input_size, kernel_size, stride, padding, dilation, output_padding):
    out = []
    for i in range(len(input_size)):
        if kernel_size[i] == -1:
            raise ValueError("deconv don't support kernel_size < 0")
        out.append((input_size[i] - 1) * stride[i] - 2 * padding[i] + kernel_size[i] + output_padding[i])
    return out


def _as_list(v, ndim):
    return list(v) if isinstance(v, (list, tuple)) else [v] * ndim


def _pad3(v, fill):
    v = [int(x) for x in v]
    return [fill] * (3 - len(v)) + v


def alloc_table(kvol, n, device):
    """[kvol, n] int32 neighbour table whose allocation continues with room for the conv kernels' tiling plan
    (include/fv2p_ops.h: fv2p_conv_plan_ints / FV2P_TAB_PLANNED)."""
    extra = int(_nat.lib().fv2p_conv_plan_ints(int(n))) if device.type == "cuda" else 0
    flat = torch.empty((kvol * n + extra,), dtype=torch.int32, device=device)
    return flat[:kvol * n].view(kvol, n)


_PLAN_CHANNELS = (64, 128)   # source channel counts whose conv kernel (conv_rows_ksplit) takes a tiling plan
PLAN_FROM_REQUEST = 3        # a table gets its plan when the N-th FORWARD conv asks for one (1: at once — tests, ahead-of-time builders)
TAB_FLIP, TAB_PLANNED = 1, 2


class Rulebook(object):
    """Neighbour tables of one (indice_key) rulebook.

    Behaves like the reference's cached 5-tuple `(outids, indices, indice_pairs, indice_pair_num,
    spatial_shape)` (conv.py:180-183) for unpacking / indexing; `indice_pairs` is materialised lazily."""

    def __init__(self, outids, indices, tab_in, tab_out, indice_num, spatial_shape, kvol, subm):
        self.outids, self.indices = outids, indices
        self.tab_in, self.tab_out = tab_in, tab_out  # tab_out None => symmetric subm: tab_out[k] == tab_in[K-1-k]
        self._num = indice_num
        self.spatial_shape = spatial_shape
        self.kvol, self.subm = kvol, subm
        self.n_in, self.n_out = int(indices.shape[0]), int(outids.shape[0])
        self._pairs = None
        self._wpairs = None   # compacted pair lists for the weight gradient (== _pairs once that exists)
        self._perm_in = None  # strided conv: input rows grouped by parity class (tile order of the backward-data conv)

    # -- tables as the kernels want them: (table, flag word: TAB_FLIP | TAB_PLANNED)
    def out_table(self, c_src=None):
        if self.tab_out is None:
            return (self.tab_in, TAB_FLIP | self._plan("tab_in", c_src, True))   # pairs per row are the same under the flip
        return (self.tab_out, self._plan("tab_out", c_src, True))

    def in_table(self, c_src=None):
        return (self.tab_in, self._plan("tab_in", c_src, False))

    def build_plan(self):
        """Builds the tiling plan of a submanifold rulebook now (input pipelines call it off the training stream)."""
        if self.subm and "tab_in" not in self.__dict__.get("_planned", {}):
            self.__dict__.setdefault("_plan_requests", {})["tab_in"] = PLAN_FROM_REQUEST
            self._plan("tab_in", _PLAN_CHANNELS[0], True)

    def _plan(self, which, c_src, forward):
        """TAB_PLANNED once the table carries a tiling plan; built when the PLAN_FROM_REQUEST-th forward conv with `c_src` source
        channels that can use one asks, if the table was allocated with room behind it (alloc_table)."""
        done = self.__dict__.setdefault("_planned", {})
        if which in done:
            return done[which]
        if c_src not in _PLAN_CHANNELS or not self.subm or not forward:
            return 0       # strided rulebooks serve one forward and one backward conv: a plan (55 - 70 us to build) cannot pay for itself
        # ... and a submanifold one saves 10 - 18 us per conv (profiles/r03_microbench_conv_kitti.txt): the two residual blocks of a
        # stage put four forward and four backward convs on one table — built at the third, the plan serves two + four of them (net
        # ~25 - 50 us per table when built in line, all of it when an input pipeline builds it ahead, prefetch.py); the two convs per
        # table of the plain backbone never pay for one (measured: 1.64 -> 2.04 ms per step with plans from the second conv on).
        # Backward convs use a plan that is there and never cause one.
        uses = self.__dict__.setdefault("_plan_requests", {})
        uses[which] = uses.get(which, 0) + 1
        if uses[which] < PLAN_FROM_REQUEST:
            return 0
        tab = getattr(self, which)
        flag = 0
        if tab is not None and tab.is_cuda and tab.numel() > 0 and tab.is_contiguous():
            kvol, n = tab.shape
            extra = int(_nat.lib().fv2p_conv_plan_ints(n))
            room = tab.untyped_storage().nbytes() // 4 - (tab.storage_offset() + kvol * n)
            if room >= extra:
                with _nat.device_guard(tab.device):
                    ws = _nat.workspace(_nat.lib().fv2p_conv_plan_ws_bytes(n), tab.device)
                    _nat.call("fv2p_conv_plan_build", tab, kvol, n, ws, ws.numel(), _nat.stream())
                flag = TAB_PLANNED
        done[which] = flag
        return flag

    @property
    def indice_pair_num(self):
        """indiceNum [K] of the reference, counted on first use (the fused kernels never need it)."""
        if self._num is None:
            num = torch.empty((self.kvol,), dtype=torch.int32, device=self.tab_in.device)
            with _nat.device_guard(num.device):
                _nat.call("fv2p_rulebook_count", self.tab_in, self.n_in, self.kvol, num, _nat.stream())
            self._num = num
        return self._num

    @property
    def indice_pairs(self):
        if self._pairs is None:
            pairs = torch.empty((self.kvol, 2, self.n_in), dtype=torch.int32, device=self.tab_in.device)
            if self.n_in > 0:
                with _nat.device_guard(pairs.device):
                    nb = _nat.lib().fv2p_rulebook_pairs_ws_bytes(self.n_in, self.kvol)
                    ws = _nat.workspace(nb, pairs.device)
                    num = self._num
                    if num is None:
                        num = torch.empty((self.kvol,), dtype=torch.int32, device=pairs.device)
                    _nat.call("fv2p_rulebook_pairs", self.tab_in, self.n_in, self.kvol, 1, pairs, num, ws, ws.numel(), _nat.stream())
                    self._num = num
            pairs._fv2p_rulebook = self
            self._pairs = pairs
            self._wpairs = pairs
        return self._pairs

    def wgrad_pairs(self):
        """(pairs [K,2,n_in], pair_num [K]) for the pair-split weight gradient: the compacted lists without the -1
        padding pass (not handed out as `indice_pairs`).  None for empty rulebooks."""
        if self._wpairs is None and self.n_in > 0:
            dev = self.tab_in.device
            pairs = torch.empty((self.kvol, 2, self.n_in), dtype=torch.int32, device=dev)
            num = self._num if self._num is not None else torch.empty((self.kvol,), dtype=torch.int32, device=dev)
            with _nat.device_guard(dev):
                ws = _nat.workspace(_nat.lib().fv2p_rulebook_pairs_ws_bytes(self.n_in, self.kvol), dev)
                _nat.call("fv2p_rulebook_pairs", self.tab_in, self.n_in, self.kvol, 0, pairs, num, ws, ws.numel(), _nat.stream())
            self._num, self._wpairs = num, pairs
        return self._wpairs

    def pairs_for_wgrad(self, cin, cout):
        """(pairs, pair_num) for a conv's weight gradient, or (None, None): the table-driven weight gradient then runs.  Lists that exist
        (an input pipeline built them with the rulebook) are used; a submanifold rulebook serving 64 / 128-channel convs in training builds
        them on first use - two launches, ~20 us, against 131 -> 66 us per 128 -> 128 weight gradient and 4 such convs per table in the
        residual backbone (profiles/r05_microbench_conv_kitti.txt); above 65 536 rows the table-driven form is not slower."""
        if self._wpairs is None and self.subm and min(cin, cout) >= 64 and 0 < self.n_in <= 65536 and torch.is_grad_enabled():
            self.wgrad_pairs()
        if self._wpairs is not None and self._num is not None:
            return self._wpairs, self._num
        return None, None

    def _tuple(self):
        return (self.outids, self.indices, self.indice_pairs, self.indice_pair_num, self.spatial_shape)

    def __iter__(self):
        return iter(self._tuple())

    def __getitem__(self, i):
        return self._tuple()[i]

    def __len__(self):
        return 5


def build_rulebook(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1, out_padding=0,
                   subm=False, transpose=False):
    """Hashed rulebook build (fv2p_rulebook_begin/finish). Returns a `Rulebook`."""
    _nat.require_cuda(indices)
    ndim = indices.shape[1] - 1
    if ndim not in (2, 3, 4):
        raise NotImplementedError("fv2p rulebook supports 2-D, 3-D and 4-D sparse tensors")
    ksize, stride, padding = _as_list(ksize, ndim), _as_list(stride, ndim), _as_list(padding, ndim)
    dilation, out_padding = _as_list(dilation, ndim), _as_list(out_padding, ndim)
    for d, s in zip(dilation, stride):
        assert any([s == 1, d == 1]), "don't support this."
    spatial_shape = [int(s) for s in spatial_shape]
    if ndim == 4:
        return _build_rulebook_4d(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose)
    if subm:
        out_shape = spatial_shape
    elif transpose:
        out_shape = get_deconv_output_size(spatial_shape, ksize, stride, padding, dilation, out_padding)
    else:
        out_shape = get_conv_output_size(spatial_shape, ksize, stride, padding, dilation)
    if indices.dtype != torch.int32:
        indices = indices.int()
    indices = indices.contiguous()
    ind4 = indices
    if ndim == 2:  # embed (b, y, x) as (b, 0, y, x)
        ind4 = torch.cat([indices[:, :1], torch.zeros_like(indices[:, :1]), indices[:, 1:]], dim=1).contiguous()
    g = dict(in_shape=_pad3(spatial_shape, 1), out_shape=_pad3(out_shape, 1), ksize=_pad3(ksize, 1),
             stride=_pad3(stride, 1), padding=_pad3(padding, 0), dilation=_pad3(dilation, 1))
    kvol = int(np.prod(ksize))
    n_in = int(indices.shape[0])
    dev = indices.device
    symmetric = bool(subm) and all(k % 2 == 1 for k in ksize) and all(d == 1 for d in dilation)
    with _nat.device_guard(dev):
        lib = _nat.lib()
        import ctypes
        arr = lambda v: (ctypes.c_int * 3)(*v)
        ws_bytes = lib.fv2p_rulebook_ws_bytes_grid(n_in, int(batch_size), arr(g["out_shape"]), arr(g["ksize"]), arr(g["stride"]),
                                                   arr(g["dilation"]), int(subm), int(transpose))
        ws = _nat.workspace(ws_bytes, dev)
        n_out_host = ctypes.c_int64(0)
        geom = (g["in_shape"], g["out_shape"], g["ksize"], g["stride"], g["padding"], g["dilation"], int(subm), int(transpose))
        _nat.call("fv2p_rulebook_begin", ind4, n_in, int(batch_size), *geom, ctypes.addressof(n_out_host), ws, ws.numel(), _nat.stream())
        n_out = int(n_out_host.value)
        tab_in = alloc_table(kvol, n_in, dev)
        num = None
        if subm:
            outids4, tab_out = None, (None if symmetric else alloc_table(kvol, n_out, dev))
        else:
            outids4 = torch.empty((n_out, 4), dtype=torch.int32, device=dev)
            tab_out = alloc_table(kvol, n_out, dev)
        _nat.call("fv2p_rulebook_finish", ind4, n_in, int(batch_size), *geom, n_out, outids4, tab_in, tab_out, num, ws, ws.numel(),
                  _nat.stream())
    if subm:
        outids = indices
    elif ndim == 2:
        outids = outids4[:, [0, 2, 3]].contiguous()
    else:
        outids = outids4
    rb = Rulebook(outids, indices, tab_in, tab_out, num, spatial_shape, kvol, bool(subm))
    rb.out_spatial_shape = out_shape
    rb.geom = (tuple(ksize), tuple(stride), tuple(padding), tuple(dilation), tuple(out_padding), bool(subm), bool(transpose))
    rb.batch_size = int(batch_size)
    return rb


def nd_tables(indices, batch_size, in_shape, out_shape, ksize, stride, padding, dilation, subm):
    """Neighbour tables of an N-D rulebook from sorted cell keys (torch sort / unique / searchsorted on the tensors' device):
    the definition spconv_ops.h:27-140 + geometry.h:25-84 reduce to when stride or dilation is 1 — output cell o and kernel
    offset k (row-major index, last dimension fastest) pair with input cell p iff o * stride = p + padding - k * dilation,
    submanifold: stride 1, padding = ksize // 2 (spconv_ops.h:76-80), outputs = the active inputs.  Output rows of a regular
    conv are in ascending (batch, cell) order, the order the reference's GPU path gets from torch::_unique.
    -> (outids [M, 1 + N] i32, tab_in [K, n_in] i32: output row fed by input i through offset k or -1,
        tab_out [K, M] i32: input row feeding output o through offset k or -1).
    Since round 6 the 4-D rulebooks of device tensors come from the hashed HIP build as well (`_native_tables_4d`,
    fv2p_rulebook4d_begin / _finish); this formulation stays as the definition the CPU tests pin against the restatement
    (tests/test_rulebook_nd_oracle.py) and the GPU test holds the kernels to (tests/test_spconv4d_gpu.py), and serves host tensors."""
    dev = indices.device
    nd = len(in_shape)
    n = int(indices.shape[0])
    lng = lambda v: torch.tensor([int(x) for x in v], dtype=torch.int64, device=dev)
    if subm:
        stride, padding = [1] * nd, [k // 2 for k in ksize]
    k_t, s_t, p_t, d_t, o_t = lng(ksize), lng(stride), lng(padding), lng(dilation), lng(out_shape)
    kvol = int(np.prod(ksize))
    vol = int(np.prod(out_shape))
    offs = torch.stack(torch.meshgrid(*[torch.arange(int(k), device=dev) for k in ksize], indexing="ij"), dim=-1).reshape(kvol, nd)
    mult = lng([int(np.prod(out_shape[i + 1:])) for i in range(nd)])
    coords, b = indices[:, 1:].long(), indices[:, 0].long()
    numer = coords.unsqueeze(0) + p_t - offs.unsqueeze(1) * d_t                      # [K, n, nd] = o * stride
    o = torch.div(numer, s_t, rounding_mode="floor")
    valid = ((numer - o * s_t == 0) & (o >= 0) & (o < o_t)).all(dim=-1)                # [K, n]
    keys = torch.where(valid, (o * mult).sum(-1) + b.unsqueeze(0) * vol, torch.full((), -1, dtype=torch.int64, device=dev))
    rows = torch.arange(n, dtype=torch.int32, device=dev).unsqueeze(0).expand(kvol, n)
    if subm:
        in_keys, order = torch.sort((coords * mult).sum(-1) + b * vol)
        pos = torch.searchsorted(in_keys, keys.reshape(-1)).reshape(kvol, n).clamp_(max=max(n - 1, 0))
        hit = valid & (in_keys[pos] == keys) if n else valid
        tab_in = torch.where(hit, order[pos].int(), torch.full((), -1, dtype=torch.int32, device=dev)) if n else rows.clone()
        outids, m = indices, n
    else:
        uniq = torch.unique(keys[valid])                                               # sorted
        m = int(uniq.numel())
        pos = torch.searchsorted(uniq, keys.reshape(-1)).reshape(kvol, n)
        tab_in = torch.where(valid, pos.int(), torch.full((), -1, dtype=torch.int32, device=dev))
        cell = uniq % vol
        outids = torch.stack([uniq // vol] + [(cell // mult[i]) % o_t[i] for i in range(nd)], dim=1).int()
        hit = valid
    tab_out = torch.full((kvol, m), -1, dtype=torch.int32, device=dev)
    kk = torch.arange(kvol, device=dev).unsqueeze(1).expand(kvol, n)
    tab_out[kk[hit], tab_in[hit].long()] = rows[hit]
    return outids.contiguous(), tab_in.contiguous(), tab_out.contiguous()


def _native_tables_4d(indices, batch_size, in_shape, out_shape, ksize, stride, padding, dilation, subm):
    """The 4-D tables from the hashed builder (fv2p_rulebook4d_begin / _finish): same definition and row order as `nd_tables`."""
    import ctypes
    dev = indices.device
    n_in = int(indices.shape[0])
    kvol = int(np.prod(ksize))
    ints = lambda v: [int(x) for x in v]
    geom = (ints(in_shape), ints(out_shape), ints(ksize), ints(stride), ints(padding), ints(dilation), int(bool(subm)))
    with _nat.device_guard(dev):
        lib = _nat.lib()
        arr = lambda v: (ctypes.c_int * 4)(*v)
        ws = _nat.workspace(lib.fv2p_rulebook4d_ws_bytes(n_in, arr(geom[2]), arr(geom[3]), arr(geom[5]), geom[6]), dev)
        n_out_host = ctypes.c_int64(0)
        _nat.call("fv2p_rulebook4d_begin", indices, n_in, int(batch_size), *geom, ctypes.addressof(n_out_host), ws, ws.numel(), _nat.stream())
        n_out = int(n_out_host.value)
        tab_in = alloc_table(kvol, n_in, dev)
        tab_out = alloc_table(kvol, n_out, dev)
        outids = indices if subm else torch.empty((n_out, 5), dtype=torch.int32, device=dev)
        _nat.call("fv2p_rulebook4d_finish", indices, n_in, int(batch_size), *geom, n_out, None if subm else outids, tab_in, tab_out, None,
                  ws, ws.numel(), _nat.stream())
    return outids, tab_in, tab_out


def _build_rulebook_4d(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose):
    if transpose:
        raise NotImplementedError("4-D transposed sparse convolution: the reference has no SparseConvTranspose4d either (conv.py:233-480)")
    out_shape = spatial_shape if subm else get_conv_output_size(spatial_shape, ksize, stride, padding, dilation)
    if indices.dtype != torch.int32:
        indices = indices.int()
    indices = indices.contiguous()
    kvol = int(np.prod(ksize))
    if indices.is_cuda:
        outids, tab_in, tab_out = _native_tables_4d(indices, batch_size, spatial_shape, out_shape, ksize, stride, padding, dilation, subm)
    else:   # host tensors (the CPU tests of the table definition): the torch formulation
        outids, tab_in, tab_out = nd_tables(indices, batch_size, spatial_shape, out_shape, ksize, stride, padding, dilation, subm)
    rb = Rulebook(outids, indices, tab_in, tab_out, None, spatial_shape, kvol, bool(subm))
    rb.out_spatial_shape = out_shape
    rb.geom = (tuple(ksize), tuple(stride), tuple(padding), tuple(dilation), tuple(out_padding), bool(subm), False)
    rb.batch_size = int(batch_size)
    return rb


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1, out_padding=0,
                     subm=False, transpose=False, grid=None):
    """Reference signature (ops.py:46-105): returns (outids, indice_pairs [K,2,N], indice_pair_num [K]).
    `grid` (pre-allocated dense grid of the reference) is accepted and ignored."""
    rb = build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose)
    return rb.outids, rb.indice_pairs, rb.indice_pair_num


def _rulebook_of(indice_pairs, indice_pair_num, n_src_rows, num_activate_out, inverse):
    """Accepts a Rulebook, a pair tensor produced by this package, or a foreign pair tensor."""
    if isinstance(indice_pairs, Rulebook):
        return indice_pairs
    rb = getattr(indice_pairs, "_fv2p_rulebook", None)
    if rb is not None:
        return rb
    # foreign rulebook in the reference format: rebuild the tables on the device
    _nat.require_cuda(indice_pairs)
    pairs = indice_pairs.int().contiguous()
    num = indice_pair_num.to(device=pairs.device, dtype=torch.int32).contiguous()
    kvol, _, plen = pairs.shape
    n_in = n_src_rows if not inverse else num_activate_out
    n_out = num_activate_out if not inverse else n_src_rows
    tab_in = alloc_table(kvol, n_in, pairs.device)
    tab_out = alloc_table(kvol, n_out, pairs.device)
    with _nat.device_guard(pairs.device):
        _nat.call("fv2p_pairs_to_tables", pairs, num, kvol, plen, n_in, n_out, tab_in, tab_out, _nat.stream())
    rb = Rulebook.__new__(Rulebook)
    rb.outids = rb.indices = None
    rb.tab_in, rb.tab_out, rb._num = tab_in, tab_out, num
    rb.spatial_shape, rb.kvol, rb.subm, rb.n_in, rb.n_out, rb._pairs, rb._wpairs = None, kvol, False, n_in, n_out, pairs, pairs
    indice_pairs._fv2p_rulebook = rb
    return rb


def _conv_rows(src, weight, table, flip, n_dst, c_dst, transpose_w, bias=None):
    _nat.require_cuda(src, weight, table)
    half = src.dtype == torch.half
    if half:  # the reference binds *_half entry points (all.cc:36-51); computed here in fp32
        src, weight = src.float(), weight.float()
        bias = None if bias is None else bias.float()
    if src.dtype != torch.float32 or weight.dtype != torch.float32:
        raise NotImplementedError("sparse conv: float32 / float16 only")
    src = src.contiguous()
    w = weight.contiguous()
    kvol = table.shape[0]
    dst = torch.empty((n_dst, c_dst), dtype=torch.float32, device=src.device)
    with _nat.device_guard(src.device):
        _nat.call("fv2p_sparse_conv_rows", src, src.shape[0], src.shape[1], w, kvol, table, n_dst, c_dst, int(flip),
                  int(transpose_w), bias.contiguous() if bias is not None else None, dst, _nat.stream())
    return dst.half() if half else dst


def indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse=False, subm=False, bias=None):
    """out[o] = sum_k feat[i] W_k over the rulebook (reference ops.py:108-126 -> spconv_ops.h:260-362)."""
    rb = _rulebook_of(indice_pairs, indice_pair_num, features.shape[0], num_activate_out, inverse)
    cin, cout = filters.shape[-2], filters.shape[-1]
    w = filters.reshape(-1, cin, cout)
    table, flip = rb.in_table(cin) if inverse else rb.out_table(cin)
    return _conv_rows(features, w, table, flip, num_activate_out, cout, False, bias)


def fused_indice_conv(features, filters, bias, indice_pairs, indice_pair_num, num_activate_out, inverse, subm):
    """conv + bias in the kernel epilogue (reference ops.py:129-139 -> fused_spconv_ops.h:28-131)."""
    return indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse, subm, bias=bias)


def indice_conv_backward(features, filters, out_bp, indice_pairs, indice_pair_num, inverse=False, subm=False):
    """Returns [d_features, d_filters] (reference ops.py:142-157 -> spconv_ops.h:364-457)."""
    rb = _rulebook_of(indice_pairs, indice_pair_num, features.shape[0], out_bp.shape[0], inverse)
    cin, cout = filters.shape[-2], filters.shape[-1]
    half = features.dtype == torch.half
    f32 = lambda t: t.float() if half else t
    feats, w, g = f32(features).contiguous(), f32(filters).reshape(-1, cin, cout).contiguous(), f32(out_bp).contiguous()
    kvol = w.shape[0]
    # forward used table F (dst rows = outputs); its transpose-direction table B has dst rows = inputs
    (tab_f, flip_f), (tab_b, flip_b) = (rb.in_table(), rb.out_table(cout)) if inverse else (rb.out_table(), rb.in_table(cout))
    if cout in _PLAN_CHANNELS and cin % 64 == 0 and cin <= 128 and kvol > 1:
        # the K-split tile's shapes: W_k^T materialised once and read as a plain weight (csrc_torch/fv2p_torch.cpp has the measurement)
        din = _conv_rows(g, w.transpose(1, 2).contiguous(), tab_b, flip_b, features.shape[0], cin, False)
    else:
        din = _conv_rows(g, w, tab_b, flip_b, features.shape[0], cin, True)
    dw = torch.empty_like(w)
    with _nat.device_guard(feats.device):
        if rb._wpairs is not None and rb._num is not None:
            # pair lists already materialised: work split by pairs, no compaction prologue
            pairs, num = rb._wpairs, rb._num
            wsb = _nat.lib().fv2p_sparse_conv_wgrad_pairs_ws_bytes(pairs.shape[2], cin, cout, kvol)
            ws = _nat.workspace(wsb, feats.device)
            _nat.call("fv2p_sparse_conv_wgrad_pairs", feats, feats.shape[0], cin, g, g.shape[0], cout, pairs, num, kvol, pairs.shape[2],
                      1 if inverse else 0, dw, ws, ws.numel(), _nat.stream())
        else:
            wsb = _nat.lib().fv2p_sparse_conv_wgrad_ws_bytes(g.shape[0], cin, cout, kvol)
            ws = _nat.workspace(wsb, feats.device)
            # submanifold conv: the centre offset pairs every row with itself (spconv_ops.h:398-402 treats it un-gathered
            # too); the kernel gives that dense offset its own finely chunked pass
            centre = (kvol // 2) if (rb.subm and rb.tab_out is None and not inverse) else -1
            _nat.call("fv2p_sparse_conv_wgrad", feats, feats.shape[0], cin, g, tab_f, g.shape[0], cout, kvol, int(flip_f), centre, dw, ws,
                      ws.numel(), _nat.stream())
    dw = dw.reshape(filters.shape)
    if half:
        din, dw = din.half(), dw.half()
    return [din, dw]


# ---- max-pool / group over the same tables (A7; reference pool_ops.h:25-94, group_ops.h:29-291) ----
def indice_maxpool(features, indice_pairs, indice_pair_num, num_activate_out):
    rb = _rulebook_of(indice_pairs, indice_pair_num, features.shape[0], num_activate_out, False)
    table, flip = rb.out_table()
    return _table_maxpool(features, table, flip, num_activate_out)


def indice_maxpool_backward(features, out_features, out_bp, indice_pairs, indice_pair_num):
    rb = _rulebook_of(indice_pairs, indice_pair_num, features.shape[0], out_features.shape[0], False)
    table, flip = rb.in_table()
    return _table_maxpool_backward(features, out_features, out_bp, table)


def _table_maxpool(features, table, flip, n_out):
    _nat.require_cuda(features)
    half = features.dtype == torch.half
    f = features.float().contiguous()
    out = torch.empty((n_out, f.shape[1]), dtype=torch.float32, device=f.device)
    with _nat.device_guard(f.device):
        _nat.call("fv2p_sparse_maxpool_fwd", f, f.shape[0], f.shape[1], table, table.shape[0], n_out, int(flip), out, _nat.stream())
    return out.half() if half else out


def _table_maxpool_backward(features, out_features, out_bp, tab_in):
    half = features.dtype == torch.half
    f, o, g = features.float().contiguous(), out_features.float().contiguous(), out_bp.float().contiguous()
    din = torch.empty_like(f)
    with _nat.device_guard(f.device):
        _nat.call("fv2p_sparse_maxpool_bwd", f, o, g, f.shape[0], f.shape[1], tab_in, tab_in.shape[0], din, _nat.stream())
    return din.half() if half else din


def indice_group(features, indice_pairs, indice_pair_num, num_activate_out, inverse=False, subm=False):
    """[K, n_out, C] gather of neighbour features, zeros where no neighbour (reference ops.py:196-211)."""
    if features.dtype != torch.float32:
        raise NotImplementedError
    rb = _rulebook_of(indice_pairs, indice_pair_num, features.shape[0], num_activate_out, inverse)
    table, flip = rb.in_table() if inverse else rb.out_table()
    f = features.contiguous()
    out = torch.empty((table.shape[0], num_activate_out, f.shape[1]), dtype=torch.float32, device=f.device)
    with _nat.device_guard(f.device):
        _nat.call("fv2p_sparse_group_fwd", f, f.shape[0], f.shape[1], table, table.shape[0], num_activate_out, int(flip), out, _nat.stream())
    return out


def indice_group_backward(features, out_bp, indice_pairs, indice_pair_num, inverse=False, subm=False):
    """d_features[i] = sum_k out_bp[k, tab_in[k][i]] (reference ops.py:214-230)."""
    if features.dtype != torch.float32:
        raise NotImplementedError
    rb = _rulebook_of(indice_pairs, indice_pair_num, features.shape[0], out_bp.shape[1], inverse)
    table, flip = rb.out_table() if inverse else rb.in_table()
    g = out_bp.contiguous()
    din = torch.empty((features.shape[0], features.shape[1]), dtype=torch.float32, device=g.device)
    with _nat.device_guard(g.device):
        _nat.call("fv2p_sparse_group_bwd", g, g.shape[1], g.shape[2], table, table.shape[0], features.shape[0], int(flip), din, _nat.stream())
    return din
