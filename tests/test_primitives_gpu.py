"""GPU: scan / radix-sort primitives vs numpy (bit-exact integer work)."""
import numpy as np
import pytest
import torch

import fv2p_native as nat

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 5, 255, 256, 2048, 2049, 100003, 1 << 20])
def test_exclusive_scan(gpu, n):
    rng = np.random.default_rng(n)
    a = rng.integers(0, 7, size=n).astype(np.int32)
    d = torch.from_numpy(a).to(gpu)
    out = torch.empty_like(d)
    tot = torch.zeros(1, dtype=torch.int32, device=gpu)
    ws = nat.workspace(nat.lib().fv2p_scan_ws_bytes(n), gpu)
    nat.call("fv2p_exclusive_scan_i32", d, out, n, tot, ws, ws.numel(), nat.stream())
    ref = np.concatenate([[0], np.cumsum(a)[:-1]]).astype(np.int32)
    assert np.array_equal(out.cpu().numpy(), ref)
    assert int(tot.item()) == int(a.sum())
    # in place
    nat.call("fv2p_exclusive_scan_i32", d, d, n, None, ws, ws.numel(), nat.stream())
    assert np.array_equal(d.cpu().numpy(), ref)


@pytest.mark.parametrize("n,lo,hi", [(2, 0, 8), (1000, 0, 13), (4096, 24, 40), (70001, 0, 31), (300000, 24, 39), (1 << 20, 0, 40)])
def test_radix_sort_stable_on_bit_range(gpu, n, lo, hi):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 1 << 62, size=n, dtype=np.int64).astype(np.uint64)
    if n > 100:
        keys[: n // 3] = keys[0]  # many ties -> exercises stability
    d = torch.from_numpy(keys.view(np.int64)).to(gpu)
    tmp = torch.empty_like(d)
    ws = nat.workspace(nat.lib().fv2p_radix_sort_ws_bytes(n), gpu)
    nat.call("fv2p_radix_sort_u64", d, tmp, n, lo, hi, ws, ws.numel(), nat.stream())
    field = (keys >> np.uint64(lo)) & np.uint64((1 << (hi - lo)) - 1)
    order = np.argsort(field, kind="stable")
    assert np.array_equal(d.cpu().numpy().view(np.uint64), keys[order])
