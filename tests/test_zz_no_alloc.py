"""cora_hip_radix_sort_noalloc (include/cora_hip.h): the sort entry point that allocates nothing -- its tile histograms
live in a static device array -- must give exactly what cora_hip_radix_sort gives (bit-exact: integer keys, stable
order), on two calls in a row, the second with a smaller input, and must refuse inputs above its stated maximum."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sort(fn, hi, lo, val, bit_lo, bit_hi):
    """Runs one of the two entry points on DEVICE arrays (torch tensors are only the device memory)."""
    import torch
    from cora_amd import hip
    n = len(hi)
    d = lambda a, dt: torch.from_numpy(a.view(dt)).cuda() if a is not None else None      # noqa: E731
    t = lambda a: torch.empty_like(a) if a is not None else None                          # noqa: E731
    dh, dl, dv = d(hi, np.int64), d(lo, np.int64), d(val, np.int32)
    th, tl, tv = t(dh), t(dl), t(dv)
    torch.cuda.synchronize()
    p = lambda a: C.c_void_p(a.data_ptr()) if a is not None else None                     # noqa: E731
    hip._check(fn(p(dh), p(dl), p(dv), p(th), p(tl), p(tv), n, bit_lo, bit_hi))
    torch.cuda.synchronize()
    out = lambda a, dt: a.cpu().numpy().view(dt) if a is not None else None               # noqa: E731
    return out(dh, np.uint64), out(dl, np.uint64), out(dv, np.uint32)


@pytest.mark.parametrize("shape", ["key", "key_val", "key128_val"])
def test_noalloc_sort_equals_the_allocating_one(shape):
    from cora_amd import hip
    hip.init(0)
    L = hip.lib()
    rs = np.random.RandomState(11)
    # two calls in a row on the shared static scratch, the second smaller (stale histogram words of the first call
    # must not leak into it); sizes straddle tile (4096 / 8192) and chunk (64 tiles) boundaries
    for n, bit_lo, bit_hi in ((3_000_001, 0, 64), (70_000, 16, 48)):
        hi = rs.randint(0, 1 << 62, size=n).astype(np.uint64) | (rs.randint(0, 4, size=n).astype(np.uint64) << np.uint64(62))
        hi[rs.randint(0, n, size=n // 3)] = hi[0]                          # many equal keys: stability shows in val
        lo = rs.randint(0, 1 << 62, size=n).astype(np.uint64) if shape == "key128_val" else None
        val = np.arange(n, dtype=np.uint32) if shape != "key" else None
        hb = bit_hi + (64 if lo is not None else 0)
        want = _sort(L.cora_hip_radix_sort, hi.copy(), None if lo is None else lo.copy(), None if val is None else val.copy(), bit_lo, hb)
        got = _sort(L.cora_hip_radix_sort_noalloc, hi.copy(), None if lo is None else lo.copy(), None if val is None else val.copy(), bit_lo, hb)
        for w, g in zip(want, got):
            assert (w is None) == (g is None)
            if w is not None:
                assert np.array_equal(w, g)
        # ... and both are what numpy says: a stable sort on the chosen key bits (the allocating entry point shares the kernels,
        # so equality with it alone would not show a fault they have in common)
        def field(a, lo_bit, hi_bit):
            w = hi_bit - lo_bit
            return (a >> np.uint64(lo_bit)) & np.uint64((1 << w) - 1 if w < 64 else 0xFFFFFFFFFFFFFFFF)
        if lo is None:
            order = np.argsort(field(hi, bit_lo, bit_hi), kind="stable")
        else:                                                                # logical key hi:lo, bits [bit_lo, 64 + bit_hi): bit_lo lies in lo
            order = np.lexsort((field(lo, bit_lo, 64), field(hi, 0, bit_hi)))   # (lexsort is stable; last key is the primary one)
        assert np.array_equal(got[0], hi[order])
        if lo is not None:
            assert np.array_equal(got[1], lo[order])
        if val is not None:
            assert np.array_equal(got[2], val[order])


def test_noalloc_sort_refuses_inputs_above_its_maximum():
    import torch
    from cora_amd import hip
    hip.init(0)
    L = hip.lib()
    nmax = L.cora_hip_radix_sort_noalloc_max()
    assert nmax == 1 << 24
    a = torch.zeros(8, dtype=torch.int64, device="cuda")       # never touched: the size check comes first
    rc = L.cora_hip_radix_sort_noalloc(C.c_void_p(a.data_ptr()), None, None, C.c_void_p(a.data_ptr()), None, None, nmax + 1, 0, 64)
    assert rc == 19 and b"at most" in L.cora_hip_last_error()
    # the maximum itself is accepted
    n = int(nmax)
    k = torch.randint(0, 1 << 40, (n,), dtype=torch.int64, device="cuda")
    t = torch.empty_like(k)
    want = torch.sort(k).values
    hip._check(L.cora_hip_radix_sort_noalloc(C.c_void_p(k.data_ptr()), None, None, C.c_void_p(t.data_ptr()), None, None, n, 0, 40))
    assert torch.equal(k, want)
