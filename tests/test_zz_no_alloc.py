"""lgicp_nn_search_static (the entry point that allocates nothing) against lgicp_nn_search.

Device buffers of the caller are made with the HIP runtime the library itself is linked to
(ctypes on libamdhip64), so no second GPU runtime enters the process.  Two calls in a row on the
same GPU, the second with smaller clouds: the static scratch keeps the first call's contents and
the second call's results must not see them.  Integer / byte work: bit-exact."""
import ctypes as C

import numpy as np
import pytest

from tests import fixtures as F

pytestmark = pytest.mark.gpu


class _Hip:
    def __init__(self):
        self.rt = C.CDLL("libamdhip64.so.7")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipFree.argtypes = [C.c_void_p]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.rt.hipDeviceSynchronize.argtypes = []
        self.rt.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.live = []

    def free_bytes(self):
        free, total = C.c_size_t(), C.c_size_t()
        assert self.rt.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    def upload(self, a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), max(a.nbytes, 16)) == 0
        self.live.append(p)
        if a.nbytes:
            assert self.rt.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
        return p.value

    def empty(self, nbytes, fill=0xAB):
        p = C.c_void_p()
        assert self.rt.hipMalloc(C.byref(p), max(nbytes, 16)) == 0
        assert self.rt.hipMemset(p, fill, max(nbytes, 16)) == 0
        self.live.append(p)
        return p.value

    def download(self, ptr, dtype, n):
        out = np.empty(n, dtype)
        if out.nbytes:
            assert self.rt.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, 2) == 0
        return out

    def free_all(self):
        for p in self.live:
            self.rt.hipFree(p)
        self.live = []


def _cloud(rng, n, dup=0):
    pts = (rng.random((n, 3)) * 20 - 10).astype(np.float32)
    if dup:
        pts[-dup:] = pts[:dup]          # exact ties between targets: lowest index must win
    return pts


def _check(ctx, hip, T, src, tgt):
    ref_idx, ref_d2 = ctx.nn_search(T, src, tgt, 0)
    d_src, d_tgt = hip.upload(src), hip.upload(tgt)
    d_idx, d_d2 = hip.empty(4 * len(src)), hip.empty(4 * len(src))
    # the entry point's contract: no device allocation inside the call, none left behind
    before = hip.free_bytes()
    ctx.nn_search_static(T, d_src, len(src), d_tgt, len(tgt), d_idx, d_d2)
    assert hip.free_bytes() == before
    idx = hip.download(d_idx, np.int32, len(src))
    d2 = hip.download(d_d2, np.float32, len(src))
    assert np.array_equal(idx, ref_idx)
    assert np.array_equal(d2.view(np.uint32), ref_d2.view(np.uint32))


def test_static_entry_point_matches_nn_search_twice_in_a_row():
    from lamp_amd import capi
    ctx = capi.Context(init_method=capi.INIT_IDENTITY)
    hip = _Hip()
    try:
        rng = np.random.default_rng(20261111)
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = F.rot_zyx(0.02, -0.03, 0.05).astype(np.float32)
        T[:3, 3] = [0.1, -0.2, 0.05]
        # first call: large clouds (several query tiles, split target), ties among the targets
        _check(ctx, hip, T, _cloud(rng, 20000), _cloud(rng, 33000, dup=500))
        # second call, smaller in both clouds: stale points / keys of the first call sit right
        # behind the new ones in the static array
        _check(ctx, hip, T, _cloud(rng, 777), _cloud(rng, 1500, dup=20))
        # third: a single query and a target smaller than one tile
        _check(ctx, hip, np.eye(4, dtype=np.float32), _cloud(rng, 1), _cloud(rng, 3))
    finally:
        hip.free_all()
        ctx.close()


def test_static_entry_point_rejects_oversized_input():
    from lamp_amd import capi
    ctx = capi.Context(init_method=capi.INIT_IDENTITY)
    hip = _Hip()
    try:
        n = 65536 + 1
        d = hip.empty(12 * n)
        o = hip.empty(4 * n)
        with pytest.raises(capi.LgicpError, match="at most 65536 points"):
            ctx.nn_search_static(np.eye(4), d, n, d, 100, o, o)
        with pytest.raises(capi.LgicpError, match="at most 65536 points"):
            ctx.nn_search_static(np.eye(4), d, 100, d, n, o, o)
    finally:
        hip.free_all()
        ctx.close()
