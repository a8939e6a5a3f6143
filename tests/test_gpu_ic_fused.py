"""Inverse consistency in one launch (option ic_fused, convex.hip::k_ic_persistent): the same fields, bit for bit, as one launch per
iteration (reference: convex_adam_utils.py:114-129) -- on several shapes, repeated (a stale read between iterations would show up as a
mismatch in some repetition), with the device-side fallback forced (ic_fused = 2), and against the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEV)


@pytest.fixture(scope="module")
def U():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from convexadam_amd import convex_adam_utils
    return convex_adam_utils


@pytest.fixture(scope="module")
def L():
    from convexadam_amd import _lib
    return _lib.lib()


def run(U, L, a, b, it, mode):
    old = L.cvx_get_option(b"ic_fused")
    L.cvx_set_option(b"ic_fused", mode)
    try:
        o1, o2 = U.inverse_consistency(a[None], b[None], iter=it)
        torch.cuda.synchronize()
        return o1[0].clone(), o2[0].clone()
    finally:
        L.cvx_set_option(b"ic_fused", old)


@pytest.mark.parametrize("shape,it,amp", [((26, 32, 37), 15, 0.15), ((26, 32, 37), 2, 0.4), ((16, 16, 16), 15, 0.3), ((5, 7, 9), 3, 0.5),
                                          ((37, 32, 37), 15, 0.2), ((40, 48, 40), 14, 0.1), ((26, 32, 37), 1, 0.2)])
def test_one_launch_equals_one_launch_per_iteration(U, L, orc, shape, it, amp):
    rng = np.random.default_rng(sum(shape) + it)
    a = (amp * rng.standard_normal((3,) + shape)).astype(np.float32)
    b = (amp * rng.standard_normal((3,) + shape)).astype(np.float32)
    da, db = dev(a), dev(b)
    r1, r2 = run(U, L, da, db, it, 0)
    for rep in range(10):
        o1, o2 = run(U, L, da, db, it, 1)
        assert torch.equal(o1, r1) and torch.equal(o2, r2), "repetition %d differs" % rep
    f1, f2 = run(U, L, da, db, it, 2)                       # placement check forced to fail: the fallback kernel computes everything
    assert torch.equal(f1, r1) and torch.equal(f2, r2)
    q1, q2 = orc.inverse_consistency(a, b, it)
    assert np.array_equal(r1.cpu().numpy(), q1) and np.array_equal(r2.cpu().numpy(), q2)


def test_concurrent_streams(U, L):
    """Four registrations' worth of inverse consistency on four streams at once: each takes another XCD, none waits for the other."""
    rng = np.random.default_rng(0)
    shape = (26, 32, 37)
    fields = [(dev((0.2 * rng.standard_normal((3,) + shape)).astype(np.float32)), dev((0.2 * rng.standard_normal((3,) + shape)).astype(np.float32))) for _ in range(4)]
    ref = [run(U, L, a, b, 15, 0) for a, b in fields]
    old = L.cvx_get_option(b"ic_fused")
    L.cvx_set_option(b"ic_fused", 1)
    try:
        streams = [torch.cuda.Stream(DEV) for _ in range(4)]
        outs = [None] * 4
        for rep in range(5):
            for k, (a, b) in enumerate(fields):
                with torch.cuda.stream(streams[k]):
                    outs[k] = U.inverse_consistency(a[None], b[None], iter=15)
            torch.cuda.synchronize()
            for k in range(4):
                assert torch.equal(outs[k][0][0], ref[k][0]) and torch.equal(outs[k][1][0], ref[k][1])
    finally:
        L.cvx_set_option(b"ic_fused", old)
