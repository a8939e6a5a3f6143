"""Reproducing the bits of a reference run on a given host (opt-in).

Every operation of the hot path is restated in the reference's own evaluation order except a global mean whose
rounding depends on the reference's thread count (set_mean_threads below restates torch's sum for a given count) and two library
calls of the reference BUILD: `torch.exp` at MINDSSC (src/convexAdam/convex_adam_utils.py:63) and the `sqrt` inside `torch.optim.Adam`
(src/convexAdam/convex_adam_MIND.py:179).  On the CPU torch evaluates both with Intel MKL VML (vsExp / vsSqrt), whose results are
at most one ulp from this library's (`cvx_expf_f32`, IEEE sqrt), do not depend on the position in the tensor, and DO depend on the
host: MKL picks its code path by CPU model (a Xeon and an EPYC host of the same image give different tables).  Because the deviation
is a pure function of the argument it can be tabulated from torch itself, exhaustively:

    sqrt: 2 x 2^23 normal (exponent parity, mantissa) classes + 2^23 denormals  -> 6 MiB of 2-bit codes, < 1 s
    exp : every float32 argument with |x| in [2^-30, 128) -> 310 M two-bit entries, 74 MiB on the device, ~15 s

With both tables installed (and the thread count of the reference run) the pipeline's output equals the reference's bit for bit --
checked at the full benchmark size through 80 Adam iterations and on the masked large-motion configuration against fields captured
from the reference (tests/golden/fullsize.npz; tests/test_gpu_parity.py).  The default
(no tables) is within one ulp at those two sites, which the Adam loop amplifies to 1e-3 voxels after 80 iterations (the reference
differs by as much from itself across hosts).  Cost when installed: a dependent 2-bit lookup per exp and per sqrt.

    from convexadam_amd import reference_bits
    reference_bits.enable("cuda:0")          # tables of THIS host's torch
    ...
    reference_bits.disable()
"""
import numpy as np
import torch

from ._lib import check, lib, ptr, stream_ptr
from .convex_adam_utils import set_adam_sqrt_table


def _bits(f):
    return int(np.float32(f).view(np.uint32))


EXP_FIRST = _bits(2.0 ** -30)            # below: exp(x) == 1.0f for both
EXP_COUNT = _bits(128.0) - EXP_FIRST     # at and beyond 128: 0 for both
_CHUNK = 1 << 24
_exp_table = None


def device_expf(x):
    """The library's expf on a float32 device tensor (cvx_expf_f32)."""
    x = x.contiguous()
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib().cvx_expf_f32(ptr(x), ptr(out), x.numel(), stream_ptr(x.device)))
    return out


def build_exp_table(device="cuda", host_exp=None):
    """uint8 device tensor, 4 two-bit entries per byte: host torch.exp minus the device expf over the whole domain."""
    dev = torch.device(device)
    host_exp = host_exp or torch.exp
    tbl = torch.zeros((EXP_COUNT + 3) // 4, dtype=torch.uint8, device=dev)
    for off in range(0, EXP_COUNT, _CHUNK):
        n = min(_CHUNK, EXP_COUNT - off)
        keys = torch.arange(EXP_FIRST + off, EXP_FIRST + off + n, dtype=torch.int64).to(torch.int32)
        x = -keys.view(torch.float32)
        ref = host_exp(x).to(dev)                                                   # CPU: MKL vsExp
        own = device_expf(x.to(dev))
        d = ref.view(torch.int32) - own.view(torch.int32)
        if int((d.abs() > 1).sum()):
            raise RuntimeError("host exp differs from the library's expf by more than one ulp")
        code = torch.where(d == 1, 1, torch.where(d == -1, 2, 0)).to(torch.uint8)
        if n % 4:
            code = torch.cat([code, torch.zeros(4 - n % 4, dtype=torch.uint8, device=dev)])
        c = code.view(-1, 4)
        tbl[off // 4: off // 4 + c.shape[0]] = c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)
    return tbl


def _pack2(code):
    c = code.reshape(-1, 4)
    return (c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8)


def build_sqrt_table(host_sqrt=None):
    """6 MiB 2-bit table: the host's torch.sqrt relative to the IEEE root (0 equal, 1 one ulp above, 2 one ulp below) for the 2 x 2^23
    normal (exponent parity, mantissa) classes and the 2^23 denormals."""
    host_sqrt = host_sqrt or torch.sqrt

    def code(e):
        x = (np.arange(1 << 23, dtype=np.uint32) | np.uint32(e << 23)).view(np.float32)
        a = host_sqrt(torch.from_numpy(x.copy())).numpy()
        d = a.view(np.int32).astype(np.int64) - np.sqrt(x).view(np.int32).astype(np.int64)
        if d.min() < -1 or d.max() > 1:
            raise RuntimeError("host sqrt is more than one ulp from the IEEE root")
        return np.where(d == 1, 1, np.where(d == -1, 2, 0)).astype(np.uint8)
    return _pack2(np.concatenate([code(126), code(127), code(0)]))


def set_mind_exp_table(table=None, first=EXP_FIRST, count=EXP_COUNT, device="cuda"):
    """Installs a 2-bit table (numpy / torch uint8, any device) for the exp of MINDSSC; None restores the library's expf."""
    global _exp_table
    if table is None:
        check(lib().cvx_set_mind_exp_table(None, 0, 0))
        _exp_table = None
        return
    t = table if isinstance(table, torch.Tensor) else torch.from_numpy(np.array(table, dtype=np.uint8, order="C"))
    t = t.to(device).contiguous()
    assert t.dtype == torch.uint8 and t.numel() * 4 >= count
    with torch.cuda.device(t.device):                                                # (the library keeps its own copy)
        check(lib().cvx_context_set_mind_exp_table(None, ptr(t), int(first), int(count), stream_ptr(t.device)))


def set_mean_threads(threads=0):
    """MINDSSC's global mean (`mind_var.mean()`, convex_adam_utils.py:61) as torch's CPU kernel evaluates it with `threads` threads
    instead of the exactly rounded, order-independent mean (0, default).  The value differs by a few ulps and only reaches the result
    through voxels whose variance is clamped to [mean / 1000, 1000 mean] (flat regions, e.g. masked images)."""
    check(lib().cvx_set_option(b"mind_mean_threads", int(threads)))


def enable(device="cuda", threads=None):
    """Builds this host's tables from torch and installs them; `threads`: the thread count of the reference run to reproduce
    (default: torch.get_num_threads() of this process)."""
    set_mind_exp_table(build_exp_table(device), device=device)
    set_adam_sqrt_table(build_sqrt_table(), device=device)
    set_mean_threads(torch.get_num_threads() if threads is None else threads)


def disable():
    set_mind_exp_table(None)
    set_adam_sqrt_table(None)
    set_mean_threads(0)


def context(device="cuda", threads=None, exp_table=None, sqrt_table=None, exp_first=EXP_FIRST, exp_count=EXP_COUNT):
    """A `convexadam_amd.context.Context` in reference-bits mode, leaving the process default context alone: `with context(...):`
    around the calls of one thread.  Tables default to this host's (built from torch, ~15 s); pass recorded ones to reproduce another
    host (tests/golden/mkl_vsexp_codes.xz, mkl_vssqrt_low.npz through convex_adam_utils.sqrt_codes_from_low_bitmaps)."""
    from .context import Context
    ctx = Context(mind_mean_threads=torch.get_num_threads() if threads is None else threads)
    ctx.set_mind_exp_table(build_exp_table(device) if exp_table is None else exp_table, exp_first, exp_count, device=device)
    ctx.set_adam_sqrt_table(build_sqrt_table() if sqrt_table is None else sqrt_table, device=device)
    return ctx
