"""Per-caller state of the HIP library: variant switches and the two reference-build tables (include/convexadam_hip.h, "State model").

    ctx = Context(mind_mean_threads=8)
    ctx.set_adam_sqrt_table(codes); ctx.set_mind_exp_table(table)
    with ctx:                               # bound to THIS thread for the duration of the block
        field = register_pair_device(fix, mov, ...)

Threads that enter different contexts (and launch on different streams) are independent: a call reads its switches and tables when
it enqueues its kernels.  Outside any `with` block a thread uses the process default context (cvx_set_option, reference_bits.enable).
The library copies tables into memory the context owns, so nothing has to be kept alive on the Python side.
"""
import threading

import numpy as np
import torch

from ._lib import check, lib, ptr, stream_ptr


class Context:
    def __init__(self, **options):
        self._h = lib().cvx_context_create()
        if not self._h:
            raise MemoryError("cvx_context_create failed")
        self._tls = threading.local()
        for k, v in options.items():
            self.set_option(k, v)

    # -- switches ---------------------------------------------------------------------------------------
    def set_option(self, name, value):
        check(lib().cvx_context_set_option(self._h, name.encode(), int(value)))
        return self

    def get_option(self, name):
        return int(lib().cvx_context_get_option(self._h, name.encode()))

    # -- tables (see reference_bits.py) -------------------------------------------------------------------
    def set_adam_sqrt_table(self, codes=None, device="cuda"):
        """6 MiB 2-bit table of cvx_context_set_adam_sqrt_table (numpy / torch uint8); None restores the IEEE sqrt."""
        if codes is None:
            check(lib().cvx_context_set_adam_sqrt_table(self._h, None, None))
            return self
        t = _as_device_u8(codes, device)
        assert t.numel() == 6 * 1024 * 1024, "sqrt table: 2 bits x (2^24 + 2^23) classes = 6 MiB"
        with torch.cuda.device(t.device):
            check(lib().cvx_context_set_adam_sqrt_table(self._h, ptr(t), stream_ptr(t.device)))
        return self

    def set_mind_exp_table(self, table=None, first=0, count=0, device="cuda"):
        if table is None:
            check(lib().cvx_context_set_mind_exp_table(self._h, None, 0, 0, None))
            return self
        t = _as_device_u8(table, device)
        assert t.numel() * 4 >= count > 0
        with torch.cuda.device(t.device):
            check(lib().cvx_context_set_mind_exp_table(self._h, ptr(t), int(first), int(count), stream_ptr(t.device)))
        return self

    # -- binding ------------------------------------------------------------------------------------------
    @property
    def handle(self):
        """cvx_context* for cvx_pair_params.ctx."""
        return self._h

    def __enter__(self):
        # the saved bindings are PER THREAD (one Context object -- e.g. the 80 MB reference-bits tables -- may be entered by several
        # worker threads at once; the library's binding is thread-local too)
        stack = getattr(self._tls, "prev", None)
        if stack is None:
            stack = self._tls.prev = []
        stack.append(lib().cvx_context_bind(self._h))
        return self

    def __exit__(self, *exc):
        lib().cvx_context_bind(self._tls.prev.pop())
        return False

    def close(self):
        if self._h:
            lib().cvx_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _as_device_u8(x, device):
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.array(x, dtype=np.uint8, order="C"))
    t = t.to(device).contiguous()
    assert t.dtype == torch.uint8
    return t
