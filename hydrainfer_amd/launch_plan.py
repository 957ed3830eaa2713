"""Launch plans: record the launches of a decode step once, replay them with one native loop — and, unlike a
captured hipGraph, with LAUNCH CHAINS (include/hydra_hip.h "Launch plans", csrc/launch_plan.hip).

The reference never finished its CUDA-graph runner (hydrainfer/model_runner/cuda_graph_model_runner.py:1-72); the
graphs of engine/graph_decode.py and model/runner.py do that job here.  A hipGraph, however, always separates
consecutive kernels by the AQL barrier bit: the next kernel's workgroups are dispatched only after the previous
kernel has drained completely.  A plan launches the five kernels of a decode layer without it
(hipExtAnyOrderLaunch) and each of them takes its data dependency inside the kernel: prefetch weights / KV pages,
wait for the predecessor's done flag, consume its output.  Same kernels, same arithmetic, bit-identical results.

    plan = LaunchPlan(device)
    out = plan.capture(step_fn)       # step_fn's hx_* launches are recorded, nothing runs; torch ops are not allowed
    plan.replay(); plan.replay()      # ... except through plan.host_op(fn), which is replayed in its place

Everything step_fn allocates comes from a private torch memory pool that lives as long as the plan (the recorded
launches hold raw pointers), exactly like the private pool of a captured graph."""
import ctypes
import threading
from typing import Callable, List, Optional

import torch

from hydrainfer_amd import _lib

_tls = threading.local()


def current() -> Optional["LaunchPlan"]:
    """The plan being recorded on this thread, or None."""
    return getattr(_tls, "plan", None)


def host_op(fn: Callable[[], None]) -> None:
    """A torch (non-hx) operation inside a step: executed now when no plan is being recorded, otherwise appended to
    the plan — replayed between the native launch segments, on the replay stream."""
    plan = current()
    if plan is None:
        fn()
    else:
        plan._host_op(fn)


class LaunchPlan:
    def __init__(self, device, chain: bool = True, max_chained_launches: int = 1024):
        self.device = torch.device(device)
        self.chain = bool(chain)
        # link areas of the chained launches (the plan zeroes what it uses at the start of every replay)
        self.sync = torch.zeros(_lib.HX_PLAN_SYNC_BYTES_PER_LAUNCH * max_chained_launches // 4, dtype=torch.int32,
                                device=self.device)
        self.error_word = torch.zeros(32, dtype=torch.int32, device=self.device)[:1]   # nonzero iff a wait gave up
        self.items: List = []                   # ctypes plan handles and host callables, in order
        self.pool = torch.cuda.MemPool()
        self._open: Optional[ctypes.c_void_p] = None
        self.n_launches = self.n_any_order = 0
        self._sync_used = 0

    # ------------------------------------------------------------------ recording
    def _begin_segment(self) -> None:
        h = ctypes.c_void_p()
        off = self._sync_used                  # every native segment gets the unused tail of the sync buffer
        _lib.check(_lib.lib().hx_plan_begin(ctypes.byref(h), ctypes.c_void_p(self.sync.data_ptr() + off),
                                            self.sync.numel() * 4 - off, ctypes.c_void_p(self.error_word.data_ptr()),
                                            1 if self.chain else 0), "plan_begin")
        self._open = h

    def _end_segment(self) -> None:
        h, self._open = self._open, None
        rc = _lib.lib().hx_plan_end(h)
        nl, na, used = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
        _lib.lib().hx_plan_info(h, ctypes.byref(nl), ctypes.byref(na), ctypes.byref(used))
        if rc != 0:
            _lib.lib().hx_plan_destroy(h)
            _lib.check(rc, "plan_end (raise max_chained_launches)")
        if nl.value:
            self.items.append(h)
            self.n_launches += nl.value
            self.n_any_order += na.value
            self._sync_used += used.value
        else:
            _lib.lib().hx_plan_destroy(h)

    def _host_op(self, fn: Callable[[], None]) -> None:
        self._end_segment()
        self.items.append(fn)
        self._begin_segment()

    def capture(self, step_fn: Callable[[], object]):
        """Runs step_fn with recording on: its hx_* launches go into the plan (not executed); returns step_fn's
        result (tensors in it are the plan's static buffers)."""
        if current() is not None:
            raise _lib.HydraHipError("a launch plan is already being recorded on this thread")
        if self.items:
            raise _lib.HydraHipError("this plan has been recorded already")
        torch.cuda.synchronize(self.device)
        _tls.plan = self
        try:
            with torch.cuda.use_mem_pool(self.pool, device=self.device):
                self._begin_segment()
                try:
                    result = step_fn()
                finally:
                    self._end_segment()
        finally:
            _tls.plan = None
        return result

    # ------------------------------------------------------------------ replay
    def replay(self) -> None:
        stream = _lib.current_stream()
        lib = _lib.lib()
        for it in self.items:
            if callable(it):
                it()
            else:
                rc = lib.hx_plan_launch(it, stream)
                if rc:
                    _lib.check(rc, "plan_launch")

    def failed(self) -> bool:
        """True iff an in-kernel wait of some replay gave up (one D2H sync); clears the word."""
        bad = bool(int(self.error_word[0]) != 0)
        if bad:
            self.error_word.zero_()
        return bad

    def __del__(self):
        try:
            lib = _lib.lib()
            for it in self.items:
                if not callable(it):
                    lib.hx_plan_destroy(it)
            self.items = []
        except Exception:
            pass
