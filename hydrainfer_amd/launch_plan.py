"""Launch plans: record the launches of a decode step once, replay them with one native loop
(include/hydra_hip.h "Launch plans", csrc/launch_plan.hip).

The reference never finished its CUDA-graph runner (hydrainfer/model_runner/cuda_graph_model_runner.py:1-72); here a
decode step is replayed either from a captured hipGraph or from a plan — the same launches in the same stream order,
issued by a C loop over the recorded (kernel, geometry, arguments) triples.  Measured (profiles/
r3_launch_chain_experiment.md): equal to the hipGraph when both exist in one process, 0.5-1 % faster than it in a
process that replays only plans (bench.py, the engine).

    plan = LaunchPlan(device)
    out = plan.capture(step_fn)       # step_fn's hx_* launches are recorded, nothing runs; torch ops raise PlanNotRecordable
    plan.replay(); plan.replay()      # ... except through launch_plan.host_op(fn), which is replayed in its place

Everything step_fn allocates comes from a private torch memory pool that lives as long as the plan (the recorded
launches hold raw pointers), exactly like the private pool of a captured graph."""
import ctypes
import threading
from typing import Callable, List, Optional

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from hydrainfer_amd import _lib

_tls = threading.local()


class PlanNotRecordable(_lib.HydraHipError):
    """The step contains work a launch plan cannot record (a torch op that launches a kernel outside host_op)."""


# aten operators that only allocate or re-view memory: they launch nothing, so a recording may contain them
_VIEW_OR_ALLOC = frozenset((
    "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "view", "_unsafe_view", "reshape",
    "_reshape_alias", "as_strided", "slice", "select", "narrow", "expand", "t", "transpose", "permute", "unsqueeze",
    "squeeze", "detach", "alias", "unbind", "split", "split_with_sizes", "chunk", "unfold", "view_as", "lift_fresh",
    "flatten", "unflatten", "movedim", "swapaxes", "diagonal", "real", "imag", "view_as_real", "view_as_complex",
    "sym_size", "sym_stride", "sym_numel", "sym_storage_offset", "is_contiguous", "size", "stride", "numel", "dim"))


class _RecordingGuard(TorchDispatchMode):
    """While a plan is recorded nothing executes: an hx_* launch goes into the plan, a torch op handed to host_op is
    replayed in its place — and any OTHER torch op would run once now and be missing from every replay, silently
    (round-3 ADVICE: bench.py --lib-gemm replayed a step without its GEMMs).  This mode turns that into an error."""

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name.split("::")[-1]
        if name not in _VIEW_OR_ALLOC:
            raise PlanNotRecordable(f"torch op aten::{name} inside a launch-plan recording: it would run once now and "
                                    "be dropped on replay — wrap it in launch_plan.host_op or use the hipGraph executor")
        return func(*args, **(kwargs or {}))


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
    def __init__(self, device):
        self.device = torch.device(device)
        self.items: List = []                   # ctypes plan handles and host callables, in order
        self.pool = torch.cuda.MemPool()
        self._open: Optional[ctypes.c_void_p] = None
        self.n_launches = 0

    # ------------------------------------------------------------------ recording
    def _begin_segment(self) -> None:
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().hx_plan_begin(ctypes.byref(h)), "plan_begin")
        self._open = h

    def _end_segment(self) -> None:
        h, self._open = self._open, None
        _lib.check(_lib.lib().hx_plan_end(h), "plan_end")
        n = _lib.lib().hx_plan_size(h)
        if n > 0:
            self.items.append(h)
            self.n_launches += n
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
                    with _RecordingGuard():
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

    def __del__(self):
        try:
            lib = _lib.lib()
            for it in self.items:
                if not callable(it):
                    lib.hx_plan_destroy(it)
            self.items = []
        except Exception:
            pass
