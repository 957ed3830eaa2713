"""Named ranges for profiles — the role of hydrainfer/utils/profiler.py:15-33 (NVTX push/pop +
accumulated wall clock printed at exit), with roctx markers so `rocprofv3 --marker-trace` shows the
engine's phases.  Off unless HX_PROFILE=1: the ranges then cost two library calls and a clock read;
with HX_PROFILE_SYNC=1 each range also synchronises the device on exit (true GPU time per phase,
serialises the pipeline — for analysis runs only)."""
import atexit
import contextlib
import ctypes
import os
import time
from collections import defaultdict

ENABLED = os.environ.get("HX_PROFILE") == "1"
_SYNC = os.environ.get("HX_PROFILE_SYNC") == "1"
_totals = defaultdict(lambda: [0, 0.0])
_roctx = None
if ENABLED:
    # rocprofv3 intercepts the rocprofiler-sdk flavour of roctx; the legacy library is the fallback
    for _name in ("librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                  "libroctx64.so", "/opt/rocm/lib/libroctx64.so"):
        try:
            _roctx = ctypes.CDLL(_name)
            _roctx.roctxRangePushA.argtypes = [ctypes.c_char_p]
            break
        except OSError:
            _roctx = None

    @atexit.register
    def _report():
        for name, (n, seconds) in sorted(_totals.items(), key=lambda kv: -kv[1][1]):
            print(f"[hx profile] {name:24s} calls {n:7d}  total {seconds * 1e3:10.2f} ms  "
                  f"avg {seconds / max(n, 1) * 1e6:9.1f} us")


@contextlib.contextmanager
def profile(name: str):
    if not ENABLED:
        yield
        return
    if _roctx is not None:
        _roctx.roctxRangePushA(name.encode())
    t0 = time.perf_counter()
    try:
        yield
    finally:
        if _SYNC:
            import torch
            torch.cuda.synchronize()
        rec = _totals[name]
        rec[0] += 1
        rec[1] += time.perf_counter() - t0
        if _roctx is not None:
            _roctx.roctxRangePop()
