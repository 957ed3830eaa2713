"""hydrainfer._C.data_transfer.block_migration — drop-in surface
(reference stub: hydrainfer/_C/data_transfer/block_migration/__init__.pyi:6-18;
CUDA original: csrc/data_transfer/block_migration.cpp:55-59, 69-80, 194-245).

Wire format kept: an IPC handle travels through Python/Ray as `list[int]` of 64 byte
values (block_migration.cpp:34-49).  A torch allocation may sit at an offset inside its
hipMalloc segment; the offset is appended to the list as 8 extra little-endian bytes
(72 ints) — the reference assumes offset 0, which holds only for a tensor that owns its
segment, and silently reads the wrong bytes otherwise.  A pool whose (layer, k/v) planes are not
back to back (memory/kv_pool.py) appends its plane stride in bytes the same way (80 ints); 72 or 64
ints mean the reference's contiguous pool."""
import ctypes
import threading
from typing import List

import torch
from torch import Tensor

from hydrainfer_amd import _lib
import hydrainfer_amd.memory.kv_pool as kv_pool

cudaMemoryIpcHandle = List[int]
_registered: List[int] = []
# handles exported by THIS process -> local base pointer: HIP refuses to open a handle in the
# process that created it, and a same-process "migration" (EPD node, tests) needs no mapping
_exported: dict = {}
_opened: dict = {}          # peer handles mapped by this process -> base pointer (the C side caches them too)
_wedged = False
OPEN_TIMEOUT_S = 60.0


def get_ipc_mem_handle(tensor: Tensor) -> cudaMemoryIpcHandle:
    _lib.require_gpu(tensor)
    buf = (ctypes.c_uint8 * _lib.HX_IPC_HANDLE_BYTES)()
    off = ctypes.c_int64(0)
    with torch.cuda.device(tensor.device):
        _lib.check(_lib.lib().hx_ipc_get_mem_handle(tensor.data_ptr(), buf, ctypes.byref(off)),
                   "get_ipc_mem_handle")
    _exported[bytes(buf)] = tensor.data_ptr() - int(off.value)
    handle = list(buf) + list(int(off.value).to_bytes(8, "little"))
    if tensor.dim() == 6 and not tensor.is_contiguous():
        handle += list(int(kv_pool.plane_bytes_of(tensor)).to_bytes(8, "little"))
    return handle


def handle_plane_bytes(handle: cudaMemoryIpcHandle) -> int:
    """Plane stride a pool's handle carries; 0 = contiguous planes (n_blocks * block_bytes)."""
    n = _lib.HX_IPC_HANDLE_BYTES
    return int.from_bytes(bytes(handle[n + 8:n + 16]), "little") if len(handle) >= n + 16 else 0


def _open(handle: cudaMemoryIpcHandle) -> int:
    if len(handle) not in (_lib.HX_IPC_HANDLE_BYTES, _lib.HX_IPC_HANDLE_BYTES + 8, _lib.HX_IPC_HANDLE_BYTES + 16):
        raise _lib.HydraHipError("IPC handle must be 64 (+8 offset, +8 plane stride) byte values")
    buf = (ctypes.c_uint8 * _lib.HX_IPC_HANDLE_BYTES)(*handle[:_lib.HX_IPC_HANDLE_BYTES])
    off = int.from_bytes(bytes(handle[_lib.HX_IPC_HANDLE_BYTES:_lib.HX_IPC_HANDLE_BYTES + 8]), "little") if len(handle) > 64 else 0
    key = bytes(buf)
    local = _exported.get(key)
    if local is not None:
        return local + off
    if key in _opened:
        return _opened[key] + off
    if _wedged:
        raise _lib.HydraHipError("an earlier hipIpcOpenMemHandle never returned: this process cannot map peer pools")
    # First mapping of this handle: bounded.  hipIpcOpenMemHandle was seen to never return for
    # allocations whose size lies in [7/8 * 2^k, 2^k) (memory/token_cache_manger.ipc_safe_n_blocks
    # sizes pools past that window); a rank that waits forever inside the driver is worse than one
    # that fails, so the call runs in a helper thread and a timeout raises.  The thread cannot be
    # cancelled; the process should exit with a non-zero status (never re-exec: it has touched the GPU).
    ptr = ctypes.c_void_p(0)
    box = {}
    dev = torch.cuda.current_device()

    def work():
        torch.cuda.set_device(dev)             # the current device is per thread
        box["rc"] = _lib.lib().hx_ipc_open_mem_handle(buf, ctypes.byref(ptr))
    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout=OPEN_TIMEOUT_S)
    if th.is_alive():
        globals()["_wedged"] = True
        raise _lib.HydraHipError(
            f"hipIpcOpenMemHandle did not return within {OPEN_TIMEOUT_S:.0f} s (peer pool size in the window "
            "[7/8 * 2^k, 2^k)? see ipc_safe_n_blocks); exit this process with a non-zero status")
    _lib.check(box["rc"], "open ipc handle")
    _opened[key] = int(ptr.value)
    return int(ptr.value) + off


def register_ipc_mem_handle(kv_cache_handle_vec: cudaMemoryIpcHandle) -> int:
    """Maps a peer handle and returns its index (block_migration.cpp:69-80)."""
    _registered.append(_open(kv_cache_handle_vec))
    return len(_registered) - 1


def _pool_planes(cache: Tensor, what: str):
    """(n_planes, n_blocks, block_bytes, plane_bytes) of a 6-D pool: contiguous, or planes apart (memory/kv_pool.py)."""
    try:
        pb = kv_pool.plane_bytes_of(cache)
    except ValueError as e:
        raise _lib.HydraHipError(f"{what}: {e}")
    n_layers, n_tokens, n_blocks, block_size, n_heads, head_size = cache.shape
    return n_layers * n_tokens, n_blocks, block_size * n_heads * head_size * cache.element_size(), pb


def migrate_blocks(src_block_table: List[int], dst_block_table: List[int],
                   src_cache: cudaMemoryIpcHandle, dst_cache: Tensor,
                   src_cache_n_blocks: int) -> None:
    """dst_cache[l, t, dst_block_table[i]] = src[l, t, src_block_table[i]] on the current stream."""
    _lib.require_gpu(dst_cache)
    n_planes, dst_n_blocks, block_bytes, dst_pb = _pool_planes(dst_cache, "migrate_blocks: dst_cache")
    if len(src_block_table) != len(dst_block_table):
        raise _lib.HydraHipError("migrate_blocks: block tables must have equal length")
    n = len(src_block_table)
    if n == 0:
        return
    with torch.cuda.device(dst_cache.device):
        src_ptr = _open(src_cache)
        src_pb = handle_plane_bytes(src_cache) or int(src_cache_n_blocks) * block_bytes
        src_tbl = (ctypes.c_int32 * n)(*src_block_table)
        dst_tbl = (ctypes.c_int32 * n)(*dst_block_table)
        _lib.check(_lib.lib().hx_migrate_blocks_planes(
            src_tbl, dst_tbl, n, src_ptr, dst_cache.data_ptr(), n_planes, int(src_cache_n_blocks), dst_n_blocks,
            src_pb, dst_pb, block_bytes, _lib.current_stream()), "migrate_blocks")


def migrate_blocks_local(src_block_table: List[int], dst_block_table: List[int],
                         src_cache: Tensor, dst_cache: Tensor) -> None:
    """Same copy with a directly addressable source pool (same process / peer-enabled)."""
    _lib.require_gpu(src_cache, dst_cache)
    n_planes, dst_n_blocks, block_bytes, dst_pb = _pool_planes(dst_cache, "migrate_blocks: dst_cache")
    s_planes, src_n_blocks, s_block_bytes, src_pb = _pool_planes(src_cache, "migrate_blocks: src_cache")
    if s_planes != n_planes or s_block_bytes != block_bytes:
        raise _lib.HydraHipError("migrate_blocks: pools differ in layers / block shape")
    n = len(src_block_table)
    if n != len(dst_block_table):
        raise _lib.HydraHipError("migrate_blocks: block tables must have equal length")
    if n == 0:
        return
    src_tbl = (ctypes.c_int32 * n)(*src_block_table)
    dst_tbl = (ctypes.c_int32 * n)(*dst_block_table)
    with torch.cuda.device(dst_cache.device):
        _lib.check(_lib.lib().hx_migrate_blocks_planes(
            src_tbl, dst_tbl, n, src_cache.data_ptr(), dst_cache.data_ptr(), n_planes, src_n_blocks, dst_n_blocks,
            src_pb, dst_pb, block_bytes, _lib.current_stream()), "migrate_blocks")


def pack_blocks(block_table: List[int], cache: Tensor, staging: Tensor) -> None:
    """staging[l, t, i] = cache[l, t, block_table[i]] — send side of the RCCL path."""
    _lib.require_gpu(cache, staging)
    n = len(block_table)
    if n == 0:
        return
    n_planes, n_blocks, block_bytes, pb = _pool_planes(cache, "pack_blocks: cache")
    if staging.numel() * staging.element_size() < n_planes * n * block_bytes:
        raise _lib.HydraHipError("pack_blocks: staging buffer too small")
    tbl = (ctypes.c_int32 * n)(*block_table)
    with torch.cuda.device(cache.device):
        _lib.check(_lib.lib().hx_pack_blocks_planes(
            tbl, n, cache.data_ptr(), staging.data_ptr(), n_planes, n_blocks, pb,
            block_bytes, _lib.current_stream()), "pack_blocks")


def unpack_blocks(block_table: List[int], staging: Tensor, cache: Tensor) -> None:
    """cache[l, t, block_table[i]] = staging[l, t, i] — receive side of the RCCL path."""
    _lib.require_gpu(cache, staging)
    n = len(block_table)
    if n == 0:
        return
    n_planes, n_blocks, block_bytes, pb = _pool_planes(cache, "unpack_blocks: cache")
    if staging.numel() * staging.element_size() < n_planes * n * block_bytes:
        raise _lib.HydraHipError("unpack_blocks: staging buffer too small")
    tbl = (ctypes.c_int32 * n)(*block_table)
    with torch.cuda.device(cache.device):
        _lib.check(_lib.lib().hx_unpack_blocks_planes(
            tbl, n, staging.data_ptr(), cache.data_ptr(), n_planes, n_blocks, pb,
            block_bytes, _lib.current_stream()), "unpack_blocks")
