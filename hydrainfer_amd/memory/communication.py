"""Cache-block migration backends — host-side mirror of hydrainfer/memory/communication.py.

Reference behaviour kept: pull model (the receiver moves the data, the sender side of the
IPC backend is a no-op), intra-node 'ipc' vs inter-node 'nccl' selection by rank2host.
MI355X redesign of the data path:
  * ipc  : ONE gather-copy kernel over the IPC-mapped peer pool (xGMI peer read) instead of
           n_layers*n_tokens*n_blocks cudaMemcpyAsync calls (communication.py:23-45 ->
           block_migration.cpp:222-244).
  * rccl : pack kernel -> ONE send/recv of the packed buffer -> unpack kernel, instead of one
           P2POp per (block, layer, k/v) view (communication.py:57-74)."""
import os
import time
from dataclasses import dataclass
from typing import Dict, Literal, Optional

import torch
import torch.distributed as dist
from torch import Tensor

from hydrainfer_amd._C.data_transfer import block_migration
from hydrainfer_amd.memory.token_cache import VirtualTokenCache


class MigrationTimeout(RuntimeError):
    """The peer of a send/recv block transfer did not take part within the bound (it died, or never got the
    request): the caller's rank must fail loudly — a fresh process is started on retry, nothing is re-exec'd."""


class CommunicationBackend:
    def migrate_blocks(self, src_virtual_cache: VirtualTokenCache,
                       dst_virtual_cache: VirtualTokenCache, is_send: bool) -> None:
        raise NotImplementedError()


class IPCHandleMemoryBackend(CommunicationBackend):
    def __init__(self, migrate_stream: "torch.cuda.Stream", cache: Tensor, n_blocks: int):
        self.migrate_stream = migrate_stream
        self.cache = cache
        self.n_blocks = n_blocks

    def migrate_blocks(self, src_virtual_cache, dst_virtual_cache, is_send: bool) -> None:
        if is_send:
            return  # pull model: the destination reads the peer pool
        assert src_virtual_cache.memory_handle is not None
        with torch.cuda.stream(self.migrate_stream):
            block_migration.migrate_blocks(
                src_virtual_cache.block_table, dst_virtual_cache.block_table,
                src_virtual_cache.memory_handle, self.cache,
                src_virtual_cache.n_blocks_of_cache_manager)


class RCCLBackend(CommunicationBackend):
    """Packed point-to-point transfer over torch.distributed (backend "nccl" == RCCL on ROCm,
    "gloo" in CPU tests of the protocol).  pack/unpack are HIP kernels; the staging buffer is
    reused across calls."""

    def __init__(self, migrate_stream: Optional["torch.cuda.Stream"], cache: Tensor,
                 group: Optional[dist.ProcessGroup] = None, timeout_s: Optional[float] = None):
        self.migrate_stream = migrate_stream
        self.cache = cache
        self.group = group
        self._staging: Optional[Tensor] = None
        # a transfer whose peer never shows up must not hang the rank (the reference's batch_isend_irecv +
        # req.wait(), hydrainfer/memory/communication.py:66-74, waits for ever)
        self.timeout_s = float(os.environ.get("HX_MIGRATE_TIMEOUT_S", "120")) if timeout_s is None else timeout_s
        # set by a MigrationTimeout: the abandoned send / recv (or the helper thread of connect()) may still sit in the
        # process group's queue, and anything issued behind it would wait for it — the backend refuses further transfers
        # instead of queueing them (round-4 ADVICE); the rank fails loudly, a fresh process is the retry
        self.broken: Optional[str] = None

    def _bounded_wait(self, work, what: str, peer: int) -> None:
        """RCCL: poll the work's completion (it is stream-ordered; wait() would only enqueue a stream wait).  gloo
        (CPU protocol tests): its wait takes the bound itself; its is_completed() stays False until waited on."""
        try:
            if dist.get_backend(self.group) == "gloo":
                from datetime import timedelta
                work.wait(timeout=timedelta(seconds=self.timeout_s))
                return
            deadline = time.monotonic() + self.timeout_s
            while not work.is_completed():
                if time.monotonic() > deadline:
                    raise TimeoutError("not completed")
                time.sleep(0.0005)
            work.wait()        # completed: surfaces a transport error, if any, as an exception
        except (RuntimeError, TimeoutError) as e:
            self.broken = f"{what} rank {peer} timed out"
            raise MigrationTimeout(f"{what} rank {peer}: no progress within {self.timeout_s:.0f} s or transport error "
                                   f"({str(e)[:120]}) — peer dead or never asked to take part in the transfer") from e

    def connect(self, peer: int, my_rank: int, timeout_s: Optional[float] = None) -> None:
        """Creates the point-to-point communicator with `peer` NOW, under a bound (round-3 ADVICE): RCCL builds it
        lazily inside the first isend / irecv, which blocks there — before _bounded_wait's polling loop is ever
        reached — so a peer that never shows up would hang the FIRST transfer for ever.  Both ranks call this at
        start-up (the lower rank sends one element, the higher one receives it) from a helper thread that is abandoned
        after the bound.  Known limit that stays: a transfer is awaited on the calling (engine) thread for its whole
        duration; the engine overlaps pulls with compute only through the IPC backend, which has no sender half."""
        import threading
        bound = self.timeout_s if timeout_s is None else timeout_s
        dev = self.cache.device
        box = {}

        def _go():
            try:
                if dev.type == "cuda":
                    torch.cuda.set_device(dev)
                t = torch.zeros(1, dtype=torch.int32, device=dev)
                if my_rank < peer:
                    dist.send(t, dst=peer, group=self.group)
                else:
                    dist.recv(t, src=peer, group=self.group)
                if dev.type == "cuda":
                    torch.cuda.synchronize(dev)
                box["ok"] = True
            except Exception as e:       # reported by the caller's thread
                box["err"] = e
        th = threading.Thread(target=_go, daemon=True)
        th.start()
        th.join(timeout=bound)
        if th.is_alive() or "err" in box:
            self.broken = f"connect to rank {peer} failed"
            raise MigrationTimeout(f"connect to rank {peer}: no communicator within {bound:.0f} s "
                                   f"({str(box.get('err', 'peer never answered'))[:120]})")

    def _staging_for(self, n_blocks: int) -> Tensor:
        L, T, _, bs, H, D = self.cache.shape
        numel = L * T * n_blocks * bs * H * D
        if self._staging is None or self._staging.numel() < numel:
            self._staging = torch.empty(numel, dtype=self.cache.dtype, device=self.cache.device)
        return self._staging[:numel]

    def migrate_blocks(self, src_virtual_cache, dst_virtual_cache, is_send: bool) -> None:
        table = src_virtual_cache.block_table if is_send else dst_virtual_cache.block_table
        peer = dst_virtual_cache.rank if is_send else src_virtual_cache.rank
        if self.broken is not None:
            raise MigrationTimeout(f"send/recv backend unusable after an earlier failure ({self.broken}): an abandoned "
                                   "operation may still hold the process group")
        if len(table) == 0:
            return
        staging = self._staging_for(len(table))
        ctx = torch.cuda.stream(self.migrate_stream) if self.migrate_stream is not None else _Null()
        with ctx:
            if is_send:
                block_migration.pack_blocks(table, self.cache, staging)
                self._bounded_wait(dist.isend(staging, dst=peer, group=self.group), "send to", peer)
            else:
                self._bounded_wait(dist.irecv(staging, src=peer, group=self.group), "recv from", peer)
                block_migration.unpack_blocks(table, staging, self.cache)


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


@dataclass
class CommunicationBackendManagerContext:
    migrate_stream: "torch.cuda.Stream"
    cache: Tensor  # (n_layers, n_tokens, n_blocks, block_size, n_heads, head_size)
    n_blocks: int
    rank2host: Dict[int, str]


@dataclass
class CommunicationBackendManagerConfig:
    intranode_migrate_backend: Literal["auto", "ipc", "nccl"] = "auto"
    internode_migrate_backend: Literal["nccl"] = "nccl"
    debug: bool = False


def get_migrate_backend(backend: str, context: CommunicationBackendManagerContext) -> CommunicationBackend:
    if backend == "ipc":
        return IPCHandleMemoryBackend(context.migrate_stream, context.cache, context.n_blocks)
    if backend == "nccl":
        return RCCLBackend(context.migrate_stream, context.cache)
    raise Exception(f"invalid migrate backend {backend}")


class CommunicationBackendManager(CommunicationBackend):
    def __init__(self, config: CommunicationBackendManagerConfig,
                 context: CommunicationBackendManagerContext):
        self.context = context
        self.rank2host = context.rank2host
        intranode = config.intranode_migrate_backend
        if intranode == "auto":
            intranode = "ipc"  # the HIP library is mandatory, so ipc is always available
        self.intranode_backend = get_migrate_backend(intranode, context)
        self.internode_backend = get_migrate_backend(config.internode_migrate_backend, context)

    def in_same_machine(self, rank1: int, rank2: int) -> bool:
        if rank1 not in self.rank2host or rank2 not in self.rank2host:
            return False
        return self.rank2host[rank1] == self.rank2host[rank2]

    def needs_sender(self, src_rank: int, dst_rank: int) -> bool:
        """True when the transfer is a send/recv pair: the sender must run its half
        (migrate_blocks(..., is_send=True)) — hydrainfer/cluster/epdnode.py:362-378 — before
        the receiver's half can complete.  The IPC pull has no sender half."""
        backend = self.intranode_backend if self.in_same_machine(src_rank, dst_rank) else self.internode_backend
        return not isinstance(backend, IPCHandleMemoryBackend)

    def connect_peers(self, my_rank: int, pairs, timeout_s: Optional[float] = None) -> int:
        """Node start-up (round-4 ADVICE: nothing called RCCLBackend.connect outside the tests): for every (src, dst)
        rank pair of `pairs` that this rank is part of and whose transfer is a send/recv pair, create the point-to-point
        communicator NOW under a bound — RCCL would otherwise build it inside the first isend / irecv, where a missing
        peer hangs the engine thread for ever.  Every rank walks the same sorted pair list, so the blocking
        handshakes match up.  Returns the number of communicators created; raises MigrationTimeout."""
        n = 0
        for src, dst in sorted(set(pairs)):
            if my_rank not in (src, dst) or src == dst or not self.needs_sender(src, dst):
                continue
            backend = self.intranode_backend if self.in_same_machine(src, dst) else self.internode_backend
            backend.connect(dst if my_rank == src else src, my_rank, timeout_s)
            n += 1
        return n

    def migrate_blocks(self, src_virtual_cache, dst_virtual_cache, is_send: bool) -> None:
        assert src_virtual_cache.n_cache_tokens == dst_virtual_cache.n_cache_tokens, \
            f"{src_virtual_cache.n_cache_tokens} {dst_virtual_cache.n_cache_tokens}"
        if self.in_same_machine(src_virtual_cache.rank, dst_virtual_cache.rank):
            self.intranode_backend.migrate_blocks(src_virtual_cache, dst_virtual_cache, is_send)
        else:
            self.internode_backend.migrate_blocks(src_virtual_cache, dst_virtual_cache, is_send)
