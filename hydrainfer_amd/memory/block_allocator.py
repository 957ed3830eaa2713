"""LIFO block allocator — host-side mirror of hydrainfer/memory/block_allocator.py:11-39.
Free ids are kept in a stack initialised [n-1 … 0]; allocate pops n ids from the TAIL in
stored order (so a fresh allocator hands out ascending runs ending at 0: [n-1..] tail =
[..., 2, 1, 0] -> allocate(3) == [2, 1, 0]); free pushes back."""
from dataclasses import dataclass
from typing import List


@dataclass
class BlockAllocatorMetrics:
    n_used_blocks: int
    n_total_blocks: int
    block_usage: float


class BlockAllocator:
    def __init__(self, total_blocks: int):
        self.total_blocks = total_blocks
        self.free_blocks: List[int] = list(range(total_blocks - 1, -1, -1))

    def get_metrics(self) -> BlockAllocatorMetrics:
        used = self.total_blocks - len(self.free_blocks)
        return BlockAllocatorMetrics(used, self.total_blocks, used / self.total_blocks)

    def allocate(self, n_blocks: int) -> List[int]:
        # at most n_blocks; fewer when the pool runs dry (block_allocator.py:25-32)
        if n_blocks <= 0:
            return []
        n_blocks = min(n_blocks, len(self.free_blocks))
        if n_blocks == 0:
            return []
        blocks = self.free_blocks[-n_blocks:]
        del self.free_blocks[-n_blocks:]
        return blocks

    def free(self, blocks: List[int]) -> None:
        self.free_blocks += blocks
        assert len(self.free_blocks) <= self.total_blocks

    def get_num_avaiable_blocks(self) -> int:
        return len(self.free_blocks)
