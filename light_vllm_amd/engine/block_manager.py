"""Physical KV-block allocator with a chained-hash prefix cache.

Produces the same block tables, cached-token counts, reference counts and free-list order as the
reference's BlockManager (lightvllm/engine/block_manager.py:41-214) on the same call sequence:

  * block hash = xxh64( [8-byte little-endian hash of the previous full block] ++ token ids as
    little-endian int64 ) — block_manager.py:64-81; only full blocks are hashed;
  * `allocate` walks the logical blocks, reuses a block on a hash hit whose stored token ids also
    match, and after the first miss treats every later block as a miss (:123-149);
  * new blocks are taken from the head of the free list, released blocks go to its tail, and
    a released block keeps its hash entry so a later request can revive it (:95-102, :153-166);
  * `may_append` opens a new block when the sequence length is 1 past a block boundary and seals
    (hashes) the last block when the length reaches a boundary (:183-214).

Differences, both outside the reference's crash-free behaviour:
  * the free list is an insertion-ordered dict, so reviving a released block is O(1) instead of the
    reference's O(n) deque.remove (:91); the order of ids is unchanged;
  * `can_append` asks for a free block exactly when `may_append` will take one (len % bs == 1).
    The reference tests len % bs == 0 (:180), which reserves one step early and lets `may_append`
    raise IndexError when the pool is empty at len % bs == 1. `strict_reference=True` restores
    the reference's test for trace comparison.
"""
from __future__ import annotations

import numpy as np
import xxhash

from .sequence import Sequence


class Block:
    __slots__ = ("block_id", "ref_count", "hash", "token_ids")

    def __init__(self, block_id: int):
        self.block_id = block_id
        self.ref_count = 0
        self.hash = -1
        self.token_ids: list[int] = []

    def update(self, hash: int, token_ids: list[int]) -> None:
        self.hash = hash
        self.token_ids = token_ids

    def reset(self) -> None:
        self.ref_count = 1
        self.hash = -1
        self.token_ids = []


class BlockManager:

    def __init__(self, num_blocks: int, block_size: int, strict_reference: bool = False):
        assert num_blocks > 0
        self.block_size = block_size
        self.strict_reference = strict_reference
        self.blocks: list[Block] = [Block(i) for i in range(num_blocks)]
        self.hash_to_block_id: dict[int, int] = {}
        self._free: dict[int, None] = dict.fromkeys(range(num_blocks))   # ordered set, head = oldest
        self.used_block_ids: set[int] = set()

    # -- views kept for callers that look at the reference's attributes
    @property
    def free_block_ids(self) -> list[int]:
        return list(self._free)

    @property
    def num_free_blocks(self) -> int:
        return len(self._free)

    @classmethod
    def compute_hash(cls, token_ids: list[int], prefix: int = -1) -> int:
        h = xxhash.xxh64()
        if prefix != -1:
            h.update(prefix.to_bytes(8, "little"))
        h.update(np.asarray(token_ids, dtype=np.int64).tobytes())
        return h.intdigest()

    def _take(self, block_id: int) -> Block:
        block = self.blocks[block_id]
        assert block.ref_count == 0, "block is still referenced"
        block.reset()
        del self._free[block_id]
        self.used_block_ids.add(block_id)
        return block

    def _take_head(self) -> Block:
        return self._take(next(iter(self._free)))   # StopIteration here = pool exhausted

    def _release(self, block_id: int) -> None:
        assert self.blocks[block_id].ref_count == 0
        self.used_block_ids.remove(block_id)
        self._free[block_id] = None

    def can_allocate(self, seq: Sequence) -> bool:
        # conservative: ignores prefix-cache hits (reference :104-108)
        return len(self._free) >= seq.num_blocks

    def allocate(self, seq: Sequence) -> None:
        assert not seq.block_table, "sequence already owns blocks"
        h = -1
        missed = False
        for i in range(seq.num_blocks):
            token_ids = seq.get_token_ids_from_block(i)
            full = len(token_ids) == self.block_size
            h = self.compute_hash(token_ids, h) if full else -1
            block_id = self.hash_to_block_id.get(h, -1)
            if block_id == -1 or self.blocks[block_id].token_ids != token_ids:
                missed = True
            if missed:
                block = self._take_head()
                block_id = block.block_id
            else:
                seq.num_cached_tokens += self.block_size
                if block_id in self.used_block_ids:
                    block = self.blocks[block_id]
                    block.ref_count += 1
                else:
                    block = self._take(block_id)   # revive a released block that still holds the data
            if full:
                block.update(h, token_ids)
                self.hash_to_block_id[h] = block_id
            seq.block_table.append(block_id)

    def deallocate(self, seq: Sequence) -> None:
        for block_id in seq.block_table:
            block = self.blocks[block_id]
            block.ref_count -= 1
            if block.ref_count == 0:
                self._release(block_id)     # hash entry stays: the block can be revived
        seq.num_cached_tokens = 0
        seq.block_table.clear()

    def can_append(self, seq: Sequence) -> bool:
        boundary = 0 if self.strict_reference else 1
        needs_new_block = len(seq) % self.block_size == boundary
        return len(self._free) >= int(needs_new_block)

    def may_append(self, seq: Sequence) -> None:
        block_table = seq.block_table
        last = self.blocks[block_table[-1]]
        r = len(seq) % self.block_size
        if r == 1:
            # the newest token starts a fresh block
            assert last.hash != -1
            block_table.append(self._take_head().block_id)
        elif r == 0:
            # the newest token fills the last block: seal it so later requests can share it
            assert last.hash == -1
            token_ids = seq.get_token_ids_from_block(seq.num_blocks - 1)
            prefix = self.blocks[block_table[-2]].hash if len(block_table) > 1 else -1
            h = self.compute_hash(token_ids, prefix)
            last.update(h, token_ids)
            self.hash_to_block_id[h] = last.block_id
        else:
            assert last.hash == -1
