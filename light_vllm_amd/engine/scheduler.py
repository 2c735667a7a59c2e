"""Prefill-first continuous batching with LIFO preemption.

Same policy as the reference's Scheduler (lightvllm/engine/scheduler.py:55-134): a step is either a
prefill of as many waiting sequences as fit (`max_num_seqs`, `max_num_batched_tokens` counting only
un-cached tokens, free KV blocks), or one decode token for the running sequences, preempting from
the tail of the running queue when the KV pool cannot grow. It decides the batch composition that
the attention kernels see; nothing here touches the device.
"""
from __future__ import annotations

from collections import deque

from .block_manager import BlockManager
from .sequence import Sequence, SequenceStatus


class Scheduler:

    def __init__(self, config):
        self.max_num_seqs = config.max_num_seqs
        self.max_num_batched_tokens = config.max_num_batched_tokens
        self.eos = config.eos
        self.block_manager = BlockManager(config.num_kvcache_blocks, config.kvcache_block_size,
                                          getattr(config, "strict_reference", False))
        self.waiting: deque[Sequence] = deque()
        self.running: deque[Sequence] = deque()

    def is_finished(self) -> bool:
        return not self.waiting and not self.running

    def add(self, seq: Sequence) -> None:
        self.waiting.append(seq)

    def preempt(self, seq: Sequence) -> None:
        seq.status = SequenceStatus.WAITING
        seq.needs_prefill = True          # recomputed from all its tokens when it is scheduled again
        self.block_manager.deallocate(seq)
        self.waiting.appendleft(seq)

    def schedule(self) -> tuple[list[Sequence], bool]:
        batch: list[Sequence] = []
        num_batched_tokens = 0
        # prefill
        while self.waiting and len(batch) < self.max_num_seqs:
            seq = self.waiting[0]
            if (num_batched_tokens + len(seq) > self.max_num_batched_tokens
                    or not self.block_manager.can_allocate(seq)):
                break
            self.block_manager.allocate(seq)
            num_batched_tokens += len(seq) - seq.num_cached_tokens
            seq.status = SequenceStatus.RUNNING
            self.waiting.popleft()
            self.running.append(seq)
            batch.append(seq)
        if batch:
            return batch, True
        # decode
        while self.running and len(batch) < self.max_num_seqs:
            seq = self.running.popleft()
            while not self.block_manager.can_append(seq):
                if self.running:
                    self.preempt(self.running.pop())
                else:
                    self.preempt(seq)
                    break
            else:
                self.block_manager.may_append(seq)
                batch.append(seq)
        assert batch, "decode step scheduled nothing"
        self.running.extendleft(reversed(batch))
        return batch, False

    def postprocess(self, seqs: list[Sequence], token_ids: list[int]) -> None:
        for seq, token_id in zip(seqs, token_ids):
            seq.append_token(token_id)
            seq.needs_prefill = False
            if (not seq.ignore_eos and token_id == self.eos) or seq.num_completion_tokens == seq.max_tokens:
                seq.status = SequenceStatus.FINISHED
                self.block_manager.deallocate(seq)
                self.running.remove(seq)
