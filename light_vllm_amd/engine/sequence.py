"""Request state and paged-KV block geometry.

Mirrors the reference's Sequence (lightvllm/engine/sequence.py:18-174): same attributes, same
block arithmetic (`num_blocks`, `last_block_num_tokens`, `num_cached_blocks`,
`get_token_ids_from_block`) and the same compact pickle state used by the step RPC.
`block_size` stays a class attribute (the reference never wires it to the config, sequence.py:28);
`Sequence.set_block_size` is the one place to change it together with the runner's block size.
"""
from __future__ import annotations

from copy import copy
from enum import Enum, auto
from itertools import count

from ..sampling_params import SamplingParams


class SequenceStatus(Enum):
    WAITING = auto()
    RUNNING = auto()
    FINISHED = auto()


class Sequence:
    block_size = 256
    counter = count()

    @classmethod
    def set_block_size(cls, block_size: int) -> None:
        assert block_size > 0
        cls.block_size = block_size

    def __init__(self, token_ids: list[int], sampling_params: SamplingParams = SamplingParams()):
        self.seq_id = next(Sequence.counter)
        self.status = SequenceStatus.WAITING
        self.token_ids = copy(token_ids)
        self.last_token = token_ids[-1]
        self.num_tokens = len(self.token_ids)
        self.num_prompt_tokens = len(token_ids)
        self.num_cached_tokens = 0           # prefix-cache hits, always a multiple of block_size
        self.block_table: list[int] = []     # physical block id of each logical block
        self.temperature = sampling_params.temperature
        self.max_tokens = sampling_params.max_tokens
        self.ignore_eos = sampling_params.ignore_eos
        # True until the sequence's KV is in the cache: at creation and again after a preemption
        # (the scheduler sets it). Decides what the step RPC ships to the workers, see __getstate__.
        self.needs_prefill = True

    def __len__(self) -> int:
        return self.num_tokens

    def __getitem__(self, key):
        return self.token_ids[key]

    @property
    def is_finished(self) -> bool:
        return self.status == SequenceStatus.FINISHED

    @property
    def is_running(self) -> bool:
        return self.status == SequenceStatus.RUNNING

    @property
    def is_waiting(self) -> bool:
        return self.status == SequenceStatus.WAITING

    @property
    def num_completion_tokens(self) -> int:
        return self.num_tokens - self.num_prompt_tokens

    @property
    def prompt_token_ids(self) -> list[int]:
        return self.token_ids[:self.num_prompt_tokens]

    @property
    def completion_token_ids(self) -> list[int]:
        return self.token_ids[self.num_prompt_tokens:]

    @property
    def num_cached_blocks(self) -> int:
        return self.num_cached_tokens // self.block_size

    @property
    def num_blocks(self) -> int:
        return -(-self.num_tokens // self.block_size)

    @property
    def last_block_num_tokens(self) -> int:
        return self.num_tokens - (self.num_blocks - 1) * self.block_size

    def get_token_ids_from_block(self, i: int) -> list[int]:
        assert 0 <= i < self.num_blocks
        return self.token_ids[i * self.block_size:(i + 1) * self.block_size]

    def append_token(self, token_id: int) -> None:
        self.token_ids.append(token_id)
        self.last_token = token_id
        self.num_tokens += 1

    # Workers need every token of a sequence that is about to be prefilled: the first time, and
    # again after a preemption, when the prompt is recomputed together with the tokens generated
    # so far. (The reference ships the full list only while nothing has been generated,
    # sequence.py:145-174, so its workers get a one-token list for a preempted sequence.) For a
    # decode step the last token is enough.
    def __getstate__(self):
        tail = self.token_ids if self.needs_prefill else self.last_token
        return (self.num_tokens, self.num_prompt_tokens, self.num_cached_tokens, self.block_table,
                self.needs_prefill, tail)

    def __setstate__(self, state):
        (self.num_tokens, self.num_prompt_tokens, self.num_cached_tokens, self.block_table,
         self.needs_prefill) = state[:-1]
        if self.needs_prefill:
            self.token_ids = state[-1]
            self.last_token = self.token_ids[-1]
        else:
            self.last_token = state[-1]
            self.token_ids = [self.last_token]
