"""Step-input builders (prepare_prefill / prepare_decode / prepare_block_tables) against the worked
examples in the reference's docstrings and against the restatement in oracle/host_oracle.py."""
import random

import torch

from light_vllm_amd.engine.block_manager import BlockManager
from light_vllm_amd.engine.model_runner import ModelRunner
from light_vllm_amd.engine.sequence import Sequence
from light_vllm_amd.utils.context import get_context, reset_context
from oracle.host_oracle import prepare_decode_ref, prepare_prefill_ref


def _runner(block_size):
    r = ModelRunner.__new__(ModelRunner)     # only the step-input builders are exercised
    r.block_size = block_size
    r.on_gpu = False
    r.device = torch.device("cpu")
    return r


def _ctx_lists():
    c = get_context()
    f = lambda t: None if t is None else t.tolist()
    return dict(cu_seqlens_q=f(c.cu_seqlens_q), cu_seqlens_k=f(c.cu_seqlens_k), max_seqlen_q=c.max_seqlen_q,
                max_seqlen_k=c.max_seqlen_k, slot_mapping=f(c.slot_mapping), context_lens=f(c.context_lens),
                block_tables=f(c.block_tables), is_prefill=c.is_prefill)


def test_prefill_docstring_example():
    """model_runner.py:294-304: block_size 4; seq0 = [10..14] with 4 cached tokens, seq1 = [20, 21]."""
    Sequence.block_size = 4
    try:
        s0, s1 = Sequence([10, 11, 12, 13, 14]), Sequence([20, 21])
        s0.num_cached_tokens, s0.block_table = 4, [3, 6]
        s1.block_table = [2]
        ids, pos = _runner(4).prepare_prefill([s0, s1])
        c = _ctx_lists()
        assert ids.tolist() == [14, 20, 21] and pos.tolist() == [4, 0, 1]
        assert ids.dtype == torch.int64 and pos.dtype == torch.int64
        assert c["cu_seqlens_q"] == [0, 1, 3] and c["cu_seqlens_k"] == [0, 5, 7]
        assert c["max_seqlen_q"] == 2 and c["max_seqlen_k"] == 5
        assert c["slot_mapping"] == [6 * 4 + 0, 2 * 4 + 0, 2 * 4 + 1]
        assert c["block_tables"] == [[3, 6], [2, -1]] and c["context_lens"] is None and c["is_prefill"]
        assert get_context().slot_mapping.dtype == torch.int32 and get_context().block_tables.dtype == torch.int32
    finally:
        Sequence.block_size = 256
        reset_context()


def test_slot_mapping_docstring_example():
    """model_runner.py:357-363: block_table [5, 7, 9], one cached block, 2 tokens in the last block
    -> slots [28, 29, 30, 31, 36, 37]."""
    Sequence.block_size = 4
    try:
        s = Sequence(list(range(10)))
        s.num_cached_tokens, s.block_table = 4, [5, 7, 9]
        _runner(4).prepare_prefill([s])
        assert _ctx_lists()["slot_mapping"] == [28, 29, 30, 31, 36, 37]
    finally:
        Sequence.block_size = 256
        reset_context()


def test_no_prefix_hit_means_no_block_tables():
    s = Sequence(list(range(300)))
    s.block_table = [4, 1]
    _runner(256).prepare_prefill([s])
    c = _ctx_lists()
    assert c["block_tables"] is None and c["cu_seqlens_q"] == c["cu_seqlens_k"] == [0, 300]
    assert c["slot_mapping"] == list(range(4 * 256, 5 * 256)) + list(range(256, 256 + 44))
    reset_context()


def test_decode_first_step_after_a_full_block():
    """config-1 geometry: 256-token prompt + 1 sampled token -> len 257, new block, slot offset 0,
    RoPE position len(seq) as the reference feeds it."""
    s = Sequence(list(range(256)))
    s.block_table = [7]
    s.append_token(99)
    s.block_table.append(3)
    ids, pos = _runner(256).prepare_decode([s])
    c = _ctx_lists()
    assert ids.tolist() == [99] and pos.tolist() == [257]
    assert c["context_lens"] == [257] and c["slot_mapping"] == [3 * 256] and c["block_tables"] == [[7, 3]]
    assert not c["is_prefill"]
    reset_context()


def test_random_batches_match_restatement():
    rng = random.Random(0)
    Sequence.block_size = 16
    try:
        for trial in range(20):
            bm = BlockManager(64, 16)
            shared = [rng.randrange(1000) for _ in range(32)]
            seqs = []
            for _ in range(rng.randrange(1, 6)):
                toks = (shared if rng.random() < 0.5 else []) + [rng.randrange(1000) for _ in range(rng.randrange(1, 40))]
                s = Sequence(toks)
                bm.allocate(s)
                seqs.append(s)
            r = _runner(16)
            ids, pos = r.prepare_prefill(seqs)
            want = prepare_prefill_ref(seqs, 16)
            c = _ctx_lists()
            assert ids.tolist() == want["input_ids"] and pos.tolist() == want["positions"]
            for k in ("cu_seqlens_q", "cu_seqlens_k", "max_seqlen_q", "max_seqlen_k", "slot_mapping", "block_tables"):
                assert c[k] == want[k], (trial, k)
            for s in seqs:
                s.append_token(rng.randrange(1000))
                bm.may_append(s)
            ids, pos = r.prepare_decode(seqs)
            want = prepare_decode_ref(seqs, 16)
            c = _ctx_lists()
            assert ids.tolist() == want["input_ids"] and pos.tolist() == want["positions"]
            for k in ("context_lens", "slot_mapping", "block_tables"):
                assert c[k] == want[k], (trial, k)
    finally:
        Sequence.block_size = 256
        reset_context()


def test_positions_beyond_the_rotary_table_are_refused():
    """The fused norm/RoPE kernel clamps positions instead of faulting, so the runner checks the
    range where the lengths are known. Decode feeds position len(seq) (model_runner.py:405)."""
    import pytest
    Sequence.block_size = 4
    try:
        r = _runner(4)
        r.max_position = 8
        s = Sequence(list(range(7)))
        s.block_table = [0, 1]
        r.prepare_decode([s])                       # position 7: the last row of the table
        reset_context()
        s.append_token(1)
        with pytest.raises(IndexError, match="rotary table"):
            r.prepare_decode([s])                   # position 8
        long = Sequence(list(range(9)))
        long.block_table = [0, 1, 2]
        with pytest.raises(IndexError, match="rotary table"):
            r.prepare_prefill([long])               # positions 0..8
    finally:
        Sequence.block_size = 256
        reset_context()


def test_run_refuses_tokens_after_a_failed_direct_all_reduce():
    """A direct all-reduce that gave up inside a replayed graph leaves no host error behind; the
    runner reads the communicator's failure word around every step and raises instead of handing
    tokens of such a step to the scheduler."""
    import pytest

    class _Comm:
        def __init__(self):
            self.dead = False

        def failed(self):
            return self.dead

    r = _runner(256)
    r.world_size, r.rank = 2, 0
    r.direct_comm = _Comm()
    calls = []
    r.prepare_decode = lambda seqs: (torch.zeros(1, dtype=torch.int64), torch.zeros(1, dtype=torch.int64))
    r.prepare_sample = lambda seqs: torch.zeros(1)

    def run_model(input_ids, positions, is_prefill, greedy=False):
        calls.append(greedy)
        r.direct_comm.dead = len(calls) == 2        # the second step's all-reduce times out
        return torch.tensor([3])
    r.run_model = run_model
    assert r.run([object()], False, all_greedy=True) == [3]
    with pytest.raises(RuntimeError, match="direct all-reduce timed out"):
        r.run([object()], False, all_greedy=True)
    with pytest.raises(RuntimeError, match="direct all-reduce timed out"):      # and every step after it, before any work
        r.run([object()], False, all_greedy=True)
    assert len(calls) == 2
