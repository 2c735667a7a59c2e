"""BlockManager / Sequence / Scheduler parity with the reference, on CPU.

tests/golden/host_traces.json was produced by running the same scenarios against the reference's
own classes (oracle/gen_golden.py); here they run against this package's classes."""
import json
import os
from types import SimpleNamespace

import pytest

from light_vllm_amd.engine.block_manager import BlockManager
from light_vllm_amd.engine.scheduler import Scheduler
from light_vllm_amd.engine.sequence import Sequence
from light_vllm_amd.sampling_params import SamplingParams
from tests._host_scenarios import SCENARIOS

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "host_traces.json")
NS = SimpleNamespace(BlockManager=BlockManager, Scheduler=Scheduler, Sequence=Sequence,
                     SamplingParams=SamplingParams)


@pytest.fixture(scope="module")
def golden():
    with open(GOLDEN) as f:
        return json.load(f)


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_scenario_matches_reference_trace(golden, name):
    got = json.loads(json.dumps(SCENARIOS[name](NS)))   # normalise tuples -> lists
    assert got == golden[name]


def test_known_answer_hashes():
    # values recorded from the reference's BlockManager.compute_hash (xxh64 over int64-LE ids)
    assert BlockManager.compute_hash(list(range(256))) == 5218229187174952702
    assert BlockManager.compute_hash(list(range(256)), 1) == 13893007031947601487


def test_prefix_sharing_known_answer():
    bm = BlockManager(8, 256)
    s1 = Sequence(list(range(600)))
    s2 = Sequence(list(range(512)) + [7] * 100)
    bm.allocate(s1)
    bm.allocate(s2)
    assert s1.block_table == [0, 1, 2] and s2.block_table == [0, 1, 3]
    assert s2.num_cached_tokens == 512
    assert [b.ref_count for b in bm.blocks][:5] == [2, 2, 1, 1, 0]
    assert bm.free_block_ids == [4, 5, 6, 7]
    bm.deallocate(s1)
    assert bm.free_block_ids == [4, 5, 6, 7, 2]


def test_can_append_matches_may_append_when_pool_is_empty():
    """Where the reference raises IndexError (pool empty at len % bs == 1) the default mode asks
    the scheduler to preempt instead."""
    Sequence.block_size = 4
    try:
        bm = BlockManager(1, 4)
        s = Sequence([1, 2, 3])
        bm.allocate(s)
        s.append_token(4)
        assert bm.can_append(s)          # len 4: seals the block, needs no new one
        bm.may_append(s)
        s.append_token(5)
        assert not bm.can_append(s)      # len 5 needs a block and none is free
        strict = BlockManager(1, 4, strict_reference=True)
        t = Sequence([1, 2, 3])
        strict.allocate(t)
        t.append_token(4)
        assert not strict.can_append(t)  # the reference reserves one step early
    finally:
        Sequence.block_size = 256


def test_sequence_pickle_state_round_trip():
    import pickle
    s = Sequence([5, 6, 7], SamplingParams(temperature=0.0, max_tokens=4))
    s.block_table = [3]
    t = pickle.loads(pickle.dumps(s))
    assert t.token_ids == [5, 6, 7] and t.block_table == [3] and len(t) == 3
    s.append_token(9)
    t = pickle.loads(pickle.dumps(s))
    assert t.last_token == 9 and len(t) == 4 and t.num_completion_tokens == 1
