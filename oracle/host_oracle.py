"""Plain-list restatement of the reference's step-input builders, ModelRunner.prepare_block_tables /
prepare_prefill / prepare_decode (lightvllm/engine/model_runner.py:256-269, 271-381, 383-418),
without pin_memory/.cuda (those make the originals unrunnable without a GPU). TEST INFRASTRUCTURE:
used by oracle/gen_e2e_golden.py to drive the reference model and by tests as the checker for
light_vllm_amd.engine.model_runner. Pinned by the worked examples in the reference's docstrings
(model_runner.py:294-304, 357-363), which tests/test_prepare_cpu.py replays."""
from __future__ import annotations


def block_tables_ref(seqs):
    width = max(len(s.block_table) for s in seqs)
    return [list(s.block_table) + [-1] * (width - len(s.block_table)) for s in seqs]


def prepare_prefill_ref(seqs, block_size):
    ids, pos, cq, ck, slots = [], [], [0], [0], []
    mq = mk = 0
    for s in seqs:
        n = len(s)
        ids.extend(s[s.num_cached_tokens:])
        pos.extend(range(s.num_cached_tokens, n))
        lq = n - s.num_cached_tokens
        cq.append(cq[-1] + lq)
        ck.append(ck[-1] + n)
        mq, mk = max(mq, lq), max(mk, n)
        if not s.block_table:
            continue
        for i in range(s.num_cached_blocks, s.num_blocks):
            start = s.block_table[i] * block_size
            end = start + (block_size if i != s.num_blocks - 1 else s.last_block_num_tokens)
            slots.extend(range(start, end))
    bt = block_tables_ref(seqs) if ck[-1] > cq[-1] else None
    return dict(input_ids=ids, positions=pos, cu_seqlens_q=cq, cu_seqlens_k=ck, max_seqlen_q=mq,
                max_seqlen_k=mk, slot_mapping=slots, block_tables=bt)


def prepare_decode_ref(seqs, block_size):
    return dict(input_ids=[s.last_token for s in seqs],
                positions=[len(s) for s in seqs],            # len(seq), not len-1 (model_runner.py:405)
                context_lens=[len(s) for s in seqs],
                slot_mapping=[s.block_table[-1] * block_size + s.last_block_num_tokens - 1 for s in seqs],
                block_tables=block_tables_ref(seqs))
