"""Host-logic scenarios run twice: by oracle/gen_golden.py against the reference's own classes
(to write tests/golden/host_traces.json) and by the CPU tests against this package's classes.
Each scenario returns a JSON-able trace; the two traces must be equal."""
from __future__ import annotations

from types import SimpleNamespace


def _snap_bm(bm):
    return {
        "free": list(bm.free_block_ids),
        "used": sorted(bm.used_block_ids),
        "ref": [b.ref_count for b in bm.blocks],
        "hash": [b.hash for b in bm.blocks],
        "hash_map": sorted([k, v] for k, v in bm.hash_to_block_id.items()),
    }


def _snap_seq(s):
    return {"len": len(s), "table": list(s.block_table), "cached": s.num_cached_tokens,
            "num_blocks": s.num_blocks, "last": s.last_block_num_tokens,
            "cached_blocks": s.num_cached_blocks}


def scenario_hash(ns):
    BM = ns.BlockManager
    return {
        "h_range256": BM.compute_hash(list(range(256))),
        "h_range256_prefix1": BM.compute_hash(list(range(256)), 1),
        "h_small": BM.compute_hash([5, 6, 7, 8]),
        "h_chain": BM.compute_hash([9, 10, 11, 12], BM.compute_hash([5, 6, 7, 8])),
        "h_big_ids": BM.compute_hash([151935, 0, 2**31, 17]),
    }


def scenario_geometry(ns):
    ns.Sequence.block_size = 256
    out = []
    for n in (1, 255, 256, 257, 512, 513, 1024):
        s = ns.Sequence(list(range(n)))
        out.append({"n": n, "num_blocks": s.num_blocks, "last": s.last_block_num_tokens,
                    "blk0_len": len(s.get_token_ids_from_block(0)),
                    "blkN_len": len(s.get_token_ids_from_block(s.num_blocks - 1))})
    return out


def scenario_prefix_sharing(ns):
    """8 blocks; s1 = range(600); s2 shares two full blocks; then release and revive."""
    ns.Sequence.block_size = 256
    bm = ns.BlockManager(8, 256)
    trace = []
    s1 = ns.Sequence(list(range(600)))
    s2 = ns.Sequence(list(range(512)) + [7] * 100)
    bm.allocate(s1)
    trace.append(["alloc s1", _snap_seq(s1), _snap_bm(bm)])
    bm.allocate(s2)
    trace.append(["alloc s2", _snap_seq(s2), _snap_bm(bm)])
    bm.deallocate(s1)
    trace.append(["free s1", _snap_seq(s1), _snap_bm(bm)])
    bm.deallocate(s2)
    trace.append(["free s2", _snap_seq(s2), _snap_bm(bm)])
    s3 = ns.Sequence(list(range(512)) + [9] * 30)     # revives two released blocks
    bm.allocate(s3)
    trace.append(["alloc s3", _snap_seq(s3), _snap_bm(bm)])
    s4 = ns.Sequence(list(range(256)) + [1] * 256 + list(range(512, 600)))  # hit, then miss, then never hit
    bm.allocate(s4)
    trace.append(["alloc s4", _snap_seq(s4), _snap_bm(bm)])
    s5 = ns.Sequence(list(range(512)))                # exact multiple, fully cached
    trace.append(["can_allocate s5", bm.can_allocate(s5)])
    bm.allocate(s5)
    trace.append(["alloc s5", _snap_seq(s5), _snap_bm(bm)])
    return trace


def scenario_may_append(ns):
    """Walk a sequence across two block boundaries with block_size 4 (the size the reference's
    docstring examples use)."""
    ns.Sequence.block_size = 4
    try:
        bm = ns.BlockManager(6, 4)
        s = ns.Sequence([10, 11, 12])
        bm.allocate(s)
        trace = [["alloc", _snap_seq(s), _snap_bm(bm)]]
        for tok in (13, 14, 15, 16, 17, 18, 19):
            s.append_token(tok)
            ok = bm.can_append(s)
            bm.may_append(s)
            trace.append([f"append {tok}", ok, _snap_seq(s), _snap_bm(bm)])
        other = ns.Sequence([10, 11, 12, 13, 14, 15, 16, 17, 99])
        bm.allocate(other)                      # shares the two sealed blocks
        trace.append(["alloc other", _snap_seq(other), _snap_bm(bm)])
        return trace
    finally:
        ns.Sequence.block_size = 256


def _fake_token(seq_index, length):
    return (seq_index * 7919 + length * 31 + 3) % 1000


def _run_scheduler(ns, prompts, *, num_blocks, block_size, max_num_seqs, max_num_batched_tokens,
                   max_tokens, max_steps, strict_reference=False):
    ns.Sequence.block_size = block_size
    try:
        cfg = SimpleNamespace(max_num_seqs=max_num_seqs, max_num_batched_tokens=max_num_batched_tokens,
                              eos=-1, num_kvcache_blocks=num_blocks, kvcache_block_size=block_size,
                              strict_reference=strict_reference)
        sch = ns.Scheduler(cfg)
        sp = ns.SamplingParams(temperature=0.0, max_tokens=max_tokens, ignore_eos=True)
        seqs = [ns.Sequence(p, sp) for p in prompts]
        index = {id(s): i for i, s in enumerate(seqs)}
        for s in seqs:
            sch.add(s)
        trace = []
        steps = 0
        while not sch.is_finished() and steps < max_steps:
            batch, is_prefill = sch.schedule()
            rec = {"prefill": is_prefill, "seqs": [index[id(s)] for s in batch],
                   "state": [_snap_seq(s) for s in batch],
                   "free": len(sch.block_manager.free_block_ids),
                   "waiting": [index[id(s)] for s in sch.waiting],
                   "running": [index[id(s)] for s in sch.running]}
            toks = [_fake_token(index[id(s)], len(s)) for s in batch]
            sch.postprocess(batch, toks)
            rec["finished"] = [index[id(s)] for s in batch if s.is_finished]
            trace.append(rec)
            steps += 1
        return {"steps": trace, "outputs": [list(s.completion_token_ids) for s in seqs]}
    finally:
        ns.Sequence.block_size = 256


def scenario_scheduler_config1(ns):
    """BASELINE config 1 batch shape: 4 distinct 256-token prompts, 64 greedy steps."""
    prompts = [[(i * 1000 + j * 7) % 5000 for j in range(256)] for i in range(4)]
    return _run_scheduler(ns, prompts, num_blocks=64, block_size=256, max_num_seqs=512,
                          max_num_batched_tokens=16384, max_tokens=64, max_steps=200)


def scenario_scheduler_prefix_cache(ns):
    """Shared 512-token prefix, staggered arrival through a small max_num_seqs."""
    shared = [(j * 13) % 3000 for j in range(512)]
    prompts = [shared + [i + 1] * (20 + 5 * i) for i in range(5)] + [[4000 + j for j in range(300)]]
    return _run_scheduler(ns, prompts, num_blocks=32, block_size=256, max_num_seqs=3,
                          max_num_batched_tokens=2048, max_tokens=12, max_steps=200)


def scenario_scheduler_preemption(ns):
    """KV pool too small for all sequences to finish together (block_size 4): exercises
    preemption and recompute. Uses the reference's can_append test (strict mode); sized so that
    the reference's own may_append never meets an empty pool."""
    prompts = [[100 * i + j for j in range(L)] for i, L in enumerate((5, 6, 7))]
    return _run_scheduler(ns, prompts, num_blocks=6, block_size=4, max_num_seqs=8,
                          max_num_batched_tokens=64, max_tokens=12, max_steps=400, strict_reference=True)


SCENARIOS = {
    "hash": scenario_hash,
    "geometry": scenario_geometry,
    "prefix_sharing": scenario_prefix_sharing,
    "may_append": scenario_may_append,
    "scheduler_config1": scenario_scheduler_config1,
    "scheduler_prefix_cache": scenario_scheduler_prefix_cache,
    "scheduler_preemption": scenario_scheduler_preemption,
}


def run_all(ns):
    return {name: fn(ns) for name, fn in SCENARIOS.items()}
