"""Tensor-parallel path on CPU: world_size-2 gloo process groups (one process per rank).

  * layer level: the head sharding of QKVParallelLinear, the all-reduce of RowParallelLinear, the
    masked all-reduce of VocabParallelEmbedding and the rank-0 gather of ParallelLMHead reproduce
    the unsharded layers (reference: layers/linear.py:240-395, layers/embed_head.py:18-197);
  * engine level: a TP=2 run (rank 0 + one worker driven over the shared-memory step RPC, kernels
    substituted by the oracle) generates the same greedy tokens as TP=1.
"""
import os
import socket
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tiny_config(dtype=torch.float32):
    return SimpleNamespace(
        vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
        num_attention_heads=4, num_key_value_heads=2, head_dim=128, hidden_act="silu",
        max_position_embeddings=1024, rms_norm_eps=1e-6, rope_theta=1000000, rope_scaling=None,
        attention_bias=False, tie_word_embeddings=False, torch_dtype=dtype)


# ------------------------------------------------------------------ layer level
def _layers_worker(rank, world, port, out_dir):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    from light_vllm_amd.layers.embed_head import ParallelLMHead, VocabParallelEmbedding
    from light_vllm_amd.layers.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
    from light_vllm_amd.utils.context import reset_context, set_context
    g = torch.Generator().manual_seed(0)
    E, D, Hq, Hkv, V, I = 64, 16, 4, 2, 40, 96
    x = torch.randn(5, E, generator=g)
    wq, wk, wv = (torch.randn(n * D, E, generator=g) for n in (Hq, Hkv, Hkv))
    qkv = QKVParallelLinear(E, D, Hq, Hkv)
    for w, sid in ((wq, "q"), (wk, "k"), (wv, "v")):
        qkv.weight.weight_loader(qkv.weight, w, sid)
    y = qkv(x)
    q, k, v = y.split([Hq // world * D, Hkv // world * D, Hkv // world * D], dim=-1)
    full = (x @ wq.T, x @ wk.T, x @ wv.T)
    for part, ref, n in ((q, full[0], Hq), (k, full[1], Hkv), (v, full[2], Hkv)):
        w_ = n // world * D
        assert torch.allclose(part, ref[:, rank * w_:(rank + 1) * w_], atol=1e-5)

    wo = torch.randn(E, Hq * D, generator=g)
    o_in = torch.randn(5, Hq * D, generator=g)
    row = RowParallelLinear(Hq * D, E)
    row.weight.weight_loader(row.weight, wo)
    shard = Hq * D // world
    y = row(o_in[:, rank * shard:(rank + 1) * shard])
    assert torch.allclose(y, o_in @ wo.T, atol=1e-4)            # identical on every rank after all-reduce

    wg, wu = torch.randn(I, E, generator=g), torch.randn(I, E, generator=g)
    gu = MergedColumnParallelLinear(E, [I, I])
    gu.weight.weight_loader(gu.weight, wg, 0)
    gu.weight.weight_loader(gu.weight, wu, 1)
    a, b = gu(x).chunk(2, -1)
    s = I // world
    assert torch.allclose(a, (x @ wg.T)[:, rank * s:(rank + 1) * s], atol=1e-5)
    assert torch.allclose(b, (x @ wu.T)[:, rank * s:(rank + 1) * s], atol=1e-5)

    we = torch.randn(V, E, generator=g)
    emb = VocabParallelEmbedding(V, E)
    emb.weight.weight_loader(emb.weight, we)
    ids = torch.tensor([0, V // world - 1, V // world, V - 1, 3])
    assert torch.equal(emb(ids), we[ids])

    head = ParallelLMHead(V, E)
    head.weight.weight_loader(head.weight, we)
    set_context(True, cu_seqlens_q=torch.tensor([0, 2, 5], dtype=torch.int32))
    logits = head(x)
    reset_context()
    if rank == 0:
        assert torch.allclose(logits, x[[1, 4]] @ we.T, atol=1e-5)
    else:
        assert logits is None
    # sharded greedy pick: per-shard maxima + one (value, index) pair per sequence and rank; every
    # rank gets torch.argmax of the full logits, also when two shards hold the same maximum
    set_context(True, cu_seqlens_q=torch.tensor([0, 2, 5], dtype=torch.int32))
    picked = head.greedy_tokens(x)
    reset_context()
    assert torch.equal(picked, (x[[1, 4]] @ we.T).argmax(-1))
    we_tie = we.clone()
    we_tie[V // world + 3] = we_tie[5]                       # the same row in both shards
    xt = torch.stack([we_tie[5] * 3, we_tie[V - 1] * 3, we_tie[5] * 3])
    head_tie = ParallelLMHead(V, E)
    head_tie.weight.weight_loader(head_tie.weight, we_tie)
    set_context(False)
    picked = head_tie.greedy_tokens(xt)
    reset_context()
    full = xt @ we_tie.T
    assert full[0, 5] == full[0, V // world + 3]            # a genuine tie across shards
    assert torch.equal(picked, full.argmax(-1)) and picked[0].item() == 5
    open(os.path.join(out_dir, f"layers_ok_{rank}"), "w").close()
    dist.destroy_process_group()


def test_tp_layers_world2(tmp_path):
    mp.spawn(_layers_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"layers_ok_{r}").exists() for r in range(2))


# ------------------------------------------------------------------ engine level
def _prompts(lens=(300, 40, 257)):
    g = torch.Generator().manual_seed(7)
    return [torch.randint(0, 512, (n,), generator=g).tolist() for n in lens]


def _generate(runner, cfg, max_tokens, lens=(300, 40, 257), trace=None, all_greedy=False):
    from light_vllm_amd.engine.scheduler import Scheduler
    from light_vllm_amd.engine.sequence import Sequence
    from light_vllm_amd.sampling_params import SamplingParams
    sch = Scheduler(cfg)
    sp = SamplingParams(temperature=0.0, max_tokens=max_tokens, ignore_eos=True)
    seqs = [Sequence(p, sp) for p in _prompts(lens)]
    for s in seqs:
        sch.add(s)
    while not sch.is_finished():
        batch, is_prefill = sch.schedule()
        if trace is not None:
            trace.append((is_prefill, [(seqs.index(s), s.num_completion_tokens) for s in batch]))
        toks = runner.call("run", batch, is_prefill, all_greedy)
        sch.postprocess(batch, toks)
    return [list(s.completion_token_ids) for s in seqs]


def _engine_worker(rank, world, port, event, out_dir):
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from tests._cpu_kernels import cpu_kernels
    cfg = Config(None, hf_config=_tiny_config(), device="cpu", tensor_parallel_size=world, num_kvcache_blocks=8,
                 max_model_len=512, max_num_batched_tokens=1024, random_seed=3,
                 dist_init_method=f"tcp://127.0.0.1:{port}")
    with cpu_kernels():
        if rank == 0:
            runner = ModelRunner(cfg, 0, [event])
            out = _generate(runner, cfg, max_tokens=6)                        # logits gathered to rank 0
            out_sharded = _generate(runner, cfg, max_tokens=6, all_greedy=True)  # per-shard maxima exchanged
            runner.call("exit")
            torch.save(out, os.path.join(out_dir, "tp2.pt"))
            torch.save(out_sharded, os.path.join(out_dir, "tp2_sharded.pt"))
        else:
            ModelRunner(cfg, rank, event)      # serves the step RPC until "exit"


def test_tp2_engine_matches_tp1(tmp_path):
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from tests._cpu_kernels import cpu_kernels
    ctx = mp.get_context("spawn")
    event = ctx.Event()
    port = _free_port()
    procs = [ctx.Process(target=_engine_worker, args=(r, 2, port, event, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    tp2 = torch.load(os.path.join(str(tmp_path), "tp2.pt"))
    cfg = Config(None, hf_config=_tiny_config(), device="cpu", num_kvcache_blocks=8, max_model_len=512,
                 max_num_batched_tokens=1024, random_seed=3)
    with cpu_kernels():
        runner = ModelRunner(cfg, 0, [])
        tp1 = _generate(runner, cfg, max_tokens=6)
    assert tp2 == tp1
    assert torch.load(os.path.join(str(tmp_path), "tp2_sharded.pt")) == tp1


# ------------------------------------------------------------------ preemption under TP
# Three one-block prompts in a three-block pool: when the first sequence crosses its block boundary
# nothing is free, the scheduler preempts the last running sequence, and that sequence is later
# prefilled again together with the tokens it had generated. The worker rank must then receive all
# of those tokens over the step RPC (a one-token list would shape-error in its prefill and leave
# rank 0 waiting in the next all-reduce).
_PREEMPT_LENS = (250, 40, 250)


def _preempt_worker(rank, world, port, event, out_dir):
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from tests._cpu_kernels import cpu_kernels
    cfg = Config(None, hf_config=_tiny_config(), device="cpu", tensor_parallel_size=world, num_kvcache_blocks=3,
                 max_model_len=512, max_num_batched_tokens=1024, random_seed=3,
                 dist_init_method=f"tcp://127.0.0.1:{port}")
    with cpu_kernels():
        if rank == 0:
            runner = ModelRunner(cfg, 0, [event])
            trace = []
            out = _generate(runner, cfg, max_tokens=10, lens=_PREEMPT_LENS, trace=trace)
            runner.call("exit")
            torch.save((out, trace), os.path.join(out_dir, "tp2_preempt.pt"))
        else:
            ModelRunner(cfg, rank, event)


def test_tp2_engine_preemption_reprefills_generated_tokens(tmp_path):
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from tests._cpu_kernels import cpu_kernels
    ctx = mp.get_context("spawn")
    event = ctx.Event()
    port = _free_port()
    procs = [ctx.Process(target=_preempt_worker, args=(r, 2, port, event, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        if p.is_alive():
            p.kill()
        assert p.exitcode == 0
    tp2, trace = torch.load(os.path.join(str(tmp_path), "tp2_preempt.pt"))
    # a sequence that had already generated tokens went through a prefill step again
    assert any(is_prefill and any(done > 0 for _, done in batch) for is_prefill, batch in trace), trace
    cfg = Config(None, hf_config=_tiny_config(), device="cpu", num_kvcache_blocks=3, max_model_len=512,
                 max_num_batched_tokens=1024, random_seed=3)
    with cpu_kernels():
        runner = ModelRunner(cfg, 0, [])
        tp1 = _generate(runner, cfg, max_tokens=10, lens=_PREEMPT_LENS)
    assert tp2 == tp1


def test_sequence_pickle_ships_all_tokens_until_prefilled():
    import pickle
    from light_vllm_amd.engine.sequence import Sequence
    s = Sequence([5, 6, 7])
    assert pickle.loads(pickle.dumps(s)).token_ids == [5, 6, 7]
    s.append_token(8)
    s.needs_prefill = False                      # what Scheduler.postprocess does
    w = pickle.loads(pickle.dumps(s))
    assert w.token_ids == [8] and w.last_token == 8 and len(w) == 4
    s.needs_prefill = True                       # what Scheduler.preempt does
    w = pickle.loads(pickle.dumps(s))
    assert w.token_ids == [5, 6, 7, 8] and len(w) == 4
