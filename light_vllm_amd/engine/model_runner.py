"""Per-GPU worker: owns the model shard, the paged KV tensor and the per-step Context.

Same responsibilities and step protocol as the reference's ModelRunner
(lightvllm/engine/model_runner.py:17-548); the parts the attention path depends on are re-done for
MI355X:

  * prepare_prefill / prepare_decode / prepare_block_tables build exactly the reference's tensors
    (:256-418) — input_ids, positions (decode uses position len(seq), :405), cu_seqlens_q/k,
    slot_mapping, context_lens, -1 padded block tables — through one pinned staging buffer and
    asynchronous H2D copies;
  * allocate_kv_cache sizes the cache from free HBM with the reference's formula (:216-236) in
    64-bit arithmetic (a 288 GB device holds > 2^31 elements per layer) and binds
    kv_cache[0, i] / kv_cache[1, i] to every module that has k_cache and v_cache (:249-254);
  * decode runs from hipGraphs captured per batch-size bucket (:489-548). Padding rows of a bucket
    carry slot -1, which lv_store_kvcache skips, and context_len 0, which lv_decode_attn turns
    into a zero row. (The reference pads with slot 0 and so overwrites cache slot 0 on every
    padded replay, :448-455.) The split-KV workspace (decode and prefill) is reserved before the KV
    cache is sized, so nothing allocates once serving has begun;
  * a decode batch with a few long contexts among short ones is served from a host-built work list
    (decode_plan.py) that cuts the long ones into pieces: lengths are python ints here (:404-408),
    where the kernels alone could only give every sequence the same split count.
"""
from __future__ import annotations

import pickle
import re
from multiprocessing.shared_memory import SharedMemory

import torch
import torch.distributed as dist

from .. import ops
from ..config import Config
from ..layers.sampler import Sampler
from ..models.qwen3 import Qwen3ForCausalLM
from ..utils.context import get_context, reset_context, set_context
from ..utils.loader import load_model, load_random, load_random_on_device
from .sequence import Sequence


def _shm_name(init_method: str) -> str:
    """Name of the step-RPC segment, derived from the rendezvous address so that two engines on
    one host (different ports) never share or unlink each other's segment."""
    return "lightvllm_amd_" + re.sub(r"[^0-9A-Za-z]+", "_", init_method).strip("_")


class ModelRunner:

    def __init__(self, config: Config, rank: int, event):
        self.config = config
        hf_config = config.hf_config
        self.block_size = config.kvcache_block_size
        self.max_position = hf_config.max_position_embeddings
        Sequence.set_block_size(self.block_size)
        self.enforce_eager = config.enforce_eager
        self.world_size = config.tensor_parallel_size
        self.rank = rank
        self.event = event
        self.on_gpu = config.device != "cpu"

        if self.on_gpu:
            self.device = torch.device(config.device if ":" in config.device else f"cuda:{rank}")
            torch.cuda.set_device(self.device)
        else:
            self.device = torch.device("cpu")
        self._own_group = False
        if self.world_size > 1 and not dist.is_initialized():
            backend = config.dist_backend or ("nccl" if self.on_gpu else "gloo")   # "nccl" is RCCL on ROCm
            kwargs = {"device_id": self.device} if backend == "nccl" else {}
            dist.init_process_group(backend, config.dist_init_method, world_size=self.world_size,
                                    rank=rank, **kwargs)
            self._own_group = True
        # Decode-sized all-reduces go over the peers' IPC-mapped buffers (one kernel, capturable into
        # the decode hipGraph); RCCL keeps the prefill-sized ones. Collective: all ranks or none.
        self.direct_comm = None
        if self.world_size > 1 and self.on_gpu and config.direct_all_reduce:
            from .. import comm as lvcomm
            # a minute of grace: ranks reach their first all-reduce (the warmup prefill) as unevenly
            # as their weights load
            self.direct_comm = lvcomm.create_for_group(self.device, timeout_ms=60000)
            lvcomm.install(self.direct_comm)

        default_dtype = torch.get_default_dtype()
        dtype = getattr(hf_config, "torch_dtype", None) or getattr(hf_config, "dtype", None) or torch.bfloat16
        self.dtype = dtype
        torch.set_default_dtype(dtype)
        torch.set_default_device(self.device)
        try:
            self.model = Qwen3ForCausalLM(hf_config)
            if config.random_seed is not None and config.random_on_device:
                load_random_on_device(self.model, config.random_seed)
            elif config.random_seed is not None:
                load_random(self.model, hf_config, config.random_seed, kind=config.random_kind)
            else:
                load_model(self.model, config.model)
            self.sampler = Sampler()
            self.kv_cache = None
            if self.world_size > 1:
                dist.barrier()          # every rank has its weights: the first collective starts together
            self.warmup_model()
            self.reserve_workspace()
            self.allocate_kv_cache()
            if not self.enforce_eager:
                self.capture_graphs()
        finally:
            torch.set_default_device("cpu")
            torch.set_default_dtype(default_dtype)

        if self.world_size > 1:
            name = _shm_name(config.dist_init_method)
            if rank == 0:
                try:
                    self.shm = SharedMemory(name=name, create=True, size=2 ** 20)
                except FileExistsError:
                    # Same rendezvous address, which this process group has just bound: the segment
                    # can only be left over from a run that died. Replace it.
                    stale = SharedMemory(name=name)
                    stale.close()
                    stale.unlink()
                    self.shm = SharedMemory(name=name, create=True, size=2 ** 20)
                dist.barrier()
            else:
                dist.barrier()
                self.shm = SharedMemory(name=name)
                self.loop()

    # ------------------------------------------------------------------ lifecycle / step RPC
    def exit(self):
        if self.world_size > 1:
            self.shm.close()
            dist.barrier()
            if self.rank == 0:
                self.shm.unlink()
        if not self.enforce_eager:
            del self.graphs, self.plan_graphs, self.graph_pool
        if self.on_gpu:
            torch.cuda.synchronize()
        if self.direct_comm is not None:
            from .. import comm as lvcomm
            failed = self.direct_comm.failed()
            dist.barrier()                 # no peer is still reading this rank's buffers
            lvcomm.install(None)
            self.direct_comm.close()
            self.direct_comm = None
            assert not failed, "a direct all-reduce timed out waiting for a peer during this run"
        if self._own_group:
            dist.destroy_process_group()

    def loop(self):
        while True:
            method_name, args = self.read_shm()
            self.call(method_name, *args)
            if method_name == "exit":
                break

    def read_shm(self):
        assert self.world_size > 1 and self.rank > 0
        self.event.wait()
        n = int.from_bytes(self.shm.buf[0:4], "little")
        method_name, *args = pickle.loads(self.shm.buf[4:n + 4])
        self.event.clear()
        return method_name, args

    def write_shm(self, method_name, *args):
        assert self.world_size > 1 and self.rank == 0
        data = pickle.dumps([method_name, *args])
        n = len(data)
        assert n + 4 <= self.shm.size, "step message exceeds the shared-memory segment"
        self.shm.buf[0:4] = n.to_bytes(4, "little")
        self.shm.buf[4:n + 4] = data
        for event in self.event:
            event.set()

    def call(self, method_name, *args):
        if self.world_size > 1 and self.rank == 0:
            self.write_shm(method_name, *args)
        method = getattr(self, method_name, None)
        if method:
            return method(*args)

    # ------------------------------------------------------------------ memory
    def warmup_model(self):
        """One prefill of the largest batch with empty caches (Attention skips the store),
        so the peak activation footprint is known before the KV cache is sized."""
        if self.on_gpu:
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
        cfg = self.config
        num_seqs = min(cfg.max_num_batched_tokens // cfg.max_model_len, cfg.max_num_seqs)
        seqs = [Sequence([0] * cfg.max_model_len) for _ in range(num_seqs)]
        self.run(seqs, True)
        if self.on_gpu:
            torch.cuda.empty_cache()

    def _head_geometry(self):
        hf = self.config.hf_config
        hq = hf.num_attention_heads // self.world_size
        hkv = hf.num_key_value_heads // self.world_size
        head_dim = getattr(hf, "head_dim", None) or hf.hidden_size // hf.num_attention_heads
        return hq, hkv, head_dim

    def decode_batch_buckets(self):
        max_bs = min(self.config.max_num_seqs, 512)
        return [b for b in (1, 2, 4, 8) if b <= max_bs] + list(range(16, max_bs + 1, 16))

    def plan_row_splits(self) -> int:
        """Most pieces a work list cuts one sequence into (= split capacity of a workspace row)."""
        from ..decode_plan import MIN_PIECE
        return max(2, min(64, -(-self.config.max_model_len // MIN_PIECE)))

    def reserve_workspace(self) -> None:
        """One split-KV workspace for everything this runner will launch, reserved BEFORE the KV cache
        takes the rest of the device: decode at every batch size (the split count the library picks
        for that shape, and the row capacity of a work list), and prefill at max_num_batched_tokens
        under the split count the library's heuristic can pick. Nothing grows mid-serving, and a
        captured graph's workspace address stays valid."""
        if not self.on_gpu:
            return
        cfg = self.config
        hq, hkv, head_dim = self._head_geometry()
        max_num_blocks = -(-cfg.max_model_len // self.block_size)
        sizes = sorted(set(self.decode_batch_buckets() + [cfg.max_num_seqs]))
        need = max(ops.decode_workspace_bytes(b, hq, head_dim, ops.decode_num_splits(b, hkv, max_num_blocks, self.block_size))
                   for b in sizes)
        need = max(need, ops.decode_workspace_bytes(max(sizes), hq, head_dim, self.plan_row_splits()))
        need = max(need, ops._lib.load().lv_prefill_max_workspace_bytes(cfg.max_num_batched_tokens, hq, head_dim))
        ops.reserve_decode_workspace(self.device, need)
        self.workspace_bytes = need

    def kv_block_bytes(self) -> int:
        hf = self.config.hf_config
        num_kv_heads = hf.num_key_value_heads // self.world_size
        head_dim = getattr(hf, "head_dim", None) or hf.hidden_size // hf.num_attention_heads
        itemsize = torch.empty((), dtype=self.dtype).element_size()
        return 2 * hf.num_hidden_layers * self.block_size * num_kv_heads * head_dim * itemsize

    def allocate_kv_cache(self):
        cfg = self.config
        hf = cfg.hf_config
        block_bytes = self.kv_block_bytes()
        if self.on_gpu:
            free, total = torch.cuda.mem_get_info()
            used = total - free
            stats = torch.cuda.memory_stats()
            peak = stats["allocated_bytes.all.peak"]
            current = stats["allocated_bytes.all.current"]
            available = int(total * cfg.gpu_memory_utilization) - used - peak + current
            num_blocks = available // block_bytes
            # what the count was computed from, kept for logs and tests
            self.kv_sizing = dict(total=total, used=used, peak=peak, current=current, block_bytes=block_bytes,
                                  utilization=cfg.gpu_memory_utilization, blocks_from_memory=int(num_blocks),
                                  cap=cfg.num_kvcache_blocks)
            if cfg.num_kvcache_blocks > 0:
                num_blocks = min(num_blocks, cfg.num_kvcache_blocks)
        else:
            assert cfg.num_kvcache_blocks > 0, "CPU plumbing runs need an explicit num_kvcache_blocks"
            num_blocks = cfg.num_kvcache_blocks
        assert num_blocks > 0, "not enough memory for the KV cache"
        cfg.num_kvcache_blocks = int(num_blocks)
        num_kv_heads = hf.num_key_value_heads // self.world_size
        head_dim = getattr(hf, "head_dim", None) or hf.hidden_size // hf.num_attention_heads
        self.kv_cache = torch.zeros(2, hf.num_hidden_layers, cfg.num_kvcache_blocks, self.block_size,
                                    num_kv_heads, head_dim, dtype=self.dtype, device=self.device)
        layer_id = 0
        for module in self.model.modules():
            if hasattr(module, "k_cache") and hasattr(module, "v_cache"):
                module.k_cache = self.kv_cache[0, layer_id]
                module.v_cache = self.kv_cache[1, layer_id]
                layer_id += 1
        assert layer_id == hf.num_hidden_layers

    # ------------------------------------------------------------------ step inputs
    def _check_positions(self, last_position: int) -> None:
        """Positions index the rotary table (max_position_embeddings rows). The fused kernel clamps
        instead of faulting, so the range is enforced here, where the lengths are known."""
        limit = getattr(self, "max_position", None)     # None: a bare runner built for its step-input builders
        if limit is not None and last_position >= limit:
            raise IndexError(f"position {last_position} is outside the rotary table of {limit} rows "
                             "(sequence longer than max_position_embeddings)")

    def _to_device(self, values, dtype) -> torch.Tensor:
        if self.on_gpu:
            return torch.tensor(values, dtype=dtype, pin_memory=True).cuda(self.device, non_blocking=True)
        return torch.tensor(values, dtype=dtype)

    def prepare_block_tables(self, seqs: list[Sequence]) -> torch.Tensor:
        width = max(len(seq.block_table) for seq in seqs)
        rows = [seq.block_table + [-1] * (width - len(seq.block_table)) for seq in seqs]
        return self._to_device(rows, torch.int32)

    def prepare_prefill(self, seqs: list[Sequence]):
        input_ids, positions, slot_mapping = [], [], []
        cu_q, cu_k = [0], [0]
        lens_q, lens_k = [], []
        max_q = max_k = 0
        bs = self.block_size
        for seq in seqs:
            n = len(seq)
            cached = seq.num_cached_tokens
            input_ids.extend(seq[cached:])
            positions.extend(range(cached, n))
            lq, lk = n - cached, n
            lens_q.append(lq)
            lens_k.append(lk)
            cu_q.append(cu_q[-1] + lq)
            cu_k.append(cu_k[-1] + lk)
            max_q, max_k = max(max_q, lq), max(max_k, lk)
            if not seq.block_table:      # warmup: nothing to store
                continue
            last = seq.num_blocks - 1
            for i in range(seq.num_cached_blocks, seq.num_blocks):
                start = seq.block_table[i] * bs
                count = bs if i != last else seq.last_block_num_tokens
                slot_mapping.extend(range(start, start + count))
        self._check_positions(max_k - 1)
        block_tables = self.prepare_block_tables(seqs) if cu_k[-1] > cu_q[-1] else None
        input_ids = self._to_device(input_ids, torch.int64)
        positions = self._to_device(positions, torch.int64)
        cu_q_t = self._to_device(cu_q, torch.int32)
        cu_k_t = self._to_device(cu_k, torch.int32)
        slot_mapping = self._to_device(slot_mapping, torch.int32)
        # launch order of the query tiles for the prefill kernels: on a tensor-parallel shard with
        # fewer than 8 kv heads it keeps each sequence's K/V stream on one or two XCDs; for long prompts
        # on a grid larger than the chip, the balanced tile lists of the persistent kernel (None otherwise)
        tile_order = None
        if self.on_gpu and max_q > 0:
            hf = self.config.hf_config
            tile_order = ops.prefill_tile_order(lens_q, lens_k, hf.num_attention_heads // self.world_size,
                                                hf.num_key_value_heads // self.world_size, self.device,
                                                paged=block_tables is not None)
        set_context(True, cu_q_t, cu_k_t, max_q, max_k, slot_mapping, None, block_tables, tile_order=tile_order)
        return input_ids, positions

    def prepare_decode(self, seqs: list[Sequence]):
        bs = self.block_size
        input_ids = [seq.last_token for seq in seqs]
        positions = [len(seq) for seq in seqs]
        context_lens = context_lens_host = [len(seq) for seq in seqs]
        slot_mapping = [seq.block_table[-1] * bs + seq.last_block_num_tokens - 1 for seq in seqs]
        self._check_positions(max(positions))
        input_ids = self._to_device(input_ids, torch.int64)
        positions = self._to_device(positions, torch.int64)
        slot_mapping = self._to_device(slot_mapping, torch.int32)
        context_lens = self._to_device(context_lens, torch.int32)
        block_tables = self.prepare_block_tables(seqs)
        # launch order for the decode kernel: longest sequences first (stable), so a ragged batch
        # does not end on a long straggler; the host already has the lengths. Batches of nearly
        # equal lengths are faster in the kernel's default order (no order passed).
        seq_order = None
        plan = self._decode_plan(context_lens_host)
        if plan is None and max(context_lens_host) - min(context_lens_host) > max(context_lens_host) // 4:
            order = sorted(range(len(seqs)), key=lambda j: -len(seqs[j]))
            seq_order = self._to_device(order, torch.int32)
        set_context(False, slot_mapping=slot_mapping, context_lens=context_lens, block_tables=block_tables,
                    seq_order=seq_order, decode_plan=plan)
        return input_ids, positions

    def _decode_plan(self, lens: list[int]):
        """A work list for this step's decode attention when the batch holds long outliers (a few
        contexts far longer than the rest): those are cut into pieces, everything runs longest
        piece first. None otherwise. For a step that will replay a captured graph the list is built
        for the graph's bucket: padding rows (length 0) included, capacity and row splits the ones
        the graph was captured with."""
        if not self.on_gpu or getattr(self, "config", None) is None:
            return None
        from ..decode_plan import build_plan, worth_planning
        hq, hkv, _ = self._head_geometry()
        if not worth_planning(lens, hkv, hq // hkv):
            return None
        bs = len(lens)
        if self.enforce_eager or not hasattr(self, "graph_bs") or bs > self.graph_bs[-1]:
            return build_plan(lens, hkv, hq // hkv, max_splits=self.plan_row_splits()).to(self.device)
        bucket = next(x for x in self.graph_bs if x >= bs)
        if bucket not in self.plan_graphs:
            return None
        return build_plan(lens + [0] * (bucket - bs), hkv, hq // hkv, max_splits=self.plan_row_splits(),
                          capacity=bucket + self.plan_extra_items)

    def prepare_sample(self, seqs: list[Sequence]) -> torch.Tensor:
        return self._to_device([seq.temperature for seq in seqs], torch.float32)

    # ------------------------------------------------------------------ forward
    @torch.inference_mode()
    def run_model(self, input_ids: torch.Tensor, positions: torch.Tensor, is_prefill: bool, greedy: bool = False):
        """Logits of the step (rank 0; None elsewhere), or with `greedy` its argmax tokens (every rank)."""
        head = self.model.greedy_tokens if greedy else self.model.compute_logits
        if is_prefill or self.enforce_eager or input_ids.size(0) > self.graph_bs[-1]:
            return head(self.model(input_ids, positions))
        bs = input_ids.size(0)
        context = get_context()
        bucket = next(x for x in self.graph_bs if x >= bs)
        gv = self.graph_vars
        gv["input_ids"].zero_()
        gv["positions"].zero_()
        gv["slot_mapping"].fill_(-1)      # padding rows store nothing
        gv["context_lens"].zero_()        # ... and attend to nothing
        gv["block_tables"].zero_()
        gv["input_ids"][:bs] = input_ids
        gv["positions"][:bs] = positions
        gv["slot_mapping"][:bs] = context.slot_mapping
        gv["context_lens"][:bs] = context.context_lens
        gv["block_tables"][:bs, :context.block_tables.size(1)] = context.block_tables
        plan = context.decode_plan
        if plan is not None:
            # host lists (built for this bucket in prepare_decode) -> the buffers the second graph of
            # the bucket was captured on
            gv["plan_items"][:2 + 2 * plan.capacity] = self._to_device(plan.items, torch.int32)
            gv["plan_seq_splits"][:bucket] = self._to_device(plan.seq_splits, torch.int32)
            self.plan_graphs[bucket].replay()
            return head(gv["outputs"][:bs])
        if context.seq_order is not None:
            gv["seq_order"][:bucket] = self._iota[:bucket]  # padding rows keep their own index ...
            gv["seq_order"][:bs] = context.seq_order        # ... behind the real rows, longest first
        else:
            gv["seq_order"][:1] = -1                        # nearly equal lengths: the kernel's default order
        self.graphs[bucket].replay()
        return head(gv["outputs"][:bs])

    def _check_direct_comm(self) -> None:
        """A direct all-reduce whose peer never arrives gives up inside its kernel (bounded wait) and
        leaves its output undefined; inside a replayed hipGraph no host call is there to return the
        error. The failure word is host-mapped, so reading it costs no device sync: every rank reads
        it around every step and refuses to hand tokens from such a step to the scheduler."""
        if self.direct_comm is not None and self.direct_comm.failed():
            raise RuntimeError("a direct all-reduce timed out waiting for a peer: this step's output is undefined "
                               "(tensor-parallel rank %d)" % self.rank)

    def run(self, seqs: list[Sequence], is_prefill: bool, all_greedy: bool = False):
        """all_greedy: rank 0 knows every sequence of the step samples at temperature 0 and says so
        to the other ranks through this argument (sampling parameters do not travel with the step's
        sequences). Under tensor parallelism the ranks then pick the token from per-shard maxima
        instead of gathering [batch, vocab] logits to rank 0 (ParallelLMHead.greedy_tokens): the
        same token, 16 bytes per sequence and rank on the wire instead of 2 * vocab / tp."""
        self._check_direct_comm()       # ranks that never sync see the steps before this one here
        input_ids, positions = self.prepare_prefill(seqs) if is_prefill else self.prepare_decode(seqs)
        if all_greedy and self.world_size > 1:
            tokens = self.run_model(input_ids, positions, is_prefill, greedy=True)
            reset_context()
            token_ids = tokens.tolist() if self.rank == 0 else None     # rank 0: the step's host sync
            self._check_direct_comm()
            return token_ids
        temperatures = self.prepare_sample(seqs) if self.rank == 0 else None
        logits = self.run_model(input_ids, positions, is_prefill)
        token_ids = None
        if self.rank == 0:      # only rank 0's sequences carry their sampling parameters
            all_greedy = all(seq.temperature == 0 for seq in seqs)
            token_ids = self.sampler(logits, temperatures, all_greedy).tolist()
        reset_context()
        self._check_direct_comm()       # after rank 0's host sync: covers this step's all-reduces
        return token_ids

    @torch.inference_mode()
    def capture_graphs(self):
        """hipGraph capture of the decode forward pass for batch buckets 1, 2, 4, 8, 16, 32, ...
        Largest first so smaller graphs reuse its memory pool."""
        assert self.on_gpu
        cfg, hf = self.config, self.config.hf_config
        max_bs = min(cfg.max_num_seqs, 512)
        max_num_blocks = -(-cfg.max_model_len // self.block_size)
        dev = self.device
        input_ids = torch.zeros(max_bs, dtype=torch.int64, device=dev)
        positions = torch.zeros(max_bs, dtype=torch.int64, device=dev)
        slot_mapping = torch.full((max_bs,), -1, dtype=torch.int32, device=dev)
        context_lens = torch.zeros(max_bs, dtype=torch.int32, device=dev)
        block_tables = torch.zeros(max_bs, max_num_blocks, dtype=torch.int32, device=dev)
        outputs = torch.zeros(max_bs, hf.hidden_size, dtype=self.dtype, device=dev)
        self._iota = torch.arange(max_bs, dtype=torch.int32, device=dev)
        seq_order = self._iota.clone()
        self.graph_bs = self.decode_batch_buckets()
        self.graphs = {}
        self.graph_pool = None
        # A second graph per bucket (from 4 sequences up) runs the decode attention from a work list:
        # the step's list and per-sequence split counts are written into these buffers before the
        # replay (decode_plan.py). Capacity: one item per row of the bucket plus plan_extra_items
        # pieces; a workspace row holds plan_row_splits() partials.
        from ..decode_plan import DecodePlan, natural_plan
        self.plan_graphs = {}
        self.plan_extra_items = 512
        plan_cap = max_bs + self.plan_extra_items
        plan_items = torch.zeros(2 + 2 * plan_cap, dtype=torch.int32, device=dev)
        plan_seq_splits = torch.ones(max_bs, dtype=torch.int32, device=dev)
        row_splits = self.plan_row_splits()
        self.reserve_workspace()        # a no-op after __init__; covers runners built piecewise (tests)

        def capture(bs, plan):
            graph = torch.cuda.CUDAGraph()
            set_context(False, slot_mapping=slot_mapping[:bs], context_lens=context_lens[:bs],
                        block_tables=block_tables[:bs], seq_order=seq_order[:bs], decode_plan=plan)
            outputs[:bs] = self.model(input_ids[:bs], positions[:bs])   # warm run outside capture
            with torch.cuda.graph(graph, pool=self.graph_pool):
                outputs[:bs] = self.model(input_ids[:bs], positions[:bs])
            if self.graph_pool is None:
                self.graph_pool = graph.pool()
            torch.cuda.synchronize()
            reset_context()
            return graph

        for bs in reversed(self.graph_bs):
            self.graphs[bs] = capture(bs, None)
            if bs >= 4:
                cap = bs + self.plan_extra_items
                nat = natural_plan(bs, 1, capacity=cap)      # a valid list for the warm run
                plan_items[:2 + 2 * cap] = torch.tensor(nat.items, dtype=torch.int32, device=dev)
                static_plan = DecodePlan(plan_items[:2 + 2 * cap], plan_seq_splits[:bs], cap, row_splits, nat.used, 0)
                static_plan.workspace_splits = row_splits
                self.plan_graphs[bs] = capture(bs, static_plan)
        self.graph_vars = dict(input_ids=input_ids, positions=positions, slot_mapping=slot_mapping,
                               context_lens=context_lens, block_tables=block_tables, outputs=outputs,
                               seq_order=seq_order, plan_items=plan_items, plan_seq_splits=plan_seq_splits)
