"""Tensor-level wrappers over the C ABI (include/lv_attention.h): shape/dtype checks that mirror
the reference's asserts (lightvllm/layers/attention.py:94-97), raw pointers + torch's current
stream passed down. All work is stream-ordered; nothing here synchronises."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _lib

_BF16 = torch.bfloat16

# kernel_hint values of include/lv_attention.h
DECODE_KERNELS = {"auto": 0, "valu": 1, "mfma": 2}
PREFILL_KERNELS = {"auto": 0, "w4": 1, "w8": 2, "pp": 3, "w64": 5, "w4_dma": 1 | 0x100, "w8_dma": 2 | 0x100}
# Process-wide default for calls that do not name a kernel (the parity tests iterate over the
# kernels by setting these; "valu" applies where that kernel exists, group sizes 1, 2, 4, 8).
DEFAULT_KERNEL = {"decode": "auto", "prefill": "auto"}
STATS = {"tile_order_dropped": 0, "workspace_allocations": 0}
_PERSISTENT_PREFILL = False     # prefill_tile_order(persistent=None) plans item tables for the persistent long-prompt
                                # kernel only when this is set: parity-green, but not faster yet (see prefill_tile_order)


def prefill_tile_order(lens_q, lens_k, num_heads: int, num_kv_heads: int, device,
                       num_splits: int = 0, kernel: Optional[str] = None, force: bool = False,
                       paged: bool = False, persistent: Optional[bool] = None):
    """Launch order of the query tiles for a prefill call on sequences of lens_q new rows over lens_k
    keys (python ints; the caller of the reference's flash_attn_varlen_func has them when it builds
    cu_seqlens, lightvllm/engine/model_runner.py:325-328), as a prefill_order.TileOrder on `device`
    to pass to prefill_attention. None when the library's own mapping serves as well: 8 or more
    kv heads per rank (one XCD per kv head already; measured equal or 1 % slower with a table).
    With 1, 2 or 4 kv heads the table keeps each sequence's K/V stream on one or two XCDs
    (4 x 4096 tokens, 8q/1kv: 0.192 -> 0.147 ms; 16 x 1024, 4q/1kv: 0.039 -> 0.032 ms), and a launch
    that would put at most one 256-row workgroup on a CU is planned on 128-row tiles instead, so
    that every CU gets a heavy and a light tile (4 x 4096, 4q/1kv: 0.107 -> 0.099 ms).
    persistent=True (or the module switch _PERSISTENT_PREFILL with persistent=None): when the launch runs
    the long-prompt kernel in one KV pass (max_seqlen_q >= 2048, or kernel "pp") the result is an ITEM
    table for that kernel's persistent form instead (whatever the head count): one workgroup per compute
    unit walks a host-balanced list of tiles, the next tile's Q rows and first K/V tiles arriving while
    the current one is computed. Bit-identical results; measured on 4 x 4096 tokens, 32q/8kv: what a
    workgroup pays per tile outside its tile loop falls from 7.6 us to 2.5 us, but the tile loop itself
    runs 3-10 % longer and the static lists end 30 us apart, so the launch is 2-3 % SLOWER than one
    tile per workgroup: off by default. paged: the launch reads K/V through a block table (the table's K
    references are then sequence indices).
    force: build a table whatever the head count (tests)."""
    from .prefill_order import build_item_table, build_tile_order, order_columns
    lens_q, lens_k = [int(x) for x in lens_q], [int(x) for x in lens_k]
    if not lens_q or max(lens_q) <= 0:
        return None
    lib = _lib.load()
    max_q, max_k = max(lens_q), max(lens_k)
    nsplit = num_splits if num_splits > 0 else lib.lv_prefill_num_splits(len(lens_q), max_q, max_k, num_heads)
    name = kernel or DEFAULT_KERNEL["prefill"]
    runs_pp = nsplit == 1 and (name == "pp" or (name == "auto" and max_q >= 2048))
    if persistent is None:
        tiles = sum(-(-lq // 256) for lq in lens_q) * num_heads
        persistent = runs_pp and tiles > 256 and _PERSISTENT_PREFILL
    if persistent and runs_pp:
        order = build_item_table(lens_q, lens_k, max_q, 256, num_heads, num_kv_heads, paged)
        if order is None:
            return None
        order.num_splits = 1
        return order.to(device)
    if order_columns(num_kv_heads) == 1 and not force:
        return None
    if name == "auto" and nsplit == 1:
        wgs256 = sum(-(-lq // 256) for lq in lens_q) * num_heads
        if 128 < wgs256 <= 256 and lib.lv_prefill_tile_rows(max_q, 1, 0) == 256:
            name = "w4"
    rows = lib.lv_prefill_tile_rows(max_q, nsplit, PREFILL_KERNELS[name])
    order = build_tile_order(lens_q, lens_k, max_q, rows, num_kv_heads)
    if order is None:
        return None
    order.kernel = name if name != (kernel or DEFAULT_KERNEL["prefill"]) else None
    order.num_splits = nsplit
    return order.to(device)


def _decode_hint(kernel: Optional[str], group: int) -> int:
    name = kernel or DEFAULT_KERNEL["decode"]
    if kernel is None and name == "valu" and group not in (1, 2, 4, 8):
        name = "auto"
    return DECODE_KERNELS[name]


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_gpu(*tensors: torch.Tensor) -> torch.device:
    dev = tensors[0].device
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("light_vllm_amd ops run on the GPU only (got a CPU tensor); "
                               "there is no CPU fallback")
        if t.device != dev:
            raise RuntimeError("all tensors must live on the same device")
    return dev


def store_kvcache(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor,
                  v_cache: torch.Tensor, slot_mapping: torch.Tensor) -> None:
    """k_cache.view(-1, Hkv*D)[slot_mapping[i]] = key[i]; same for value. Slots < 0 are skipped.

    key/value: [N, Hkv, D] bf16, last two dims contiguous, row stride free.
    k_cache/v_cache: [num_blocks, block_size, Hkv, D] contiguous. slot_mapping: int32 [N].
    """
    dev = _require_gpu(key, value, k_cache, v_cache, slot_mapping)
    N, num_heads, head_dim = key.shape
    D = num_heads * head_dim
    assert value.shape == key.shape
    assert key.dtype == _BF16 and value.dtype == _BF16 and k_cache.dtype == _BF16 and v_cache.dtype == _BF16
    assert key.stride(-1) == 1 and value.stride(-1) == 1
    assert key.stride(1) == head_dim and value.stride(1) == head_dim
    assert k_cache.is_contiguous() and v_cache.is_contiguous()
    assert k_cache.shape == v_cache.shape and k_cache.numel() % D == 0
    assert slot_mapping.dtype == torch.int32 and slot_mapping.is_contiguous()
    assert slot_mapping.numel() == N
    if N == 0:
        return
    lib = _lib.load()
    with torch.cuda.device(dev):
        rc = lib.lv_store_kvcache(key.data_ptr(), value.data_ptr(), key.stride(0), value.stride(0),
                                  k_cache.data_ptr(), v_cache.data_ptr(), slot_mapping.data_ptr(),
                                  N, num_heads, head_dim, k_cache.numel() // D, _stream_ptr(dev))
    _lib.check(rc, "lv_store_kvcache")


def prefill_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                      cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor,
                      max_seqlen_q: int, max_seqlen_k: int, softmax_scale: float,
                      block_table: Optional[torch.Tensor] = None,
                      out: Optional[torch.Tensor] = None, kernel: Optional[str] = None,
                      num_splits: int = 0, tile_order=None) -> torch.Tensor:
    """Varlen causal attention (bottom-right aligned mask), GQA, optional paged K/V.

    q: [total_q, Hq, D]. Packed mode: k/v [total_k, Hkv, D]. Paged mode (block_table given):
    k/v are the caches [num_blocks, block_size, Hkv, D] and block_table is int32 [B, max_blocks].
    kernel: a key of PREFILL_KERNELS to force one kernel (default: DEFAULT_KERNEL["prefill"]).
    num_splits: KV splits per query tile (0 = library heuristic: only for few query rows against
    a long context; 1 = one pass; n = n runs + a merge pass through the per-device workspace).
    tile_order: optional prefill_order.TileOrder (its table an int32 tensor on q's device): the launch
    order of the query tiles built from the host's sequence lengths. A performance hint only - it is
    dropped (counted in STATS) when it was built for another tile height than this launch's kernel
    uses or for another max_seqlen_q.
    """
    dev = _require_gpu(q, k, v, cu_seqlens_q, cu_seqlens_k)
    total_q, Hq, D = q.shape
    assert q.dtype == _BF16 and k.dtype == _BF16 and v.dtype == _BF16
    assert q.stride(-1) == 1 and q.stride(1) == D
    assert cu_seqlens_q.dtype == torch.int32 and cu_seqlens_k.dtype == torch.int32
    assert cu_seqlens_q.is_contiguous() and cu_seqlens_k.is_contiguous()
    B = cu_seqlens_q.numel() - 1
    assert cu_seqlens_k.numel() == B + 1
    if block_table is not None:
        _require_gpu(q, block_table)
        assert k.dim() == 4 and v.shape == k.shape and k.is_contiguous() and v.is_contiguous()
        assert block_table.dtype == torch.int32 and block_table.dim() == 2 and block_table.stride(1) == 1
        assert block_table.size(0) == B
        block_size, Hkv = k.size(1), k.size(2)
        assert k.size(3) == D
        k_stride0 = v_stride0 = Hkv * D
        bt_ptr, bt_stride = block_table.data_ptr(), block_table.stride(0)
    else:
        assert k.dim() == 3 and v.shape == k.shape
        Hkv = k.size(1)
        assert k.size(2) == D
        assert k.stride(-1) == 1 and v.stride(-1) == 1 and k.stride(1) == D and v.stride(1) == D
        k_stride0, v_stride0 = k.stride(0), v.stride(0)
        block_size, bt_ptr, bt_stride = 0, None, 0
    if out is None:
        out = torch.empty((total_q, Hq, D), dtype=_BF16, device=dev)
    else:
        assert out.shape == (total_q, Hq, D) and out.dtype == _BF16 and out.stride(-1) == 1 and out.stride(1) == D
    if total_q == 0 or B == 0:
        return out
    lib = _lib.load()
    nsplit = num_splits if num_splits > 0 else lib.lv_prefill_num_splits(B, int(max_seqlen_q), int(max_seqlen_k), Hq)
    ws_ptr, ws_bytes, keep = _workspace(dev, lib.lv_prefill_workspace_bytes(total_q, Hq, D, nsplit))
    name = kernel or DEFAULT_KERNEL["prefill"]
    if tile_order is not None and tile_order.kernel and kernel is None and name == "auto":
        name = tile_order.kernel        # the table was planned on another tile height than the default kernel's
    hint = PREFILL_KERNELS[name]
    to_ptr, to_len, to_rows, to_lanes = None, 0, 0, 0
    if tile_order is not None:
        lanes = getattr(tile_order, "lanes", 0)
        fits = (tile_order.tile_rows == lib.lv_prefill_tile_rows(int(max_seqlen_q), nsplit, hint)
                and tile_order.max_seqlen_q == int(max_seqlen_q)
                and tile_order.cols == lib.lv_prefill_order_columns(Hkv))
        if lanes:       # an item table: the phase-locked kernel in one pass, the same addressing mode it was built for
            pp = (hint & 0xff) == PREFILL_KERNELS["pp"] or ((hint & 0xff) == 0 and int(max_seqlen_q) >= 2048)
            fits = (fits and nsplit == 1 and pp and lanes == lib.lv_prefill_order_lanes(Hq, Hkv)
                    and bool(tile_order.paged) == (block_table is not None))
        if fits:
            tab = tile_order.table
            assert tab.dtype == torch.int32 and tab.is_contiguous() and tab.device == q.device
            to_ptr, to_len, to_rows, to_lanes = tab.data_ptr(), tab.numel(), tile_order.tile_rows, lanes
        else:
            STATS["tile_order_dropped"] += 1
    with torch.cuda.device(dev):
        rc = lib.lv_prefill_attn(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
                                 q.stride(0), k_stride0, v_stride0, out.stride(0),
                                 cu_seqlens_q.data_ptr(), cu_seqlens_k.data_ptr(), B,
                                 int(max_seqlen_q), int(max_seqlen_k), bt_ptr, bt_stride, block_size,
                                 Hq, Hkv, D, float(softmax_scale), total_q, ws_ptr, ws_bytes, nsplit,
                                 to_ptr, to_len, to_rows, to_lanes, hint, _stream_ptr(dev))
    del keep
    _lib.check(rc, "lv_prefill_attn")
    return out


# One split-KV workspace per device, shared by all layers (launches are stream-ordered and each
# decode call finishes reading it in its own merge kernel before the next layer writes it; callers
# that decode on several streams of one device at once must pass their own `workspace`).
# A buffer that has been handed to a launch is never freed: a captured hipGraph keeps the address
# it was captured with, so a larger request retires the old buffer into _RETIRED instead of
# dropping it.
_WORKSPACE: Dict[int, torch.Tensor] = {}
_RETIRED: list = []


def decode_num_splits(batch: int, num_kv_heads: int, max_blocks_per_seq: int, block_size: int = 256) -> int:
    return _lib.load().lv_decode_num_splits(batch, num_kv_heads, max_blocks_per_seq, block_size)


def decode_workspace_bytes(batch: int, num_heads: int, head_dim: int, num_splits: int) -> int:
    return _lib.load().lv_decode_workspace_bytes(batch, num_heads, head_dim, num_splits)


def reserve_decode_workspace(device: torch.device, nbytes: int) -> None:
    """Grow the cached workspace of `device` to at least nbytes. Call before hipGraph capture
    with the largest (batch, splits) the process will see, so capture never has to allocate."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    cur = _WORKSPACE.get(idx)
    if cur is None or cur.numel() < nbytes:
        if cur is not None:
            _RETIRED.append(cur)
        _WORKSPACE[idx] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=f"cuda:{idx}")
        STATS["workspace_allocations"] += 1     # a runner that reserved once sees this stay put


def _workspace(device: torch.device, nbytes: int) -> Tuple[Optional[int], int, Optional[torch.Tensor]]:
    if nbytes == 0:
        return None, 0, None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    cur = _WORKSPACE.get(idx)
    if cur is None or cur.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            # served from the capturing graph's private pool; lives as long as the graph does
            tmp = torch.empty(nbytes, dtype=torch.uint8, device=device)
            return tmp.data_ptr(), nbytes, tmp
        reserve_decode_workspace(device, nbytes)
        cur = _WORKSPACE[idx]
    return cur.data_ptr(), cur.numel(), cur


def decode_attention(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                     context_lens: torch.Tensor, block_table: torch.Tensor, softmax_scale: float,
                     num_splits: int = 0, out: Optional[torch.Tensor] = None,
                     seq_order: Optional[torch.Tensor] = None, kernel: Optional[str] = None,
                     workspace: Optional[torch.Tensor] = None,
                     static_scratch: bool = False, plan=None) -> torch.Tensor:
    """One query token per sequence over the paged cache.

    q: [B, Hq, D]; caches [num_blocks, block_size, Hkv, D] contiguous; context_lens int32 [B]
    (already counting the token stored this step); block_table int32 [B, max_blocks].
    num_splits: 0 = library heuristic, 1 = single pass, n = split-KV + merge.
    seq_order: optional int32 [B] permutation, the order sequences are launched in (longest first
    evens out ragged batches; leave it out for batches of equal lengths, which run ~3 % faster in
    the default order); the result does not depend on it.
    kernel: a key of DECODE_KERNELS to force one kernel (default: DEFAULT_KERNEL["decode"]).
    workspace: caller-owned uint8 scratch for the split partials (default: the per-device one).
    static_scratch: use lv_decode_attn_static (the library's built-in device array holds the
    partials, nothing is allocated here or there; `out` must then be given).
    plan: optional decode_plan.DecodePlan on q's device (lv_decode_attn_planned): a work list built
    from the host's lengths that cuts long sequences into more pieces than short ones; num_splits
    and seq_order are then ignored. With plan_max_splits (a captured graph's fixed workspace row
    capacity) attribute absent the plan's own max_splits sizes the workspace.
    """
    dev = _require_gpu(q, k_cache, v_cache, context_lens, block_table)
    B, Hq, D = q.shape
    assert q.dtype == _BF16 and k_cache.dtype == _BF16 and v_cache.dtype == _BF16
    assert q.stride(-1) == 1 and q.stride(1) == D
    assert k_cache.dim() == 4 and k_cache.shape == v_cache.shape
    assert k_cache.is_contiguous() and v_cache.is_contiguous()
    block_size, Hkv = k_cache.size(1), k_cache.size(2)
    assert k_cache.size(3) == D
    assert context_lens.dtype == torch.int32 and context_lens.is_contiguous() and context_lens.numel() == B
    assert block_table.dtype == torch.int32 and block_table.dim() == 2 and block_table.stride(1) == 1
    assert block_table.size(0) == B
    max_blocks = block_table.size(1)
    if seq_order is not None:
        assert seq_order.dtype == torch.int32 and seq_order.is_contiguous() and seq_order.numel() == B
        assert seq_order.device == q.device
    if out is None:
        out = torch.empty((B, Hq, D), dtype=_BF16, device=dev)
    else:
        assert out.shape == (B, Hq, D) and out.dtype == _BF16 and out.stride(-1) == 1 and out.stride(1) == D
    if B == 0:
        return out
    lib = _lib.load()
    if plan is not None:
        items, ss = plan.items, plan.seq_splits
        assert items.dtype == torch.int32 and items.is_contiguous() and items.device == q.device
        assert ss.dtype == torch.int32 and ss.is_contiguous() and ss.device == q.device
        assert items.numel() == 2 + 2 * plan.capacity and ss.numel() >= B
        cap_splits = int(getattr(plan, "workspace_splits", 0) or plan.max_splits)
        assert cap_splits >= plan.max_splits
        need = lib.lv_decode_workspace_bytes(B, Hq, D, cap_splits)
        if workspace is not None:
            assert workspace.is_cuda and workspace.device == dev and workspace.is_contiguous()
            ws_ptr, ws_bytes, keep = workspace.data_ptr(), workspace.numel() * workspace.element_size(), workspace
        else:
            ws_ptr, ws_bytes, keep = _workspace(dev, need)
        with torch.cuda.device(dev):
            rc = lib.lv_decode_attn_planned(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(),
                                            q.stride(0), out.stride(0), context_lens.data_ptr(),
                                            block_table.data_ptr(), block_table.stride(0), B, max_blocks, block_size,
                                            Hq, Hkv, D, float(softmax_scale), ws_ptr, ws_bytes, cap_splits,
                                            items.data_ptr(), plan.capacity, ss.data_ptr(),
                                            _decode_hint(kernel, Hq // Hkv), _stream_ptr(dev))
        _lib.check(rc, "lv_decode_attn_planned")
        del keep
        return out
    nsplit = num_splits if num_splits > 0 else lib.lv_decode_num_splits(B, Hkv, max_blocks, block_size)
    if static_scratch:
        with torch.cuda.device(dev):
            rc = lib.lv_decode_attn_static(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(),
                                           q.stride(0), out.stride(0), context_lens.data_ptr(),
                                           block_table.data_ptr(), block_table.stride(0), B, max_blocks,
                                           block_size, Hq, Hkv, D, float(softmax_scale), nsplit,
                                           seq_order.data_ptr() if seq_order is not None else None,
                                           _decode_hint(kernel, Hq // Hkv), _stream_ptr(dev))
        _lib.check(rc, "lv_decode_attn_static")
        return out
    need = lib.lv_decode_workspace_bytes(B, Hq, D, nsplit)
    if workspace is not None:
        assert workspace.is_cuda and workspace.device == dev and workspace.is_contiguous()
        ws_ptr, ws_bytes, keep = workspace.data_ptr(), workspace.numel() * workspace.element_size(), workspace
    else:
        ws_ptr, ws_bytes, keep = _workspace(dev, need)
    with torch.cuda.device(dev):
        rc = lib.lv_decode_attn(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(),
                                q.stride(0), out.stride(0), context_lens.data_ptr(),
                                block_table.data_ptr(), block_table.stride(0), B, max_blocks, block_size,
                                Hq, Hkv, D, float(softmax_scale), ws_ptr, ws_bytes, nsplit,
                                seq_order.data_ptr() if seq_order is not None else None,
                                _decode_hint(kernel, Hq // Hkv), _stream_ptr(dev))
    _lib.check(rc, "lv_decode_attn")
    del keep
    return out


def qk_norm_rope_store(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, positions: torch.Tensor,
                       q_weight: torch.Tensor, k_weight: torch.Tensor, eps: float,
                       cos_sin_cache: torch.Tensor, k_cache: Optional[torch.Tensor] = None,
                       v_cache: Optional[torch.Tensor] = None,
                       slot_mapping: Optional[torch.Tensor] = None) -> None:
    """Per-head RMSNorm of q and k, neox RoPE of both (in place), then the K/V cache scatter.

    q: [N, Hq, 128], k/v: [N, Hkv, 128] bf16 views (row stride free, normally slices of the fused
    qkv activation). positions int64 [N]. q_weight/k_weight bf16 [128]. cos_sin_cache fp32
    [max_position, 128] (cos | sin). With k_cache/v_cache/slot_mapping None (or empty caches) the
    store is skipped, as the reference does during warmup (lightvllm/layers/attention.py:149).
    """
    dev = _require_gpu(q, k, v, positions, q_weight, k_weight, cos_sin_cache)
    N, Hq, D = q.shape
    Hkv = k.size(1)
    assert k.shape == (N, Hkv, D) and v.shape == (N, Hkv, D)
    assert q.dtype == _BF16 and k.dtype == _BF16 and v.dtype == _BF16
    assert q_weight.dtype == _BF16 and k_weight.dtype == _BF16
    assert q_weight.numel() == D and k_weight.numel() == D and q_weight.is_contiguous() and k_weight.is_contiguous()
    for t in (q, k, v):
        assert t.stride(-1) == 1 and t.stride(1) == D
    assert positions.dtype == torch.int64 and positions.is_contiguous() and positions.numel() == N
    assert cos_sin_cache.dtype == torch.float32 and cos_sin_cache.is_contiguous()
    assert cos_sin_cache.dim() == 2 and cos_sin_cache.size(1) == D
    store = k_cache is not None and v_cache is not None and k_cache.numel() > 0 and v_cache.numel() > 0
    kc_ptr = vc_ptr = sm_ptr = None
    num_slots = 0
    if store:
        _require_gpu(q, k_cache, v_cache, slot_mapping)
        assert k_cache.dtype == _BF16 and v_cache.dtype == _BF16
        assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape == v_cache.shape
        assert k_cache.numel() % (Hkv * D) == 0
        assert slot_mapping.dtype == torch.int32 and slot_mapping.is_contiguous() and slot_mapping.numel() == N
        kc_ptr, vc_ptr, sm_ptr = k_cache.data_ptr(), v_cache.data_ptr(), slot_mapping.data_ptr()
        num_slots = k_cache.numel() // (Hkv * D)
    if N == 0:
        return
    lib = _lib.load()
    with torch.cuda.device(dev):
        rc = lib.lv_qk_norm_rope_store(q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0),
                                       v.stride(0), positions.data_ptr(), q_weight.data_ptr(),
                                       k_weight.data_ptr(), float(eps), cos_sin_cache.data_ptr(),
                                       cos_sin_cache.size(0), kc_ptr, vc_ptr, sm_ptr, num_slots, N,
                                       Hq, Hkv, D, _stream_ptr(dev))
    _lib.check(rc, "lv_qk_norm_rope_store")


def add_rms_norm(x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor,
                 eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """RMSNorm of x (+ residual). With a residual, `residual` is updated in place to the rounded
    sum x + residual and the normalised sum is returned. x/residual: [N, hidden] bf16."""
    dev = _require_gpu(x, weight) if residual is None else _require_gpu(x, residual, weight)
    hidden = x.size(-1)
    x2 = x.view(-1, hidden)
    assert x2.dtype == _BF16 and weight.dtype == _BF16 and x2.stride(-1) == 1
    assert weight.numel() == hidden and weight.is_contiguous()
    N = x2.size(0)
    res_ptr, res_stride = None, 0
    if residual is not None:
        r2 = residual.view(-1, hidden)
        assert r2.shape == x2.shape and r2.dtype == _BF16 and r2.stride(-1) == 1
        res_ptr, res_stride = r2.data_ptr(), r2.stride(0)
    if out is None:
        out = torch.empty((N, hidden), dtype=_BF16, device=dev)
    o2 = out.view(-1, hidden)
    assert o2.shape == x2.shape and o2.dtype == _BF16 and o2.stride(-1) == 1
    if N:
        lib = _lib.load()
        with torch.cuda.device(dev):
            rc = lib.lv_add_rms_norm(x2.data_ptr(), res_ptr, weight.data_ptr(), o2.data_ptr(),
                                     x2.stride(0), res_stride, o2.stride(0), N, hidden, float(eps),
                                     _stream_ptr(dev))
        _lib.check(rc, "lv_add_rms_norm")
    return out.view(x.shape)


def silu_and_mul(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """silu(x[..., :I]) * x[..., I:] for x [N, 2I] bf16 -> [N, I]."""
    dev = _require_gpu(x)
    inter = x.size(-1) // 2
    assert x.size(-1) == 2 * inter
    x2 = x.view(-1, 2 * inter)
    assert x2.dtype == _BF16 and x2.stride(-1) == 1
    N = x2.size(0)
    if out is None:
        out = torch.empty((*x.shape[:-1], inter), dtype=_BF16, device=dev)
    o2 = out.view(-1, inter)
    assert o2.size(0) == N and o2.dtype == _BF16 and o2.stride(-1) == 1
    if N:
        lib = _lib.load()
        with torch.cuda.device(dev):
            rc = lib.lv_silu_and_mul(x2.data_ptr(), o2.data_ptr(), x2.stride(0), o2.stride(0), N, inter,
                                     _stream_ptr(dev))
        _lib.check(rc, "lv_silu_and_mul")
    return out
