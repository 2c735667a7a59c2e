"""Host-side launch order of the query tiles of a prefill attention call (the `tile_order` argument
of lv_prefill_attn, include/lv_attention.h).

The caller of the reference's flash_attn_varlen_func (lightvllm/layers/attention.py:159-165) has
every sequence length on the host when it builds cu_seqlens (engine/model_runner.py:325-328); the
kernels see only max_seqlen_q. From the lengths this module builds the table that tells the
kernels which (sequence, query tile) each workgroup serves:

  * with 1, 2 or 4 kv heads per rank (tensor-parallel shards) the table has 8 / kv-heads columns,
    one per XCD group; a sequence's tiles are kept in as few columns as balance allows, so one XCD's
    L2 fetches that sequence's K/V stream instead of all eight;
  * a column's rows go heaviest first (most key tiles to walk). (Measured on 4 x 4096 tokens, 8q/1kv:
    0.192 ms with the library's mapping, 0.147 ms this way; putting the second half of a column
    lightest first, to pair heavy and light tiles on a compute unit, 0.160 ms.)
  * no workgroup is launched for tiles beyond a sequence's end (the library's own mapping launches
    ceil(max_seqlen_q / rows) tiles for every sequence and lets the surplus exit).

Pure Python on plain ints; results of the attention call never depend on the table.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

KEY_TILE = 64            # keys per K/V tile of the prefill kernels


@dataclass
class TileOrder:
    """table: flat [rows * cols] list (or device tensor) of b * qtiles + t, -1 = empty place; with
    lanes > 0 an item table, flat [depth * lanes * cols * 8] (build_item_table).
    kernel / num_splits: what the table was planned for (None / 0: the library's own choice)."""
    table: object
    tile_rows: int
    cols: int
    max_seqlen_q: int
    kernel: Optional[str] = None
    num_splits: int = 0
    lanes: int = 0
    paged: bool = False      # item tables only: K references are sequence indices (block-table rows)

    def to(self, device):
        import torch
        t = torch.tensor(self.table, dtype=torch.int32, device="cpu")
        if torch.device(device).type != "cpu":
            t = t.pin_memory().to(device, non_blocking=True)
        return TileOrder(t, self.tile_rows, self.cols, self.max_seqlen_q, self.kernel, self.num_splits, self.lanes, self.paged)


def tile_weight(lq: int, lk: int, t: int, tile_rows: int) -> int:
    """Key tiles the workgroup of query tile t walks (bottom-right aligned causal mask)."""
    last_row = min((t + 1) * tile_rows, lq) - 1
    max_key = min(last_row + (lk - lq), lk - 1)
    return max_key // KEY_TILE + 1 if max_key >= 0 else 0


def order_columns(num_kv_heads: int) -> int:
    """lv_prefill_order_columns: table columns for this many kv heads per rank."""
    return 8 // num_kv_heads if num_kv_heads in (1, 2, 4) else 1


def build_tile_order(lens_q: Sequence[int], lens_k: Sequence[int], max_seqlen_q: int, tile_rows: int,
                     num_kv_heads: int) -> Optional[TileOrder]:
    """Table for sequences of lens_q new rows on lens_k keys. tile_rows = lv_prefill_tile_rows(...) of
    the launch. None when there is no tile."""
    cols = order_columns(num_kv_heads)
    assert len(lens_q) == len(lens_k) and tile_rows > 0
    qtiles = -(-max_seqlen_q // tile_rows)
    seqs = []        # (total weight, b, [(weight, code)] heaviest first)
    for b, (lq, lk) in enumerate(zip(lens_q, lens_k)):
        assert 0 <= lq <= max_seqlen_q
        nt = -(-lq // tile_rows)
        items = [(tile_weight(lq, lk, t, tile_rows), b * qtiles + t) for t in range(nt - 1, -1, -1)]
        if items:
            seqs.append((sum(w for w, _ in items), b, items))
    if not seqs:
        return None
    seqs.sort(key=lambda s: -s[0])
    # one stream of items, sequence after sequence; inside a sequence heavy and light tiles alternate,
    # so any contiguous piece of it carries about the sequence's mean weight
    stream = []
    for _, _, items in seqs:
        lo, hi = 0, len(items) - 1
        while lo <= hi:
            stream.append(items[lo])
            if hi != lo:
                stream.append(items[hi])
            lo, hi = lo + 1, hi - 1
    total = sum(w for w, _ in stream)
    columns: List[List] = [[] for _ in range(cols)]
    if cols == 1:
        columns[0] = stream
    else:
        # cut the stream into `cols` pieces of equal weight (an item goes where its midpoint falls)
        cum = 0
        for w, code in stream:
            c = min(cols - 1, int((cum + w / 2) * cols / max(total, 1)))
            columns[c].append((w, code))
            cum += w
    nrows = max(len(c) for c in columns)
    table = [-1] * (nrows * cols)
    for c, col in enumerate(columns):
        col.sort(key=lambda it: -it[0])
        for r, (_, code) in enumerate(col):
            table[r * cols + c] = code
    return TileOrder(table, tile_rows, cols, max_seqlen_q)


def order_lanes(num_heads: int, num_kv_heads: int) -> int:
    """lv_prefill_order_lanes: lists per (query head, column) of an item table."""
    per_lane = num_heads * order_columns(num_kv_heads)
    return 1 if per_lane >= 256 else 256 // per_lane


def build_item_table(lens_q: Sequence[int], lens_k: Sequence[int], max_seqlen_q: int, tile_rows: int,
                     num_heads: int, num_kv_heads: int, paged: bool) -> Optional[TileOrder]:
    """Item table for the persistent long-prompt kernel (lv_prefill_attn, tile_order_lanes > 0): the
    query tiles of sequences of lens_q new rows over lens_k keys, dealt to `lanes` lists per table
    column so that every list carries about the same number of key tiles (longest processing time
    first: heaviest tile to the list that is lightest so far), each list heaviest first. A sequence's
    tiles stay inside one or two columns as in build_tile_order (one XCD's L2 per K/V stream). An entry
    is {first q row, valid rows, causal limit of the first row, K reference, keys, 0, 0, 0}, the K
    reference being the sequence's first packed K/V row, or its index when K/V are paged. None when
    there is no tile."""
    cols = order_columns(num_kv_heads)
    lanes = order_lanes(num_heads, num_kv_heads)
    assert len(lens_q) == len(lens_k) and tile_rows > 0
    q_start = k_start = 0
    seqs = []        # (total weight, [(weight, entry)])
    for b, (lq, lk) in enumerate(zip(lens_q, lens_k)):
        assert 0 <= lq <= max_seqlen_q
        items = []
        for t in range(-(-lq // tile_rows) - 1, -1, -1):
            q0 = t * tile_rows
            entry = [q_start + q0, min(tile_rows, lq - q0), q0 + lk - lq, b if paged else k_start, lk, 0, 0, 0]
            items.append((tile_weight(lq, lk, t, tile_rows), entry))
        if items:
            seqs.append((sum(w for w, _ in items), items))
        q_start += lq
        k_start += lk
    if not seqs:
        return None
    seqs.sort(key=lambda s: -s[0])
    stream = []
    for _, items in seqs:       # heavy and light tiles of a sequence alternate: any piece carries its mean weight
        lo, hi = 0, len(items) - 1
        while lo <= hi:
            stream.append(items[lo])
            if hi != lo:
                stream.append(items[hi])
            lo, hi = lo + 1, hi - 1
    total = sum(w for w, _ in stream)
    columns: List[List] = [[] for _ in range(cols)]
    cum = 0
    for w, entry in stream:
        c = 0 if cols == 1 else min(cols - 1, int((cum + w / 2) * cols / max(total, 1)))
        columns[c].append((w, entry))
        cum += w
    lists = []       # [col][lane] -> entries
    for col in columns:
        col.sort(key=lambda it: -it[0])
        load = [0] * lanes
        mine: List[List] = [[] for _ in range(lanes)]
        for w, entry in col:
            r = min(range(lanes), key=lambda j: (load[j], j))
            mine[r].append(entry)
            load[r] += max(w, 1)
        lists.append(mine)
    depth = max(len(lst) for col in lists for lst in col)
    empty = [0] * 8
    table: List[int] = []
    for k in range(depth):
        for r in range(lanes):
            for c in range(cols):
                lst = lists[c][r]
                table += lst[k] if k < len(lst) else empty
    return TileOrder(table, tile_rows, cols, max_seqlen_q, lanes=lanes, paged=paged)
