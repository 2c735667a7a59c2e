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
    """table: flat [rows * cols] list (or device tensor) of b * qtiles + t, -1 = empty place.
    kernel / num_splits: what the table was planned for (None / 0: the library's own choice)."""
    table: object
    tile_rows: int
    cols: int
    max_seqlen_q: int
    kernel: Optional[str] = None
    num_splits: int = 0

    def to(self, device):
        import torch
        t = torch.tensor(self.table, dtype=torch.int32, device="cpu")
        if torch.device(device).type != "cpu":
            t = t.pin_memory().to(device, non_blocking=True)
        return TileOrder(t, self.tile_rows, self.cols, self.max_seqlen_q, self.kernel, self.num_splits)


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
