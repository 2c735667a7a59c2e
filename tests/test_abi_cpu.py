"""The C-ABI library loads without a GPU and exports exactly what include/lv_attention.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", h) for h in ("lv_attention.h", "lv_comm.h")]


def declared_functions():
    names = set()
    for header in HEADERS:
        text = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
        names.update(re.findall(r"\b(lv_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_header_declares_the_expected_entry_points():
    assert declared_functions() == sorted([
        "lv_abi_version", "lv_error_string", "lv_last_error", "lv_store_kvcache", "lv_prefill_attn",
        "lv_qk_norm_rope_store", "lv_add_rms_norm", "lv_silu_and_mul",
        "lv_decode_num_splits", "lv_decode_workspace_bytes", "lv_decode_attn", "lv_decode_attn_static",
        "lv_decode_attn_planned", "lv_prefill_max_workspace_bytes",
        "lv_prefill_num_splits", "lv_prefill_workspace_bytes", "lv_prefill_tile_rows", "lv_prefill_order_columns", "lv_prefill_order_lanes",
        "lv_comm_create", "lv_comm_get_handle", "lv_comm_connect", "lv_comm_all_reduce", "lv_comm_failed",
        "lv_comm_destroy", "lv_comm_all_reduce_add_rms_norm", "lv_comm_set_one_shot_inflow", "lv_comm_one_shot_inflow"])


def test_library_exports_every_declared_symbol():
    from light_vllm_amd import _lib
    lib = _lib.load()
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} missing from liblvattn.so"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == declared_functions()


def test_host_only_entry_points():
    """Entry points that never touch the device can be called without one."""
    from light_vllm_amd import _lib
    lib = _lib.load()
    assert lib.lv_abi_version() == _lib.ABI_VERSION
    assert lib.lv_error_string(0) == b"ok"
    assert lib.lv_error_string(-2) == b"unsupported configuration"
    assert lib.lv_decode_workspace_bytes(4, 32, 128, 1) == 0
    assert lib.lv_decode_workspace_bytes(4, 32, 128, 8) == 4 * 32 * 8 * 130 * 4
    assert lib.lv_decode_workspace_bytes(4, 32, 64, 8) == 0
    assert lib.lv_decode_num_splits(256, 8, 4, 256) == 1      # enough (sequence, kv head) units already
    assert lib.lv_decode_num_splits(1, 1, 128, 256) == 256    # one long sequence: a workgroup per CU
    assert lib.lv_decode_num_splits(8, 8, 128, 256) == 32     # 32k-token capacity: ~1024 tokens per split
    assert lib.lv_decode_num_splits(1, 8, 1, 256) == 4        # a single block: the kernel's 64-token unit
    assert lib.lv_decode_num_splits(64, 8, 16, 256) == 4      # 64 sequences: cut in 4 for ~256 read streams
    assert lib.lv_decode_num_splits(256, 1, 16, 256) == 1     # 256 sequences are streams enough
    assert lib.lv_decode_num_splits(8, 1, 16, 256) == 32      # few units: one workgroup per CU
    # prefill: no KV splits while the query tiles alone give every CU a workgroup or the context is short ...
    assert lib.lv_prefill_num_splits(4, 4096, 4096, 32) == 1
    assert lib.lv_prefill_num_splits(1, 100, 400, 8) == 1
    # ... else about two workgroups per CU, runs of at least eight 64-key tiles, at most 32
    assert lib.lv_prefill_num_splits(8, 100, 4196, 8) == 8       # 64 workgroups, 66 tiles
    assert lib.lv_prefill_num_splits(8, 100, 4196, 32) == 2      # 256 workgroups
    assert lib.lv_prefill_num_splits(1, 100, 32768, 8) == 32     # 8 workgroups, 512 tiles
    # the reservation bound covers every launch the heuristic splits
    for batch, max_q, max_k, heads in ((8, 100, 4196, 8), (8, 100, 4196, 32), (1, 100, 32768, 8), (1, 4096, 4096, 4),
                                       (2, 512, 8192, 32), (1, 16384, 16384, 4), (3, 300, 40000, 2), (1, 1, 32768, 1)):
        n = lib.lv_prefill_num_splits(batch, max_q, max_k, heads)
        assert lib.lv_prefill_workspace_bytes(batch * max_q, heads, 128, n) <= \
            lib.lv_prefill_max_workspace_bytes(batch * max_q, heads, 128), (batch, max_q, max_k, heads, n)
    assert lib.lv_prefill_max_workspace_bytes(16384, 32, 128) == 256 * 768 * 130 * 4
    assert lib.lv_prefill_max_workspace_bytes(10, 2, 128) == 10 * 2 * 32 * 130 * 4
    # query rows per tile of the kernel a launch runs; columns of a tile_order table
    assert lib.lv_prefill_tile_rows(1024, 1, 0) == 128 and lib.lv_prefill_tile_rows(1025, 1, 0) == 256
    assert lib.lv_prefill_tile_rows(4096, 1, 0) == 256 and lib.lv_prefill_tile_rows(100, 8, 0) == 128
    assert lib.lv_prefill_tile_rows(100, 1, 3) == 256 and lib.lv_prefill_tile_rows(4096, 1, 1) == 128
    assert lib.lv_prefill_tile_rows(100, 1, 77) == 0
    assert [lib.lv_prefill_order_columns(h) for h in (1, 2, 3, 4, 8, 16)] == [8, 4, 1, 2, 1, 1]
    assert lib.lv_prefill_num_splits(1, 40, 1064, 8) == 2        # 17 tiles: runs of >= 8
    # whole prompts: up to 128 workgroups as above; 129..256 only from 8192 keys
    assert lib.lv_prefill_num_splits(1, 4096, 4096, 4) == 8      # 64 workgroups
    assert lib.lv_prefill_num_splits(2, 4096, 4096, 4) == 4      # 128 workgroups
    assert lib.lv_prefill_num_splits(4, 4096, 4096, 4) == 1      # 256 workgroups, 64 tiles
    assert lib.lv_prefill_num_splits(1, 16384, 16384, 4) == 2    # 256 workgroups, 256 tiles
    assert lib.lv_prefill_workspace_bytes(300, 8, 128, 4) == 300 * 8 * 4 * 130 * 4
    assert lib.lv_prefill_workspace_bytes(300, 8, 128, 1) == 0


def test_argument_validation_happens_before_any_launch():
    """Bad arguments are rejected on the host with a message (no GPU needed)."""
    from light_vllm_amd import _lib
    lib = _lib.load()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    # head_dim 64 is not supported
    rc = lib.lv_decode_attn(p16, p16, p16, p16, 1024, 1024, p16, p16, 4, 1, 4, 256, 8, 8, 64, 0.1, None, 0, 1, None, 0, None)
    assert rc == -2 and b"head_dim" in lib.lv_last_error()
    # block size must be a multiple of 32
    rc = lib.lv_decode_attn(p16, p16, p16, p16, 1024, 1024, p16, p16, 4, 1, 4, 48, 8, 8, 128, 0.1, None, 0, 1, None, 0, None)
    assert rc == -2
    # split-KV without a workspace
    rc = lib.lv_decode_attn(p16, p16, p16, p16, 1024, 1024, p16, p16, 4, 1, 4, 256, 8, 8, 128, 0.1, None, 0, 4, None, 0, None)
    assert rc == -4 and b"workspace" in lib.lv_last_error()
    # misaligned key pointer
    rc = lib.lv_store_kvcache(p16 + 2, p16, 1024, 1024, p16, p16, p16, 1, 8, 128, 256, None)
    assert rc == -2
    # row stride smaller than a row
    rc = lib.lv_store_kvcache(p16, p16, 512, 1024, p16, p16, p16, 1, 8, 128, 256, None)
    assert rc == -1
    # heads not divisible
    rc = lib.lv_prefill_attn(p16, p16, p16, p16, 1024, 1024, 1024, 1024, p16, p16, 1, 8, 8, None, 0, 0, 12, 8, 128, 0.1, 8, None, 0, 1, None, 0, 0, 0, 0, None)
    assert rc == -1
    # scale must be finite and positive (a zero scale would turn the pre-scaled prefill kernel's constants into NaN)
    rc = lib.lv_prefill_attn(p16, p16, p16, p16, 1024, 1024, 1024, 1024, p16, p16, 1, 8, 8, None, 0, 0, 8, 8, 128, 0.0, 8, None, 0, 1, None, 0, 0, 0, 0, None)
    assert rc == -1 and b"scale" in lib.lv_last_error()
    rc = lib.lv_decode_attn(p16, p16, p16, p16, 1024, 1024, p16, p16, 4, 1, 4, 256, 8, 8, 128, float("nan"), None, 0, 1, None, 0, None)
    assert rc == -1 and b"scale" in lib.lv_last_error()
    # unknown kernel hints, and the vector-ALU decode kernel for a group size it does not serve
    rc = lib.lv_decode_attn(p16, p16, p16, p16, 1024, 1024, p16, p16, 4, 1, 4, 256, 8, 8, 128, 0.1, None, 0, 1, None, 7, None)
    assert rc == -1 and b"kernel_hint" in lib.lv_last_error()
    rc = lib.lv_decode_attn(p16, p16, p16, p16, 1024, 1024, p16, p16, 4, 1, 4, 256, 24, 8, 128, 0.1, None, 0, 1, None, 1, None)
    assert rc == -2
    # zero-size calls are no-ops
    assert lib.lv_store_kvcache(None, None, 0, 0, None, None, None, 0, 8, 128, 256, None) == 0
    assert lib.lv_decode_attn(None, None, None, None, 0, 0, None, None, 0, 0, 0, 256, 8, 8, 128, 0.1, None, 0, 1, None, 0, None) == 0


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from light_vllm_amd import ops
    x = torch.zeros(1, 8, 128, dtype=torch.bfloat16)
    c = torch.zeros(1, 256, 8, 128, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.store_kvcache(x, x, c, c, torch.zeros(1, dtype=torch.int32))


def test_comm_argument_validation_without_a_device():
    """The communicator's entry points reject bad arguments on the host before any HIP call."""
    from light_vllm_amd import _lib
    lib = _lib.load()
    comm = ctypes.c_void_p()
    assert lib.lv_comm_create(5, 2, 0, ctypes.byref(comm)) == -1 and not comm.value       # rank outside the world
    assert lib.lv_comm_create(0, 9, 0, ctypes.byref(comm)) == -1 and b"2..8" in lib.lv_last_error()
    assert lib.lv_comm_create(0, 2, -1, ctypes.byref(comm)) == -1
    assert lib.lv_comm_create(0, 2, 0, None) == -1
    assert lib.lv_comm_all_reduce(None, None, 8, 0, None) == -1
    assert lib.lv_comm_all_reduce_add_rms_norm(None, None, None, None, 4, 4096, 1e-6, None) == -1
    assert lib.lv_comm_get_handle(None, None) == -1 and lib.lv_comm_connect(None, None) == -1
    assert lib.lv_comm_failed(None) == 1 and lib.lv_comm_destroy(None) == 0
    assert lib.lv_error_string(-5) == b"communicator failure"
