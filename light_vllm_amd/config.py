"""Engine configuration (reference: lightvllm/config.py:6-55) with the MI355X-specific knobs.

`model` is a local HuggingFace directory, or None together with a ready `hf_config` object
(a namespace with the Qwen3 fields) for random-initialised runs without weights on disk.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, Optional


@dataclass
class Config:
    model: Optional[str] = None
    tensor_parallel_size: int = 1
    max_num_batched_tokens: int = 16384
    max_num_seqs: int = 512
    max_model_len: int = 4096
    gpu_memory_utilization: float = 0.9
    kvcache_block_size: int = 256
    num_kvcache_blocks: int = -1      # -1: size from free HBM; > 0: upper bound on the computed count
    enforce_eager: bool = True
    hf_config: Any = None
    eos: int = -1
    kernel_backend: str = "native"    # accepted for signature parity; there is a single kernel path
    # --- additions
    device: str = "cuda"              # "cuda": rank r runs on cuda:r; "cuda:<i>": every rank on that device; "cpu": plumbing tests that substitute the kernels
    strict_reference: bool = False    # BlockManager.can_append as the reference tests it
    random_seed: Optional[int] = None  # random-init weights (seeded) instead of loading `model`
    random_on_device: bool = False    # benchmarks: draw the random weights on the GPU (fast, not portable)
    random_kind: str = "normal"       # "lagcopy": utils.loader.lagcopy_checkpoint (greedy output decided with a margin)
    dist_init_method: str = "tcp://127.0.0.1:2333"
    dist_backend: Optional[str] = None  # None: "nccl" (RCCL) on GPUs, "gloo" on CPU
    direct_all_reduce: bool = True    # TP > 1 on GPUs: decode-sized all-reduces over IPC-mapped peer buffers (lv_comm)

    def __post_init__(self):
        assert self.kvcache_block_size % 256 == 0, "kvcache_block_size must be a multiple of 256"
        assert 1 <= self.tensor_parallel_size <= 8
        assert self.device == "cpu" or self.device == "cuda" or self.device.startswith("cuda:")
        if self.hf_config is None:
            assert self.model is not None and os.path.isdir(self.model), f"'{self.model}' is not a model directory"
            from transformers import AutoConfig
            self.hf_config = AutoConfig.from_pretrained(self.model)
        self.max_model_len = min(self.max_model_len, self.hf_config.max_position_embeddings)
        assert self.max_num_batched_tokens >= self.max_model_len
