"""generate() loop: tokenise, schedule, run one model step, post-process — the reference's
LLMEngine (lightvllm/engine/llm_engine.py:19-297) with the same step accounting: a prefill step
reports the number of prompt tokens (positive), a decode step the negated batch size (:166), and
the averages of both rates are kept for the caller (`last_stats`) as well as printed.
Rank 0 lives in the caller's process; ranks > 0 are spawned, one process per GPU.
"""
from __future__ import annotations

import atexit
from dataclasses import fields
from time import perf_counter

import torch.multiprocessing as mp

from ..config import Config
from ..sampling_params import SamplingParams
from .model_runner import ModelRunner
from .scheduler import Scheduler
from .sequence import Sequence


class LLMEngine:

    def __init__(self, model=None, **kwargs):
        names = {f.name for f in fields(Config)}
        config = Config(model, **{k: v for k, v in kwargs.items() if k in names})
        self.config = config
        self.ps, self.events = [], []
        ctx = mp.get_context("spawn")
        for i in range(1, config.tensor_parallel_size):
            event = ctx.Event()
            process = ctx.Process(target=ModelRunner, args=(config, i, event))
            process.start()
            self.ps.append(process)
            self.events.append(event)
        self.model_runner = ModelRunner(config, 0, self.events)
        self.tokenizer = None
        if config.model is not None:
            from transformers import AutoTokenizer
            self.tokenizer = AutoTokenizer.from_pretrained(config.model, use_fast=True)
            config.eos = self.tokenizer.eos_token_id
        self.scheduler = Scheduler(config)
        self.last_stats = {}
        self._closed = False
        atexit.register(self.exit)

    def exit(self):
        if self._closed:
            return
        self._closed = True
        self.model_runner.call("exit")
        del self.model_runner
        for p in self.ps:
            p.join()

    def add_request(self, prompt, sampling_params: SamplingParams):
        if isinstance(prompt, str):
            assert self.tokenizer is not None, "string prompts need a tokenizer (model directory)"
            prompt = self.tokenizer.encode(prompt)
        self.scheduler.add(Sequence(prompt, sampling_params))

    def step(self):
        seqs, is_prefill = self.scheduler.schedule()
        all_greedy = all(seq.temperature == 0 for seq in seqs)
        token_ids = self.model_runner.call("run", seqs, is_prefill, all_greedy)
        self.scheduler.postprocess(seqs, token_ids)
        outputs = [(seq.seq_id, seq.completion_token_ids) for seq in seqs if seq.is_finished]
        num_tokens = sum(len(seq) for seq in seqs) if is_prefill else -len(seqs)
        return outputs, num_tokens

    def is_finished(self):
        return self.scheduler.is_finished()

    def generate(self, prompts, sampling_params, use_tqdm: bool = False):
        if not isinstance(sampling_params, list):
            sampling_params = [sampling_params] * len(prompts)
        for prompt, sp in zip(prompts, sampling_params):
            self.add_request(prompt, sp)
        pbar = None
        if use_tqdm:
            from tqdm.auto import tqdm
            pbar = tqdm(total=len(prompts), desc="Generating", dynamic_ncols=True)
        outputs = {}
        pre_tok = dec_tok = 0
        pre_t = dec_t = 0.0
        while not self.is_finished():
            t = perf_counter()
            output, num_tokens = self.step()
            dt = perf_counter() - t
            if num_tokens > 0:
                pre_tok += num_tokens
                pre_t += dt
            else:
                dec_tok += -num_tokens
                dec_t += dt
            if pbar is not None:
                pbar.set_postfix({"Prefill": f"{int(pre_tok / pre_t) if pre_t else 0} tokens/s",
                                  "Decode": f"{int(dec_tok / dec_t) if dec_t else 0} tokens/s"})
            for seq_id, token_ids in output:
                outputs[seq_id] = token_ids
                if pbar is not None:
                    pbar.update(1)
        if pbar is not None:
            pbar.close()
        self.last_stats = {"prefill_tokens": pre_tok, "prefill_time": pre_t, "decode_tokens": dec_tok,
                           "decode_time": dec_t,
                           "avg_prefill_tok_s": pre_tok / pre_t if pre_t else 0.0,
                           "avg_decode_tok_s": dec_tok / dec_t if dec_t else 0.0}
        ordered = [outputs[seq_id] for seq_id in sorted(outputs)]
        decode = self.tokenizer.decode if self.tokenizer is not None else (lambda ids: "")
        return [{"text": decode(ids), "token_ids": ids} for ids in ordered]
