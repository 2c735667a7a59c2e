"""Public entry point, as in the reference (lightvllm/llm.py:4): LLM is the engine."""
from .engine.llm_engine import LLMEngine


class LLM(LLMEngine):
    pass
