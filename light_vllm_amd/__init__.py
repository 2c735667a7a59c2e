"""MI355X-native paged attention for light-vllm: the Attention module, its Context contract and
the host pieces around it, backed by hand-written gfx950 HIP kernels in lib/liblvattn.so."""
from .utils.context import Context, get_context, set_context, reset_context  # noqa: F401
from .layers.attention import Attention, store_kvcache  # noqa: F401

__all__ = ["Attention", "store_kvcache", "Context", "get_context", "set_context", "reset_context"]
